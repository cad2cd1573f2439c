#!/bin/bash
# same-call A/B of the staged assembly: one chunk (serial) against chunked with / without the row sums on a second stream
# tools/tune_overlap.sh <tag> [chunks...]
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-tune_overlap}
mkdir -p $O
shift
CH=${@:-"16384 32768 65536 131072 262144"}
run() {  # name, bench args
  n=$1; shift
  timeout -k 10 200 python3 bench.py --no-cpu --steps 20 --warmup 3 "$@" > $O/$n.json 2> $O/$n.err || { tail -3 $O/$n.err; exit 1; }
  python3 - $O/$n.json $n <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-28s %.3f ms/step  kernel %.3f  assign %.3f  ad-form %.3f" % (sys.argv[2], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"],
      d["also"].get("ms_per_step_assign_mode", 0), d["also"].get("ms_per_step_iterated_ad_form", 0)))
P
}
run one_chunk --stage-overlap 0
for c in $CH; do
  run serial_$c --stage-overlap 0 --stage-chunk $c
  run overlap_$c --stage-overlap 1 --stage-chunk $c
done
run one_chunk_again --stage-overlap 0
