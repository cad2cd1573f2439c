#!/bin/bash
# kernel trace of the overlapped staged assembly: do the row sums really run beside the next chunk's assembly?
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-trace_overlap}
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --no-cpu --steps 3 --warmup 1 --stage-overlap 1 --stage-chunk ${2:-65536} > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
python3 - $O <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "closed" in r["Kernel_Name"] or "gather_rows" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[:80]:
    print("%-8s q%-3s %10.1f -> %10.1f us (%7.1f)" % ("K1" if "closed" in r["Kernel_Name"] else "rows", r.get("Queue_Id", "?"),
          (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
P
