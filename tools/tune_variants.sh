#!/bin/bash
# Tuning builds on the GPU box: every flag set is compiled there (hipcc is in the image) and timed with the bench
# workload; results are valid assemblies (C8_TUNE_* switches change timing only).  Usage:
#   gpurun -- 'bash tools/tune_variants.sh "" "-DC8_TUNE_GATHER_PAD=4" ...'
# Each argument is one flag set (C8_EXTRA_FLAGS); extra bench arguments go in BENCH_ARGS.
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  C8_EXTRA_FLAGS="$f" python3 -m calibr8_amd.build > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  for rep in 1 2; do
    timeout -k 10 200 python3 bench.py --no-cpu --steps 10 --warmup 3 $BENCH_ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-60s step %.3f ms  kernels %.3f ms  assign %.3f ms' % ('$f', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['also'].get('ms_per_step_assign_mode', float('nan'))))"
  done
done
