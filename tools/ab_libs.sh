#!/bin/bash
# same-call A/B of two builds of libc8.so: tools/_ab/before.so against tools/_ab/after.so (bench workload, two runs each, twice)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in after before; do
    cp tools/_ab/$v.so calibr8_amd/libc8.so
    for rep in 1 2; do
      timeout -k 10 200 python3 bench.py --no-cpu --steps 20 --warmup 3 $BENCH_ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-8s step %.3f ms  kernels %.3f ms  assign %.3f ms  ad-form %.3f' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['also'].get('ms_per_step_assign_mode', float('nan')), d['also'].get('ms_per_step_iterated_ad_form', float('nan'))))"
    done
  done
done
cp tools/_ab/after.so calibr8_amd/libc8.so
