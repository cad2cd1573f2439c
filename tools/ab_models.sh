#!/bin/bash
# same-call A/B of the builds in tools/_ab/*.so on the per-entry-point timings of several models (tools/bench_kernels.py, staged mode)
# usage (GPU box): bash tools/ab_models.sh "hyper_J2 small_hill hypo_hill"
cd $GRAFT_REPO_ROOT
cp calibr8_amd/libc8.so /tmp/libc8_keep.so
for f in tools/_ab/*.so; do
  cp $f calibr8_amd/libc8.so
  for m in $1; do
    timeout -k 10 300 python3 tools/bench_kernels.py --model $m --scatter gather --reps 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())['ms']
print('%-10s %-12s ' % ('$(basename $f .so)', '$m') + '  '.join('%s %.2f' % (k.replace('forward_jacobian', 'K1').replace('adjoint_jacobian', 'K3').replace('solve_adjoint_local', 'K4').replace('param_gradient', 'K5'), v) for k, v in d.items()))"
  done
done
cp /tmp/libc8_keep.so calibr8_amd/libc8.so
