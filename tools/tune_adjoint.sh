#!/bin/bash
# same-call A/B of flag sets on the per-entry-point timings of one model in the default (staged) mode:
# MODEL=small_J2 tools/tune_adjoint.sh "" "-DC8_TUNE_..."
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  C8_EXTRA_FLAGS="$f" python3 -m calibr8_amd.build > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  for rep in 1 2; do
    timeout -k 10 300 python3 tools/bench_kernels.py --model ${MODEL:-small_J2} --scatter gather 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())['ms']
print('%-36s' % '$f', {k.replace('_wave', ''): round(v, 2) for k, v in d.items()})"
  done
done
