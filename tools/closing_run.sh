#!/bin/bash
# closing run of a round on the GPU box: tools/closing_run.sh <tag> <profiles prefix>
# GPU test suite, smoke, the profile batch, and rehearsals of bench.py --gpus 2 / 4 through the host transport
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-closing}
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -5 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
bash tools/profile_batch.sh $1 $2 || exit 1
P=$((20000 + RANDOM % 20000))
# rehearsals of the N > 1 line (strong scaling as value, weak under also.weak) with all ranks on this one card: host transport
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --edge 48 --steps 10 --warmup 2 > $O/bench_n2.json 2> $O/bench_n2.err || exit 1
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((P + 1)) bench.py --gpus 4 --edge 48 --steps 10 --warmup 2 > $O/bench_n4.json 2> $O/bench_n4.err || exit 1
echo rehearsals done
