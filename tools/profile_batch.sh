#!/bin/bash
# closing profile batch of a round (runs on the GPU box from the repo root): tools/profile_batch.sh <tag> [<profiles prefix>]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-batch}
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_atomic -- python3 bench.py --no-cpu --scatter atomic > $O/prof_atomic.log 2>&1 || exit 1
echo prof_atomic done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gather -- python3 bench.py --no-cpu --scatter gather > $O/prof_gather.log 2>&1 || exit 1
echo prof_gather done
timeout -k 10 900 python3 tools/collect_pmc.py --out $O/pmc_atomic.json -- --scatter atomic > $O/pmc_atomic.log 2>&1 || exit 1
timeout -k 10 900 python3 tools/collect_pmc.py --out $O/pmc_gather.json -- --scatter gather > $O/pmc_gather.log 2>&1 || exit 1
echo pmc done
# the bench line takes roofline.traffic / roofline.valu from a PMC profile of THIS build under profiles/: put the one just
# collected there (on the box; tools/save_profiles.py does the same in the repository) before the bench runs
if [ -n "$2" ]; then cp $O/pmc_gather.json profiles/$2_traffic_gather_100cube.json; fi
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo bench default done
timeout -k 10 200 python3 bench.py --no-cpu --scatter atomic > $O/bench_atomic.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --no-cpu --scatter colored > $O/bench_colored.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --no-cpu --workload notch > $O/bench_notch.json 2>/dev/null || exit 1
echo bench modes done
timeout -k 10 300 python3 tools/bench_kernels.py > $O/kernels_small_J2.json 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/bench_kernels.py --model hyper_J2 > $O/kernels_hyper_J2.json 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/bench_kernels.py --model small_hill > $O/kernels_small_hill.json 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/bench_kernels.py --model hypo_hill > $O/kernels_hypo_hill.json 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/bench_kernels.py --tet --edge 56 --scatter gather > $O/kernels_tet4_gather.json 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/time_k1.py --kernels auto,wave,wave_ad > $O/k1_kernels.log 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/time_k1.py --kernels auto,wave --adjoint > $O/k3_kernels.log 2>/dev/null || exit 1
# the same in the library's default mode (staged assembly + row sums) for the four hex8 models
for m in small_J2 hyper_J2 small_hill hypo_hill; do
  timeout -k 10 300 python3 tools/bench_kernels.py --model $m --scatter gather > $O/kernels_${m}_gather.json 2>/dev/null || exit 1
done
for m in hypo_barlat small_hosford hypo_hosford; do  # the line-search models (wave-per-element kernels on hex8), library default mode
  timeout -k 10 300 python3 tools/bench_kernels.py --model $m --scatter default --reps 3 > $O/kernels_${m}_default.json 2>/dev/null || exit 1
done
echo kernels done
# the vector-memory pipeline of the row-per-node kernels (address unit / L1 busy and stall cycles)
timeout -k 10 600 python3 tools/pmc_k1.py --match node_rows --groups 9,10,11,12,8,3 --out $O/pmc_node_ta.json -- --kernel auto > $O/pmc_node_ta.log 2>&1 || exit 1
echo ta counters done
timeout -k 10 400 python3 tools/bench_fractions.py > $O/fractions.json 2>/dev/null || exit 1
echo all done
