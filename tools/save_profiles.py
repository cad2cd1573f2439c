"""Copy the summaries of one `tools/profile_batch.sh <tag>` run from gpurun_out/<tag>/ (scratch) into profiles/ (tracked),
named per round: python tools/save_profiles.py r02a r02_a"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = os.path.join(ROOT, "gpurun_out", sys.argv[1]), sys.argv[2]
dst = os.path.join(ROOT, "profiles")


def put(name, out):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, out)))
        print("saved", out)


for mode in ("atomic", "gather"):
    for f in glob.glob(os.path.join(src, "prof_" + mode, "*", "*kernel_stats.csv")):
        rows = list(csv.reader(open(f)))
        keep = [rows[0]] + [r for r in rows[1:] if "c8::" in r[0]]
        csv.writer(open(os.path.join(dst, "%s_kernel_stats_wave_%s_100cube.csv" % (tag, mode)), "w")).writerows(keep)
        print("saved kernel stats", mode)
    put("pmc_%s.json" % mode, "traffic_%s_100cube.json" % mode)
put("bench_default.json", "bench_default_wave_gather_100cube.json")
put("bench_atomic.json", "bench_wave_atomic_100cube.json")
put("bench_colored.json", "bench_wave_colored_100cube.json")
put("bench_notch.json", "bench_notch_specimen_1M.json")
put("fractions.json", "plastic_fractions_100cube.json")
for m in ("small_J2", "hyper_J2", "small_hill", "hypo_hill"):
    put("kernels_%s.json" % m, "all_kernels_%s_100cube.json" % m)
    put("kernels_%s_gather.json" % m, "all_kernels_%s_100cube_gather.json" % m)
put("kernels_tet4_gather.json", "all_kernels_tet4_1M_gather.json")
for m in ("hypo_barlat", "small_hosford", "hypo_hosford"):
    put("kernels_%s_default.json" % m, "all_kernels_%s_100cube_default.json" % m)
put("pmc_node_ta.json", "pmc_node_ta.json")
put("k1_kernels.log", "forward_kernels_100cube.log")
put("k3_kernels.log", "adjoint_kernels_100cube.log")
put("bench_n2.json", "bench_n2_rehearsal_48cube_host_transport.json")
put("bench_n4.json", "bench_n4_rehearsal_48cube_host_transport.json")
put("gpu_tests.log", "gpu_tests.log")
