"""rocprofv3 PMC passes (one counter group per pass, kernel-trace only) over tools/time_k1.py: per-launch means of the
counters of every kernel whose name contains --match.  GPU box:  python3 tools/pmc_k1.py --match node_rows -- --kernels auto"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["FETCH_SIZE"], ["WRITE_SIZE"],
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"],
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    ["TCC_HIT_sum", "TCC_MISS_sum"], ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"], ["GRBM_GUI_ACTIVE"],
    # 9-12: the vector memory pipeline (address unit, L1): busy and stall cycles summed over the CUs
    ["TA_TA_BUSY_sum", "TA_BUSY_avr", "TA_FLAT_WAVEFRONTS_sum"], ["TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TOTAL_ACCESSES_sum"],
    ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"],
    ["TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TD_TD_BUSY_sum", "TCP_TCC_READ_REQ_LATENCY_sum"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--match", default="c8::")
    ap.add_argument("--out", default="")
    ap.add_argument("--groups", default="", help="comma-separated indices of the counter groups to run (default all)")
    ap.add_argument("args", nargs="*")
    a = ap.parse_args()
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    groups = [GROUPS[int(i)] for i in a.groups.split(",")] if a.groups else GROUPS
    for grp in groups:
        d = os.path.join(ROOT, "gpurun_out", "pmc_tmp")
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc"] + grp + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
                                             sys.executable, os.path.join(ROOT, "tools", "time_k1.py"), "--reps", "3"] + a.args
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print("pass", grp, "FAILED", r.stderr[-600:], flush=True)
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                if a.match not in name:
                    continue
                short = name.split("(")[0].replace("void ", "")[:90]
                e = res.setdefault(short, {}).setdefault(row["Counter_Name"], [0.0, 0])
                e[0] += float(row["Counter_Value"])
                e[1] += 1
        shutil.rmtree(d, ignore_errors=True)
    out = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in res.items()}
    for k, cs in out.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            cs["hbm_GB_fetch_doubled"] = 1024.0 * (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) / 1e9
            cs["hbm_GB_fetch_as_reported"] = 1024.0 * (cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) / 1e9
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
