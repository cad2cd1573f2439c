"""Collect rocprofv3 PMC counters for the bench workload, one counter group per pass (MI355X_MICROARCH.md: counters in
their own runs, never together with sys/runtime/hip traces), and write per-launch means per kernel as JSON.

Runs ON THE GPU BOX from the repo root:   python3 tools/collect_pmc.py --out gpurun_out/pmc_atomic.json -- --scatter atomic
(everything after `--` goes to bench.py).  rocprofv3 is started as a child with the program directly after its `--`."""
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
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"],
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--kernels", default="c8::", help="substring of the kernel names to keep")
    ap.add_argument("bench_args", nargs="*")
    args = ap.parse_args()
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    for grp in GROUPS:
        d = os.path.join(ROOT, "gpurun_out", "pmc_tmp")
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc"] + grp + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
                                             sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--steps", "2", "--warmup", "1"] + args.bench_args
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit("rocprofv3 failed for %s" % grp)
        print("pass", grp, "done", flush=True)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                if args.kernels not in name:
                    continue
                short = name.split("(")[0].replace("void ", "")
                e = res.setdefault(short, {}).setdefault(row["Counter_Name"], [0.0, 0])
                e[0] += float(row["Counter_Value"])
                e[1] += 1
        shutil.rmtree(d, ignore_errors=True)
    out = {"bench_args": args.bench_args, "per_launch_mean": {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in res.items()},
           "launches_seen": {k: max(v[1] for v in cs.values()) for k, cs in res.items()}}
    for k, cs in out["per_launch_mean"].items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            # FETCH_SIZE/WRITE_SIZE are in KB; gfx950 reports half of wide reads (MI355X_MICROARCH.md): doubled = upper bound
            cs["hbm_bytes_corrected"] = 1024.0 * (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"])
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out["per_launch_mean"], indent=1))


if __name__ == "__main__":
    main()
