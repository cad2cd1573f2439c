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
                                             sys.executable, os.path.join(ROOT, "bench.py"), "--headline-only", "--steps", "3", "--warmup", "1"] + args.bench_args
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
    # what bench.py looks up: the configuration, the library build the counters belong to, and the HBM bytes of one
    # assembly = the launches of one c8_assemble_forward_jacobian call
    sys.path.insert(0, ROOT)
    from calibr8_amd import lib as c8lib
    info = c8lib.load_library().c8_build_info().decode()
    ba = args.bench_args
    opt = lambda name, default: ba[ba.index(name) + 1] if name in ba else default
    scatter, kernel = opt("--scatter", "gather"), opt("--kernel", "auto")
    kkey = kernel if kernel in ("slot", "wave_ad") else ("node" if kernel in ("auto", "node") and scatter == "gather" else "wave")
    out["config"] = {"edge": int(opt("--edge", 100)), "scatter": scatter, "kernel": kkey}
    out["build_id"] = info.split()[0].split("=")[1]
    out["library_build"] = info
    # bench.py also times the iterated form of the kernel beside the headline: keep the kernel of the configuration named
    fwd = [k for k in out["per_launch_mean"] if "k_forward_jacobian" in k and "SmallJ2" in k]
    fwd = [k for k in fwd if ("_closed" in k) == (kkey == "wave")] or fwd
    rows = [k for k in out["per_launch_mean"] if "k_gather_rows" in k] if scatter == "gather" else []
    if kkey == "node":  # one kernel forms and writes the rows: no stage, no row-sum launch
        fwd, rows = [k for k in out["per_launch_mean"] if "k_node_rows_closed" in k], []
    out["launches_per_assembly"] = fwd[:1] + rows[:1]
    ncol = {"colored": 8}.get(scatter, 1)  # per_launch_mean of a colour-batched assembly is the mean over its launches
    if all("hbm_bytes_corrected" in out["per_launch_mean"][k] for k in out["launches_per_assembly"]) and out["launches_per_assembly"]:
        out["traffic_bytes_per_launch"] = sum(out["per_launch_mean"][k]["hbm_bytes_corrected"] for k in out["launches_per_assembly"]) * ncol
    out["note"] = ("rocprofv3 --pmc passes (one counter group per pass, kernel-trace only) on `python3 bench.py --headline-only --steps 3 "
                   "--warmup 1 %s` (the timed configuration alone: no `also` timings in the profiled process); traffic_bytes_per_launch = HBM-side bytes per ASSEMBLY = sum over the assembly's kernel "
                   "launches of 1024*(2*FETCH_SIZE + WRITE_SIZE): FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for "
                   "gfx950 (an upper bound for our gathers), WRITE_SIZE exact for 16-B stores" % " ".join(ba))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out["per_launch_mean"], indent=1))


if __name__ == "__main__":
    main()
