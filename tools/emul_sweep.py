"""Random parity sweep on the CPU: the kernel source run by the lane emulator (tests/emul) against the oracle, every
case of tests/test_gpu_fuzz.random_case.  Prints the worst relative deviation per failing seed.

    python tools/emul_sweep.py [--seeds 0:1000] [--model hyper_J2] [--tol 1e-12] [--jobs 8]

TEST TOOL: it drives the oracle and the emulator; nothing in calibr8_amd/ is involved."""
import argparse
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(args):
    seed, tol, only = args
    import emul_lib as em
    import oracle_lib as ol
    from parity_cases import check_adjoint_chain, check_forward, check_residual
    from test_gpu_fuzz import random_case
    model, params, kind, c, conn, eps, scatter, kernel = random_case(seed)
    if only and model != only:
        return seed, model, None, None
    et = ol.HEX8 if kind == "hex8" else ol.TET4
    orc = ol.Oracle(et, c, conn, model, params)
    dut = em.Emul(et, c, conn, model, params)
    dut.wave = kind == "hex8" and kernel in ("auto", "wave", "wave_ad", "node")
    dut.closed = kernel != "wave_ad"
    dut.node = kernel == "node" or (kernel == "auto" and kind == "hex8" and model == "small_J2" and scatter == "gather")
    dut.staged = scatter == "gather"
    from parity_cases import AUDIT
    AUDIT.ctx = "emul %s %s %s" % (kind, kernel, scatter)
    try:
        check_forward(orc, dut, c, model, eps, tol)
        check_residual(orc, dut, c, eps, tol)
        if not (kind == "hex8" and not dut.wave and dut.staged):
            check_adjoint_chain(orc, dut, c, model, eps, tol)
    except AssertionError as e:
        return seed, model, "%s %s %s: %s" % (kind, scatter, kernel, str(e)[:300]), dict(AUDIT.table(), pid=os.getpid())
    return seed, model, "", dict(AUDIT.table(), pid=os.getpid())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:200")
    ap.add_argument("--model", default="")
    ap.add_argument("--tol", type=float, default=1e-12)
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--audit", default="", help="write the checker's allowance table (parity_cases.AUDIT) here")
    a = ap.parse_args()
    lo, hi = (int(v) for v in a.seeds.split(":"))
    import emul_lib as em
    em.lib()  # build once, before the workers start
    nrun = nfail = 0
    from collections import Counter
    cases, used = Counter(), Counter()
    with ProcessPoolExecutor(a.jobs) as ex:
        # (a worker's audit table accumulates over the cases it has run: keep the last table of every worker process)
        last = {}
        for seed, model, msg, table in ex.map(run, [(s, a.tol, a.model) for s in range(lo, hi)], chunksize=4):
            if msg is None:
                continue
            last[table.pop("pid")] = table
            nrun += 1
            if msg:
                nfail += 1
                print("seed %d %s FAILED %s" % (seed, model, msg), flush=True)
    for t in last.values():
        cases.update(t["cases"])
        used.update(t["allowances_used"])
    print("%d cases run, %d failed at tol %g" % (nrun, nfail, a.tol))
    print("allowances used:", dict(used) if used else "none")
    if a.audit:
        import json
        json.dump({"seeds": a.seeds, "cases": dict(cases), "allowances_used": dict(used), "failed": nfail}, open(a.audit, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
