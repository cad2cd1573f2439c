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
        return seed, model, None
    et = ol.HEX8 if kind == "hex8" else ol.TET4
    orc = ol.Oracle(et, c, conn, model, params)
    dut = em.Emul(et, c, conn, model, params)
    dut.wave = kind == "hex8" and kernel == "auto"
    dut.staged = scatter == "gather"
    try:
        check_forward(orc, dut, c, model, eps, tol)
        check_residual(orc, dut, c, eps, tol)
        if not (kind == "hex8" and not dut.wave and dut.staged):
            check_adjoint_chain(orc, dut, c, model, eps, tol)
    except AssertionError as e:
        return seed, model, "%s %s wave=%s: %s" % (kind, scatter, dut.wave, str(e)[:300])
    return seed, model, ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:200")
    ap.add_argument("--model", default="")
    ap.add_argument("--tol", type=float, default=1e-12)
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    lo, hi = (int(v) for v in a.seeds.split(":"))
    import emul_lib as em
    em.lib()  # build once, before the workers start
    nrun = nfail = 0
    with ProcessPoolExecutor(a.jobs) as ex:
        for seed, model, msg in ex.map(run, [(s, a.tol, a.model) for s in range(lo, hi)], chunksize=4):
            if msg is None:
                continue
            nrun += 1
            if msg:
                nfail += 1
                print("seed %d %s FAILED %s" % (seed, model, msg), flush=True)
    print("%d cases run, %d failed at tol %g" % (nrun, nfail, a.tol))


if __name__ == "__main__":
    main()
