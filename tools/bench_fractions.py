"""SURVEY.md 8d asks for the K1 throughput at three plastic fractions (Newton iteration counts differ) as the
median of >= 20 timed calls after 3 warm-ups: all elastic (eps 0.001), ramp (~50 % of points plastic, the bench
workload) and all plastic (eps 0.004), in each scatter mode.  100^3 hex8 brick, small_J2, one MI355X."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
J2 = [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", type=int, default=100)
    ap.add_argument("--calls", type=int, default=20)
    args = ap.parse_args()
    import torch
    from calibr8_amd import Assembler, brick_mesh
    from meshes import prescribed_fields
    n = args.edge
    coords, conn = brick_mesh(n, n, n)
    states = {"all_elastic_eps0.001": dict(eps_bar=0.001, ramp=False), "ramp_eps0.004": dict(eps_bar=0.004, ramp=True),
              "all_plastic_eps0.004": dict(eps_bar=0.004, ramp=False)}
    out = {"elements": len(conn), "calls": args.calls, "warmup": 3, "results": {}}
    for scatter in ("atomic", "gather", "colored"):
        asm = Assembler(8, coords, conn, "small_J2", J2, scatter=scatter)
        asm.set_async(True)
        ls, xi0, xi = asm.new_linsys(), asm.new_state(), asm.new_state()
        for name, kw in states.items():
            u_h, p_h = prescribed_fields(coords, kw["eps_bar"], ramp=kw["ramp"])
            u, p = asm.dev(u_h), asm.dev(p_h)
            u0, p0 = torch.zeros_like(u), torch.zeros_like(p)
            for _ in range(3):
                asm.forward_jacobian(u, p, u0, p0, xi0, xi, ls)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.calls)]
            for a, b in ev:
                a.record()
                asm.forward_jacobian(u, p, u0, p0, xi0, xi, ls)
                b.record()
            torch.cuda.synchronize()
            assert asm.status() == 0
            ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
            out["results"]["%s/%s" % (scatter, name)] = {
                "median_ms": ms, "Melem_per_s": len(conn) / ms / 1e3,
                "plastic_point_fraction": float((xi[:, :, 6] > 0).double().mean().item())}
        del asm
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
