"""Times c8_assemble_forward_jacobian on the bench workload (100^3 hex8 brick, small_J2, ramped state) for a list of kernel
variants: median / min of HIP-event times over --reps calls.  GPU box.  tools/ab_k1.sh runs it once per library build."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", type=int, default=100)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--kernels", default="auto,wave")
    ap.add_argument("--tag", default="")
    ap.add_argument("--assign", action="store_true")
    ap.add_argument("--eps", type=float, default=0.004)
    ap.add_argument("--adjoint", action="store_true", help="time c8_assemble_adjoint_jacobian (K3) instead of the forward assembly")
    a = ap.parse_args()
    import torch
    from calibr8_amd import Assembler
    from calibr8_amd.assembly import brick_mesh
    from meshes import prescribed_fields
    n = a.edge
    c, conn = brick_mesh(n, n, n)
    asm = Assembler(8, c, conn, "small_J2", [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0], scatter="gather")
    u_h, p_h = prescribed_fields(c, a.eps, ramp=True)
    u, p = asm.dev(u_h), asm.dev(p_h)
    u0, p0 = torch.zeros_like(u), torch.zeros_like(p)
    xi_prev, xi = asm.new_state(), asm.new_state()
    ls = asm.new_linsys()
    asm.set_async(True)
    if a.assign:
        asm.set_assign_mode(True)
    asm.forward_jacobian(u, p, u0, p0, xi_prev, xi, ls)  # the converged state of the step
    g_h = torch.zeros(asm.nelems, asm.npts, asm.nloc, dtype=torch.float64, device=u.device)
    f_h = torch.zeros(asm.nelems, asm.npts, asm.ndofs, dtype=torch.float64, device=u.device)
    call = (lambda: asm.adjoint_jacobian(u, p, u0, p0, xi_prev, xi, g_h, f_h, ls)) if a.adjoint else \
        (lambda: asm.forward_jacobian(u, p, u0, p0, xi_prev, xi, ls))
    for k in a.kernels.split(","):
        asm.set_kernel(k)
        for _ in range(3):
            call()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
        for s, e in ev:
            s.record()
            call()
            e.record()
        torch.cuda.synchronize()
        assert asm.status() == 0
        t = np.array([s.elapsed_time(e) for s, e in ev])
        print("%-14s %-8s %s median %.3f ms  min %.3f  max %.3f   (%.1f M elements/s)" % (a.tag, k, "K3" if a.adjoint else "K1", np.median(t), t.min(), t.max(), len(conn) / np.median(t) / 1e3), flush=True)


if __name__ == "__main__":
    main()
