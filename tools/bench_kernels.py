"""Times every entry point of the C ABI on the bench workload (100^3 hex8 brick, small_J2).
Not the contract bench (that is ../bench.py): a per-kernel table for DESIGN.md / profiles."""
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
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--scatter", default="atomic", help="atomic | colored | gather | default (the library's choice per entry point)")
    ap.add_argument("--model", default="small_J2", choices=["small_J2", "hyper_J2", "small_hill", "elastic", "hypo_hill", "small_hosford", "hypo_hosford",
                                                            "hypo_barlat", "small_hill_plane_strain", "hyper_J2_plane_strain", "hypo_hill_plane_strain",
                                                            "small_hill_plane_stress", "hyper_J2_plane_stress", "hypo_hill_plane_stress"])
    ap.add_argument("--eps", type=float, default=None, help="strain of the prescribed state (default 0.004; hypo_barlat 0.006; the Hosford models 0.002: one step to 0.004 is beyond their local Newton iteration)")
    ap.add_argument("--tet", action="store_true", help="split every hex into 6 tet4 (the reference's element type)")
    ap.add_argument("--tri", action="store_true", help="a tri3 mesh of 2 * (7 * edge)^2 elements (the reference's 2-D element type; 2-D models)")
    args = ap.parse_args()
    import torch
    from calibr8_amd import Assembler, brick_mesh
    from meshes import prescribed_fields
    n = args.edge
    coords, conn = brick_mesh(n, n, n)
    et = 8
    if args.tet:
        # Kuhn split of each hex into 6 positively oriented tets sharing the 0-6 diagonal
        tets = [[0, 1, 2, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 7, 4, 6], [0, 4, 5, 6], [0, 5, 1, 6]]
        conn = np.concatenate([conn[:, t] for t in tets]).astype(np.int32)
        X = coords[conn]
        vol = np.einsum("ij,ij->i", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0])
        assert (vol > 0).all()
        et = 4
    from parity_cases import ACTIVE, BARLAT, EL, HILL, HILL_PS, HJ2, HJ2_PS, HJ2_PSS, HOSFORD, HYPO_PSS, LOCAL_LINE_SEARCH
    if args.tri:
        from meshes import tri_mesh
        coords, conn, _ = tri_mesh(7 * n, 7 * n, 1.0, 1.0)
        et = 3
    params = {"small_J2": J2, "hyper_J2": HJ2, "small_hill": HILL, "elastic": EL, "hypo_hill": HILL, "small_hosford": HOSFORD, "hypo_hosford": HOSFORD,
              "hypo_barlat": BARLAT, "small_hill_plane_strain": HILL_PS, "hyper_J2_plane_strain": HJ2_PS, "hypo_hill_plane_strain": HILL_PS,
              "small_hill_plane_stress": HILL_PS, "hyper_J2_plane_stress": HJ2_PSS, "hypo_hill_plane_stress": HYPO_PSS}[args.model]
    line_search = args.model in ("small_hosford", "hypo_hosford", "hypo_barlat")
    asm = Assembler(et, coords, conn, args.model, params, scatter=None if args.scatter == "default" else args.scatter,
                    **({"line_search": LOCAL_LINE_SEARCH} if line_search else {}))
    asm.set_active(0, ACTIVE[args.model][:4])
    asm.set_async(True)
    u_h, p_h = prescribed_fields(coords, args.eps if args.eps else {"hypo_barlat": 0.006, "small_hosford": 0.002, "hypo_hosford": 0.002}.get(args.model, 0.004), ramp=True)
    if args.tri:
        from meshes import fields_for
        u_h, p_h = fields_for(2, u_h, p_h)
    u, p = asm.dev(u_h), asm.dev(p_h)
    u0, p0 = torch.zeros_like(u), torch.zeros_like(p)
    xi0, xi = asm.new_state(), asm.new_state()
    ls = asm.new_linsys()
    g = torch.zeros(asm.nelems, asm.npts, asm.nloc, dtype=torch.float64, device=asm.device)
    f = torch.zeros(asm.nelems, asm.npts, asm.ndofs, dtype=torch.float64, device=asm.device)
    phi = torch.zeros_like(g)
    z_u, z_p = torch.randn_like(u) * 1e-3, torch.randn_like(p) * 1e-3
    grad = torch.zeros(len(ACTIVE[args.model][:4]), dtype=torch.float64, device=asm.device)
    J = torch.zeros(1, dtype=torch.float64, device=asm.device)

    def timeit(fn):
        fn()
        assert asm.status() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        assert asm.status() == 0
        return a.elapsed_time(b) / args.reps

    res = {}
    # staged (gather) mode: the hex8 lane-group adjoint kernel cannot stage, time the wave-per-element kernels only
    variants = ("slot",) if (args.tet or args.tri) else (("wave",) if args.scatter in ("gather", "default") else ("wave", "slot"))
    for k in variants:
        asm.set_kernel(k)
        res["forward_jacobian_" + k] = timeit(lambda: asm.forward_jacobian(u, p, u0, p0, xi0, xi, ls))
    asm.set_kernel("auto")  # the library's choice (a model's closed form where it has one)
    res["forward_jacobian_auto"] = timeit(lambda: asm.forward_jacobian(u, p, u0, p0, xi0, xi, ls))
    for k in variants:
        asm.set_kernel(k)
        res["adjoint_jacobian_" + k] = timeit(lambda: asm.adjoint_jacobian(u, p, u0, p0, xi0, xi, g, f, ls))
    for k in variants:
        asm.set_kernel(k)
        res["solve_adjoint_local_" + k] = timeit(lambda: asm.solve_adjoint_local(u, p, u0, p0, xi0, xi, z_u, z_p, phi, g, f))
        res["param_gradient_" + k] = timeit(lambda: asm.qoi_gradient(u, p, u0, p0, xi0, xi, z_u, z_p, phi, grad))
    asm.set_kernel("auto")
    res["residual"] = timeit(lambda: asm.global_residual(u, p, u0, p0, xi0, xi, ls))
    res["eval_qoi"] = timeit(lambda: asm.eval_qoi(u, p, J))
    out = {"elements": asm.nelems, "element_type": "tri3" if args.tri else ("tet4" if args.tet else "hex8"), "model": args.model, "scatter": args.scatter,
           "ms": res, "Melem_per_s": {k: asm.nelems / v / 1e3 for k, v in res.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
