"""`-m gpu`: the 2-D path on the device (tri3 elements, `mechanics` with 2 + 1 equations per node; SURVEY.md section 8
f4): every entry point against the oracle at 1e-12, and the reference's two `mechanics` decks on the notch2D mesh end to
end on the GPU (HIP assembly, device boundary conditions, the C++ Newton driver)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from parity_cases import CASES_2D, CASES_PLANE_STRESS, check_adjoint_chain, check_forward, check_residual, mesh_2d

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("scatter", ["colored", "atomic"])
@pytest.mark.parametrize("mesh", ["structured", "notch2D"])
@pytest.mark.parametrize("model,params,eps", CASES_2D)
def test_2d_kernels_match_oracle(model, params, eps, mesh, scatter):
    from gpu_backend import GpuBackend
    et, c, conn = mesh_2d(mesh)
    orc = ol.Oracle(et, c, conn, model, params)
    gpu = GpuBackend(et, c, conn, model, params, scatter=scatter)
    assert gpu.ndims == 2 and gpu.nloc in (4, 5)
    check_forward(orc, gpu, c, model, eps, 1e-12)
    check_residual(orc, gpu, c, eps, 1e-12)
    check_adjoint_chain(orc, gpu, c, model, eps, 1e-12)


@pytest.mark.parametrize("scatter", ["colored", "atomic"])
@pytest.mark.parametrize("mesh", ["structured", "notch2D"])
@pytest.mark.parametrize("model,params,eps", CASES_PLANE_STRESS)
def test_plane_stress_kernels_match_oracle(model, params, eps, mesh, scatter):
    # `mechanics_plane_stress` (one global residual): Tri3PlaneStress lane groups of six, every entry point at 1e-12;
    # the p arrays and the three other blocks of the test containers must stay untouched (compared with the oracle's zeros)
    from gpu_backend import GpuBackend
    et, c, conn = mesh_2d(mesh)
    orc = ol.Oracle(et, c, conn, model, params)
    gpu = GpuBackend(et, c, conn, model, params, scatter=scatter)
    assert gpu.asm.nres == 1 and gpu.ndims == 2 and gpu.asm.ndofs == 6 and gpu.nloc == orc.nloc
    check_forward(orc, gpu, c, model, eps, 1e-12)
    check_residual(orc, gpu, c, eps, 1e-12)
    check_adjoint_chain(orc, gpu, c, model, eps, 1e-12)


def test_plane_stress_abi_ignores_the_second_residual_and_refuses_mismatched_pairs():
    import ctypes as C
    import torch
    from calibr8_amd import Assembler, lib as _l
    from calibr8_amd.lib import C8Error
    from parity_cases import HILL_PS, two_steps
    et, c, conn = mesh_2d("structured")
    asm = Assembler(3, c, conn, "small_hill_plane_stress", HILL_PS, thickness=0.7)
    assert asm.nres == 1 and asm.scatter == "colored"
    with pytest.raises(C8Error):  # the pairing is fixed: plane-stress models under mechanics_plane_stress only ...
        Assembler(3, c, conn, "small_hill_plane_stress", HILL_PS, global_type="mechanics")
    with pytest.raises(C8Error):  # ... and only they
        Assembler(3, c, conn, "small_hill_plane_strain", HILL_PS, global_type="mechanics_plane_stress")
    with pytest.raises(C8Error):  # six lanes per element group: at most six active parameters
        asm.set_active(0, [0, 1, 2, 3, 4, 5, 6])
    # NULL in every [1] entry of state and system; thickness scales A and b
    orc = ol.Oracle(et, c, conn, "small_hill_plane_stress", HILL_PS)
    orc.set_thickness(0.7)
    (u, p, xi), (up, pp, xip) = two_steps(orc, c, 0.004)[1], two_steps(orc, c, 0.004)[0]
    ls_o, xo = orc.new_linsys(), orc.new_state()
    assert orc.forward_jacobian(u, p, up, pp, xip, xo, ls_o) == 0
    d = lambda a: asm.dev(a)
    du, dup, dxip, dxi = d(u), d(up), d(xip), asm.new_state()
    A00 = torch.zeros(asm.nnz[0][0], dtype=torch.float64, device=asm.device)
    b0 = torch.zeros(asm.nnodes * 2, dtype=torch.float64, device=asm.device)
    st, sy = _l.State(), _l.System()
    st.x[0], st.x_prev[0], st.xi_prev, st.xi = du.data_ptr(), dup.data_ptr(), dxip.data_ptr(), dxi.data_ptr()
    sy.A[0][0], sy.b[0] = A00.data_ptr(), b0.data_ptr()
    assert asm.L.c8_assemble_forward_jacobian(asm.h, C.byref(st), C.byref(sy)) == 0
    torch.cuda.synchronize()
    from parity import rel_csr_rows, rel_vec
    assert rel_vec(b0.cpu().numpy(), ls_o.b[0]) < 1e-12
    assert rel_csr_rows(A00.cpu().numpy(), ls_o.A[0][0], orc.rowptr[0][0]) < 1e-12
    # a Dirichlet condition on residual 1 does not exist here
    with pytest.raises(C8Error):
        nodes = torch.zeros(1, dtype=torch.int32, device=asm.device)
        asm.apply_dirichlet([(1, 0, nodes, d(np.zeros(1)))], du, du, asm.new_linsys())


def test_2d_default_mode_and_refusals():
    from calibr8_amd import Assembler
    from calibr8_amd.lib import C8Error
    et, c, conn = mesh_2d("structured")
    asm = Assembler(3, c, conn, "small_J2", [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0])
    assert asm.scatter == "colored" and asm.ndims == 2 and asm.ndofs == 9  # no staged assembly for 2 + 1 equations per node
    with pytest.raises(C8Error):
        asm.set_scatter("gather")
    with pytest.raises(C8Error):  # 3-D-only models are refused on a 2-D mesh
        Assembler(3, c, conn, "hyper_J2", [1000.0, 0.25, 2.0, 1.0, 5.0, 0.5, 0.5, 100.0])
    with pytest.raises(C8Error):  # and the plane-strain model on a 3-D mesh
        from meshes import brick
        c3, conn3, _ = brick(2, 2, 2)
        Assembler(8, c3, conn3, "small_hill_plane_strain", [1000.0, 0.25, 2.0, 10.0, 2.0, 1.0, 1.0, 1.0, 1.0])


def notch2d():
    d = json.load(open(os.path.join(HERE, "golden", "notch2D_tri3.json")))
    return np.array(d["coords"]), np.array(d["conn"], dtype=np.int32), {k: np.array(v, dtype=np.int32) for k, v in d["node_sets"].items()}


@pytest.mark.parametrize("deck", ["notch2D_small_J2", "notch2D_small_J2_plane_strain", "notch2D_hyper_J2_plane_strain", "notch2D_hypo_J2_plane_strain"])
def test_notch2D_regressions_with_device_newton_driver(deck):
    # test/primal/notch2D_small_J2.yaml.in (8 steps, Y 10) and notch2D_small_J2_plane_strain.yaml.in
    # (`small_hill_plane_strain`, 4 steps): assembly, Dirichlet rows, norms, updates and line search on the device
    from calibr8_amd import Assembler
    from calibr8_amd.primal import PrimalDriver
    c, conn, ns = notch2d()
    rate = 0.001
    if deck == "notch2D_small_J2":
        model, params, nsteps, iters, pin, tol = "small_J2", [1000.0, 0.25, 100.0, 10.0, 0.0, 0.0], 8, 15, 6.55208497250819866e-03, 2e-5
    elif deck == "notch2D_hypo_J2_plane_strain":
        model, params, nsteps, iters, pin, tol = "hypo_hill_plane_strain", [1000.0, 0.25, 2.0, 10.0, 2.0, 1.0, 1.0, 1.0, 1.0], 4, 30, 7.10226176768509899e-03, 1e-8
        rate = 0.005
    elif deck == "notch2D_hyper_J2_plane_strain":
        model, params, nsteps, iters, pin, tol = "hyper_J2_plane_strain", [1000.0, 0.25, 100.0, 10.0, 0.0, 0.0], 8, 15, 6.5626182813091150e-03, 1e-9
    else:
        model, params, nsteps, iters, pin, tol = "small_hill_plane_strain", [1000.0, 0.25, 2.0, 10.0, 2.0, 1.0, 1.0, 1.0, 1.0], 4, 30, 1.7664579853744898e-03, 1e-10
    asm = Assembler(3, c, conn, model, params)
    dbcs = [(0, 0, ns["xmin"], lambda x, y, z, t: 0.0), (0, 1, ns["ymin"], lambda x, y, z, t: 0.0),
            (0, 1, ns["ymax"], lambda x, y, z, t: rate * t)]
    drv = PrimalDriver(asm, dbcs, max_iters=iters, abs_tol=1e-8, rel_tol=1e-8).solve(nsteps)
    J = drv.qoi()
    assert abs(J - pin) / pin < tol, (deck, J, pin)
    # the oracle-driven host driver takes the same Newton iterations and lands on the same objective
    from fe_driver import Dbc, Primal
    orc = ol.Oracle(ol.TRI3, c, conn, model, params)
    pr = Primal(orc, c, [Dbc(r, e, n, f) for r, e, n, f in dbcs], max_iters=iters, abs_tol=1e-8, rel_tol=1e-8).solve(nsteps)
    assert drv.newton_iters == pr.newton_iters, (drv.newton_iters, pr.newton_iters)
    assert abs(J - pr.qoi()) < 1e-10 * abs(J)
    assert float(drv.xi[-1][:, :, -1].max()) > 1e-3


@pytest.mark.parametrize("deck", ["notch2D_small_J2_plane_stress", "notch2D_hyper_J2_plane_stress", "notch2D_hypo_J2_plane_stress"])
def test_notch2D_plane_stress_regressions_with_device_newton_driver(deck):
    # the reference's three `mechanics_plane_stress` decks end to end on the device: one-residual system, the C++ Newton
    # driver with the line search that restores the local state before each trial (primal.cpp:146-156; the
    # hyper_J2_plane_stress deck needs it: its local solves fail at the full step of the second Newton iteration)
    from calibr8_amd import Assembler
    from calibr8_amd.primal import PrimalDriver
    c, conn, ns = notch2d()
    if deck == "notch2D_small_J2_plane_stress":
        model, params, rate, nsteps, pin, tol = "small_hill_plane_stress", [1000.0, 0.25, 2.0, 10.0, 2.0, 1.0, 1.0, 1.0, 1.0], 0.001, 4, 2.2831790025047405e-03, 1e-9
    elif deck == "notch2D_hyper_J2_plane_stress":
        model, params, rate, nsteps, pin, tol = "hyper_J2_plane_stress", [1000.0, 0.25, 2.0, 10.0, 2.0, 0.0, 0.0, 0.0], 0.005, 5, 1.7493199283412385e-02, 1e-8
    else:
        model, params, rate, nsteps, pin, tol = "hypo_hill_plane_stress", [1000.0, 0.25, 2.0, 10.0, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 1.0], 0.005, 4, 1.1852379652063684e-02, 1e-8
    asm = Assembler(3, c, conn, model, params)
    dbcs = [(0, 0, ns["xmin"], lambda x, y, z, t: 0.0), (0, 1, ns["ymin"], lambda x, y, z, t: 0.0),
            (0, 1, ns["ymax"], lambda x, y, z, t: rate * t)]
    drv = PrimalDriver(asm, dbcs, max_iters=30, abs_tol=1e-8, rel_tol=1e-8).solve(nsteps)
    J = drv.qoi()
    assert abs(J - pin) / pin < tol, (deck, J, pin)
    from fe_driver import Dbc, Primal
    orc = ol.Oracle(ol.TRI3, c, conn, model, params)
    pr = Primal(orc, c, [Dbc(r, e, n, f) for r, e, n, f in dbcs], max_iters=30, abs_tol=1e-8, rel_tol=1e-8).solve(nsteps)
    assert drv.newton_iters == pr.newton_iters, (drv.newton_iters, pr.newton_iters)
    assert abs(J - pr.qoi()) < 1e-10 * abs(J)
    assert not bool(drv.p[-1].any())


def test_plane_stress_adjoint_gradient_on_device_passes_fd_check():
    # the adjoint chain of a one-residual system (K3 -> solve -> K4 -> K5 through c8_adjoint_solve_step) against central
    # differences of the device objective
    from calibr8_amd import Assembler
    from calibr8_amd.primal import PrimalDriver, adjoint_gradient
    c, conn, ns = notch2d()
    dbcs = [(0, 0, ns["xmin"], lambda x, y, z, t: 0.0), (0, 1, ns["ymin"], lambda x, y, z, t: 0.0),
            (0, 1, ns["ymax"], lambda x, y, z, t: 0.001 * t)]
    p0 = np.array([1000.0, 0.25, 2.0, 10.0, 2.0, 1.0, 1.0, 1.0, 1.0])
    act = [0, 1, 2, 3]
    span = np.array([100.0, 0.05, 1.0, 2.0])

    def solve(p):
        asm = Assembler(3, c, conn, "small_hill_plane_stress", list(p))
        asm.set_active(0, act)
        return PrimalDriver(asm, dbcs, max_iters=30, abs_tol=1e-12, rel_tol=1e-12).solve(3)

    gd = float(adjoint_gradient(solve(p0), 4) * span @ np.full(4, 0.1))
    errs = []
    for h in (1e-2, 1e-3, 1e-4):
        pp, pm = p0.copy(), p0.copy()
        pp[act] += h * 0.1 * span
        pm[act] -= h * 0.1 * span
        errs.append(abs((solve(pp).qoi() - solve(pm).qoi()) / (2 * h) - gd))
    assert min(errs) < 1e-6 * abs(gd), (errs, gd)


def test_notch2D_adjoint_gradient_on_device_passes_fd_check():
    # adjoint/notch2D_small_J2_adjoint_check.yaml.in with every step on the device: the adjoint gradient against
    # central differences of the device objective along the deck's direction (all four parameters)
    from calibr8_amd import Assembler
    from calibr8_amd.primal import PrimalDriver, adjoint_gradient
    c, conn, ns = notch2d()
    dbcs = [(0, 0, ns["xmin"], lambda x, y, z, t: 0.0), (0, 1, ns["ymin"], lambda x, y, z, t: 0.0),
            (0, 1, ns["ymax"], lambda x, y, z, t: 0.001 * t)]
    p0 = np.array([1000.0, 0.25, 100.0, 2.0, 0.0, 0.0])
    span = np.array([100.0, 0.05, 10.0, 1.0])

    def solve(p):
        asm = Assembler(3, c, conn, "small_J2", list(p))
        asm.set_active(0, [0, 1, 2, 3])
        return PrimalDriver(asm, dbcs, max_iters=15, abs_tol=1e-12, rel_tol=1e-12).solve(4)

    drv = solve(p0)
    gd = float(adjoint_gradient(drv, 4) * span @ np.full(4, 0.1))
    errs = []
    for h in (1e-2, 1e-3, 1e-4):
        pp, pm = p0.copy(), p0.copy()
        pp[:4] += h * 0.1 * span
        pm[:4] -= h * 0.1 * span
        errs.append(abs((solve(pp).qoi() - solve(pm).qoi()) / (2 * h) - gd))
    assert min(errs) < 1e-6 * abs(gd), (errs, gd)


@pytest.mark.parametrize("model,listed,thickness", [("small_J2", False, 0.0), ("small_hill_plane_stress", True, 0.7)])
def test_calibration_objective_2d_matches_oracle(model, listed, thickness):
    # Calibration's 2-D branch (calibration.cpp:76-104, :163-222) on the device: element integral of the displacement
    # mismatch (all elements / a listed subset), reaction load on the line y = ymin, under both 2-D global residuals
    from gpu_backend import GpuBackend
    from parity import compare_systems, rel_vec
    from parity_cases import HILL_PS, J2, two_steps
    et, c, conn = mesh_2d("structured")
    params = J2 if model == "small_J2" else HILL_PS
    elems = [e for e in range(len(conn)) if c[conn[e]].mean(axis=0)[0] > 0.4] if listed else None
    kw = dict(weights=(1.0, 2.0, 0.0), balance=0.3, coord_idx=1, coord_value=float(c[:, 1].min()), coord_tol=0.06, comp=1, dt_over_T=0.5)
    orc = ol.Oracle(et, c, conn, model, params)
    gpu = GpuBackend(et, c, conn, model, params, **({"thickness": thickness} if thickness else {}))
    if thickness:
        orc.set_thickness(thickness)
    orc.set_calibration(elems, **kw)
    gpu.set_calibration(elems, **kw)
    st = two_steps(orc, c, 0.004)
    (u, p, xi), (up, pp, xip) = st[2], st[1]
    u_meas = u + 1e-4 * np.random.default_rng(3).standard_normal(len(u))
    for b in (orc, gpu):
        b.set_active(0, [0, 1, 2, 3])
        b.set_measured(u_meas, -0.7)
    po, pd = orc.qoi_preprocess(u, p, up, pp, xip, xi), gpu.qoi_preprocess(u, p, up, pp, xip, xi)
    assert po[0] > 0 and abs(po[1]) > 1e-3 and np.abs(po - pd).max() < 1e-12 * max(1.0, np.abs(po).max()), (po, pd)
    Jo, Jd = orc.eval_qoi(u, p), gpu.eval_qoi(u, p)
    assert abs(Jo - Jd) < 1e-12 * abs(Jo), (Jo, Jd)
    res = []
    for b in (orc, gpu):
        g = np.full((orc.nelems, orc.npts, orc.nloc), 0.01)
        f = np.full((orc.nelems, orc.npts, orc.ndofs), 0.02)
        ls = b.new_linsys()
        b.adjoint_jacobian(u, p, up, pp, xip, xi, g, f, ls)
        z_u, z_p = np.linspace(-1e-3, 1e-3, len(u)), np.linspace(2e-3, -1e-3, len(p))
        phi = np.zeros_like(g)
        b.solve_adjoint_local(u, p, up, pp, xip, xi, z_u, z_p, phi, g, f)
        res.append((ls, g, f, phi, b.qoi_gradient(u, p, up, pp, xip, xi, z_u, z_p, phi, 4)))
    (lo, go, fo, pho, gro), (ld, gd, fd, phd, grd) = res
    errs = compare_systems(orc, ld, lo)
    errs["g"], errs["f"], errs["phi"] = rel_vec(gd, go), rel_vec(fd, fo), rel_vec(phd, pho)
    errs["grad"] = float(np.abs(grd - gro).max() / np.abs(gro).max())
    assert max(errs.values()) < 1e-12, errs


def test_synthetic_calibration_2d_objective_and_gradient_on_device():
    # the FEMU set-up of test/python/notch2D_small_J2_FEMU.yaml.in in small: measurements generated with the true
    # parameters on the notch2D mesh (full-field displacements, reaction load on ymin); at perturbed parameters the 2-D
    # Calibration objective and its adjoint gradient, checked by central differences
    import torch
    from calibr8_amd import Assembler
    from calibr8_amd.primal import PrimalDriver, adjoint_gradient
    c, conn, ns = notch2d()
    dbcs = [(0, 0, ns["xmin"], lambda x, y, z, t: 0.0), (0, 1, ns["ymin"], lambda x, y, z, t: 0.0),
            (0, 1, ns["ymax"], lambda x, y, z, t: 0.001 * t)]
    act, nsteps = [0, 1, 2, 3], 3
    truth = np.array([1000.0, 0.25, 100.0, 2.0, 0.0, 0.0])

    def solve(params, measured=None):
        asm = Assembler(3, c, conn, "small_J2", list(params))
        asm.set_active(0, act)
        asm.set_qoi_calibration(None, weights=(1.0, 1.0, 0.0), balance=1e-2, coord_idx=1, coord_value=0.0, coord_tol=1e-8,
                                comp=1, dt_over_T=1.0 / nsteps)
        pr = PrimalDriver(asm, dbcs, max_iters=15, abs_tol=1e-12, rel_tol=1e-12).solve(nsteps)
        if measured is not None:
            pr.set_measured(*measured)
        return pr

    pt = solve(truth)
    loads, zero_meas = [0.0], torch.zeros_like(pt.u[1])
    for s in range(1, nsteps + 1):
        pt.asm.set_measured(zero_meas, 0.0)
        area, total, _ = pt.asm.qoi_preprocess(pt.u[s], pt.p[s], pt.u[s - 1], pt.p[s - 1], pt.xi[s - 1], pt.xi[s])
        loads.append(total)
    assert area > 0.5 and abs(loads[-1]) > 1e-2
    measured = ([None] + [u.clone() for u in pt.u[1:]], loads)
    pt.set_measured(*measured)
    assert abs(pt.qoi()) < 1e-18
    base = truth * np.array([1.1, 1.0, 0.8, 0.9, 1.0, 1.0])
    pr = solve(base, measured)
    J0 = pr.qoi()
    assert J0 > 1e-14 and float(pr.xi[-1][:, :, 3].max()) > 1e-4
    grad = adjoint_gradient(pr, len(act))
    direction = np.array([100.0, 0.02, 10.0, 0.2])
    gd = float(grad @ direction)
    errs = []
    for k in (2, 3, 4):
        h = 10.0 ** (-k)
        pp, pm = base.copy(), base.copy()
        pp[act] += h * direction
        pm[act] -= h * direction
        errs.append(abs((solve(pp, measured).qoi() - solve(pm, measured).qoi()) / (2 * h) - gd))
    assert min(errs) < 1e-6 * abs(gd), (errs, gd)


def boundary_edges_on(conn, nodes):
    """edges of a tri3 mesh that belong to one triangle only and whose two nodes lie in `nodes` (a side set of a 2-D mesh)"""
    count = {}
    for tri in conn:
        for a, b in ((tri[0], tri[1]), (tri[1], tri[2]), (tri[2], tri[0])):
            key = (min(int(a), int(b)), max(int(a), int(b)))
            count[key] = count.get(key, 0) + 1
    inside = set(int(n) for n in nodes)
    return np.array([k for k, v in count.items() if v == 1 and k[0] in inside and k[1] in inside], dtype=np.int32)


@pytest.mark.parametrize("global_type,model,params", [
    ("mechanics", "small_J2", [1000.0, 0.25, 100.0, 10.0, 0.0, 0.0]),
    ("mechanics_plane_stress", "small_hill_plane_stress", [1000.0, 0.25, 2.0, 10.0, 2.0, 1.0, 1.0, 1.0, 1.0])])
def test_2d_tractions_on_device(global_type, model, params):
    # tbcs.cpp:17-86 on a 2-D mesh: the sides are edges (order-1 rule: midpoint, N = 1/2, w dv = length), two equations per
    # node.  (1) the device kernel equals the host restatement, a constant traction integrates to T * length of the side
    # set; (2) a traction-driven load history through c8_primal_solve_step takes the Newton iterations of the
    # oracle-driven host driver and lands on the same objective.
    import torch
    from calibr8_amd import Assembler
    from calibr8_amd import lib as L
    from calibr8_amd.primal import PrimalDriver
    from fe_driver import Dbc, Primal, Tbc, apply_tbcs
    c, conn, ns = notch2d()
    edges = boundary_edges_on(conn, ns["ymax"])
    assert len(edges) == len(ns["ymax"]) - 1
    asm = Assembler(3, c, conn, model, params)  # the global residual follows the model's name, as in the reference's decks
    orc = ol.Oracle(ol.TRI3, c, conn, model, params)
    assert asm.nres == (1 if global_type == "mechanics_plane_stress" else 2)
    # (1) kernel against the host restatement
    fn = lambda x, y, z, t: (0.3 * x + 0.1, 2.0 - 0.5 * x * x, 0.0)
    pts = np.zeros((len(edges), 1, 3))
    L.check(asm.L.c8_face_points(2, len(edges), asm.coords.ctypes.data_as(L.dp), edges.ctypes.data_as(L.i32p), pts.ctypes.data_as(L.dp)))
    assert np.allclose(pts[:, 0, :], 0.5 * (c[edges[:, 0]] + c[edges[:, 1]]), atol=1e-15)
    tr = np.array([[fn(*p, 0.0) for p in fp] for fp in pts])
    ls = asm.new_linsys()
    asm.apply_traction([(0, torch.as_tensor(edges, device=asm.device), asm.dev(tr.ravel()))], ls)
    ref = orc.new_linsys()
    apply_tbcs(ref, [Tbc(0, [tuple(e) for e in edges], fn)], c, 0.0)
    b = ls.b[0].cpu().numpy()
    assert np.abs(b - ref.b[0]).max() < 1e-14 * np.abs(ref.b[0]).max()
    ls.zero()
    asm.apply_traction([(0, torch.as_tensor(edges, device=asm.device), asm.dev(np.tile([0.0, 2.0, 0.0], len(edges))))], ls)
    length = np.linalg.norm(c[edges[:, 1]] - c[edges[:, 0]], axis=1).sum()
    b = ls.b[0].cpu().numpy().reshape(-1, 2)
    assert abs(b[:, 1].sum() + 2.0 * length) < 1e-12 and abs(b[:, 0].sum()) < 1e-14
    # a 3-D side on a 2-D mesh is refused
    with pytest.raises(L.C8Error):
        asm.apply_traction([(0, torch.zeros((1, 3), dtype=torch.int32, device=asm.device), asm.dev(np.zeros(3)))], ls)
    # (2) traction-driven steps into the plastic range
    zero = lambda x, y, z, t: 0.0
    dbcs = [(0, 0, ns["xmin"], zero), (0, 1, ns["ymin"], zero)]
    pull = lambda x, y, z, t: (0.0, 3.0 * t, 0.0)
    drv = PrimalDriver(asm, dbcs, [(0, edges, pull)], max_iters=20).solve(3)
    pr = Primal(orc, c, [Dbc(r, e, n, f) for r, e, n, f in dbcs], [Tbc(0, [tuple(e) for e in edges], pull)], max_iters=20).solve(3)
    assert drv.newton_iters == pr.newton_iters, (drv.newton_iters, pr.newton_iters)
    assert abs(drv.qoi() - pr.qoi()) < 1e-9 * abs(pr.qoi())
    assert float(drv.xi[-1][:, :, -1].max()) > 1e-4  # plastic
