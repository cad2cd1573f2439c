"""The 2-D path (tri3, `mechanics` with 2 + 1 equations per node) on the CPU: the kernel source
run by the lane emulator against the oracle, every entry point, on a jiggled structured triangle mesh and on the
reference's notch2D mesh.  The oracle's 2-D path is pinned to the reference in test_oracle_pins.py."""
import numpy as np
import pytest

import emul_lib as em
import oracle_lib as ol
from parity_cases import ACTIVE, CASES_2D, CASES_PLANE_STRESS, check_adjoint_chain, check_forward, check_residual, mesh_2d


@pytest.mark.parametrize("mesh", ["structured", "notch2D"])
@pytest.mark.parametrize("model,params,eps", CASES_2D)
def test_emulated_2d_kernels_match_oracle(model, params, eps, mesh):
    et, c, conn = mesh_2d(mesh)
    orc = ol.Oracle(et, c, conn, model, params)
    dut = em.Emul(et, c, conn, model, params)
    assert orc.ndims == 2 and orc.nloc in (4, 5) and orc.ndofs == 9
    check_forward(orc, dut, c, model, eps, 1e-12)
    check_residual(orc, dut, c, eps, 1e-12)
    check_adjoint_chain(orc, dut, c, model, eps, 1e-12)


@pytest.mark.parametrize("mesh", ["structured", "notch2D"])
@pytest.mark.parametrize("model,params,eps", CASES_PLANE_STRESS)
def test_emulated_plane_stress_kernels_match_oracle(model, params, eps, mesh):
    # `mechanics_plane_stress`: one global residual (u), one ip set, six element DOFs; the p arrays and the other three
    # blocks of the test containers must come back untouched (compare_systems checks them against the oracle's zeros)
    et, c, conn = mesh_2d(mesh)
    orc = ol.Oracle(et, c, conn, model, params)
    dut = em.Emul(et, c, conn, model, params)
    assert orc.nres == 1 and orc.ndims == 2 and orc.nloc in (4, 5, 6) and orc.ndofs == 6
    check_forward(orc, dut, c, model, eps, 1e-12)
    check_residual(orc, dut, c, eps, 1e-12)
    check_adjoint_chain(orc, dut, c, model, eps, 1e-12)


def test_plane_stress_plastic_branch_ran():
    from parity_cases import two_steps
    et, c, conn = mesh_2d("structured")
    for model, params, eps in CASES_PLANE_STRESS[1:]:
        st = two_steps(ol.Oracle(et, c, conn, model, params), c, eps)
        alpha = st[2][2][:, :, 3 if model != "hyper_J2_plane_stress" else 5]
        assert (alpha > 0).mean() > 0.3, model


def test_2d_plastic_branch_ran():
    from parity_cases import J2, two_steps
    et, c, conn = mesh_2d("structured")
    orc = ol.Oracle(et, c, conn, "small_J2", J2)
    st = two_steps(orc, c, 0.004)
    assert (st[2][2][:, :, 3] > 0).mean() > 0.3


@pytest.mark.parametrize("model,listed", [("small_J2", False), ("small_hill_plane_stress", True)])
def test_calibration_objective_2d(model, listed):
    # Calibration's 2-D branch (calibration.cpp:76-104, :163-222): the displacement mismatch integrated over the elements
    # (all of them, or a listed subset: the distance-threshold form) with the triangle's order-2 rule, the reaction load on a
    # coordinate line, under `mechanics` (2 + 1 equations) and `mechanics_plane_stress` (one residual, thickness)
    from parity import compare_systems, rel_vec
    from parity_cases import HILL_PS, J2, two_steps
    et, c, conn = mesh_2d("structured")
    params = J2 if model == "small_J2" else HILL_PS
    ymin = c[:, 1].min()
    elems = [e for e in range(len(conn)) if c[conn[e]].mean(axis=0)[0] > 0.4] if listed else None
    kw = dict(weights=(1.0, 2.0, 0.0), balance=0.3, coord_idx=1, coord_value=float(ymin), coord_tol=0.06, comp=1, dt_over_T=0.5)
    orc, dut = ol.Oracle(et, c, conn, model, params), em.Emul(et, c, conn, model, params)
    orc.set_calibration(elems, **kw)
    dut.set_calibration(elems, **kw)
    st = two_steps(orc, c, 0.004)
    (u, p, xi), (up, pp, xip) = st[2], st[1]
    u_meas = u + 1e-4 * np.random.default_rng(3).standard_normal(len(u))
    for b in (orc, dut):
        b.set_active(0, [0, 1, 2, 3])
        b.set_measured(u_meas, -0.7)
    po, pd = orc.qoi_preprocess(u, p, up, pp, xip, xi), dut.qoi_preprocess(u, p, up, pp, xip, xi)
    assert po[0] > 0 and abs(po[1]) > 1e-3 and np.abs(po - pd).max() < 1e-12 * max(1.0, np.abs(po).max()), (po, pd)
    Jo, Jd = orc.eval_qoi(u, p), dut.eval_qoi(u, p)
    assert abs(Jo - Jd) < 1e-12 * abs(Jo), (Jo, Jd)
    res = []
    for b in (orc, dut):
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
