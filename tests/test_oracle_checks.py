"""Self-consistency checks of the CPU oracle:
quadrature exactness (reference test/unit/quadrature.cpp.in:47-65), finite-difference
check of the condensed Jacobian, patch test, and the reference-style finite-difference
check of the adjoint gradient (main_inverse.cpp:126-158)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from fe_driver import Dbc, Primal, adjoint_gradient, block_matrix
from meshes import brick, jiggle, prescribed_fields

HERE = os.path.dirname(os.path.abspath(__file__))
J2 = [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0]  # adjoint/notch2D_small_J2_adjoint_check.yaml.in:27-33


def integrate(elem_type, coords, conn, ip_set, fn):
    pts, wts = ol.kit_points(elem_type, ip_set)
    tot = 0.0
    for e in conn:
        X = coords[e]
        for q, w in zip(pts, wts):
            N, dN, dv = ol.shape(elem_type, X, q)
            tot += fn(N @ X) * w * dv
    return tot


def test_quadrature_exactness_tet4():
    d = json.load(open(os.path.join(HERE, "golden", "cube_tet4.json")))
    c, conn = np.array(d["coords"]), np.array(d["conn"])
    f = lambda p: (lambda x: x[0] ** p + x[1] ** p + x[2] ** p)
    assert abs(integrate(ol.TET4, c, conn, 0, f(1)) - 1.5) < 1e-14   # order-1 rule
    assert abs(integrate(ol.TET4, c, conn, 1, f(1)) - 1.5) < 1e-14   # order-2 rule
    assert abs(integrate(ol.TET4, c, conn, 1, f(2)) - 1.0) < 1e-14


def test_quadrature_exactness_hex8():
    c, conn, _ = brick(2, 3, 2)
    f = lambda p: (lambda x: x[0] ** p + x[1] ** p + x[2] ** p)
    for s in (0, 1):
        assert abs(integrate(ol.HEX8, c, conn, s, f(1)) - 1.5) < 1e-14
        assert abs(integrate(ol.HEX8, c, conn, s, f(2)) - 1.0) < 1e-14
        assert abs(integrate(ol.HEX8, c, conn, s, f(3)) - 0.75) < 1e-14


def test_shape_partition_of_unity_and_gradients():
    rng = np.random.default_rng(0)
    c, conn, sets = brick(2, 2, 2)
    c = jiggle(c, sets, 0.1)
    for et, X in ((ol.HEX8, c[conn[0]]), (ol.TET4, np.array([[0, 0, 0], [1, 0, 0.1], [0.2, 1, 0], [0, 0.1, 1.]]))):
        xi = rng.random(3) * 0.2 + 0.1
        N, dN, dv = ol.shape(et, X, xi)
        assert abs(N.sum() - 1) < 1e-14 and np.abs(dN.sum(0)).max() < 1e-13 and dv > 0
        # a linear field is reproduced exactly with its gradient
        a = np.array([0.3, -1.2, 0.7])
        assert np.abs(dN.T @ (X @ a) - a).max() < 1e-12


@pytest.mark.parametrize("model,params,eps", [("small_J2", J2, 0.004), ("small_J2", J2, 0.001),
                                              ("elastic", [1000.0, 0.25, 1e-3, 10.0], 0.002),
                                              ("hyper_J2", [1000.0, 0.25, 2.0, 1.0, 5.0, 0.5, 0.5, 100.0], 0.004),
                                              ("small_hill", [1000.0, 0.25, 2.0, 1.0, 1.1, 0.9, 1.05, 0.95, 1.0, 1.0, 50.0], 0.004),
                                              ("isotropic_elastic", [1000.0, 0.25], 0.002),
                                              ("hypo_hill", [1000.0, 0.25, 2.0, 1.0, 1.1, 0.9, 1.05, 0.95, 1.0, 1.0, 50.0], 0.004)])
def test_jacobian_matches_finite_differences_hex8(model, params, eps):
    c, conn, sets = brick(3, 2, 2, 1.0, 0.8, 0.7)
    c = jiggle(c, sets, 0.05)
    be = ol.Oracle(ol.HEX8, c, conn, model, params)
    u, p = prescribed_fields(c, eps, ramp=True, perturb=5e-2)
    u0, p0 = np.zeros_like(u), np.zeros_like(p)
    xi_prev = be.new_state()

    def resid(uu, pp):
        ls, xi = be.new_linsys(), be.new_state()
        assert be.forward_jacobian(uu, pp, u0, p0, xi_prev, xi, ls) == 0
        return np.concatenate(ls.b), ls, xi

    R, ls, xi = resid(u, p)
    if model in ("small_J2", "small_hill", "hypo_hill"):
        frac = (xi[:, :, 6] > 0).mean()
        assert (frac > 0.2) if eps > 0.003 else (frac == 0.0)
    A = block_matrix(be, ls).toarray()
    rng = np.random.default_rng(3)
    x = np.concatenate([u, p])
    worst = 0.0
    for _ in range(6):
        v = rng.standard_normal(len(x))
        v /= np.linalg.norm(v)
        h = 1e-6 * max(1.0, np.abs(x).max())
        xp, xm = x + h * v, x - h * v
        Rp = resid(np.ascontiguousarray(xp[:len(u)]), np.ascontiguousarray(xp[len(u):]))[0]
        Rm = resid(np.ascontiguousarray(xm[:len(u)]), np.ascontiguousarray(xm[len(u):]))[0]
        fd = (Rp - Rm) / (2 * h)
        worst = max(worst, np.abs(A @ v - fd).max() / np.abs(A @ v).max())
    assert worst < 2e-6, worst


def test_patch_test_hex8_elastic():
    # a homogeneous strain state with p = -kappa tr(eps) gives zero interior residual on a distorted mesh
    c, conn, sets = brick(3, 3, 3)
    c = jiggle(c, sets, 0.08)
    E, nu = 1000.0, 0.25
    be = ol.Oracle(ol.HEX8, c, conn, "elastic", [E, nu, 0.0, 0.0])
    G = np.array([[0.001, 0.0004, 0.0], [-0.0002, 0.002, 0.0003], [0.0, 0.0001, -0.0015]])
    u = (c @ G.T).ravel().copy()
    kappa = E / (3 * (1 - 2 * nu))
    p = np.full(len(c), -kappa * np.trace(G))
    ls, xi = be.new_linsys(), be.new_state()
    be.forward_jacobian(u, p, np.zeros_like(u), np.zeros_like(p), be.new_state(), xi, ls)
    interior = np.ones(len(c), dtype=bool)
    for v in sets.values():
        interior[v] = False
    Ru = ls.b[0].reshape(-1, 3)
    assert np.abs(Ru[interior]).max() < 1e-12 * np.abs(Ru).max()
    assert np.abs(ls.b[1][interior]).max() < 1e-14
    # and the residual-only path agrees with the residual of the Jacobian path
    ls2 = be.new_linsys()
    be.global_residual(u, p, np.zeros_like(u), np.zeros_like(p), be.new_state(), xi, ls2)
    assert np.abs(ls2.b[0] - ls.b[0]).max() < 1e-13 * np.abs(ls.b[0]).max()


def small_j2_bar(params):
    c, conn, sets = brick(2, 3, 2, 1.0, 1.5, 1.0)
    c = jiggle(c, sets, 0.05)
    be = ol.Oracle(ol.HEX8, c, conn, "small_J2", params)
    dbcs = [Dbc(0, 0, sets["xmin"], lambda x, y, z, t: 0.0), Dbc(0, 1, sets["ymin"], lambda x, y, z, t: 0.0),
            Dbc(0, 2, sets["zmin"], lambda x, y, z, t: 0.0), Dbc(0, 1, sets["ymax"], lambda x, y, z, t: 0.002 * t)]
    return be, c, dbcs


def objective(params, nsteps=3):
    be, c, dbcs = small_j2_bar(params)
    pr = Primal(be, c, dbcs, max_iters=15, abs_tol=1e-12, rel_tol=1e-12).solve(nsteps)
    return pr.qoi(), pr


def test_adjoint_gradient_fd_check_small_J2():
    # the reference's recipe: 13 step sizes, 2nd-order FD along a fixed direction; the error must
    # drop by >= 7 decades between the worst and best step (cf. adjoint check pin 7.738 +- 10 %)
    base = np.array(J2)
    J0, pr = objective(base)
    assert pr.xi[-1][:, :, 6].max() > 1e-4  # plastic
    active = [0, 1, 2, 3]
    pr.be.set_active(0, active)
    grad = adjoint_gradient(pr, len(active))
    direction = np.array([100.0, 0.02, 10.0, 0.2])  # ~10 % of each parameter
    gd = float(grad @ direction)
    errs = []
    for k in range(13):
        h = 10.0 ** (-k)
        pp, pm = base.copy(), base.copy()
        pp[active] += h * direction
        pm[active] -= h * direction
        fd = (objective(pp)[0] - objective(pm)[0]) / (2 * h)
        errs.append(abs(fd - gd))
    errs = np.array(errs)
    drop = np.log10(errs.max() / errs.min())
    assert drop > 7.0, (drop, errs, gd)
    assert errs.min() < 1e-7 * abs(gd), (errs.min(), gd)


def canonical_fd_drop(solve, adjoint_gradient_fn, p0, bounds):
    """The reference's gradient check (main_inverse.cpp:126-158: ROL checkGradient, 13 steps 1e0 .. 1e-12, order 2 =
    central differences, direction 0.1 in every CANONICAL variable c = (p - mean) / span of objective.cpp:41-61):
    log10(max |error| / min |error|) over the 13 steps.  solve(params) -> solved Primal; bounds {param index: (lo, hi)}."""
    idx = sorted(bounds)
    span = np.array([0.5 * (bounds[k][1] - bounds[k][0]) for k in idx])
    pr = solve(p0)
    grad_canonical = adjoint_gradient_fn(pr, len(idx)) * span  # objective.cpp:125-137
    d = np.full(len(idx), 0.1)
    gd = float(grad_canonical @ d)
    errs = []
    for k in range(13):
        h = 10.0 ** (-k)
        pp, pm = np.array(p0, dtype=float), np.array(p0, dtype=float)
        pp[idx] += h * d * span
        pm[idx] -= h * d * span
        fd = (solve(pp).qoi() - solve(pm).qoi()) / (2 * h)
        errs.append(abs(fd - gd))
    errs = np.array(errs)
    return float(np.log10(errs.max() / errs.min()))


def calibration_bar(params, measured=None, nsteps=3, kind="hex8"):
    """A bar pulled in y: displacement mismatch on the xmax face, reaction load on the ymin plane (component y)."""
    c, conn, sets = brick(2, 3, 2, 1.0, 1.5, 1.0)
    c = jiggle(c, sets, 0.0)
    be = ol.Oracle(ol.HEX8, c, conn, "small_J2", params)
    dbcs = [Dbc(0, 0, sets["xmin"], lambda x, y, z, t: 0.0), Dbc(0, 1, sets["ymin"], lambda x, y, z, t: 0.0),
            Dbc(0, 2, sets["zmin"], lambda x, y, z, t: 0.0), Dbc(0, 1, sets["ymax"], lambda x, y, z, t: 0.002 * t)]
    xmax = set(sets["xmax"].tolist())
    faces = []
    for e in conn:
        for f in ([0, 1, 2, 3], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7], [4, 5, 6, 7]):
            if all(int(e[k]) in xmax for k in f):
                faces.append([int(e[k]) for k in f])
    be.set_calibration(faces, weights=(1.0, 2.0, 0.5), balance=1e-4, coord_idx=1, coord_value=0.0, coord_tol=1e-8,
                       comp=1, dt_over_T=1.0 / nsteps)
    pr = Primal(be, c, dbcs, max_iters=15, abs_tol=1e-12, rel_tol=1e-12).solve(nsteps)
    if measured is not None:
        pr.measured = measured
    return pr


def test_calibration_qoi_value_and_fd_gradient():
    # synthetic calibration (the reference's example set-up in small): "measurements" from the true parameters,
    # objective = surface displacement mismatch + balance * load mismatch (calibration.cpp)
    truth = np.array(J2)
    pt = calibration_bar(truth)
    nsteps = len(pt.u) - 1
    loads = [0.0]
    for s in range(1, nsteps + 1):
        pt.be.set_measured(np.zeros_like(pt.u[s]), 0.0)
        area, total, mism = pt.be.qoi_preprocess(pt.u[s], pt.p[s], pt.u[s - 1], pt.p[s - 1], pt.xi[s - 1], pt.xi[s])
        loads.append(total)
    assert abs(area - 1.5) < 1e-12            # the xmax face of the 1 x 1.5 x 1 bar
    assert loads[-1] < 0 < abs(loads[1])      # pulling in +y: the reaction on the ymin plane points to -y
    measured = ([None] + [u.copy() for u in pt.u[1:]], loads)
    pt.measured = measured
    assert abs(pt.qoi()) < 1e-20              # zero at the truth
    base = truth * np.array([1.1, 1.0, 0.8, 0.9, 1.0, 1.0])

    def objective(params):
        return calibration_bar(params, measured).qoi()

    pr = calibration_bar(base, measured)
    J0 = pr.qoi()
    assert J0 > 1e-12
    assert pr.xi[-1][:, :, 6].max() > 1e-4    # plastic
    active = [0, 1, 2, 3]
    pr.be.set_active(0, active)
    grad = adjoint_gradient(pr, len(active))
    direction = np.array([100.0, 0.02, 10.0, 0.2])
    gd = float(grad @ direction)
    errs = []
    for k in range(1, 9):
        h = 10.0 ** (-k)
        pp, pm = base.copy(), base.copy()
        pp[active] += h * direction
        pm[active] -= h * direction
        errs.append(abs((objective(pp) - objective(pm)) / (2 * h) - gd))
    errs = np.array(errs)
    assert errs.min() < 1e-6 * abs(gd), (errs, gd)
    assert np.log10(errs.max() / errs.min()) > 4.0, (errs, gd)


def test_isotropic_elastic_equals_elastic_without_thermal_strain():
    # two formulations of the same material: `elastic` condenses nothing, `isotropic_elastic` carries the stress as
    # its local unknown (isotropic_elastic.cpp); residual and condensed Jacobian must coincide
    c, conn, sets = brick(3, 2, 2, 1.0, 0.8, 0.7)
    c = jiggle(c, sets, 0.05)
    u, p = prescribed_fields(c, 0.002, ramp=True, perturb=5e-2)
    z, zp = np.zeros_like(u), np.zeros_like(p)
    out = []
    for model, params in (("elastic", [1000.0, 0.25, 0.0, 0.0]), ("isotropic_elastic", [1000.0, 0.25])):
        be = ol.Oracle(ol.HEX8, c, conn, model, params)
        ls, xi = be.new_linsys(), be.new_state()
        assert be.forward_jacobian(u, p, z, zp, be.new_state(), xi, ls) == 0
        out.append(ls)
    for i in range(2):
        assert np.abs(out[0].b[i] - out[1].b[i]).max() < 1e-12 * np.abs(out[0].b[i]).max()
        for j in range(2):
            assert np.abs(out[0].A[i][j] - out[1].A[i][j]).max() < 1e-12 * np.abs(out[0].A[i][j]).max()
