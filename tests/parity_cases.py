"""Shared parity cases: the device-under-test (emulated kernels on the CPU, or the HIP library on a
GPU) against the oracle on identical seeded inputs.  Each check feeds BOTH sides the oracle's
inputs at every stage, so a deviation is attributed to the stage that produced it."""
import json
import os

import numpy as np

import oracle_lib as ol
from meshes import brick, fields_for, jiggle, prescribed_fields
from parity import compare_systems, rel_csr_rows, rel_vec

HERE = os.path.dirname(os.path.abspath(__file__))


class Audit:
    """Which of the checker's allowances a green run actually used.  The bar is 1e-12 on every quantity, compared DIRECTLY;
    four allowances exist for cases where the algorithm itself has finite resolution (documented where they are applied):
      at_state      a Jacobian block / residual above the bar directly, within it against the oracle AT THE DEVICE'S OWN
                    converged local state
      xi_10x        the local state above the bar, within ten times the local Newton tolerance (only after at_state)
      ulp_sens      an entry above the bar, within four times the oracle's own deviation under a one-ulp input change
      k3_rhs_scale  the adjoint right-hand side above the bar against its own norm, within it against its largest summand
    Every use is counted per (context, model, allowance); the models of BASELINE.json's configurations and their elastic
    companions (STRICT) must pass with none: for them "within 1e-12" is the direct comparison, nothing else."""
    STRICT = ("elastic", "small_J2", "small_hill", "isotropic_elastic")

    def __init__(self):
        from collections import Counter
        self.used, self.cases, self.ctx = Counter(), Counter(), "unset"

    def case(self, model, what):
        self.cases["%s|%s|%s" % (self.ctx, model, what)] += 1

    def fire(self, model, allowance, detail=None):
        self.used["%s|%s|%s" % (self.ctx, model, allowance)] += 1
        assert model not in self.STRICT, "allowance %s used for %s (%s): %r -- this model must meet the bar directly" % (allowance, model, self.ctx, detail)

    def table(self):
        return {"cases": dict(self.cases), "allowances_used": dict(self.used), "strict_models": list(self.STRICT)}

    def dump(self, path):
        json.dump(self.table(), open(path, "w"), indent=1, sort_keys=True)


AUDIT = Audit()
J2 = [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0]
HJ2 = [1000.0, 0.25, 2.0, 1.0, 5.0, 0.5, 0.5, 100.0]
EL = [1000.0, 0.25, 1e-3, 10.0]
# small_hill: E nu Y R00 R11 R22 R01 R02 R12 S D (small_hill.cpp:78-88)
HILL = [1000.0, 0.25, 2.0, 1.0, 1.1, 0.9, 1.05, 0.95, 1.0, 1.0, 50.0]
CASES = [("small_J2", J2, 0.001), ("small_J2", J2, 0.004), ("elastic", EL, 0.002), ("hyper_J2", HJ2, 0.004),
         ("small_hill", HILL, 0.004), ("isotropic_elastic", [1000.0, 0.25], 0.002), ("hypo_hill", HILL, 0.004)]
MESHES = ["hex8", "tet4"]
ACTIVE = {"small_J2": [0, 1, 2, 3], "elastic": [0, 1], "hyper_J2": [0, 1, 2, 3, 4, 7], "small_hill": [0, 2, 3, 6, 9, 10], "isotropic_elastic": [0, 1],
          "hypo_hill": [0, 2, 3, 6, 9, 10], "small_hill_plane_strain": [0, 2, 3, 4, 5, 8], "hyper_J2_plane_strain": [0, 1, 2, 3, 4, 5],
          "hypo_hill_plane_strain": [0, 2, 3, 4, 5, 8], "small_hill_plane_stress": [0, 2, 3, 4, 5, 8],
          "hyper_J2_plane_stress": [0, 1, 2, 3, 4, 7], "hypo_hill_plane_stress": [0, 2, 3, 8, 9, 12]}
# Hosford / Barlat family (3-D): yield functions on principal stresses, local Newton iteration with line search
HOSFORD = [1000.0, 0.25, 2.0, 8.0, 50.0, 10.0, 2.0]        # E nu Y a K S D
HOSFORD_100 = [1000.0, 0.25, 2.0, 100.0, 0.0, 10.0, 2.0]   # the exponent of notch_small_hosford.yaml.in
BARLAT = [70e3, 0.3, 200.0, 8.0, 100.0, 200.0, 20.0,       # E nu Y a K S D + Yld2004-18p coefficients of 6111-T4 aluminium
          1.241024, 1.078271, 1.216463, 1.223867, 1.093105, 0.889161, 0.501909, 0.557173, 1.349094,
          0.775366, 0.922743, 0.765487, 0.793356, 0.918689, 1.027625, 1.115833, 1.112273, 0.589787]
LOCAL_LINE_SEARCH = (1e-4, 0.5, 0.9, 100)  # the `line search:` sublist of the two decks
CASES_LINE_SEARCH = [("small_hosford", HOSFORD, 0.004), ("small_hosford", HOSFORD_100, 0.004), ("hypo_hosford", HOSFORD, 0.004),
                     ("hypo_barlat", BARLAT, 0.006)]
ACTIVE.update({"small_hosford": [0, 1, 2, 3, 5, 6], "hypo_hosford": [0, 1, 2, 3, 5, 6], "hypo_barlat": [0, 2, 3, 7, 13, 16, 22, 24]})
# 2-D (tri3) cases: the models of the reference's 2-D decks that run on `mechanics` with 2 + 1 equations per node
HILL_PS = [1000.0, 0.25, 2.0, 10.0, 2.0, 1.0, 1.1, 0.9, 1.05]  # E nu Y S D R00 R11 R22 R01
HJ2_PS = [1000.0, 0.25, 100.0, 2.0, 3.0, 40.0]  # E nu K Y Y_inf delta
CASES_2D = [("small_J2", J2, 0.001), ("small_J2", J2, 0.004), ("small_hill_plane_strain", HILL_PS, 0.004),
            ("hyper_J2_plane_strain", HJ2_PS, 0.004), ("hypo_hill_plane_strain", HILL_PS, 0.004)]
# the models of `mechanics_plane_stress` (ONE global residual: six element DOFs on tri3)
HJ2_PSS = [1000.0, 0.25, 2.0, 10.0, 2.0, 3.0, 0.6, 40.0]  # E nu Y S D A n K
HYPO_PSS = HILL_PS + [0.96, -0.28, 0.28, 0.96]              # ... + Q00 Q01 Q10 Q11: material axes rotated by 16 degrees
CASES_PLANE_STRESS = [("small_hill_plane_stress", HILL_PS, 0.001), ("small_hill_plane_stress", HILL_PS, 0.004),
                      ("hyper_J2_plane_stress", HJ2_PSS, 0.004), ("hypo_hill_plane_stress", HYPO_PSS, 0.004)]


def mesh_2d(kind="structured"):
    """tri3 test meshes: a jiggled structured one, or the reference's notch2D mesh (tests/golden/notch2D_tri3.json)"""
    from meshes import jiggle_2d, tri_mesh
    if kind == "notch2D":
        d = json.load(open(os.path.join(HERE, "golden", "notch2D_tri3.json")))
        return ol.TRI3, np.array(d["coords"]), np.array(d["conn"], dtype=np.int32)
    c, conn, sets = tri_mesh(5, 4, 1.0, 0.8)
    return ol.TRI3, jiggle_2d(c, sets, 0.04), conn


def mesh_of(kind, n=(4, 3, 3)):
    if kind == "hex8":
        c, conn, sets = brick(n[0], n[1], n[2], 1.0, 0.8, 0.7)
        return ol.HEX8, jiggle(c, sets, 0.04), conn
    d = json.load(open(os.path.join(HERE, "golden", "cube_tet4.json")))
    return ol.TET4, np.array(d["coords"]), np.array(d["conn"], dtype=np.int32)


def make_pair(factory, kind, model, params, **kw):
    et, c, conn = mesh_of(kind)
    return ol.Oracle(et, c, conn, model, params, **kw), factory(et, c, conn, model, params, **kw), c


def two_steps(orc, c, eps):
    """Two consecutive load steps solved locally by the oracle: returns per-step (u, p, xi)."""
    u1, p1 = fields_for(getattr(orc, "ndims", 3), *prescribed_fields(c, eps, ramp=True, perturb=5e-2))
    u0, p0 = np.zeros_like(u1), np.zeros_like(p1)
    xi0, xi1, xi2 = orc.new_state(), orc.new_state(), orc.new_state()
    assert orc.forward_jacobian(u1, p1, u0, p0, xi0, xi1, orc.new_linsys()) == 0
    u2, p2 = 1.5 * u1, 1.5 * p1
    assert orc.forward_jacobian(u2, p2, u1, p1, xi1, xi2, orc.new_linsys()) == 0
    return [(u0, p0, xi0), (u1, p1, xi1), (u2, p2, xi2)]


def jacobian_at_state(orc, u, p, up, pp, xip, xi):
    """The oracle's condensed Jacobian dR/dx at a GIVEN converged local state, as scipy blocks: its adjoint assembly
    (evaluations.cpp:349-526) evaluates at the stored state and scatters the transpose."""
    import scipy.sparse as sp
    ls = orc.new_linsys()
    orc.adjoint_jacobian(u, p, up, pp, xip, xi, np.zeros((orc.nelems, orc.npts, orc.nloc)),
                         np.zeros((orc.nelems, orc.npts, orc.ndofs)), ls)
    shape = lambda i, j: (len(orc.rowptr[i][j]) - 1, len(orc.rowptr[j][i]) - 1)
    T = [[sp.csr_matrix((ls.A[j][i], orc.colidx[j][i], orc.rowptr[j][i]), shape=shape(j, i)).T.tocsr() for j in range(2)]
         for i in range(2)]
    for i in range(2):
        for j in range(2):
            T[i][j].sort_indices()
            assert np.array_equal(T[i][j].indices, orc.colidx[i][j])  # structurally symmetric graphs
    return [[T[i][j].data for j in range(2)] for i in range(2)]


def forward_errors(orc, ls_d, xd, ls_o, xo, u, p, up, pp, xip):
    """Deviations of a forward assembly (ls_d, xd) from the oracle's (ls_o, xo).  The Jacobian is compared directly and,
    where that exceeds the bar, against the oracle's Jacobian AT THE DEVICE'S converged local state: the map from the
    local state to dR/dx can be ill-conditioned (hyper_J2 at plastic onset: the power-law flow stress A (alpha +
    1e-12)^n has curvature ~alpha^(n-2), 1e12 at a first plastic increment of 1e-8), so two states that agree to one
    unit in the last place (1e-15) give Jacobians that differ by 1e-11 -- in the oracle itself.  Returned: the direct
    deviations with every A block replaced by min(direct, at-the-same-state), plus both under their own names."""
    errs = compare_systems(orc, ls_d, ls_o)
    errs["xi"] = rel_vec(xd, xo)
    direct = max(max(errs["A%d%d" % (i, j)] for i in range(2) for j in range(2)), errs["b_u"], errs["b_p"])  # the direct comparison
    same = None
    if max(errs["A%d%d" % (i, j)] for i in range(2) for j in range(2)) > 1e-13 or errs["xi"] > 1e-13:
        A_at = jacobian_at_state(orc, u, p, up, pp, xip, xd)
        same = {}
        for i in range(2):
            for j in range(2):
                k = "A%d%d" % (i, j)
                same[k] = rel_csr_rows(ls_d.A[i][j], A_at[i][j], orc.rowptr[i][j])
                errs[k] = min(errs[k], same[k])
        # the residual at the DEVICE'S converged state (eval_global_residual evaluates at a stored state): the local
        # Newton iteration stops at |C| < abs_tol, so two correct runs may stop one iteration apart and their states then
        # differ by ~abs_tol |dC/dxi^-1| -- the residual follows the state (round-2 sweep seed 214: states 6e-13 apart,
        # residuals 8e-11 apart, both agreeing with their own state to 1e-15)
        ls_s = orc.new_linsys()
        orc.global_residual(u, p, up, pp, xip, xd, ls_s)
        for i, k in enumerate(("b_u", "b_p")):
            same[k] = rel_vec(ls_d.b[i], ls_s.b[i])
            errs[k] = min(errs[k], same[k])
    return errs, direct, same


def ulp_perturbed(u, seed=1):
    """u with every entry moved by one unit in the last place, sign at random"""
    return u * (1.0 + 2.2e-16 * np.random.default_rng(seed).choice([-1.0, 1.0], len(u)))


def bar_with_sensitivity(errs, tol, sens):
    """Entries of `errs` above `tol` are accepted up to four times the ORACLE'S OWN deviation under a one-ulp perturbation of
    the displacement field (`sens`, same keys): where the oracle cannot reproduce itself to 1e-12 from inputs that differ
    in the last bit, no second implementation can be asked to.  Seen only with Hosford's exponent a = 100 (x^a has
    condition number a; 5 of 1000 random cases reach 1.1e-12 .. 1.4e-12, the oracle's own sensitivity there is 8e-13)."""
    return {k: v for k, v in errs.items() if v >= max(tol, 4.0 * sens.get(k, 0.0))}


def check_forward(orc, dut, c, model, eps, tol):
    st = two_steps(orc, c, eps)
    for n in (1, 2):
        (u, p, xi_ref), (up, pp, xip) = st[n], st[n - 1]
        ls_o, ls_d = orc.new_linsys(), dut.new_linsys()
        xo, xd = orc.new_state(), dut.new_state()
        assert orc.forward_jacobian(u, p, up, pp, xip, xo, ls_o) == 0
        assert dut.forward_jacobian(u, p, up, pp, xip, xd, ls_d) == 0
        errs, direct, same = forward_errors(orc, ls_d, xd, ls_o, xo, u, p, up, pp, xip)
        AUDIT.case(model, "forward")
        if direct >= tol and max(errs.values()) < tol:
            AUDIT.fire(model, "at_state", (direct, same))
        # the local state agrees to `tol`, or -- when system and state have been checked against each other at the
        # device's own state -- to ten times the local Newton tolerance (the iteration's own resolution)
        xi_err = errs.pop("xi")
        if max(errs.values()) >= tol:  # rare: judge against the oracle's own one-ulp sensitivity (bar_with_sensitivity)
            ls_p = orc.new_linsys()
            assert orc.forward_jacobian(ulp_perturbed(u), p, up, pp, xip, orc.new_state(), ls_p) == 0
            bad = bar_with_sensitivity(errs, tol, compare_systems(orc, ls_p, ls_o))
            assert not bad, (n, bad, errs, direct, same, xi_err)
            AUDIT.fire(model, "ulp_sens", errs)
        assert xi_err < tol or (same is not None and xi_err < 10.0 * getattr(orc, "local_abs_tol", 1e-12)), (n, xi_err, errs)
        if xi_err >= tol:
            AUDIT.fire(model, "xi_10x", xi_err)
    if model == "small_J2" and eps > 0.003:
        assert (st[2][2][:, :, -1] > 0).mean() > 0.3  # the plastic branch really ran (alpha is the last local unknown)


def check_residual(orc, dut, c, eps, tol):
    st = two_steps(orc, c, eps)
    (u, p, xi), (up, pp, xip) = st[2], st[1]
    ls_o, ls_d = orc.new_linsys(), dut.new_linsys()
    orc.global_residual(u, p, up, pp, xip, xi, ls_o)
    assert dut.global_residual(u, p, up, pp, xip, xi, ls_d) == 0
    assert rel_vec(ls_d.b[0], ls_o.b[0]) < tol and rel_vec(ls_d.b[1], ls_o.b[1]) < tol
    # the residual-only path reproduces the residual of the Jacobian path at the converged state
    ls_j = orc.new_linsys()
    orc.forward_jacobian(u, p, up, pp, xip, orc.new_state(), ls_j)
    assert rel_vec(ls_d.b[0], ls_j.b[0]) < 1e-10 and rel_vec(ls_d.b[1], ls_j.b[1]) < 1e-10


def check_adjoint_chain(orc, dut, c, model, eps, tol):
    st = two_steps(orc, c, eps)
    act = ACTIVE[model]
    orc.set_active(0, act)
    dut.set_active(0, act)
    nd = orc.ndofs
    rng = np.random.default_rng(11)
    g_o = np.zeros((orc.nelems, orc.npts, orc.nloc))
    f_o = np.zeros((orc.nelems, orc.npts, nd))
    for n in (2, 1):
        (u, p, xi), (up, pp, xip) = st[n], st[n - 1]
        # K3
        g_d, f_d = g_o.copy(), f_o.copy()
        g_in, f_in = g_o.copy(), f_o.copy()
        ls_o, ls_d = orc.new_linsys(), dut.new_linsys()
        orc.adjoint_jacobian(u, p, up, pp, xip, xi, g_o, f_o, ls_o)
        assert dut.adjoint_jacobian(u, p, up, pp, xip, xi, g_d, f_d, ls_d) == 0
        errs = compare_systems(orc, ls_d, ls_o)
        errs["g"] = rel_vec(g_d, g_o)
        AUDIT.case(model, "adjoint")
        # the right-hand side is a sum over elements of entries of the history f (among others): judged against the
        # largest summand, not against the sum (with the stand-in adjoint solution below the summands cancel by three
        # orders of magnitude: |f| ~ 20, |b| ~ 0.02; the order of the scatter then shows at 4e-12 of |b|)
        unscaled = max(errs["b_u"], errs["b_p"])
        for i, k in enumerate(("b_u", "b_p")):
            top = np.abs(ls_o.b[i]).max()
            errs[k] *= top / max(top, np.abs(f_in).max()) if top > 0 else 1.0
        if unscaled >= tol and max(errs["b_u"], errs["b_p"]) < tol:
            AUDIT.fire(model, "k3_rhs_scale", unscaled)
        if max(errs.values()) >= tol:
            ls_p = orc.new_linsys()
            orc.adjoint_jacobian(ulp_perturbed(u), p, up, pp, xip, xi, g_in.copy(), f_in, ls_p)
            bad = bar_with_sensitivity(errs, tol, compare_systems(orc, ls_p, ls_o))
            assert not bad, ("adjoint_jacobian", n, bad, errs)
            AUDIT.fire(model, "ulp_sens", errs)
        # a stand-in global adjoint solution (the linear solve is out of scope)
        z_u = rng.standard_normal(len(u)) * 1e-3
        z_p = rng.standard_normal(len(p)) * 1e-3
        # K4
        phi_o, phi_d = np.zeros_like(g_o), np.zeros_like(g_o)
        g_d, f_d = g_o.copy(), f_o.copy()
        orc.solve_adjoint_local(u, p, up, pp, xip, xi, z_u, z_p, phi_o, g_o, f_o)
        assert dut.solve_adjoint_local(u, p, up, pp, xip, xi, z_u, z_p, phi_d, g_d, f_d) == 0
        errs = {"phi": rel_vec(phi_d, phi_o), "g": rel_vec(g_d, g_o), "f": rel_vec(f_d, f_o)}
        assert max(errs.values()) < tol, ("solve_adjoint_local", n, errs)
        # K5
        gr_o, gr_scale = orc.qoi_gradient_with_scale(u, p, up, pp, xip, xi, z_u, z_p, phi_o, len(act))
        gr_d = dut.qoi_gradient(u, p, up, pp, xip, xi, z_u, z_p, phi_o, len(act))
        # per component at 1e-12.  A component is a sum over all points of products of either sign and may cancel
        # (sweep seed 478: dJ/dE = 3e-11 beside dJ/dnu = -9e-8): the scale of component i is the sum of the MAGNITUDES
        # of the products summed into it (>= |component|, equal to it when nothing cancels) -- the forward error bound
        # of any evaluation order of that sum
        assert (np.abs(gr_d - gr_o) / np.maximum(gr_scale, 1e-300)).max() < tol, ("qoi_gradient", n, gr_d, gr_o, gr_scale)
        # K6
        assert abs(dut.eval_qoi(u, p) - orc.eval_qoi(u, p)) < tol * max(1.0, abs(orc.eval_qoi(u, p)))


def check_two_element_sets(factory, kind, tol, wave=None):
    """Two material blocks with different parameters and different active-parameter lists
    (local_residual.cpp:96-100 per-set parameters, :812-819 seed_wrt_params(es), :859-867 scatter_es_gradient)."""
    et, c, conn = mesh_of(kind)
    nelems = len(conn)
    rng = np.random.default_rng(21)
    es = (rng.random(nelems) < 0.4).astype(np.int32)  # ragged: sets are interleaved, not contiguous
    params = [[1000.0, 0.25, 100.0, 2.0, 0.0, 0.0], [700.0, 0.3, 50.0, 1.2, 1e-4, 5.0]]
    orc = ol.Oracle(et, c, conn, "small_J2", params, elem_set=es)
    dut = factory(et, c, conn, "small_J2", params, elem_set=es)
    if wave is not None:
        dut.wave = wave
    st = two_steps(orc, c, 0.004)
    (u, p, xi), (up, pp, xip) = st[2], st[1]
    ls_o, ls_d, xo, xd = orc.new_linsys(), dut.new_linsys(), orc.new_state(), dut.new_state()
    assert orc.forward_jacobian(u, p, up, pp, xip, xo, ls_o) == 0
    assert dut.forward_jacobian(u, p, up, pp, xip, xd, ls_d) == 0
    errs = compare_systems(orc, ls_d, ls_o)
    errs["xi"] = rel_vec(xd, xo)
    assert max(errs.values()) < tol, errs
    for b, a0, a1 in ((orc, [0, 3], [1, 2, 3]), (dut, [0, 3], [1, 2, 3])):
        b.set_active(0, a0)
        b.set_active(1, a1)
    nd = 4 * orc.nn
    g = np.zeros((orc.nelems, orc.npts, orc.nloc))
    f = np.zeros((orc.nelems, orc.npts, nd))
    orc.adjoint_jacobian(u, p, up, pp, xip, xi, g, f, orc.new_linsys())
    z_u, z_p = rng.standard_normal(len(u)) * 1e-3, rng.standard_normal(len(p)) * 1e-3
    phi = np.zeros_like(g)
    g_d, f_d, phi_d = g.copy(), f.copy(), np.zeros_like(g)
    orc.solve_adjoint_local(u, p, up, pp, xip, xi, z_u, z_p, phi, g, f)
    # the local adjoint solve with per-set parameters (the device's own kernel choice: closed form where the model has one)
    dut.solve_adjoint_local(u, p, up, pp, xip, xi, z_u, z_p, phi_d, g_d, f_d)
    assert rel_vec(phi_d, phi) < tol and rel_vec(g_d, g) < tol and np.abs(f_d - f).max() <= tol * max(1.0, np.abs(f).max()), (rel_vec(phi_d, phi), rel_vec(g_d, g))
    gr_o, gr_scale = orc.qoi_gradient_with_scale(u, p, up, pp, xip, xi, z_u, z_p, phi, 5)
    gr_d = dut.qoi_gradient(u, p, up, pp, xip, xi, z_u, z_p, phi, 5)
    # per component at `tol` against the sum of the magnitudes of the products summed into it (see check_adjoint_chain)
    assert np.abs(gr_o).min() > 0 and (np.abs(gr_d - gr_o) / np.maximum(gr_scale, 1e-300)).max() < tol, (gr_d, gr_o, gr_scale)


def check_tiny_and_ragged(factory, tol):
    """One element, and element counts that do not fill a workgroup / wavefront."""
    for n in ((1, 1, 1), (3, 1, 1), (5, 1, 1), (3, 3, 1)):
        c, conn, sets = brick(n[0], n[1], n[2], 1.0 * n[0], 1.0 * n[1], 1.0 * n[2])
        orc = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2)
        dut = factory(ol.HEX8, c, conn, "small_J2", J2)
        check_forward(orc, dut, c, "small_J2", 0.0035, tol)
    c = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]])
    conn = np.array([[0, 1, 2, 3], [1, 2, 3, 4]], dtype=np.int32)  # two tets; second one re-oriented below
    v = np.linalg.det(c[conn[1]][1:] - c[conn[1]][0])
    if v < 0:
        conn[1] = conn[1][[0, 2, 1, 3]]
    orc = ol.Oracle(ol.TET4, c, conn, "hyper_J2", HJ2)
    dut = factory(ol.TET4, c, conn, "hyper_J2", HJ2)
    check_forward(orc, dut, c, "hyper_J2", 0.004, tol)
