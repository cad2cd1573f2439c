"""`-m gpu`: the rows next to the hot path on the device -- Dirichlet and traction
boundary conditions, y = A x, and the C++ Newton / line-search driver -- against the test-side
restatements in fe_driver.py and the reference's regression values."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from fe_driver import Dbc, Primal, Tbc, apply_dbcs, apply_tbcs, block_matrix
from meshes import brick, jiggle, prescribed_fields

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
J2 = [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0]


def cube():
    d = json.load(open(os.path.join(HERE, "golden", "cube_tet4.json")))
    return np.array(d["coords"]), np.array(d["conn"], dtype=np.int32), d["node_sets"], d["side_sets"]


def test_boundary_conditions_and_matvec_match_host_restatement():
    import torch
    from gpu_backend import GpuBackend
    for kind in ("tet4", "hex8"):
        if kind == "tet4":
            c, conn, ns, ss = cube()
            et, model, params = ol.TET4, "elastic", [1000.0, 0.25, 1e-3, 10.0]
            sets = {k: np.array(v) for k, v in ns.items()}
            faces = np.array(ss["ymax"], dtype=np.int32)
        else:
            c, conn, sets = brick(4, 3, 3, 1.0, 0.8, 0.7)
            c = jiggle(c, sets, 0.03)
            et, model, params = ol.HEX8, "small_J2", J2
            # quad faces of the ymax side: nodes of each boundary element face
            faces = np.array([[e[3], e[2], e[6], e[7]] for e in conn if np.isin(e[[3, 2, 6, 7]], sets["ymax"]).all()],
                             dtype=np.int32)
        g = GpuBackend(et, c, conn, model, params)
        a = g.asm
        u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
        z, zp = np.zeros_like(u), np.zeros_like(p)
        ls_h = g.new_linsys()
        assert g.forward_jacobian(u, p, z, zp, g.new_state(), g.new_state(), ls_h) == 0
        d = a.new_linsys()
        for i in range(2):
            d.b[i].copy_(a.dev(ls_h.b[i]))
            for j in range(2):
                d.A[i][j].copy_(a.dev(ls_h.A[i][j]))
        # host restatement
        fn1 = lambda x, y, zz, t: 0.01 * t + 0.1 * x
        trac = lambda x, y, zz, t: (0.3 * t, 0.1 * t + y, -0.2 * x)
        dbcs = [Dbc(0, 0, sets["xmin"], lambda x, y, zz, t: 0.0), Dbc(0, 1, sets["ymin"], fn1), Dbc(1, 0, sets["zmin"], fn1)]
        t = 2.0
        if kind == "tet4":
            apply_tbcs(ls_h, [Tbc(0, faces.tolist(), trac)], c, t)
        apply_dbcs(g, ls_h, dbcs, [u, p], c, t)
        # device
        import calibr8_amd.lib as L
        npf = faces.shape[1]
        pts = np.zeros((len(faces), 1 if npf == 3 else 4, 3))
        L.check(a.L.c8_face_points(npf, len(faces), a.coords.ctypes.data_as(L.dp), faces.ctypes.data_as(L.i32p), pts.ctypes.data_as(L.dp)))
        tr = np.array([[trac(x, y, zz, t) for x, y, zz in fp] for fp in pts])
        if kind == "tet4":
            a.apply_traction([(0, torch.as_tensor(faces, device=a.device), a.dev(tr.ravel()))], d)
        dd = [(b.resid, b.eq, torch.as_tensor(np.asarray(b.nodes, dtype=np.int32), device=a.device),
               a.dev(np.array([b.fn(*c[n], t) for n in b.nodes]))) for b in dbcs]
        a.apply_dirichlet(dd, a.dev(u), a.dev(p), d)
        torch.cuda.synchronize()
        for i in range(2):
            assert np.abs(d.b[i].cpu().numpy() - ls_h.b[i]).max() <= 1e-13 * max(1.0, np.abs(ls_h.b[i]).max())
            for j in range(2):
                assert np.array_equal(d.A[i][j].cpu().numpy(), ls_h.A[i][j])
        if kind == "hex8":  # quad4 traction: constant traction integrates to T * area, spread over the face nodes
            d.b[0].zero_()
            a.apply_traction([(0, torch.as_tensor(faces, device=a.device), a.dev(np.tile([0.0, 2.0, 0.0], len(faces) * 4)))], d)
            by = d.b[0].cpu().numpy().reshape(-1, 3)
            assert abs(by[:, 1].sum() + 2.0 * 1.0 * 0.7) < 1e-12 and np.abs(by[:, [0, 2]]).max() == 0.0
        # y = A x
        rng = np.random.default_rng(1)
        xu, xp = rng.standard_normal(len(u)), rng.standard_normal(len(p))
        yu, yp = a.dev(np.zeros_like(u)), a.dev(np.zeros_like(p))
        a.apply_A(d, a.dev(xu), a.dev(xp), yu, yp)
        ref = block_matrix(g, ls_h) @ np.concatenate([xu, xp])
        got = np.concatenate([yu.cpu().numpy(), yp.cpu().numpy()])
        assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("deck", ["cube_elastic", "cube_hyper_J2", "cube_hyperelasticity", "cube_hyperelasticity_traction"])
def test_reference_regressions_with_device_newton_driver(deck):
    # the reference's primal regression decks (test/primal/*.yaml.in) run end to end through
    # c8_primal_solve_step: HIP assembly + device boundary conditions + C++ Newton/line search
    from calibr8_amd import Assembler, PrimalDriver
    c, conn, ns, ss = cube()
    zero = lambda x, y, z, t: 0.0
    sym = [(0, 0, ns["xmin"], zero), (0, 1, ns["ymin"], zero), (0, 2, ns["zmin"], zero)]
    if deck == "cube_elastic":
        asm = Assembler(4, c, conn, "elastic", [1000.0, 0.25, 1e-3, 10.0])
        pr = PrimalDriver(asm, sym).solve(1)
        expected, tol = 5.00000000000000184e-3, 1e-6
    elif deck == "cube_hyper_J2":
        asm = Assembler(4, c, conn, "hyper_J2", [1000.0, 0.25, 10.0, 0.0, 0.0, 0.0, 0.0, 100.0], max_iters=30)
        pr = PrimalDriver(asm, sym + [(0, 1, ns["ymax"], lambda x, y, z, t: 0.01 * t)]).solve(10)
        expected, tol = 1.57817536611772440e-02, 1e-4
    elif deck == "cube_hyperelasticity":  # Y = 1e5 never yields; the deck allows three Newton iterations per step
        asm = Assembler(4, c, conn, "hyper_J2", [1000.0, 0.25, 100000.0, 0.0, 0.0, 0.0, 0.0, 100.0], max_iters=30)
        pr = PrimalDriver(asm, sym + [(0, 1, ns["ymax"], lambda x, y, z, t: 0.001 * t)], max_iters=3).solve(4)
        expected, tol = 8.34720846455980019e-04, 1e-4
    else:
        asm = Assembler(4, c, conn, "hyper_J2", [1000.0, 0.25, 100000.0, 0.0, 0.0, 0.0, 0.0, 100.0], max_iters=30)
        clamp = [(0, d, ns["ymin"], zero) for d in range(3)]
        pr = PrimalDriver(asm, clamp, [(0, np.array(ss["ymax"], dtype=np.int32), lambda x, y, z, t: (0.0, 0.1 * t, 0.0))],
                          max_iters=10).solve(4)
        expected, tol = 1.61757374785081228e-04, 1e-4
    assert abs(pr.qoi() / expected - 1) < tol, (pr.qoi(), pr.newton_iters)


@pytest.mark.parametrize("deck", ["notch_small_J2", "notch_hyper_J2", "notch_hypo_J2"])
def test_notch_regressions_with_device_newton_driver(deck):
    # the reference's 3-D notch decks (1550 tets read out of test/mesh/notch/notch0.smb, tests/golden/make_notch_fixture.py),
    # four plastic load steps, every assembly, boundary condition and Newton update on the device.  `notch_small_J2`
    # selects `small_hill` with R = 1 (notch_small_J2.yaml.in:21): the reference's 3-D small-strain plasticity regression.
    # The CPU oracle reproduces both pins to 2e-15 (test_oracle_pins.py); the deck tolerance is 1e-4.
    from calibr8_amd import Assembler, PrimalDriver
    d = json.load(open(os.path.join(HERE, "golden", "notch_tet4.json")))
    c, conn, ns = np.array(d["coords"]), np.array(d["conn"], dtype=np.int32), d["node_sets"]
    zero = lambda x, y, z, t: 0.0
    sym = [(0, 0, ns["xmin"], zero), (0, 1, ns["ymin"], zero), (0, 2, ns["zmin"], zero)]
    if deck == "notch_small_J2":
        asm = Assembler(4, c, conn, "small_hill", [1000.0, 0.25, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 10.0, 2.0], max_iters=500)
        pr = PrimalDriver(asm, sym + [(0, 1, ns["ymax"], lambda x, y, z, t: 0.001 * t)]).solve(4)
        expected = 1.4622046563394649e-04
    elif deck == "notch_hypo_J2":  # selects `hypo_hill` (notch_hypo_J2.yaml.in:21); the oracle reproduces the pin to 1.4e-11
        asm = Assembler(4, c, conn, "hypo_hill", [1000.0, 0.25, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 10.0, 2.0], max_iters=500)
        pr = PrimalDriver(asm, sym + [(0, 1, ns["ymax"], lambda x, y, z, t: 0.005 * t)]).solve(4)
        expected = 7.5441386985803955e-04
    else:
        asm = Assembler(4, c, conn, "hyper_J2", [1000.0, 0.25, 10.0, 0.0, 0.0, 0.0, 0.0, 100.0], max_iters=500)
        pr = PrimalDriver(asm, sym + [(0, 1, ns["ymax"], lambda x, y, z, t: 0.005 * t)]).solve(4)
        expected = 7.0080671510235862e-04
    assert abs(pr.qoi() / expected - 1) < 1e-9, (pr.qoi(), pr.newton_iters)


def test_device_newton_driver_matches_oracle_driven_newton():
    # same problem, Newton history and solution: C++ driver on the GPU vs the Python driver on the oracle
    from calibr8_amd import Assembler, PrimalDriver
    c, conn, sets = brick(3, 4, 3, 1.0, 1.5, 1.0)
    c = jiggle(c, sets, 0.05)
    zero = lambda x, y, z, t: 0.0
    pull = lambda x, y, z, t: 0.003 * t
    spec = [(0, 0, sets["ymin"], zero), (0, 1, sets["ymin"], zero), (0, 2, sets["ymin"], zero),
            (0, 1, sets["ymax"], pull), (0, 0, sets["ymax"], zero)]
    orc = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2)
    ref = Primal(orc, c, [Dbc(*s) for s in spec], max_iters=15, abs_tol=1e-10, rel_tol=1e-10).solve(3)
    asm = Assembler(8, c, conn, "small_J2", J2)
    pr = PrimalDriver(asm, spec, max_iters=15, abs_tol=1e-10, rel_tol=1e-10).solve(3)
    assert pr.newton_iters == ref.newton_iters
    for s in range(1, 4):
        assert np.abs(pr.u[s].cpu().numpy() - ref.u[s]).max() < 1e-10 * np.abs(ref.u[s]).max()
        assert np.abs(pr.xi[s].cpu().numpy() - ref.xi[s]).max() < 1e-10
    assert abs(pr.qoi() - ref.qoi()) < 1e-12


def test_line_search_contracts_over_failed_trials_with_two_part_row_sums():
    # a trial step whose local solves fail makes the search contract (primal.cpp:165, line_search.hpp:103-107).  With the
    # row sums in two parts (c8_set_gather_early_nodes, the mode of multi-part meshes) the failed assembly used to leave
    # its second part pending, and the next trial was refused: the step must run as in single-part mode.
    from calibr8_amd import Assembler, PrimalDriver
    c, conn, sets = brick(3, 3, 4, 1.0, 1.0, 1.5)
    zero = lambda x, y, z, t: 0.0
    spec = [(0, 0, sets["xmin"], zero), (0, 1, sets["ymin"], zero), (0, 2, sets["zmin"], zero),
            (0, 2, sets["zmax"], lambda x, y, z, t: 0.12 * t)]
    hill = [1000.0, 0.25, 2.0, 1.0, 1.1, 0.9, 1.05, 0.95, 1.0, 1.0, 50.0]
    # eight local Newton iterations: enough at the converged steps, too few at the first full Newton steps of this load
    orc = ol.Oracle(ol.HEX8, c, conn, "small_hill", hill, max_iters=8)
    failed = [0]
    plain = orc.forward_jacobian

    def counted(*a, **k):
        rc = plain(*a, **k)
        failed[0] += rc != 0
        return rc

    orc.forward_jacobian = counted
    ref = Primal(orc, c, [Dbc(*s) for s in spec], max_iters=25, abs_tol=1e-10, rel_tol=1e-10).solve(2)
    assert failed[0] > 0  # the case does exercise failed trials
    runs = []
    for early in (None, (5, 30)):
        asm = Assembler(8, c, conn, "small_hill", hill, max_iters=8)
        if early:
            asm.set_gather_early_nodes(*early)
        runs.append(PrimalDriver(asm, spec, max_iters=25, abs_tol=1e-10, rel_tol=1e-10).solve(2))
    one, two = runs
    assert one.newton_iters == two.newton_iters == ref.newton_iters
    for s in (1, 2):
        assert np.array_equal(one.u[s].cpu().numpy(), two.u[s].cpu().numpy())  # the two-part row sums are bitwise the same
        assert np.abs(one.u[s].cpu().numpy() - ref.u[s]).max() < 1e-9 * np.abs(ref.u[s]).max()


def test_device_adjoint_driver_gradient():
    # c8_adjoint_solve_step marched backwards = the oracle-driven adjoint (fe_driver.adjoint_gradient),
    # and the reference's finite-difference gradient check with everything on the device
    from calibr8_amd import Assembler, PrimalDriver, adjoint_gradient
    from fe_driver import adjoint_gradient as ref_adjoint_gradient
    c, conn, sets = brick(3, 4, 3, 1.0, 1.5, 1.0)
    c = jiggle(c, sets, 0.05)
    zero = lambda x, y, z, t: 0.0
    spec = [(0, 0, sets["ymin"], zero), (0, 1, sets["ymin"], zero), (0, 2, sets["ymin"], zero),
            (0, 1, sets["ymax"], lambda x, y, z, t: 0.003 * t), (0, 0, sets["ymax"], zero)]
    act = [0, 1, 2, 3]
    base = np.array(J2)

    def solve(params):
        asm = Assembler(8, c, conn, "small_J2", params)
        asm.set_active(0, act)
        return PrimalDriver(asm, spec, max_iters=15, abs_tol=1e-12, rel_tol=1e-12).solve(3)

    pr = solve(base)
    grad = adjoint_gradient(pr, len(act))
    orc = ol.Oracle(ol.HEX8, c, conn, "small_J2", base)
    ref = Primal(orc, c, [Dbc(*s) for s in spec], max_iters=15, abs_tol=1e-12, rel_tol=1e-12).solve(3)
    orc.set_active(0, act)
    gref = ref_adjoint_gradient(ref, len(act))
    assert (np.abs(grad - gref) / np.abs(gref)).max() < 1e-8, (grad, gref)
    direction = np.array([100.0, 0.02, 10.0, 0.2])
    gd = float(grad @ direction)
    errs = []
    for k in range(2, 7):
        h = 10.0 ** (-k)
        pp, pm = base.copy(), base.copy()
        pp[act] += h * direction
        pm[act] -= h * direction
        errs.append(abs((solve(pp).qoi() - solve(pm).qoi()) / (2 * h) - gd))
    # this bar problem is nearly linear in the parameters: central differences agree with the adjoint at every
    # step size (no visible error drop), so the check is the agreement itself
    assert min(errs) < 1e-7 * abs(gd) and max(errs) < 1e-4 * abs(gd), (errs, gd)


def test_device_synthetic_calibration_objective_and_gradient():
    # the reference's inverse-problem set-up in small, everything on the device: "measurements" (surface
    # displacements on xmax, reaction load on ymin) from the true parameters; at perturbed parameters the
    # Calibration objective (calibration.cpp) and its adjoint gradient, checked by central differences
    import torch
    from calibr8_amd import Assembler, PrimalDriver, adjoint_gradient
    c, conn, sets = brick(3, 4, 2, 1.0, 1.5, 1.0)
    zero = lambda x, y, z, t: 0.0
    spec = [(0, 0, sets["xmin"], zero), (0, 1, sets["ymin"], zero), (0, 2, sets["zmin"], zero),
            (0, 1, sets["ymax"], lambda x, y, z, t: 0.002 * t)]
    xmax = set(sets["xmax"].tolist())
    loc = ([0, 1, 2, 3], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7], [4, 5, 6, 7])
    faces = [[int(e[k]) for k in f] for e in conn for f in loc if all(int(e[k]) in xmax for k in f)]
    act, nsteps = [0, 1, 2, 3], 3
    truth = np.array(J2)

    def solve(params, measured=None):
        asm = Assembler(8, c, conn, "small_J2", params)
        asm.set_active(0, act)
        asm.set_qoi_calibration(faces, weights=(1.0, 2.0, 0.5), balance=1e-4, coord_idx=1, coord_value=0.0,
                                coord_tol=1e-8, comp=1, dt_over_T=1.0 / nsteps)
        pr = PrimalDriver(asm, spec, max_iters=15, abs_tol=1e-12, rel_tol=1e-12).solve(nsteps)
        if measured is not None:
            pr.set_measured(*measured)
        return pr

    pt = solve(truth)
    loads = [0.0]
    zero_meas = torch.zeros_like(pt.u[1])
    for s in range(1, nsteps + 1):
        pt.asm.set_measured(zero_meas, 0.0)
        area, total, _ = pt.asm.qoi_preprocess(pt.u[s], pt.p[s], pt.u[s - 1], pt.p[s - 1], pt.xi[s - 1], pt.xi[s])
        loads.append(total)
    assert abs(area - 1.5) < 1e-12 and loads[-1] < -1.0
    measured = ([None] + [u.clone() for u in pt.u[1:]], loads)
    pt.set_measured(*measured)
    assert abs(pt.qoi()) < 1e-18
    base = truth * np.array([1.1, 1.0, 0.8, 0.9, 1.0, 1.0])
    pr = solve(base, measured)
    J0 = pr.qoi()
    assert J0 > 1e-12 and float(pr.xi[-1][:, :, 6].max()) > 1e-4
    grad = adjoint_gradient(pr, len(act))
    direction = np.array([100.0, 0.02, 10.0, 0.2])
    gd = float(grad @ direction)
    errs = []
    for k in (2, 3, 4, 5):
        h = 10.0 ** (-k)
        pp, pm = base.copy(), base.copy()
        pp[act] += h * direction
        pm[act] -= h * direction
        errs.append(abs((solve(pp, measured).qoi() - solve(pm, measured).qoi()) / (2 * h) - gd))
    assert min(errs) < 1e-7 * abs(gd), (errs, gd)
    assert errs[0] > 10 * min(errs), (errs, gd)  # second-order convergence of the difference quotient is visible


def test_device_inverse_problem_recovers_parameters():
    # the reference's synthetic calibration in small, end to end on the device: measurements from the true
    # parameters, Calibration objective, adjoint gradients, canonical [-1, 1] variables, bound-constrained L-BFGS
    import torch
    from calibr8_amd import Assembler, InverseProblem, PrimalDriver
    c, conn, sets = brick(3, 4, 2, 1.0, 1.5, 1.0)
    zero = lambda x, y, z, t: 0.0
    spec = [(0, 0, sets["xmin"], zero), (0, 1, sets["ymin"], zero), (0, 2, sets["zmin"], zero),
            (0, 1, sets["ymax"], lambda x, y, z, t: 0.002 * t)]
    xmax = set(sets["xmax"].tolist())
    loc = ([0, 1, 2, 3], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7], [4, 5, 6, 7])
    faces = [[int(e[k]) for k in f] for e in conn for f in loc if all(int(e[k]) in xmax for k in f)]
    nsteps, truth = 3, np.array(J2)
    measured = [None]

    def make_primal(params):
        asm = Assembler(8, c, conn, "small_J2", params)
        asm.set_qoi_calibration(faces, weights=(1.0, 1.0, 1.0), balance=1e-2, coord_idx=1, coord_value=0.0,
                                coord_tol=1e-8, comp=1, dt_over_T=1.0 / nsteps)
        pr = PrimalDriver(asm, spec, max_iters=20, abs_tol=1e-12, rel_tol=1e-12).solve(nsteps)
        if measured[0] is not None:
            pr.set_measured(*measured[0])
        return pr

    pt = make_primal(truth)
    loads, zm = [0.0], torch.zeros_like(pt.u[1])
    for s in range(1, nsteps + 1):
        pt.asm.set_measured(zm, 0.0)
        loads.append(pt.asm.qoi_preprocess(pt.u[s], pt.p[s], pt.u[s - 1], pt.p[s - 1], pt.xi[s - 1], pt.xi[s])[1])
    measured[0] = ([None] + [u.clone() for u in pt.u[1:]], loads)
    active = [2, 3]  # hardening modulus K and yield strength Y (small_J2.cpp:70-75)
    inv = InverseProblem(make_primal, truth, active, bounds=[[50.0, 200.0], [1.0, 4.0]])
    start = np.array([150.0, 3.0])
    J_start = inv.value_and_gradient(inv.to_canonical(start))[0]
    found, info = inv.solve(start, max_iters=40, grad_tol=1e-14, step_tol=1e-12, max_ls_evals=8)
    assert info["f"] < 1e-8 * J_start, (info, J_start)
    assert np.abs(found / truth[active] - 1.0).max() < 1e-3, (found, info)


def test_femu_objective_and_several_problems_on_device():
    # FEMU_Objective (femu_objective.cpp:13-36): the value of the adjoint objective without its gradient -- finite
    # differences of it reproduce the adjoint gradient; two forward problems sharing the parameters (objective.cpp:16-39)
    # add their objectives and gradients; the value is kept for the last parameter vector (param_diff)
    from calibr8_amd import Assembler, FEMUProblem, InverseProblem, PrimalDriver
    c, conn, sets = brick(3, 3, 2, 1.0, 1.2, 1.0)
    zero = lambda x, y, z, t: 0.0
    calls = [0]

    def problem(rate):
        spec = [(0, 0, sets["xmin"], zero), (0, 1, sets["ymin"], zero), (0, 2, sets["zmin"], zero),
                (0, 1, sets["ymax"], lambda x, y, z, t: rate * t)]

        def make(params):
            calls[0] += 1
            return PrimalDriver(Assembler(8, c, conn, "small_J2", params), spec, max_iters=20, abs_tol=1e-12, rel_tol=1e-12).solve(2)
        return make

    truth, active, bounds = np.array(J2), [2, 3], [[50.0, 200.0], [1.0, 4.0]]
    adj = InverseProblem(problem(0.003), truth, active, bounds)
    femu = FEMUProblem(problem(0.003), truth, active, bounds, fd_step=1e-6)
    # (trial parameters away from the load at which the uniform bar reaches the yield surface: J has a kink there)
    x = adj.to_canonical(np.array([120.0, 2.2]))
    J_a, g_a = adj.value_and_gradient(x)
    J_f, g_f = femu.value_and_gradient(x)
    assert abs(J_f - J_a) < 1e-14 * abs(J_a)
    assert np.abs(g_f - g_a).max() < 1e-4 * np.abs(g_a).max(), (g_f, g_a)
    n = calls[0]
    assert femu.value(x) == J_f and calls[0] == n   # same parameters: no new forward solve
    both = InverseProblem([problem(0.003), problem(0.002)], truth, active, bounds)
    other = InverseProblem(problem(0.002), truth, active, bounds)
    J_b, g_b = both.value_and_gradient(x)
    J_o, g_o = other.value_and_gradient(x)
    assert abs(J_b - (J_a + J_o)) < 1e-13 * abs(J_b) and np.abs(g_b - (g_a + g_o)).max() < 1e-12 * np.abs(g_b).max()


def test_baseline_config1_hex8_elastic_bar_on_device():
    # BASELINE.json configs[0]: the 10^3 hex8 elastic bar, solved with the device Newton driver
    from calibr8_amd import Assembler, PrimalDriver
    c, conn, sets = brick(10, 10, 10)
    zero = lambda x, y, z, t: 0.0
    spec = [(0, k, sets[s], zero) for k, s in enumerate(["xmin", "ymin", "zmin"])]
    asm = Assembler(8, c, conn, "elastic", [1000.0, 0.25, 1e-3, 10.0])
    pr = PrimalDriver(asm, spec, max_iters=5, abs_tol=1e-10, rel_tol=1e-10).solve(1)
    assert abs(pr.qoi() / 5.0e-3 - 1.0) < 1e-9
    assert np.abs(pr.u[1].cpu().numpy().reshape(-1, 3) - 0.01 * c).max() < 1e-11
