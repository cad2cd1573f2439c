"""The kernel source (calibr8_amd/csrc/c8_assemble*.hpp) run lane-by-lane on the CPU must agree
with the oracle to 1e-12 (BASELINE.json tolerance).  This is the no-GPU rehearsal of the
`-m gpu` parity tests in test_gpu_parity.py: the same cases, the same checker."""
import numpy as np
import pytest

import emul_lib as em
from parity_cases import (CASES, MESHES, check_adjoint_chain, check_forward, check_residual, check_tiny_and_ragged,
                          check_two_element_sets, make_pair)

TOL = 1e-12


def factory(et, c, conn, model, params, **kw):
    return em.Emul(et, c, conn, model, params, **kw)


@pytest.mark.parametrize("mesh", MESHES)
@pytest.mark.parametrize("model,params,eps", CASES)
def test_forward_jacobian(mesh, model, params, eps):
    orc, dut, c = make_pair(factory, mesh, model, params)
    check_forward(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("mesh", MESHES)
@pytest.mark.parametrize("model,params,eps", CASES)
def test_residual_only(mesh, model, params, eps):
    orc, dut, c = make_pair(factory, mesh, model, params)
    check_residual(orc, dut, c, eps, TOL)


@pytest.mark.parametrize("model,params,eps", CASES)
def test_residual_only_wave_kernel(model, params, eps):
    # eight hex8 elements per wavefront (residual_wave8); the 4 x 3 x 3 mesh ends with a partial group of 4
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.wave = True
    check_residual(orc, dut, c, eps, TOL)


@pytest.mark.parametrize("mesh", MESHES)
@pytest.mark.parametrize("model,params,eps", CASES)
def test_adjoint_chain(mesh, model, params, eps):
    # K3 -> K4 -> K5 on two consecutive load steps with non-trivial history vectors
    orc, dut, c = make_pair(factory, mesh, model, params)
    check_adjoint_chain(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("model,params,eps", CASES)
def test_forward_jacobian_wave_kernel(model, params, eps):
    # the one-wavefront-per-element hex8 kernel (c8_assemble_wave.hpp)
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.wave = True
    check_forward(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("staged", [False, True])
@pytest.mark.parametrize("model,params,eps", CASES[:2])
def test_forward_jacobian_wave_kernel_iterated_form(model, params, eps, staged):
    # small_J2 has a closed form (radial return, consistent tangent) that the forward wave kernel runs by default
    # (test_forward_jacobian_wave_kernel above); C8_KERNEL_WAVE_AD keeps the local Newton iteration and the AD passes
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.wave, dut.closed, dut.staged = True, False, staged
    check_forward(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("two_part", [False, True])
@pytest.mark.parametrize("model,params,eps", CASES[:2])
def test_forward_jacobian_row_per_node_kernel(model, params, eps, two_part):
    # one wavefront per NODE (c8_assemble_node.hpp): the node's four CSR rows formed from its elements with the model's
    # closed form and written once, no element stage; two_part = half of the nodes first, the rest afterwards
    # (c8_set_gather_early_nodes + c8_gather_finish)
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.node, dut.staged = True, two_part
    check_forward(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("two_part", [False, True])
@pytest.mark.parametrize("model,params,eps", CASES[:2])
def test_adjoint_chain_row_per_node_kernel(model, params, eps, two_part):
    # K3 in the row-per-node form: transposed blocks from the closed form's tangent, right-hand side -dJ/dx + f +
    # (dxi/dx)^T g with (dxi/dx)^T g in closed form (no elimination of dC/dxi); K4 and K5 are the wave kernels
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.node, dut.wave, dut.staged = True, True, two_part
    check_adjoint_chain(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("force_many", [False, True])
def test_row_per_node_kernel_nodes_with_many_elements(force_many):
    # a node with sixteen elements goes through the kernel eight elements at a time (its second form, with separate
    # accumulator storage); force_many runs that form on a mesh that does not need it
    import oracle_lib as ol
    from meshes import pinched_bricks
    from parity_cases import mesh_of
    if force_many:
        et, c, conn = mesh_of("hex8")
    else:
        et = ol.HEX8
        c, conn = pinched_bricks()
    orc = ol.Oracle(et, c, conn, "small_J2", CASES[1][1])
    dut = em.Emul(et, c, conn, "small_J2", CASES[1][1])
    dut.node = dut.wave = True
    em.lib().c8emu_set_node_many(1 if force_many else 0)
    try:
        check_forward(orc, dut, c, "small_J2", 0.004, TOL)
        check_adjoint_chain(orc, dut, c, "small_J2", 0.004, TOL)
    finally:
        em.lib().c8emu_set_node_many(0)


def test_adjoint_row_per_node_kernel_against_iterated_form():
    # K3 of small_J2 in its two forms on the same stored states (second load step, plastic history, non-zero g): the
    # row-per-node kernel (closed-form tangent and (dxi/dx)^T g) against the wave kernel (dual numbers, elimination of dC/dxi)
    import oracle_lib as ol
    from parity_cases import J2, mesh_of, two_steps
    et, c, conn = mesh_of("hex8")
    orc = ol.Oracle(et, c, conn, "small_J2", J2)
    st = two_steps(orc, c, 0.004)
    (u, p, xi), (up, pp, xip) = st[2], st[1]
    rng = np.random.default_rng(3)
    g0 = 1e-3 * rng.standard_normal((orc.nelems, orc.npts, orc.nloc))
    f0 = 1e-3 * rng.standard_normal((orc.nelems, orc.npts, 4 * orc.nn))
    out = []
    for node in (True, False):
        dut = em.Emul(et, c, conn, "small_J2", J2)
        dut.wave, dut.node = True, node
        ls, g = dut.new_linsys(), g0.copy()
        assert dut.adjoint_jacobian(u, p, up, pp, xip, xi, g, f0, ls) == 0
        # K4 and K5 on the same states with a non-zero adjoint: closed forms (node) against dual numbers + elimination (wave)
        z_u, z_p = 1e-2 * np.cos(np.arange(len(u))), 1e-2 * np.sin(np.arange(len(p)))
        phi, g4, f4 = np.full_like(g0, 7.0), g.copy(), np.full_like(f0, 7.0)
        dut.solve_adjoint_local(u, p, up, pp, xip, xi, z_u, z_p, phi, g4, f4)
        assert np.abs(f4).max() == 0.0  # small strain: no dependence on the previous displacement
        dut.set_active(0, [0, 1, 2, 3])
        grad = dut.qoi_gradient(u, p, up, pp, xip, xi, z_u, z_p, phi, 4)
        out.append([g] + ls.b + [ls.A[i][j] for i in range(2) for j in range(2)] + [phi, g4, grad])
    for a, b in zip(*out):
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
    assert np.array_equal(out[0][0], g0)  # the average-displacement objective leaves g as it is
    assert (out[0][-3][:, :, 6] != out[0][-2][:, :, 6]).mean() > 0.3  # plastic points: g' differs from phi in the alpha entry


def test_row_per_node_kernel_assign_mode_and_ragged_meshes():
    import oracle_lib as ol
    from meshes import brick, notched_bar, prescribed_fields
    from parity import compare_systems, rel_vec
    # assign mode on stale content = accumulate-into on zeros
    orc, dut, c = make_pair(factory, "hex8", "small_J2", CASES[1][1])
    dut.node = True
    u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
    z, zp = np.zeros_like(u), np.zeros_like(p)
    ls_o, ls_d, xo, xd = orc.new_linsys(), dut.new_linsys(), orc.new_state(), dut.new_state()
    assert orc.forward_jacobian(u, p, z, zp, orc.new_state(), xo, ls_o) == 0
    for i in range(2):
        ls_d.b[i][:] = 7.0
        for j in range(2):
            ls_d.A[i][j][:] = -3.0
    dut.assign = True
    assert dut.forward_jacobian(u, p, z, zp, dut.new_state(), xd, ls_d) == 0
    errs = compare_systems(orc, ls_d, ls_o)
    errs["xi"] = rel_vec(xd, xo)
    assert max(errs.values()) < TOL, errs
    # accumulate-into: a second assembly doubles the system
    dut.assign = False
    assert dut.forward_jacobian(u, p, z, zp, dut.new_state(), xd, ls_d) == 0
    assert np.abs(ls_d.A[0][0] - 2 * ls_o.A[0][0]).max() < 1e-12 * np.abs(ls_o.A[0][0]).max()
    # one element, partial groups, and the notched bar (node degrees 8 .. 27, nodes with 1 .. 8 elements)
    for mesh in (brick(1, 1, 1), brick(3, 1, 1, 3.0), brick(2, 3, 1, 2.0, 3.0), notched_bar(8, 6, 2)):
        cc, conn = mesh[0], mesh[1]
        orc = ol.Oracle(ol.HEX8, cc, conn, "small_J2", CASES[1][1])
        dut = em.Emul(ol.HEX8, cc, conn, "small_J2", CASES[1][1])
        dut.node = True
        check_forward(orc, dut, cc, "small_J2", 0.0035, TOL)


@pytest.mark.parametrize("params", [[1000.0, 0.25, 100.0, 2.0, 0.0, 0.0], [1000.0, 0.25, 0.0, 2.0, 0.0, 0.0],
                                    [1000.0, 0.25, 5000.0, 0.5, 0.0, 0.0]])
def test_closed_form_against_iterated_form(params):
    # the two forms of the small_J2 kernel on the same two load steps (the second from a plastic state): same state,
    # residual and Jacobian to the local Newton tolerance -- with hardening, perfectly plastic and with stiff hardening
    import oracle_lib as ol
    from meshes import prescribed_fields
    from parity_cases import mesh_of
    et, c, conn = mesh_of("hex8")
    res = []
    for closed in (True, False):
        dut = em.Emul(et, c, conn, "small_J2", params)
        dut.wave, dut.closed = True, closed
        u1, p1 = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
        u0, p0 = np.zeros_like(u1), np.zeros_like(p1)
        xi0, xi1, xi2 = dut.new_state(), dut.new_state(), dut.new_state()
        l1, l2 = dut.new_linsys(), dut.new_linsys()
        assert dut.forward_jacobian(u1, p1, u0, p0, xi0, xi1, l1) == 0
        assert dut.forward_jacobian(1.5 * u1, 1.5 * p1, u1, p1, xi1, xi2, l2) == 0
        assert (xi2[:, :, 6] > xi1[:, :, 6]).any() and (xi1[:, :, 6] > 0).any() and (xi1[:, :, 6] == 0).any()
        res.append([xi1, xi2] + [l.b[i] for l in (l1, l2) for i in range(2)] +
                   [l.A[i][j] for l in (l1, l2) for i in range(2) for j in range(2)])
    for a, b in zip(*res):
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()


@pytest.mark.parametrize("model,params,eps", CASES)
def test_adjoint_chain_wave_kernel(model, params, eps):
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.wave = True
    check_adjoint_chain(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("model,params,eps", [CASES[1], CASES[3]])
def test_wave_kernels_without_shape_cache(model, params, eps):
    # the wave kernels read dN/dx, w dv and h from tables built once per context (c8_set_shape_cache, default and the
    # path every other test here takes); without the tables they compute them per element: same results
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.wave = True
    em.lib().c8emu_set_shape_cache(0)
    try:
        check_forward(orc, dut, c, model, eps, TOL)
        check_adjoint_chain(orc, dut, c, model, eps, TOL)
    finally:
        em.lib().c8emu_set_shape_cache(1)


@pytest.mark.parametrize("mesh", MESHES)
def test_staged_assembly_assign_mode(mesh):
    # c8_set_assign_mode: the row sums assign A and b; on stale content the result is the accumulate-into result on zeros
    import numpy as np
    from meshes import prescribed_fields
    orc, dut, c = make_pair(factory, mesh, "small_J2", CASES[1][1])
    dut.wave = mesh == "hex8"
    dut.staged = True
    u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
    z, zp = np.zeros_like(u), np.zeros_like(p)
    ref, xi_ref = dut.new_linsys(), dut.new_state()
    assert dut.forward_jacobian(u, p, z, zp, dut.new_state(), xi_ref, ref) == 0
    dut.assign = True
    ls = dut.new_linsys()
    for i in range(2):
        ls.b[i][:] = 3.25
        for j in range(2):
            ls.A[i][j][:] = -1.5
    assert dut.forward_jacobian(u, p, z, zp, dut.new_state(), dut.new_state(), ls) == 0
    for i in range(2):
        assert np.array_equal(ls.b[i], ref.b[i])
        for j in range(2):
            assert np.array_equal(ls.A[i][j], ref.A[i][j])


@pytest.mark.parametrize("model,params,eps", CASES)
def test_staged_gather_assembly(model, params, eps):
    # staged assembly: element matrices stored element-major, rows summed per node (gather_node_rows)
    orc, dut, c = make_pair(factory, "hex8", model, params)
    dut.wave = dut.staged = True
    check_forward(orc, dut, c, model, eps, TOL)
    check_adjoint_chain(orc, dut, c, model, eps, TOL)


@pytest.mark.parametrize("kind", ["tet4", "hex8"])
def test_staged_gather_assembly_slot_kernels(kind):
    # staged assembly through the slot-per-lane kernels: tet4 K1 and K3 (transposed through LDS), hex8 K1
    from parity_cases import J2
    orc, dut, c = make_pair(factory, kind, "small_J2", J2)
    dut.staged = True
    check_forward(orc, dut, c, "small_J2", 0.004, TOL)
    if kind == "tet4":
        check_adjoint_chain(orc, dut, c, "small_J2", 0.004, TOL)


def test_staged_assembly_goes_round_the_ring():
    # a long thin brick: 16 chunks through a ring of three, node rows summed as their last chunk completes
    import emul_lib
    import oracle_lib as ol
    from meshes import brick, jiggle
    from parity_cases import J2
    c, conn, sets = brick(2, 2, 16, 0.4, 0.4, 3.0)
    c = jiggle(c, sets, 0.03)
    orc, dut = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2), factory(ol.HEX8, c, conn, "small_J2", J2)
    dut.wave = dut.staged = True
    check_forward(orc, dut, c, "small_J2", 0.004, TOL)
    assert emul_lib.lib().c8emu_last_nchunks() > 4
    check_adjoint_chain(orc, dut, c, "small_J2", 0.004, TOL)


@pytest.mark.parametrize("kind,wave", [("hex8", False), ("hex8", True), ("tet4", False)])
def test_two_element_sets(kind, wave):
    check_two_element_sets(factory, kind, TOL, wave=wave)


def test_tiny_and_ragged_meshes():
    check_tiny_and_ragged(factory, TOL)


@pytest.mark.parametrize("kind,wave", [("hex8", True), ("hex8", "node"), ("hex8", False), ("tet4", False)])
def test_calibration_objective(kind, wave):
    # Calibration QoI (calibration.cpp) through the kernel source on the CPU: set-up tables, preprocess (total load),
    # value, and the x / xi / parameter derivatives through K3 -> K4 -> K5, against the oracle
    import numpy as np
    import oracle_lib as ol
    from parity import compare_systems, rel_vec
    from parity_cases import J2, mesh_of, two_steps
    et, c, conn = mesh_of(kind)
    xmax, ymin = c[:, 0].max(), c[:, 1].min()
    loc = ([0, 1, 2], [0, 1, 3], [1, 2, 3], [0, 2, 3]) if et == ol.TET4 else \
        ([0, 1, 2, 3], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7], [4, 5, 6, 7])
    faces = [[int(e[k]) for k in f] for e in conn for f in loc if all(abs(c[e[k], 0] - xmax) < 0.06 for k in f)]
    faces = [f for f in faces if all(abs(c[n, 0] - xmax) < 0.06 for n in f)]
    assert faces
    kw = dict(weights=(1.0, 2.0, 0.5), balance=0.3, coord_idx=1, coord_value=float(ymin), coord_tol=0.06, comp=1, dt_over_T=0.5)
    orc = ol.Oracle(et, c, conn, "small_J2", J2)
    dut = factory(et, c, conn, "small_J2", J2)
    dut.wave = bool(wave)
    dut.node = wave == "node"  # K3 in the row-per-node form: the load term's derivatives in closed form, g updated afterwards
    orc.set_calibration(faces, **kw)
    dut.set_calibration(faces, **kw)
    st = two_steps(orc, c, 0.004)
    (u, p, xi), (up, pp, xip) = st[2], st[1]
    rng = np.random.default_rng(3)
    u_meas = u + 1e-4 * rng.standard_normal(len(u))
    for b in (orc, dut):
        b.set_active(0, [0, 1, 2, 3])
        b.set_measured(u_meas, -0.7)
    po, pd = orc.qoi_preprocess(u, p, up, pp, xip, xi), dut.qoi_preprocess(u, p, up, pp, xip, xi)
    assert np.abs(po - pd).max() < 1e-12 * max(1.0, np.abs(po).max()), (po, pd)
    Jo, Jd = orc.eval_qoi(u, p), dut.eval_qoi(u, p)
    assert abs(Jo - Jd) < 1e-12 * abs(Jo), (Jo, Jd)
    res = []
    for b in (orc, dut):
        g = np.full((orc.nelems, orc.npts, orc.nloc), 0.01)
        f = np.full((orc.nelems, orc.npts, 4 * orc.nn), 0.02)
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
    assert max(errs.values()) < TOL, errs
