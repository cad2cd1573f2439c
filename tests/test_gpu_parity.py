"""`-m gpu` parity tests: the HIP path, called through the C ABI (libc8.so), against the CPU
oracle on the same seeded inputs.  Bar (BASELINE.json): residuals, every Jacobian entry and the
updated local state within 1e-12 relative (metrics in parity.py)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from meshes import brick, jiggle, prescribed_fields
from parity import compare_systems, rel_vec
from parity_cases import EL, HJ2, J2

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-12


from parity_cases import (CASES, MESHES, check_adjoint_chain, check_forward, check_residual, check_tiny_and_ragged,
                          check_two_element_sets, make_pair)


def hex_mesh(n=(6, 5, 4)):
    c, conn, sets = brick(n[0], n[1], n[2], 1.0, 0.8, 0.7)
    return jiggle(c, sets, 0.03), conn


def both(et, c, conn, model, params, scatter):
    from gpu_backend import GpuBackend
    return ol.Oracle(et, c, conn, model, params), GpuBackend(et, c, conn, model, params, scatter=scatter)


def factory(scatter, kernel="auto"):
    def make(et, c, conn, model, params, **kw):
        from gpu_backend import GpuBackend
        return GpuBackend(et, c, conn, model, params, scatter=scatter, kernel=kernel, **kw)
    return make


@pytest.mark.parametrize("model,params,eps", CASES)
def test_adjoint_chain_slot_kernel_hex8(model, params, eps):
    orc, gpu, c = make_pair(factory("colored", "slot"), "hex8", model, params)
    check_adjoint_chain(orc, gpu, c, model, eps, TOL)


@pytest.mark.parametrize("scatter", ["colored", "atomic"])
@pytest.mark.parametrize("model,params,eps", CASES)
def test_forward_jacobian_slot_kernel_hex8(model, params, eps, scatter):
    # hex8 defaults to the wave-per-element kernel; the slot-per-lane kernel stays covered here
    orc, gpu, c = make_pair(factory(scatter, "slot"), "hex8", model, params)
    check_forward(orc, gpu, c, model, eps, TOL)


@pytest.mark.parametrize("scatter", ["colored", "atomic"])
@pytest.mark.parametrize("mesh", MESHES)
@pytest.mark.parametrize("model,params,eps", CASES)
def test_forward_jacobian_matches_oracle(mesh, model, params, eps, scatter):
    orc, gpu, c = make_pair(factory(scatter), mesh, model, params)
    assert gpu.npts == orc.npts and gpu.nloc == orc.nloc
    for i in range(2):
        for j in range(2):
            assert np.array_equal(gpu.rowptr[i][j], orc.rowptr[i][j])
            assert np.array_equal(gpu.colidx[i][j], orc.colidx[i][j])
    check_forward(orc, gpu, c, model, eps, TOL)


@pytest.mark.parametrize("model,params,eps", CASES)
def test_initial_local_state_matches_oracle(model, params, eps):
    # init_variables_impl of every registered model (c8_init_variables)
    orc, gpu, c = make_pair(factory("colored"), "hex8", model, params)
    assert np.array_equal(gpu.new_state(), orc.new_state())


@pytest.mark.parametrize("mesh", MESHES)
@pytest.mark.parametrize("model,params,eps", CASES)
def test_residual_only_matches_oracle(mesh, model, params, eps):
    orc, gpu, c = make_pair(factory("colored"), mesh, model, params)
    check_residual(orc, gpu, c, eps, TOL)


@pytest.mark.parametrize("model,params,eps", CASES)
def test_residual_only_wave_kernel_hex8(model, params, eps):
    # atomic / staged modes route hex8 residual-only assembly to the eight-elements-per-wavefront kernel
    orc, gpu, c = make_pair(factory("atomic"), "hex8", model, params)
    check_residual(orc, gpu, c, eps, TOL)


@pytest.mark.parametrize("scatter", ["colored", "atomic"])
@pytest.mark.parametrize("mesh", MESHES)
@pytest.mark.parametrize("model,params,eps", CASES)
def test_adjoint_chain_matches_oracle(mesh, model, params, eps, scatter):
    # eval_adjoint_jacobian -> solve_adjoint_local -> eval_qoi_gradient (+ eval_qoi), two steps
    orc, gpu, c = make_pair(factory(scatter), mesh, model, params)
    check_adjoint_chain(orc, gpu, c, model, eps, TOL)


def test_adjoint_gradient_fd_check_through_gpu():
    # the reference's gradient check (main_inverse.cpp:126-158) with every assembly on the GPU
    from fe_driver import Dbc, Primal, adjoint_gradient
    from gpu_backend import GpuBackend
    c, conn, sets = brick(3, 4, 3, 1.0, 1.5, 1.0)
    c = jiggle(c, sets, 0.05)
    z = lambda x, y, zz, t: 0.0
    dbcs = [Dbc(0, 0, sets["ymin"], z), Dbc(0, 1, sets["ymin"], z), Dbc(0, 2, sets["ymin"], z),
            Dbc(0, 1, sets["ymax"], lambda x, y, zz, t: 0.003 * t), Dbc(0, 0, sets["ymax"], z)]
    base = np.array(J2)
    act = [0, 1, 2, 3]

    def objective(params):
        be = GpuBackend(ol.HEX8, c, conn, "small_J2", params)
        pr = Primal(be, c, dbcs, max_iters=15, abs_tol=1e-12, rel_tol=1e-12).solve(3)
        return pr.qoi(), pr

    J0, pr = objective(base)
    assert pr.xi[-1][:, :, 6].max() > 1e-4
    pr.be.set_active(0, act)
    grad = adjoint_gradient(pr, len(act))
    direction = np.array([100.0, 0.02, 10.0, 0.2])
    gd = float(grad @ direction)
    errs = []
    for k in range(1, 8):
        h = 10.0 ** (-k)
        pp, pm = base.copy(), base.copy()
        pp[act] += h * direction
        pm[act] -= h * direction
        errs.append(abs((objective(pp)[0] - objective(pm)[0]) / (2 * h) - gd))
    errs = np.array(errs)
    assert np.log10(errs.max() / errs.min()) > 5.0 and errs.min() < 1e-6 * abs(gd), (errs, gd)


@pytest.mark.parametrize("model,params,eps", CASES)
def test_staged_gather_assembly_hex8(model, params, eps):
    # scatter mode GATHER: element matrices staged element-major, rows summed per node, no atomics
    orc, gpu, c = make_pair(factory("gather", "wave"), "hex8", model, params)
    check_forward(orc, gpu, c, model, eps, TOL)
    check_adjoint_chain(orc, gpu, c, model, eps, TOL)


@pytest.mark.parametrize("model,params,eps", CASES)
def test_staged_gather_assembly_tet4(model, params, eps):
    # tet4 (the reference's element type): K1 and K3 of the slot-per-lane kernels into the stage, rows gathered
    orc, gpu, c = make_pair(factory("gather"), "tet4", model, params)
    check_forward(orc, gpu, c, model, eps, TOL)
    check_adjoint_chain(orc, gpu, c, model, eps, TOL)


@pytest.mark.parametrize("model,params,eps", [CASES[1], CASES[3]])
def test_wave_kernels_without_shape_cache(model, params, eps):
    # c8_set_shape_cache(ctx, 0): the wave kernels compute the shape tables per call instead of reading the cached ones
    orc, gpu, c = make_pair(factory("atomic", "wave"), "hex8", model, params)
    gpu.asm.set_shape_cache(False)
    check_forward(orc, gpu, c, model, eps, TOL)
    check_adjoint_chain(orc, gpu, c, model, eps, TOL)
    gpu.asm.set_shape_cache(True)
    check_forward(orc, gpu, c, model, eps, TOL)


def test_staged_gather_in_two_parts():
    # c8_set_gather_early_nodes: the assembly call sums the rows of an early node range only (the ghost rows of a mesh
    # part), c8_gather_finish the rest; the result is bitwise the one of the single-call staged assembly
    import torch
    from calibr8_amd import Assembler
    c, conn = hex_mesh((6, 5, 4))
    asm = Assembler(8, c, conn, "small_J2", J2, scatter="gather")
    u_h, p_h = prescribed_fields(c, 0.004, ramp=True)
    u, p = asm.dev(u_h), asm.dev(p_h)
    z, zp = torch.zeros_like(u), torch.zeros_like(p)
    xi0 = asm.new_state()
    ref, xi_ref = asm.new_linsys(), asm.new_state()
    assert asm.forward_jacobian(u, p, z, zp, xi0, xi_ref, ref) == 0
    nn = len(c)
    for lo, hi in ((nn // 3, nn // 2), (0, nn), (nn - 1, nn)):
        asm.set_gather_early_nodes(lo, hi)
        ls, xi = asm.new_linsys(), asm.new_state()
        assert asm.forward_jacobian(u, p, z, zp, xi0, xi, ls) == 0
        # after the first part exactly the early rows are complete
        r3 = asm.rowptr[0][0]
        a, b = int(r3[3 * lo]), int(r3[3 * hi])
        assert torch.equal(ls.A[0][0][a:b], ref.A[0][0][a:b]) and torch.equal(ls.b[0][3 * lo:3 * hi], ref.b[0][3 * lo:3 * hi])
        if hi - lo < nn:
            assert not torch.equal(ls.flat, ref.flat)
            with pytest.raises(RuntimeError, match="c8_gather_finish"):
                asm.forward_jacobian(u, p, z, zp, xi0, xi, ls)
        assert asm.gather_finish() == 0
        assert torch.equal(ls.flat, ref.flat) and torch.equal(xi, xi_ref)
    asm.set_gather_early_nodes(0, 0)
    ls = asm.new_linsys()
    assert asm.forward_jacobian(u, p, z, zp, xi0, asm.new_state(), ls) == 0 and torch.equal(ls.flat, ref.flat)


def test_staged_gather_assign_mode():
    # c8_set_assign_mode: zero_all + assembly in one call.  On any initial content of A and b the result is bitwise the
    # accumulate-into result on a zeroed system (every node of this mesh has elements); a Jacobian assembly in another
    # scatter mode is refused while the mode is on.
    import torch
    from calibr8_amd import Assembler
    c, conn = hex_mesh((5, 4, 4))
    asm = Assembler(8, c, conn, "small_J2", J2, scatter="gather")
    u_h, p_h = prescribed_fields(c, 0.004, ramp=True)
    u, p = asm.dev(u_h), asm.dev(p_h)
    z, zp = torch.zeros_like(u), torch.zeros_like(p)
    xi0 = asm.new_state()
    ref, xi_ref = asm.new_linsys(), asm.new_state()
    assert asm.forward_jacobian(u, p, z, zp, xi0, xi_ref, ref) == 0
    asm.set_assign_mode(True)
    ls = asm.new_linsys()
    ls.flat.fill_(123.456)  # stale content
    assert asm.forward_jacobian(u, p, z, zp, xi0, asm.new_state(), ls) == 0
    assert torch.equal(ls.flat, ref.flat)
    assert asm.forward_jacobian(u, p, z, zp, xi0, asm.new_state(), ls) == 0  # again: assigned, not doubled
    assert torch.equal(ls.flat, ref.flat)
    # K3 too
    g = torch.zeros(asm.nelems, asm.npts, asm.nloc, dtype=torch.float64, device=asm.device)
    f = torch.zeros(asm.nelems, asm.npts, asm.ndofs, dtype=torch.float64, device=asm.device)
    ls.flat.fill_(-7.0)
    asm.adjoint_jacobian(u, p, z, zp, xi0, xi_ref, g, f, ls)
    asm.set_assign_mode(False)
    ref3 = asm.new_linsys()
    asm.adjoint_jacobian(u, p, z, zp, xi0, xi_ref, torch.zeros_like(g), f, ref3)
    assert torch.equal(ls.flat, ref3.flat)
    asm.set_assign_mode(True)
    b_before = ls.b[0].clone()
    asm.global_residual(u, p, z, zp, xi0, xi_ref, ls)  # the residual-only assembly keeps adding
    assert not torch.equal(ls.b[0], b_before)
    asm.set_scatter("atomic")
    with pytest.raises(RuntimeError, match="assign mode"):
        asm.forward_jacobian(u, p, z, zp, xi0, asm.new_state(), ls)


@pytest.mark.parametrize("model,params,eps", CASES[:2])
def test_row_per_node_kernel_hex8(model, params, eps):
    # C8_KERNEL_NODE: one wavefront per node forms the node's rows from its elements (closed form) and writes them once
    orc, gpu, c = make_pair(factory("gather", "node"), "hex8", model, params)
    check_forward(orc, gpu, c, model, eps, TOL)


@pytest.mark.parametrize("model,params,eps", CASES[:2])
def test_row_per_node_kernel_adjoint_chain_hex8(model, params, eps):
    # K3 in the row-per-node form (transposed blocks, closed-form (dxi/dx)^T g), then K4 and K5 on its outputs
    orc, gpu, c = make_pair(factory("gather", "node"), "hex8", model, params)
    check_adjoint_chain(orc, gpu, c, model, eps, TOL)


def test_adjoint_row_per_node_kernel_against_iterated_form():
    # K3 of small_J2 on a 12^3 brick in its two forms on the same stored states (plastic history, non-zero g and f): the
    # row-per-node kernel (closed form) against the staged wave kernel (dual numbers); bitwise reproducible
    import torch
    from calibr8_amd import Assembler
    c, conn = hex_mesh((12, 12, 12))
    u_h, p_h = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
    res = []
    for kernel in ("node", "wave", "node"):
        asm = Assembler(8, c, conn, "small_J2", J2, scatter="gather")
        asm.set_kernel(kernel)
        u1, p1 = asm.dev(u_h), asm.dev(p_h)
        z, zp = torch.zeros_like(u1), torch.zeros_like(p1)
        xi0, xi1, xi2 = asm.new_state(), asm.new_state(), asm.new_state()
        assert asm.forward_jacobian(u1, p1, z, zp, xi0, xi1, asm.new_linsys()) == 0
        assert asm.forward_jacobian(1.5 * u1, 1.5 * p1, u1, p1, xi1, xi2, asm.new_linsys()) == 0
        gen = torch.Generator(device="cpu").manual_seed(5)
        g = (1e-3 * torch.randn(asm.nelems, asm.npts, asm.nloc, generator=gen, dtype=torch.float64)).to(asm.device)
        f = (1e-3 * torch.randn(asm.nelems, asm.npts, asm.ndofs, generator=gen, dtype=torch.float64)).to(asm.device)
        g_in = g.clone()
        ls = asm.new_linsys()
        assert asm.adjoint_jacobian(1.5 * u1, 1.5 * p1, u1, p1, xi1, xi2, g, f, ls) == 0
        assert torch.equal(g, g_in)  # average displacement: dJ/dxi = 0
        # K4 on the same states with a non-zero adjoint z: closed form (node / auto) against dual numbers + elimination (wave)
        zu, zpp = (1e-2 * torch.randn(u1.shape, generator=gen, dtype=torch.float64)).to(asm.device), \
            (1e-2 * torch.randn(p1.shape, generator=gen, dtype=torch.float64)).to(asm.device)
        phi = torch.full_like(g, 7.0)
        f.fill_(7.0)
        assert asm.solve_adjoint_local(1.5 * u1, 1.5 * p1, u1, p1, xi1, xi2, zu, zpp, phi, g, f) == 0
        assert float(f.abs().max()) == 0.0  # small strain: no dependence on the previous displacement
        res.append((ls.flat.clone(), phi.clone(), g.clone()))
    for k in range(3):
        assert torch.equal(res[0][k], res[2][k])
        assert float((res[0][k] - res[1][k]).abs().max() / res[1][k].abs().max()) < 1e-12
    assert float((res[0][1][:, :, 6] != res[0][2][:, :, 6]).double().mean()) > 0.3  # plastic points: g' differs from phi in the alpha entry


def test_row_per_node_kernel_ragged_meshes_sets_and_refusals():
    import torch
    from calibr8_amd import Assembler
    from gpu_backend import GpuBackend
    from meshes import notched_bar
    for n in ((1, 1, 1), (3, 1, 1), (5, 1, 1), (3, 3, 1)):  # one element; nodes with one, two and four elements only
        c, conn, _ = brick(n[0], n[1], n[2], 1.0 * n[0], 1.0 * n[1], 1.0 * n[2])
        check_forward(ol.Oracle(ol.HEX8, c, conn, "small_J2", J2), GpuBackend(ol.HEX8, c, conn, "small_J2", J2, scatter="gather", kernel="node"),
                      c, "small_J2", 0.0035, TOL)
    check_two_element_sets(factory("gather", "node"), "hex8", TOL)
    c, conn, _ = notched_bar(10, 6, 3)  # node degrees 8 .. 27, nodes with 1 .. 8 elements
    orc, gpu = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2), GpuBackend(ol.HEX8, c, conn, "small_J2", J2, scatter="gather", kernel="node")
    check_forward(orc, gpu, c, "small_J2", 0.004, TOL)
    check_adjoint_chain(orc, gpu, c, "small_J2", 0.004, TOL)
    from meshes import pinched_bricks
    c, conn = pinched_bricks()  # a node with sixteen elements: the kernel's form that takes a node's elements eight at a time
    orc, gpu = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2), GpuBackend(ol.HEX8, c, conn, "small_J2", J2, scatter="gather", kernel="node")
    check_forward(orc, gpu, c, "small_J2", 0.004, TOL)
    check_adjoint_chain(orc, gpu, c, "small_J2", 0.004, TOL)
    # it is what the default takes for this model and element, bit for bit, and two runs agree bit for bit
    c, conn = hex_mesh((5, 4, 3))
    u_h, p_h = prescribed_fields(c, 0.004, ramp=True)
    res = []
    for kernel in ("auto", "node", "node", "wave"):
        asm = Assembler(8, c, conn, "small_J2", J2, scatter="gather")
        asm.set_kernel(kernel)
        u, p = asm.dev(u_h), asm.dev(p_h)
        ls, xi = asm.new_linsys(), asm.new_state()
        assert asm.forward_jacobian(u, p, torch.zeros_like(u), torch.zeros_like(p), asm.new_state(), xi, ls) == 0
        res.append((ls.flat.clone(), xi.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[1][0], res[2][0]) and torch.equal(res[0][1], res[2][1])
    # the staged one-wavefront-per-element form computes the same system in another order
    assert float((res[3][0] - res[0][0]).abs().max() / res[0][0].abs().max()) < 1e-13 and not torch.equal(res[3][0], res[0][0])
    # where it does not apply: another scatter mode, previous and new state in one array, a model without a closed form
    asm = Assembler(8, c, conn, "small_J2", J2, scatter="atomic")
    asm.set_kernel("node")
    u, p = asm.dev(u_h), asm.dev(p_h)
    with pytest.raises(RuntimeError, match="C8_KERNEL_NODE"):
        asm.forward_jacobian(u, p, torch.zeros_like(u), torch.zeros_like(p), asm.new_state(), asm.new_state(), asm.new_linsys())
    asm = Assembler(8, c, conn, "small_J2", J2, scatter="gather")
    xi = asm.new_state()
    ls = asm.new_linsys()
    assert asm.forward_jacobian(u, p, torch.zeros_like(u), torch.zeros_like(p), xi, xi, ls) == 0  # default: the staged form takes over
    assert float((ls.flat - res[0][0]).abs().max() / res[0][0].abs().max()) < 1e-13
    with pytest.raises(RuntimeError, match="closed form"):
        Assembler(8, c, conn, "hyper_J2", HJ2).set_kernel("node")


def test_staged_gather_slot_kernel_hex8():
    orc, gpu, c = make_pair(factory("gather", "slot"), "hex8", "small_J2", J2)
    check_forward(orc, gpu, c, "small_J2", 0.004, TOL)


def test_staged_gather_ring_pipeline():
    # small chunks: 20+ chunks through the ring of three, gathers overlapping later chunks on their own stream
    from gpu_backend import GpuBackend
    c, conn, sets = brick(3, 3, 40, 0.3, 0.3, 4.0)
    c = jiggle(c, sets, 0.02)
    orc = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2)
    gpu = GpuBackend(ol.HEX8, c, conn, "small_J2", J2, scatter="gather")
    gpu.asm.set_stage_chunk(4)
    check_forward(orc, gpu, c, "small_J2", 0.004, TOL)
    check_adjoint_chain(orc, gpu, c, "small_J2", 0.004, TOL)


def test_staged_gather_row_sums_beside_next_chunk():
    # c8_set_stage_overlap: the row sums of chunk k on the context's second stream beside the assembly of chunk k + 1
    # (events both ways, ring slots reused only after their row sums): parity with the oracle, and the same bits as the
    # one-after-the-other form, over repeated calls into the same system
    from gpu_backend import GpuBackend
    c, conn, sets = brick(3, 3, 40, 0.3, 0.3, 4.0)
    c = jiggle(c, sets, 0.02)
    orc = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2)
    gpu = GpuBackend(ol.HEX8, c, conn, "small_J2", J2, scatter="gather")
    gpu.asm.set_stage_chunk(4)
    gpu.asm.set_stage_overlap(True)
    check_forward(orc, gpu, c, "small_J2", 0.004, TOL)
    check_adjoint_chain(orc, gpu, c, "small_J2", 0.004, TOL)
    u, p = prescribed_fields(c, 0.004, ramp=True)
    z, zp = np.zeros_like(u), np.zeros_like(p)
    out = []
    for on in (True, False):
        gpu.asm.set_stage_overlap(on)
        ls = gpu.new_linsys()
        for _ in range(3):  # accumulates: three assemblies into the same system
            assert gpu.forward_jacobian(u, p, z, zp, gpu.new_state(), gpu.new_state(), ls) == 0
        out.append(ls)
    for i in range(2):
        assert np.array_equal(out[0].b[i], out[1].b[i])
        for j in range(2):
            assert np.array_equal(out[0].A[i][j], out[1].A[i][j])


def test_staged_gather_two_sets_and_reproducible():
    check_two_element_sets(factory("gather", "wave"), "hex8", TOL)
    # two runs are bitwise identical (fixed summation order)
    from gpu_backend import GpuBackend
    from parity_cases import mesh_of
    c, conn = hex_mesh((5, 4, 3))
    g = GpuBackend(ol.HEX8, c, conn, "small_J2", J2, scatter="gather")
    u, p = prescribed_fields(c, 0.004, ramp=True)
    z, zp = np.zeros_like(u), np.zeros_like(p)
    out = []
    for _ in range(2):
        ls, xi = g.new_linsys(), g.new_state()
        assert g.forward_jacobian(u, p, z, zp, g.new_state(), xi, ls) == 0
        out.append(ls)
    for i in range(2):
        assert np.array_equal(out[0].b[i], out[1].b[i])
        for j in range(2):
            assert np.array_equal(out[0].A[i][j], out[1].A[i][j])
    # the slot-per-lane adjoint kernel of hex8 cannot stage (it holds rows, not columns): refused loudly
    gs = GpuBackend(ol.HEX8, c, conn, "small_J2", J2, scatter="gather", kernel="slot")
    with pytest.raises(RuntimeError, match="wave-per-element"):
        gs.adjoint_jacobian(u, p, z, zp, gs.new_state(), xi, np.zeros((gs.nelems, gs.npts, gs.nloc)),
                            np.zeros((gs.nelems, gs.npts, 4 * gs.nn)), gs.new_linsys())


@pytest.mark.parametrize("kind,kernel", [("hex8", "wave"), ("hex8", "slot"), ("tet4", "auto")])
@pytest.mark.parametrize("scatter", ["colored", "atomic"])
def test_two_element_sets(kind, kernel, scatter):
    check_two_element_sets(factory(scatter, kernel), kind, TOL)


@pytest.mark.parametrize("kernel", ["wave", "slot"])
def test_tiny_and_ragged_meshes(kernel):
    check_tiny_and_ragged(factory("colored", kernel), TOL)


def test_forward_jacobian_accumulates_into_outputs():
    # outputs are += (global_residual.cpp:463-479, :556-586): assembling twice doubles them
    c, conn = hex_mesh((3, 3, 3))
    orc, gpu = both(ol.HEX8, c, conn, "small_J2", J2, "colored")
    u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
    z, zp = np.zeros_like(u), np.zeros_like(p)
    ls1, ls2 = gpu.new_linsys(), gpu.new_linsys()
    gpu.forward_jacobian(u, p, z, zp, gpu.new_state(), gpu.new_state(), ls1)
    gpu.forward_jacobian(u, p, z, zp, gpu.new_state(), gpu.new_state(), ls2)
    gpu.forward_jacobian(u, p, z, zp, gpu.new_state(), gpu.new_state(), ls2)
    assert rel_vec(ls2.b[0], 2 * ls1.b[0]) < 1e-14 and rel_vec(ls2.A[0][0], 2 * ls1.A[0][0]) < 1e-14


def test_local_solve_failure_returns_minus_one():
    # one local Newton iteration is not enough on a plastic point: -1, like evaluations.cpp:95-97
    from gpu_backend import GpuBackend
    c, conn = hex_mesh((3, 3, 3))
    gpu = GpuBackend(ol.HEX8, c, conn, "small_J2", J2, max_iters=1)
    orc = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2, max_iters=1)
    u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
    z, zp = np.zeros_like(u), np.zeros_like(p)
    assert orc.forward_jacobian(u, p, z, zp, orc.new_state(), orc.new_state(), orc.new_linsys()) == -1
    assert gpu.forward_jacobian(u, p, z, zp, gpu.new_state(), gpu.new_state(), gpu.new_linsys()) == -1
    # and the context recovers
    ue, pe = prescribed_fields(c, 0.0005, perturb=5e-2)
    assert gpu.forward_jacobian(ue, pe, z, zp, gpu.new_state(), gpu.new_state(), gpu.new_linsys()) == 0


@pytest.mark.parametrize("scatter", ["gather", "colored", "atomic"])
@pytest.mark.parametrize("model,params,eps", CASES[:2])
def test_forward_jacobian_iterated_form_hex8(model, params, eps, scatter):
    # small_J2 on hex8 runs its closed form by default (every other hex8 small_J2 test of this file); C8_KERNEL_WAVE_AD
    # keeps the local Newton iteration and the AD passes in the same kernel
    orc, gpu, c = make_pair(factory(scatter, "wave_ad"), "hex8", model, params)
    check_forward(orc, gpu, c, model, eps, TOL)


@pytest.mark.parametrize("params", [[1000.0, 0.25, 100.0, 2.0, 0.0, 0.0], [1000.0, 0.25, 0.0, 2.0, 0.0, 0.0],
                                    [1000.0, 0.25, 5000.0, 0.5, 0.0, 0.0]])
def test_closed_form_against_iterated_form(params):
    # both forms of the kernel on two load steps of a 12^3 brick (the second from a plastic state): same state, residual
    # and Jacobian to the local Newton tolerance; then the variant switch back to the default
    from gpu_backend import GpuBackend
    c, conn = hex_mesh((12, 12, 12))
    gpu = GpuBackend(ol.HEX8, c, conn, "small_J2", params, scatter="gather")
    u1, p1 = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
    u0, p0 = np.zeros_like(u1), np.zeros_like(p1)

    def run(kernel):
        gpu.asm.set_kernel(kernel)
        xi0, xi1, xi2 = gpu.new_state(), gpu.new_state(), gpu.new_state()
        l1, l2 = gpu.new_linsys(), gpu.new_linsys()
        assert gpu.forward_jacobian(u1, p1, u0, p0, xi0, xi1, l1) == 0
        assert gpu.forward_jacobian(1.5 * u1, 1.5 * p1, u1, p1, xi1, xi2, l2) == 0
        assert (xi2[:, :, 6] > xi1[:, :, 6]).any() and (xi1[:, :, 6] > 0).any() and (xi1[:, :, 6] == 0).any()
        return [xi1, xi2] + [l.b[i] for l in (l1, l2) for i in range(2)] + \
               [l.A[i][j] for l in (l1, l2) for i in range(2) for j in range(2)]

    # "auto" = the row-per-node kernel (closed form), "wave" = the staged one-wavefront-per-element kernel (closed form),
    # "wave_ad" = the latter with the iterated, automatically differentiated local solve
    closed, iterated, again, staged = run("auto"), run("wave_ad"), run("auto"), run("wave")
    for a, b, a2, a3 in zip(closed, iterated, again, staged):
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
        assert np.array_equal(a, a2)  # bitwise reproducible, also after a switch of the kernel variant and back
        assert np.abs(a - a3).max() <= 1e-13 * np.abs(a3).max()  # the two closed-form kernels: the same sums in another order


@pytest.mark.parametrize("scatter", ["gather", "atomic"])
def test_closed_form_against_iterated_form_tet4(scatter):
    # the lane-group kernel's two instantiations on a tet mesh (the reference's element type): C8_KERNEL_AUTO runs small_J2's
    # closed form, an explicit C8_KERNEL_SLOT the local Newton iteration with the AD passes -- same state, residual and
    # Jacobian to the local Newton tolerance over two load steps, the second from a plastic state; a context whose local
    # Newton budget is below eight iterations always iterates
    from gpu_backend import GpuBackend
    from calibr8_amd import brick_mesh
    from parity_cases import mesh_of
    et, c0, conn0 = mesh_of("tet4")
    n = 10  # 6000 tets: every brick cell split into six
    hc, hconn = brick_mesh(n, n, n)
    T = np.array([[0, 1, 2, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 7, 4, 6], [0, 4, 5, 6], [0, 5, 1, 6]])
    conn = np.ascontiguousarray(hconn[:, T].reshape(-1, 4).astype(np.int32))
    X = hc[conn]
    vol = np.einsum("ij,ij->i", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0])
    conn[vol < 0] = conn[vol < 0][:, [0, 2, 1, 3]]
    gpu = GpuBackend(ol.TET4, hc, conn, "small_J2", J2, scatter=scatter)
    u1, p1 = prescribed_fields(hc, 0.004, ramp=True, perturb=5e-2)
    u0, p0 = np.zeros_like(u1), np.zeros_like(p1)

    def run(g, kernel):
        g.asm.set_kernel(kernel)
        xi0, xi1, xi2 = g.new_state(), g.new_state(), g.new_state()
        l1, l2 = g.new_linsys(), g.new_linsys()
        assert g.forward_jacobian(u1, p1, u0, p0, xi0, xi1, l1) == 0
        assert g.forward_jacobian(1.5 * u1, 1.5 * p1, u1, p1, xi1, xi2, l2) == 0
        assert (xi2[:, :, 6] > xi1[:, :, 6]).any() and (xi1[:, :, 6] > 0).any() and (xi1[:, :, 6] == 0).any()
        return [xi1, xi2] + [l.b[i] for l in (l1, l2) for i in range(2)] + \
               [l.A[i][j] for l in (l1, l2) for i in range(2) for j in range(2)]

    closed, iterated = run(gpu, "auto"), run(gpu, "slot")
    differs = False
    for a, b in zip(closed, iterated):
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
        differs = differs or not np.array_equal(a, b)
    assert differs  # two different kernels did run
    few = GpuBackend(ol.TET4, hc, conn, "small_J2", J2, scatter=scatter, max_iters=6)
    if scatter == "gather":  # (bitwise reproducible mode) the small budget selects the iterated form under AUTO too
        for a, b in zip(run(few, "auto"), iterated):
            assert np.array_equal(a, b)


def test_iterated_form_needs_the_wave_kernel():
    from calibr8_amd import C8Error
    from gpu_backend import GpuBackend
    from parity_cases import mesh_of
    et, c, conn = mesh_of("tet4")
    gpu = GpuBackend(et, c, conn, "small_J2", J2)
    with pytest.raises(C8Error):
        gpu.asm.set_kernel("wave_ad")


def test_cube_elastic_pin_through_gpu():
    # reference regression primal/cube_elastic.yaml.in:40-41 reproduced with the HIP assembly
    from fe_driver import Dbc, Primal
    from gpu_backend import GpuBackend
    d = json.load(open(os.path.join(HERE, "golden", "cube_tet4.json")))
    c, conn, ns = np.array(d["coords"]), np.array(d["conn"], dtype=np.int32), d["node_sets"]
    gpu = GpuBackend(ol.TET4, c, conn, "elastic", EL)
    dbcs = [Dbc(0, k, ns[s], lambda x, y, z, t: 0.0) for k, s in enumerate(["xmin", "ymin", "zmin"])]
    pr = Primal(gpu, c, dbcs).solve(1)
    assert abs(pr.qoi() / 5.00000000000000184e-3 - 1) < 1e-6


def test_large_brick_properties():
    # BASELINE-size behaviour through size-independent properties on a 40^3 brick (64k hex8):
    # (1) colour-batched and atomic scatter agree to rounding, (2) J x matches a directional
    # finite difference of R, (3) rigid translation of u leaves R unchanged (small strain).
    from gpu_backend import GpuBackend
    c, conn, sets = brick(40, 40, 40)
    gc = GpuBackend(ol.HEX8, c, conn, "small_J2", J2, scatter="colored")
    ga = GpuBackend(ol.HEX8, c, conn, "small_J2", J2, scatter="atomic")
    u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
    z, zp = np.zeros_like(u), np.zeros_like(p)
    lc, la = gc.new_linsys(), ga.new_linsys()
    xc, xa = gc.new_state(), ga.new_state()
    assert gc.forward_jacobian(u, p, z, zp, gc.new_state(), xc, lc) == 0
    assert ga.forward_jacobian(u, p, z, zp, ga.new_state(), xa, la) == 0
    errs = compare_systems(gc, la, lc)
    assert max(errs.values()) < 1e-13 and np.array_equal(xc, xa), errs
    import scipy.sparse as sp
    # the directional derivative is taken at an all-plastic state (no branch switches inside the
    # finite-difference stencil; the elastic/plastic kink is not differentiable)
    up, ppl = prescribed_fields(c, 0.008, ramp=False, perturb=1e-3)
    lq, xq = gc.new_linsys(), gc.new_state()
    assert gc.forward_jacobian(up, ppl, z, zp, gc.new_state(), xq, lq) == 0
    assert (xq[:, :, 6] > 0).all()
    A00 = sp.csr_matrix((lq.A[0][0], gc.colidx[0][0], gc.rowptr[0][0]))
    A10 = sp.csr_matrix((lq.A[1][0], gc.colidx[1][0], gc.rowptr[1][0]), shape=(len(p), len(u)))
    rng = np.random.default_rng(5)
    v = rng.standard_normal(len(u))
    v /= np.abs(v).max()
    h = 1e-7
    lp, lm = gc.new_linsys(), gc.new_linsys()
    gc.forward_jacobian(up + h * v, ppl, z, zp, gc.new_state(), gc.new_state(), lp)
    gc.forward_jacobian(up - h * v, ppl, z, zp, gc.new_state(), gc.new_state(), lm)
    fd_u, fd_p = (lp.b[0] - lm.b[0]) / (2 * h), (lp.b[1] - lm.b[1]) / (2 * h)
    assert np.abs(A00 @ v - fd_u).max() < 1e-5 * np.abs(A00 @ v).max()
    assert np.abs(A10 @ v - fd_p).max() < 1e-5 * np.abs(A10 @ v).max()
    lt = gc.new_linsys()
    shift = np.tile([0.3, -0.2, 0.1], len(p))
    gc.forward_jacobian(u + shift, p, z, zp, gc.new_state(), gc.new_state(), lt)
    assert rel_vec(lt.b[0], lc.b[0]) < 1e-12 and rel_vec(lt.A[0][0], lc.A[0][0]) < 1e-12


def _calibration_pair(kind):
    """oracle and GPU backend with the Calibration objective on the same mesh: displacement side set = the xmax
    boundary faces, load plane y = 0, component y"""
    from gpu_backend import GpuBackend
    from parity_cases import mesh_of
    et, c, conn = mesh_of(kind) if kind == "tet4" else (ol.HEX8,) + tuple(hex_mesh((3, 4, 2)))
    xmax, ymin = c[:, 0].max(), c[:, 1].min()
    loc_faces = ([0, 1, 2], [0, 1, 3], [1, 2, 3], [0, 2, 3]) if et == ol.TET4 else \
        ([0, 1, 2, 3], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7], [4, 5, 6, 7])
    faces = [[int(e[k]) for k in f] for e in conn for f in loc_faces if all(abs(c[e[k], 0] - xmax) < 1e-9 for k in f)]
    assert faces
    kw = dict(weights=(1.0, 2.0, 0.5), balance=0.3, coord_idx=1, coord_value=float(ymin), coord_tol=1e-6, comp=1,
              dt_over_T=0.5)
    orc = ol.Oracle(et, c, conn, "small_J2", J2)
    gpu = GpuBackend(et, c, conn, "small_J2", J2, scatter="atomic")
    orc.set_calibration(faces, **kw)
    gpu.set_calibration(faces, **kw)
    return orc, gpu, c


@pytest.mark.parametrize("kind", ["hex8", "tet4"])
def test_calibration_objective_matches_oracle(kind):
    # Calibration QoI (calibration.cpp): preprocess (total load), value, and its x / xi / parameter derivatives
    # through K3 -> K4 -> K5, against the oracle
    from parity_cases import two_steps
    orc, gpu, c = _calibration_pair(kind)
    if kind == "tet4":
        pytest.importorskip("numpy")
    st = two_steps(orc, c, 0.004)
    (u, p, xi), (up, pp, xip) = st[2], st[1]
    rng = np.random.default_rng(3)
    u_meas = u + 1e-4 * rng.standard_normal(len(u))
    for b in (orc, gpu):
        b.set_active(0, [0, 1, 2, 3])
        b.set_measured(u_meas, -0.7)
    po = orc.qoi_preprocess(u, p, up, pp, xip, xi)
    pg = gpu.qoi_preprocess(u, p, up, pp, xip, xi)
    assert np.abs(po - pg).max() < 1e-12 * max(1.0, np.abs(po).max()), (po, pg)
    assert abs(po[2]) > 1e-3  # a real load mismatch
    Jo, Jg = orc.eval_qoi(u, p), gpu.eval_qoi(u, p)
    assert abs(Jo - Jg) < 1e-12 * abs(Jo), (Jo, Jg)
    nd = 4 * orc.nn
    res = []
    for b in (orc, gpu):
        g = rng.standard_normal((orc.nelems, orc.npts, orc.nloc)) * 0 + 0.01
        f = np.full((orc.nelems, orc.npts, nd), 0.02)
        ls = b.new_linsys()
        b.adjoint_jacobian(u, p, up, pp, xip, xi, g, f, ls)
        z_u, z_p = np.linspace(-1e-3, 1e-3, len(u)), np.linspace(2e-3, -1e-3, len(p))
        phi = np.zeros_like(g)
        b.solve_adjoint_local(u, p, up, pp, xip, xi, z_u, z_p, phi, g, f)
        grad = b.qoi_gradient(u, p, up, pp, xip, xi, z_u, z_p, phi, 4)
        res.append((ls, g, f, phi, grad))
    (lo, go, fo, pho, gro), (lg, gg, fg, phg, grg) = res
    errs = compare_systems(orc, lg, lo)
    errs["g"], errs["f"], errs["phi"] = rel_vec(gg, go), rel_vec(fg, fo), rel_vec(phg, pho)
    errs["grad"] = float(np.abs(grg - gro).max() / np.abs(gro).max())
    assert max(errs.values()) < TOL, errs


def test_full_size_million_elements_properties():
    # BASELINE.json's full size (100^3 hex8 = 1 M elements, 4.4e8 non-zeros) through size-independent properties,
    # all on the device:
    #  (1) the three scatter modes agree to rounding and the staged mode is bitwise reproducible,
    #  (2) global equilibrium: the nodal internal forces sum to zero in every direction (partition of unity),
    #  (3) rigid translations: R unchanged, and J t = 0 for a constant displacement field t,
    #  (4) J v equals a directional finite difference of R at an all-plastic state.
    import torch
    from calibr8_amd import Assembler, brick_mesh
    n = 100
    coords, conn = brick_mesh(n, n, n)
    asm = Assembler(8, coords, conn, "small_J2", J2, scatter="atomic")
    u_h, p_h = prescribed_fields(coords, 0.004, ramp=True)  # the bench state: about half of the points plastic
    u, p = asm.dev(u_h), asm.dev(p_h)
    z, zp = torch.zeros_like(u), torch.zeros_like(p)
    xi0 = asm.new_state()

    def assemble(mode, uu=u, pp=p):
        asm.set_scatter(mode)
        ls, xi = asm.new_linsys(), asm.new_state()
        assert asm.forward_jacobian(uu, pp, z, zp, xi0, xi, ls) == 0
        return ls, xi

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())

    la, xa = assemble("atomic")
    lg, xg = assemble("gather")
    lg2, _ = assemble("gather")
    assert torch.equal(lg.flat, lg2.flat)  # bitwise reproducible
    # the row-per-node kernel sums the interpolation over the nodes in a rotated order (skewed shape table): same state to rounding
    assert rel(xg, xa) < 1e-13
    assert rel(lg.flat, la.flat) < 1e-13
    del lg2
    lc, _ = assemble("colored")
    assert rel(lc.flat, la.flat) < 1e-13
    del lc, lg
    frac = float((xa[:, :, 6] > 0).double().mean())
    assert 0.3 < frac < 0.7
    # (2)
    Ru = la.b[0].view(-1, 3)
    assert float(Ru.sum(0).abs().max()) < 1e-9 * float(Ru.abs().sum())
    # (3)
    shift = torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64, device=u.device).repeat(len(p))
    lt, _ = assemble("atomic", u + shift)
    # nodal differences of 4e-5 (strain 4e-3 x edge 1e-2) taken from values of 0.3: 1e-16 * 0.3 / 4e-5 ~ 1e-12 of
    # cancellation in the strains, so the bar here is 1e-10, not the parity bar
    assert rel(lt.b[0], la.b[0]) < 1e-10 and rel(lt.A[0][0], la.A[0][0]) < 1e-10
    del lt
    for k in range(3):
        t = torch.zeros_like(u).view(-1, 3)
        t[:, k] = 1.0
        yu, yp = torch.zeros_like(u), torch.zeros_like(p)
        asm.apply_A(la, t.view(-1).contiguous(), zp, yu, yp)
        scale = float(la.A[0][0].abs().max())
        assert float(yu.abs().max()) < 1e-10 * scale and float(yp.abs().max()) < 1e-10 * scale
    # (4)
    up_h, pp_h = prescribed_fields(coords, 0.008, ramp=False, perturb=1e-3)
    up, pp = asm.dev(up_h), asm.dev(pp_h)
    lq, xq = assemble("atomic", up, pp)
    assert bool((xq[:, :, 6] > 0).all())
    g = torch.Generator(device="cpu").manual_seed(5)
    v = torch.randn(len(u_h), generator=g, dtype=torch.float64).to(u.device)
    v /= v.abs().max()
    h = 1e-7
    lp, _ = assemble("atomic", up + h * v, pp)
    lm, _ = assemble("atomic", up - h * v, pp)
    fd_u, fd_p = (lp.b[0] - lm.b[0]) / (2 * h), (lp.b[1] - lm.b[1]) / (2 * h)
    yu, yp = torch.zeros_like(u), torch.zeros_like(p)
    asm.apply_A(lq, v, zp, yu, yp)
    assert rel(fd_u, yu) < 1e-5 and rel(fd_p, yp) < 1e-5


def test_full_size_million_elements_adjoint_path_properties():
    # BASELINE config 3, adjoint half, at full size (100^3 hex8): K3 / K4 / K5 checked, not only timed.
    #  (1) K3 with zero histories assembles the transpose of K1's Jacobian: y . (A_K3 x) = x . (A_K1 y);
    #  (2) K3 staged and K3 with atomic adds agree; the staged mode is bitwise reproducible;
    #  (3) K4 (per-point outputs) at 200 random elements equals the ORACLE on those elements as stand-alone meshes;
    #  (4) K5: the two independent kernels (eight elements per wavefront / one lane group per element) agree over the
    #      whole mesh, and the gradient is affine in (z, phi).
    import torch
    from calibr8_amd import Assembler, brick_mesh
    n = 100
    coords, conn = brick_mesh(n, n, n)
    asm = Assembler(8, coords, conn, "small_J2", J2)
    u_h, p_h = prescribed_fields(coords, 0.004, ramp=True)
    u, p = asm.dev(u_h), asm.dev(p_h)
    z, zp = torch.zeros_like(u), torch.zeros_like(p)
    xi0, xi = asm.new_state(), asm.new_state()
    dot = lambda a, b: float((a * b).sum())
    l1 = asm.new_linsys()
    assert asm.forward_jacobian(u, p, z, zp, xi0, xi, l1) == 0
    g0 = torch.zeros(asm.nelems, asm.npts, asm.nloc, dtype=torch.float64, device=u.device)
    f0 = torch.zeros(asm.nelems, asm.npts, asm.ndofs, dtype=torch.float64, device=u.device)
    # (1), (2)
    l3 = asm.new_linsys()
    g = g0.clone()
    assert asm.adjoint_jacobian(u, p, z, zp, xi0, xi, g, f0, l3) == 0
    l3b, gb = asm.new_linsys(), g0.clone()
    assert asm.adjoint_jacobian(u, p, z, zp, xi0, xi, gb, f0, l3b) == 0
    assert torch.equal(l3.flat, l3b.flat) and torch.equal(g, gb)
    del l3b
    asm.set_scatter("atomic")
    l3a, ga = asm.new_linsys(), g0.clone()
    assert asm.adjoint_jacobian(u, p, z, zp, xi0, xi, ga, f0, l3a) == 0
    assert float((l3a.flat - l3.flat).abs().max() / l3.flat.abs().max()) < 1e-13
    del l3a, ga
    asm.set_scatter("gather")
    gen = torch.Generator(device="cpu").manual_seed(9)
    rnd = lambda t: torch.randn(t.shape, generator=gen, dtype=torch.float64).to(u.device)
    xu, xp, yu, yp = rnd(u), rnd(p), rnd(u), rnd(p)
    a3u, a3p, a1u, a1p = torch.zeros_like(u), torch.zeros_like(p), torch.zeros_like(u), torch.zeros_like(p)
    asm.apply_A(l3, xu, xp, a3u, a3p)
    asm.apply_A(l1, yu, yp, a1u, a1p)
    lhs, rhs = dot(yu, a3u) + dot(yp, a3p), dot(xu, a1u) + dot(xp, a1p)
    scale = float(a3u.abs().max()) * float(yu.abs().sum())
    assert abs(lhs - rhs) < 1e-12 * scale, (lhs, rhs, scale)
    del l1, l3, a3u, a3p, a1u, a1p
    # (3) K4 at full size against the oracle on sampled elements
    z_u, z_p = rnd(u) * 1e-3, rnd(p) * 1e-3
    g_in = rnd(g0) * 1e-2
    phi, g4, f4 = torch.zeros_like(g0), g_in.clone(), f0.clone()
    assert asm.solve_adjoint_local(u, p, z, zp, xi0, xi, z_u, z_p, phi, g4, f4) == 0
    rng = np.random.default_rng(4)
    sample = np.sort(rng.choice(asm.nelems, 200, replace=False))
    nodes = conn[sample].ravel()                      # every sampled element as a mesh of its own: 8 private nodes
    sc, sconn = coords[nodes], np.arange(len(nodes), dtype=np.int32).reshape(-1, 8)
    orc = ol.Oracle(ol.HEX8, sc, sconn, "small_J2", J2)
    take3 = lambda t: np.ascontiguousarray(t.cpu().numpy().reshape(-1, 3)[nodes].ravel())
    take1 = lambda t: np.ascontiguousarray(t.cpu().numpy()[nodes])
    st = lambda t: np.ascontiguousarray(t.cpu().numpy()[sample])
    su, sp_ = take3(u), take1(p)
    phi_o, g_o, f_o = np.zeros((200, 8, 7)), st(g_in), np.zeros((200, 8, 32))
    orc.solve_adjoint_local(su, sp_, 0 * su, 0 * sp_, st(xi0), st(xi), take3(z_u), take1(z_p), phi_o, g_o, f_o)
    assert rel_vec(st(phi), phi_o) < 1e-12 and rel_vec(st(g4), g_o) < 1e-12 and np.abs(st(f4) - f_o).max() < 1e-12
    # (4) K5
    asm.set_active(0, [0, 1, 2, 3])
    def grad_of(zu_, zp_, phi_, kernel="auto"):
        asm.set_kernel(kernel)
        gr = torch.zeros(4, dtype=torch.float64, device=u.device)
        assert asm.qoi_gradient(u, p, z, zp, xi0, xi, zu_, zp_, phi_, gr) == 0
        asm.set_kernel("auto")
        return gr.cpu().numpy()
    g_w, g_s = grad_of(z_u, z_p, phi), grad_of(z_u, z_p, phi, "slot")
    g_00 = grad_of(torch.zeros_like(z_u), torch.zeros_like(z_p), torch.zeros_like(phi))
    g_2 = grad_of(2 * z_u, 2 * z_p, 2 * phi)
    mag = np.abs(g_w).max()
    assert np.abs(g_w).min() > 0 and np.abs(g_w - g_s).max() < 1e-11 * mag, (g_w, g_s)  # 8e6 terms summed in another order
    assert np.abs((g_2 - g_00) - 2 * (g_w - g_00)).max() < 1e-11 * mag


def test_full_size_million_tets_properties():
    # the reference's element type at a million elements (56^3 hexes split into 6 tets each, 1.05 M tet4):
    # scatter modes agree, staged mode bitwise reproducible, global equilibrium, rigid-translation null space
    import torch
    from calibr8_amd import Assembler, brick_mesh
    n = 56
    coords, hexes = brick_mesh(n, n, n)
    tets = [[0, 1, 2, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 7, 4, 6], [0, 4, 5, 6], [0, 5, 1, 6]]  # Kuhn split
    conn = np.concatenate([hexes[:, t] for t in tets]).astype(np.int32)
    asm = Assembler(4, coords, conn, "small_J2", J2, scatter="atomic")
    u_h, p_h = prescribed_fields(coords, 0.004, ramp=True)
    u, p = asm.dev(u_h), asm.dev(p_h)
    z, zp = torch.zeros_like(u), torch.zeros_like(p)
    xi0 = asm.new_state()

    def assemble(mode):
        asm.set_scatter(mode)
        ls, xi = asm.new_linsys(), asm.new_state()
        assert asm.forward_jacobian(u, p, z, zp, xi0, xi, ls) == 0
        return ls, xi

    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    la, xa = assemble("atomic")
    lg, xg = assemble("gather")
    lg2, _ = assemble("gather")
    lc, _ = assemble("colored")
    assert torch.equal(lg.flat, lg2.flat)  # bitwise reproducible
    # the row-per-node kernel sums the interpolation over the nodes in a rotated order (skewed shape table): same state to rounding
    assert rel(xg, xa) < 1e-13
    assert rel(lg.flat, la.flat) < 1e-13 and rel(lc.flat, la.flat) < 1e-13
    assert 0.3 < float((xa[:, :, 6] > 0).double().mean()) < 0.7
    Ru = la.b[0].view(-1, 3)
    assert float(Ru.sum(0).abs().max()) < 1e-9 * float(Ru.abs().sum())
    for k in range(3):
        t = torch.zeros_like(u).view(-1, 3)
        t[:, k] = 1.0
        yu, yp = torch.zeros_like(u), torch.zeros_like(p)
        asm.apply_A(la, t.view(-1).contiguous(), zp, yu, yp)
        scale = float(la.A[0][0].abs().max())
        assert float(yu.abs().max()) < 1e-10 * scale and float(yp.abs().max()) < 1e-10 * scale


def test_baseline_config2_bar_100k_matches_oracle():
    # BASELINE.json configs[1] (SURVEY.md 8d "Config 2"): 20 x 20 x 250 hex8 bar, h = 0.05, small_J2 with the
    # parameters of notch2D_small_J2_adjoint_check.yaml.in:27-33, prescribed ramped state: the whole 100k-element
    # assembly against the oracle (one element slice per host thread and colour)
    from gpu_backend import GpuBackend
    c, conn, sets = brick(20, 20, 250, 1.0, 1.0, 12.5)
    orc = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2)
    gpu = GpuBackend(ol.HEX8, c, conn, "small_J2", J2, scatter="atomic")
    u, p = prescribed_fields(c, 0.004, ramp=True)
    z, zp = np.zeros_like(u), np.zeros_like(p)
    lo, lg, xo, xg = orc.new_linsys(), gpu.new_linsys(), orc.new_state(), gpu.new_state()
    assert orc.forward_jacobian(u, p, z, zp, orc.new_state(), xo, lo, nthreads=min(16, os.cpu_count() or 1)) == 0
    assert gpu.forward_jacobian(u, p, z, zp, gpu.new_state(), xg, lg) == 0
    errs = compare_systems(orc, lg, lo)
    errs["xi"] = rel_vec(xg, xo)
    assert max(errs.values()) < TOL, errs
    assert 0.2 < (xo[:, :, 6] > 0).mean() < 0.99  # elastic and plastic points both present


@pytest.mark.parametrize("model,params,eps", [("small_J2", J2, 0.004), ("hyper_J2", HJ2, 0.004)])
@pytest.mark.parametrize("scatter", ["gather", "atomic", "colored"])
def test_notched_specimen_matches_oracle(model, params, eps, scatter):
    # BASELINE config 3's geometry in small: a double-edge-notched hex8 bar (elements removed from a brick, node degrees
    # varying along the notch flanks -- the unstructured case of the graphs, the colouring and the staged assembly),
    # every entry point against the oracle at 1e-12
    from gpu_backend import GpuBackend
    from meshes import notched_bar
    c, conn, sets = notched_bar(24, 12, 6)
    c = jiggle(c, sets, 0.01)
    orc = ol.Oracle(ol.HEX8, c, conn, model, params)
    gpu = GpuBackend(ol.HEX8, c, conn, model, params, scatter=scatter)
    for i in range(2):
        for j in range(2):
            assert np.array_equal(gpu.rowptr[i][j], orc.rowptr[i][j]) and np.array_equal(gpu.colidx[i][j], orc.colidx[i][j])
    deg = np.diff(orc.rowptr[1][1])
    assert deg.min() < 27 and len(np.unique(deg)) > 4   # not a brick
    check_forward(orc, gpu, c, model, eps, TOL)
    check_adjoint_chain(orc, gpu, c, model, eps, TOL)


def test_full_size_notched_specimen_properties():
    # BASELINE config 3 at its size: 1.0 M hex8 elements of a double-edge-notched bar (290 x 64 x 64 brick minus the
    # notches), small_J2, primal and adjoint assembly, through size-independent properties on the device:
    #  (1) staged assembly bitwise reproducible and equal to the atomic one to rounding, states equal;
    #  (2) global equilibrium of the internal forces; J t = 0 for rigid translations;
    #  (3) K3 with zero histories is K1's transpose: y . (A_K3 x) = x . (A_K1 y);
    #  (4) K1 at 200 sampled elements (as stand-alone meshes) equals the ORACLE's element matrices at 1e-12.
    import torch
    from calibr8_amd import Assembler
    from meshes import notched_bar
    coords, conn, _ = notched_bar(290, 64, 64)
    assert 0.95e6 < len(conn) < 1.06e6
    asm = Assembler(8, coords, conn, "small_J2", J2)
    assert asm.scatter == "gather"
    u_h, p_h = prescribed_fields(coords, 0.004, ramp=True)
    u, p = asm.dev(u_h), asm.dev(p_h)
    z, zp = torch.zeros_like(u), torch.zeros_like(p)
    xi0, xi = asm.new_state(), asm.new_state()
    l1 = asm.new_linsys()
    assert asm.forward_jacobian(u, p, z, zp, xi0, xi, l1) == 0
    frac = float((xi[:, :, 6] > 0).double().mean())
    assert 0.2 < frac < 0.8
    l1b, xib = asm.new_linsys(), asm.new_state()
    assert asm.forward_jacobian(u, p, z, zp, xi0, xib, l1b) == 0
    assert torch.equal(l1.flat, l1b.flat) and torch.equal(xi, xib)
    asm.set_scatter("atomic")
    l1b.zero()
    assert asm.forward_jacobian(u, p, z, zp, xi0, xib, l1b) == 0
    assert float((l1b.flat - l1.flat).abs().max() / l1.flat.abs().max()) < 1e-13
    assert float((xi - xib).abs().max() / xi.abs().max()) < 1e-13  # another summation order over the nodes (skewed shape table)
    del l1b, xib
    asm.set_scatter("gather")
    # (2)
    Ru = l1.b[0].view(-1, 3)
    assert float(Ru.sum(0).abs().max()) < 1e-9 * float(Ru.abs().sum())
    for k in range(3):
        t = torch.zeros_like(u).view(-1, 3)
        t[:, k] = 1.0
        yu, yp = torch.zeros_like(u), torch.zeros_like(p)
        asm.apply_A(l1, t.view(-1).contiguous(), zp, yu, yp)
        scale = float(l1.A[0][0].abs().max())
        assert float(yu.abs().max()) < 1e-10 * scale and float(yp.abs().max()) < 1e-10 * scale
    # (3)
    g0 = torch.zeros(asm.nelems, asm.npts, asm.nloc, dtype=torch.float64, device=u.device)
    f0 = torch.zeros(asm.nelems, asm.npts, asm.ndofs, dtype=torch.float64, device=u.device)
    l3 = asm.new_linsys()
    assert asm.adjoint_jacobian(u, p, z, zp, xi0, xi, g0, f0, l3) == 0
    del g0, f0
    gen = torch.Generator(device="cpu").manual_seed(11)
    rnd = lambda t: torch.randn(t.shape, generator=gen, dtype=torch.float64).to(u.device)
    dot = lambda a, b: float((a * b).sum())
    xu, xp, yu, yp = rnd(u), rnd(p), rnd(u), rnd(p)
    a3u, a3p, a1u, a1p = torch.zeros_like(u), torch.zeros_like(p), torch.zeros_like(u), torch.zeros_like(p)
    asm.apply_A(l3, xu, xp, a3u, a3p)
    asm.apply_A(l1, yu, yp, a1u, a1p)
    lhs, rhs = dot(yu, a3u) + dot(yp, a3p), dot(xu, a1u) + dot(xp, a1p)
    assert abs(lhs - rhs) < 1e-12 * float(a3u.abs().max()) * float(yu.abs().sum()), (lhs, rhs)
    del l1, l3, a3u, a3p, a1u, a1p
    # (4) sampled elements, each as a mesh of its own with 8 private nodes: assembled on the device and by the oracle
    rng = np.random.default_rng(12)
    sample = np.sort(rng.choice(asm.nelems, 200, replace=False))
    nodes = conn[sample].ravel()
    sc, sconn = np.ascontiguousarray(coords[nodes]), np.arange(len(nodes), dtype=np.int32).reshape(-1, 8)
    su = np.ascontiguousarray(u_h.reshape(-1, 3)[nodes].ravel())
    sp_ = np.ascontiguousarray(p_h[nodes])
    orc = ol.Oracle(ol.HEX8, sc, sconn, "small_J2", J2)
    small = Assembler(8, sc, sconn, "small_J2", J2)
    lo, xo = orc.new_linsys(), orc.new_state()
    assert orc.forward_jacobian(su, sp_, 0 * su, 0 * sp_, orc.new_state(), xo, lo) == 0
    ls, xs = small.new_linsys(), small.new_state()
    assert small.forward_jacobian(small.dev(su), small.dev(sp_), small.dev(0 * su), small.dev(0 * sp_), small.new_state(), xs, ls) == 0
    assert rel_vec(xs.cpu().numpy(), xo) < 1e-12
    assert rel_vec(xi.cpu().numpy()[sample], xo) < 1e-12   # the full-size run's states of those elements
    for i in range(2):
        assert rel_vec(ls.b[i].cpu().numpy(), lo.b[i]) < 1e-12
        for j in range(2):
            assert rel_vec(ls.A[i][j].cpu().numpy(), lo.A[i][j]) < 1e-12
