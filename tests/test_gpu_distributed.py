"""Two ranks sharing the GPU of the box (gloo rendezvous; RCCL refuses two ranks on one card, so the messages travel
through c8_comm's host transport -- pack, unpack-add and every index table are the ones an RCCL run uses).

  * HIP-assembled multi-part systems after the halo (C1 + C2) against the single-part HIP assembly of the whole mesh
    AND the oracle, 1e-12, all four blocks and b, in every form the bench and the drivers use: blocking exchange after a
    staged / coloured / atomic assembly; staged assembly with the row sums in two parts (ghost rows first, exchange
    overlapped with the owned rows' sums), also in assign mode; atomic assembly of the interface elements first with the
    exchange overlapping the interior elements.  The same for the adjoint Jacobian (K3).
  * C3 (owner values to ghost and phantom copies) and the packed C4 / C5 all-reduce through libc8.so.
  * The multi-part step drivers: c8_primal_solve_step / c8_adjoint_solve_step over two parts take the Newton iterations
    of the single-part run and land on its solution and gradient.
  * The Calibration objective's cross-part sums (side-set area, reaction load, load term of J: PCU_Add_Double in
    calibration.cpp:138, :351, :375-378) through the halo's communicator.
  * One rank: the RCCL transport itself (librccl loaded by dlopen, communicator, grouped send / recv to self, all-reduce).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
pytestmark = pytest.mark.gpu
J2 = [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0]
NEQ = (3, 1)
LOC = ([0, 1, 2, 3], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7], [4, 5, 6, 7])


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn(fn, world, *args):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(fn, args=(world, free_port(), out) + args, nprocs=world, join=True)
    assert len(out) == world
    return dict(out)


def setup_part(rank, world, n, pdims, model="small_J2", params=J2, jig=0.03):
    """global problem, this rank's part, plan, assembler, halo over a host-transport communicator"""
    import ctypes as C
    from calibr8_amd import Assembler
    from calibr8_amd import distributed as D
    from calibr8_amd.lib import load_library
    from meshes import brick, jiggle, prescribed_fields
    c, conn, sets = brick(n[0], n[1], n[2], 1.0, 0.8, 0.7)
    if jig:
        c = jiggle(c, sets, jig)
    u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
    L = load_library()
    ep = np.zeros(len(conn), dtype=np.int32)
    L.c8_brick_partition(n[0], n[1], n[2], pdims[0], pdims[1], pdims[2], ep.ctypes.data_as(C.POINTER(C.c_int32)))
    part = D.part_from_global(c, conn, ep, rank, world)
    plan = D.HaloPlan(part, dist)
    asm = Assembler(8, c[plan.node_gid], part.conn, model, params, extra_pairs=plan.extra_pairs)
    comm = D.Comm.host(dist, rank, world)
    halo = D.Halo(plan, asm.rowptr[1][1], asm.colidx[1][1], asm, comm)
    return dict(c=c, conn=conn, sets=sets, u=u, p=p, ep=ep, part=part, plan=plan, asm=asm, comm=comm, halo=halo)


def owned_rows_error(asm, plan, ls, ref_rowptr, ref_colidx, ref_A, ref_b, nglobal):
    """max over the OWNED rows of |x - ref| / ||ref row||_inf for the four blocks, and of |b - ref| / ||ref||_inf:
    the part's local system (device) against a global reference system (host arrays)"""
    import scipy.sparse as sp
    gid, no = plan.node_gid, plan.part.nowned
    worst = {}
    for i in range(2):
        bo = ls.b[i].cpu().numpy()[: no * NEQ[i]].reshape(no, NEQ[i])
        br = ref_b[i].reshape(-1, NEQ[i])[gid[:no]]
        worst["b%d" % i] = float(np.abs(bo - br).max() / np.abs(ref_b[i]).max())
        grow = np.repeat(gid[:no], NEQ[i]) * NEQ[i] + np.tile(np.arange(NEQ[i]), no)
        for j in range(2):
            Ag = sp.csr_matrix((ref_A[i][j], ref_colidx[i][j], ref_rowptr[i][j]), shape=(nglobal * NEQ[i], nglobal * NEQ[j]))[grow]
            rp, ci = asm.rowptr[i][j], asm.colidx[i][j]
            nrows = no * NEQ[i]
            vals = ls.A[i][j].cpu().numpy()[: rp[nrows]]
            cols = ci[: rp[nrows]]
            gcol = gid[cols // NEQ[j]] * NEQ[j] + cols % NEQ[j]
            Al = sp.csr_matrix((vals, gcol, rp[: nrows + 1]), shape=(nrows, nglobal * NEQ[j]))
            rownorm = np.maximum(abs(Ag).max(axis=1).toarray().ravel(), 1e-300)
            d = abs(Al - Ag).max(axis=1).toarray().ravel()
            worst["A%d%d" % (i, j)] = float((d / rownorm).max())
    return worst


def halo_worker(rank, world, port, out, n, pdims):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as ol
        from calibr8_amd import Assembler
        S = setup_part(rank, world, n, pdims)
        c, conn, u, p, part, plan, asm, comm, halo = (S[k] for k in ("c", "conn", "u", "p", "part", "plan", "asm", "comm", "halo"))
        gid, no, nt = plan.node_gid, part.nowned, part.ntouched
        dev = asm.device
        elems = np.nonzero(S["ep"] == rank)[0]
        sl3 = lambda v: np.ascontiguousarray(v.reshape(-1, 3)[gid].ravel())
        du, dp = asm.dev(sl3(u)), asm.dev(np.ascontiguousarray(p[gid]))
        z3, z1 = torch.zeros_like(du), torch.zeros_like(dp)
        # ---- references on the whole mesh: single-part HIP assembly and the oracle ----
        ref = Assembler(8, c, conn, "small_J2", J2)
        ru, rp_ = ref.dev(u), ref.dev(p)
        rls, rxi = ref.new_linsys(), ref.new_state()
        assert ref.forward_jacobian(ru, rp_, torch.zeros_like(ru), torch.zeros_like(rp_), ref.new_state(), rxi, rls) == 0
        hipA = [[rls.A[i][j].cpu().numpy() for j in range(2)] for i in range(2)]
        hipb = [rls.b[i].cpu().numpy() for i in range(2)]
        orc = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2)
        ols, oxi = orc.new_linsys(), orc.new_state()
        assert orc.forward_jacobian(u, p, 0 * u, 0 * p, orc.new_state(), oxi, ols) == 0
        assert all(np.array_equal(ref.colidx[i][j], orc.colidx[i][j]) for i in range(2) for j in range(2))
        res = {}

        def check(tag, ls):
            torch.cuda.synchronize()
            e_hip = owned_rows_error(asm, plan, ls, ref.rowptr, ref.colidx, hipA, hipb, len(c))
            e_orc = owned_rows_error(asm, plan, ls, orc.rowptr, orc.colidx, ols.A, ols.b, len(c))
            res[tag] = (max(e_hip.values()), max(e_orc.values()))
            res[tag + "_blocks"] = {k: (e_hip[k], e_orc[k]) for k in e_hip}  # per block, for the message of a failing assertion

        xi0 = asm.new_state()
        fwd = lambda ls, xi: asm.forward_jacobian(du, dp, z3, z1, xi0, xi, ls)
        # 1. blocking exchange after a whole assembly, every scatter mode
        for mode in ("gather", "colored", "atomic"):
            asm.set_scatter(mode)
            ls, xi = asm.new_linsys(), asm.new_state()
            assert fwd(ls, xi) == 0
            halo.gather(ls)
            check("blocking_" + mode, ls)
            if mode == "gather":
                keep = ls.flat.clone()
        # the local state of the part equals the single-part state of its elements
        res["xi"] = float((xi.cpu() - rxi.cpu()[elems]).abs().max())
        # 2. staged assembly, row sums in two parts: ghost rows first, exchange while the owned rows are summed
        asm.set_scatter("gather")
        asm.set_stage_chunk(asm.nelems)
        asm.set_gather_early_nodes(no, nt)
        for assign in (False, True):
            asm.set_assign_mode(assign)
            ls, xi = asm.new_linsys(), asm.new_state()
            assert fwd(ls, xi) == 0
            halo.gather_start(ls)
            assert asm.gather_finish() == 0
            halo.gather_finish(ls)
            check("split_assign" if assign else "split", ls)
            if not assign:  # reproducible: bitwise the blocking result on the owned rows
                torch.cuda.synchronize()
                same = True
                for k, neq in ((4, 3), (5, 1)):
                    lo = int(ls.offsets[k])
                    same = same and bool((ls.flat[lo:lo + no * neq] == keep[lo:lo + no * neq]).all())
                same = same and bool((ls.flat[: int(asm.rowptr[0][0][no * 3])] == keep[: int(asm.rowptr[0][0][no * 3])]).all())
                res["split_bitwise_equals_blocking"] = same
        asm.set_assign_mode(False)
        asm.set_gather_early_nodes(0, 0)
        # 3. atomic adds: interface elements, start the exchange, interior elements, finish
        asm.set_scatter("atomic")
        e_if = torch.as_tensor(plan.interface_elems, device=dev)
        e_in = torch.as_tensor(plan.interior_elems, device=dev)
        ls, xi = asm.new_linsys(), asm.new_state()
        assert asm.forward_jacobian_subset(du, dp, z3, z1, xi0, xi, ls, e_if) == 0
        halo.gather_start(ls)
        assert asm.forward_jacobian_subset(du, dp, z3, z1, xi0, xi, ls, e_in) == 0
        halo.gather_finish(ls)
        check("overlap_atomic", ls)
        # 4. the adjoint Jacobian (K3) the same way: blocking and in two parts, with non-trivial histories
        rng = np.random.default_rng(3)
        g_h = rng.standard_normal((len(conn), 8, 7)) * 1e-2
        f_h = rng.standard_normal((len(conn), 8, 32)) * 1e-2
        g_o, lo3 = g_h.copy(), orc.new_linsys()
        orc.adjoint_jacobian(u, p, 0 * u, 0 * p, orc.new_state(), oxi, g_o, f_h, lo3)
        ref.set_scatter("gather")
        rl3, g_r = ref.new_linsys(), ref.dev(g_h.ravel()).reshape(g_h.shape)
        assert ref.adjoint_jacobian(ru, rp_, torch.zeros_like(ru), torch.zeros_like(rp_), ref.new_state(), rxi, g_r, ref.dev(f_h.ravel()), rl3) == 0
        hipA3 = [[rl3.A[i][j].cpu().numpy() for j in range(2)] for i in range(2)]
        hipb3 = [rl3.b[i].cpu().numpy() for i in range(2)]
        xi_part = asm.dev(oxi[elems].ravel()).reshape(len(elems), 8, 7)
        for split in (False, True):
            asm.set_scatter("gather")
            asm.set_gather_early_nodes(no if split else 0, nt if split else 0)
            ls = asm.new_linsys()
            g_d = asm.dev(g_h[elems].ravel()).reshape(len(elems), 8, 7)
            assert asm.adjoint_jacobian(du, dp, z3, z1, xi0, xi_part, g_d, asm.dev(f_h[elems].ravel()), ls) == 0
            if split:
                halo.gather_start(ls)
                assert asm.gather_finish() == 0
                halo.gather_finish(ls)
            else:
                halo.gather(ls)
            torch.cuda.synchronize()
            e_hip = owned_rows_error(asm, plan, ls, ref.rowptr, ref.colidx, hipA3, hipb3, len(c))
            e_orc = owned_rows_error(asm, plan, ls, orc.rowptr, orc.colidx, lo3.A, lo3.b, len(c))
            res["k3_split" if split else "k3_blocking"] = (max(e_hip.values()), max(e_orc.values()))
            res["k3_g"] = float((g_d.cpu().numpy() - g_o[elems]).__abs__().max())
        asm.set_gather_early_nodes(0, 0)
        # 5. C1 alone
        ls = asm.new_linsys()
        assert fwd(ls, asm.new_state()) == 0
        b_before = [ls.b[i].clone() for i in range(2)]
        A_before = ls.A[0][0].clone()
        halo.gather(ls, halo.B)
        torch.cuda.synchronize()
        res["c1_leaves_A"] = bool((ls.A[0][0] == A_before).all())
        bo = [ls.b[i].cpu().numpy()[: no * NEQ[i]].reshape(no, NEQ[i]) for i in range(2)]
        res["c1"] = max(float(np.abs(bo[i] - hipb[i].reshape(-1, NEQ[i])[gid[:no]]).max() / np.abs(hipb[i]).max()) for i in range(2))
        # 6. C3: owner values to the ghost AND phantom copies
        x = [asm.dev(sl3(u)), asm.dev(np.ascontiguousarray(p[gid]))]
        x[0][no * 3:] = -7.0
        x[1][no:] = -7.0
        halo.scatter_x(x)
        torch.cuda.synchronize()
        res["c3"] = bool((x[0].cpu().numpy() == sl3(u)).all() and (x[1].cpu().numpy() == p[gid]).all())
        res["phantoms"] = len(plan.phantom_gid)
        # 7. C4 / C5
        v = comm.allreduce(np.array([1.0 + rank, 10.0 * rank, 0.0]))
        res["c45"] = bool(np.allclose(v, [sum(1.0 + r for r in range(world)), sum(10.0 * r for r in range(world)), 0.0]))
        res["bytes"] = (halo.send_bytes(3), halo.send_bytes(1), halo.send_bytes(0))
        halo.close()
        comm.close()
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_two_parts_hip_assembly_and_halo_match_single_part_and_oracle():
    out = spawn(halo_worker, 2, (6, 4, 4), (2, 1, 1))
    for r in range(2):
        res = out[r]
        for tag in ("blocking_gather", "blocking_colored", "blocking_atomic", "split", "split_assign", "overlap_atomic",
                    "k3_blocking", "k3_split"):
            assert res[tag][0] < 1e-12 and res[tag][1] < 1e-12, (r, tag, res[tag], res.get(tag + "_blocks"))
        assert res["xi"] < 1e-14 and res["k3_g"] < 1e-14, (r, res["xi"], res["k3_g"])
        assert res["split_bitwise_equals_blocking"], r
        assert res["c1"] < 1e-12 and res["c1_leaves_A"] and res["c3"] and res["c45"], (r, res)
        assert res["bytes"][0] >= res["bytes"][1] >= 0
    # rank 0 (lowest part id) owns every node it shares: it sends no ghost rows but exports values (C3); rank 1 the reverse
    assert sum(out[r]["bytes"][0] for r in range(2)) > sum(out[r]["bytes"][1] for r in range(2)) > 0
    assert sum(out[r]["bytes"][2] for r in range(2)) > 0
    assert sum(out[r]["phantoms"] for r in range(2)) > 0


def test_four_parts_sharing_the_card():
    # a 2 x 2 x 1 split: nodes shared by four parts, contributions from three ranks added in a fixed order
    out = spawn(halo_worker, 4, (6, 6, 2), (2, 2, 1))
    for r in range(4):
        for tag in ("blocking_gather", "split", "overlap_atomic", "k3_split"):
            assert out[r][tag][0] < 1e-12 and out[r][tag][1] < 1e-12, (r, tag, out[r][tag])
        assert out[r]["split_bitwise_equals_blocking"] and out[r]["c3"] and out[r]["c45"]


# ---- step drivers over two parts ---------------------------------------------------------------------------------
def bcs_for(sets_of, coords):
    """Dirichlet conditions of a uniaxial pull (node sets by coordinate): (resid, eq, nodes, fn(x, y, z, t))"""
    return [(0, 0, sets_of("xmin"), lambda x, y, z, t: 0.0), (0, 1, sets_of("ymin"), lambda x, y, z, t: 0.0),
            (0, 2, sets_of("zmin"), lambda x, y, z, t: 0.0), (0, 1, sets_of("ymax"), lambda x, y, z, t: 0.002 * t)]


def driver_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from calibr8_amd import Assembler
        from calibr8_amd.primal import PrimalDriver, adjoint_gradient, distributed_scipy_solver
        n = (6, 4, 3)
        S = setup_part(rank, world, n, (2, 1, 1), jig=0.02)
        c, conn, part, plan, asm, comm, halo = (S[k] for k in ("c", "conn", "part", "plan", "asm", "comm", "halo"))
        gid, no = plan.node_gid, part.nowned
        lc = c[gid]
        lo, hi = c.min(axis=0), c.max(axis=0)

        def local_sets(coords, nlim):
            def of(name):
                ax, side = "xyz".index(name[0]), name[1:]
                v = lo[ax] if side == "min" else hi[ax]
                return np.nonzero(np.abs(coords[:nlim, ax] - v) < 1e-9)[0].astype(np.int32)
            return of

        act = [0, 1, 2, 3]
        asm.set_active(0, act)
        # the row sums in two parts, so that the drivers' overlapped gather is what runs
        asm.set_stage_chunk(asm.nelems)
        asm.set_gather_early_nodes(no, part.ntouched)
        drv = PrimalDriver(asm, bcs_for(local_sets(lc, len(lc)), lc), solver=distributed_scipy_solver(asm, plan, dist))
        drv.solve(2)
        J = comm.allreduce(np.array([drv.qoi()]))[0]
        grad = comm.allreduce(adjoint_gradient(drv, len(act)))  # C4 (adjoint_objective.cpp:109)
        res = {"iters": list(drv.newton_iters), "J": float(J), "grad": grad}
        # single-part reference (every rank runs it: small)
        ref = Assembler(8, c, conn, "small_J2", J2)
        ref.set_active(0, act)
        rdrv = PrimalDriver(ref, bcs_for(local_sets(c, len(c)), c))
        rdrv.solve(2)
        res["ref_iters"] = list(rdrv.newton_iters)
        res["ref_J"] = rdrv.qoi()
        res["ref_grad"] = adjoint_gradient(rdrv, len(act))
        torch.cuda.synchronize()
        ug = rdrv.u[2].cpu().numpy().reshape(-1, 3)[gid].ravel()
        res["du"] = float(np.abs(drv.u[2].cpu().numpy() - ug).max() / np.abs(ug).max())   # owned, ghost AND phantom copies
        res["dxi"] = float((drv.xi[2].cpu() - rdrv.xi[2].cpu()[np.nonzero(S["ep"] == rank)[0]]).abs().max())
        halo.close()
        comm.close()
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_step_drivers_over_two_parts_reproduce_the_single_part_run():
    out = spawn(driver_worker, 2)
    for r in range(2):
        res = out[r]
        assert res["iters"] == res["ref_iters"] and max(res["iters"]) > 2, (r, res["iters"], res["ref_iters"])
        assert res["du"] < 1e-9 and res["dxi"] < 1e-9, (r, res["du"], res["dxi"])
        assert abs(res["J"] - res["ref_J"]) < 1e-9 * abs(res["ref_J"]), (r, res["J"], res["ref_J"])
        assert np.abs(res["grad"] - res["ref_grad"]).max() < 1e-7 * np.abs(res["ref_grad"]).max(), (r, res["grad"], res["ref_grad"])


# ---- 2-D meshes over parts: 2 + 1 equations per node (`mechanics` on tri3) and 2 (`mechanics_plane_stress`) -------------
def driver_worker_2d(rank, world, port, out, model, params):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from calibr8_amd import Assembler
        from calibr8_amd import distributed as D
        from calibr8_amd.primal import PrimalDriver, adjoint_gradient, distributed_scipy_solver
        from meshes import jiggle_2d, tri_mesh
        c, conn, sets = tri_mesh(8, 6, 1.0, 0.8)
        c = jiggle_2d(c, sets, 0.02)
        ep = (c[conn].mean(axis=1)[:, 0] > 0.47).astype(np.int32) if world == 2 else np.zeros(len(conn), dtype=np.int32)
        part = D.part_from_global(c, conn, ep, rank, world)
        plan = D.HaloPlan(part, dist)
        asm = Assembler(3, c[plan.node_gid], part.conn, model, params, extra_pairs=plan.extra_pairs)
        comm = D.Comm.host(dist, rank, world)
        halo = D.Halo(plan, asm.rowptr[1][1], asm.colidx[1][1], asm, comm)  # the tables take the assembler's ndims / nres
        gid, no = plan.node_gid, part.nowned
        lc = c[gid]
        lo, hi = c.min(axis=0), c.max(axis=0)

        def bcs(coords):
            of = lambda ax, v: np.nonzero(np.abs(coords[:, ax] - v) < 1e-9)[0].astype(np.int32)
            return [(0, 0, of(0, lo[0]), lambda x, y, z, t: 0.0), (0, 1, of(1, lo[1]), lambda x, y, z, t: 0.0),
                    (0, 1, of(1, hi[1]), lambda x, y, z, t: 0.002 * t)]

        act = [0, 1, 2, 3]
        asm.set_active(0, act)
        drv = PrimalDriver(asm, bcs(lc), max_iters=30, solver=distributed_scipy_solver(asm, plan, dist))
        drv.solve(2)
        J = comm.allreduce(np.array([drv.qoi()]))[0]
        grad = comm.allreduce(adjoint_gradient(drv, len(act)))
        res = {"iters": list(drv.newton_iters), "J": float(J), "grad": grad, "nres": asm.nres,
               "send_bytes": halo.send_bytes(3), "import_bytes": halo.send_bytes(0), "nghost": part.ntouched - no}
        ref = Assembler(3, c, conn, model, params)
        ref.set_active(0, act)
        rdrv = PrimalDriver(ref, bcs(c), max_iters=30)
        rdrv.solve(2)
        res["ref_iters"], res["ref_J"], res["ref_grad"] = list(rdrv.newton_iters), rdrv.qoi(), adjoint_gradient(rdrv, len(act))
        torch.cuda.synchronize()
        ug = rdrv.u[2].cpu().numpy().reshape(-1, 2)[gid].ravel()
        res["du"] = float(np.abs(drv.u[2].cpu().numpy() - ug).max() / np.abs(ug).max())
        res["dxi"] = float((drv.xi[2].cpu() - rdrv.xi[2].cpu()[np.nonzero(ep == rank)[0]]).abs().max())
        res["alpha"] = float(rdrv.xi[2][:, :, 3].max())
        halo.close()
        comm.close()
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model", ["small_J2", "small_hill_plane_stress"])
def test_step_drivers_over_two_parts_on_2d_meshes(model):
    # the halo tables for 2 + 1 equations per node (tri3 under `mechanics`) and for one residual with 2 equations
    # (`mechanics_plane_stress`): c8_halo_desc.num_dims / num_residuals; both step drivers over two parts reproduce the
    # single-part run
    params = J2 if model == "small_J2" else [1000.0, 0.25, 2.0, 10.0, 2.0, 1.0, 1.1, 0.9, 1.05]
    out = spawn(driver_worker_2d, 2, model, params)
    for r in range(2):
        res = out[r]
        assert res["nres"] == (2 if model == "small_J2" else 1)
        assert res["iters"] == res["ref_iters"] and max(res["iters"]) > 2 and res["alpha"] > 1e-4, (r, res)
        assert res["du"] < 1e-9 and res["dxi"] < 1e-9, (r, res["du"], res["dxi"])
        assert abs(res["J"] - res["ref_J"]) < 1e-9 * abs(res["ref_J"]), (r, res["J"], res["ref_J"])
        assert np.abs(res["grad"] - res["ref_grad"]).max() < 1e-7 * np.abs(res["ref_grad"]).max(), (r, res["grad"], res["ref_grad"])
    # a node travels with 3 (u_x, u_y, p) or 2 values in the import exchange
    per_node = 3 if model == "small_J2" else 2
    assert sum(out[r]["import_bytes"] for r in range(2)) % (8 * per_node) == 0 and sum(out[r]["send_bytes"] for r in range(2)) > 0


# ---- the calibration objective over two parts (through the halo's communicator) ---------------------------------------
def evaluate(asm, c, conn, u, p, xi_prev, z_u, z_p, u_meas):
    """objective value and parameter gradient of one (part of a) mesh at a prescribed state"""
    faces = [[int(e[k]) for k in f] for e in conn for f in LOC if all(abs(c[e[k], 0] - 1.0) < 1e-9 for k in f)]
    faces = np.array(faces, dtype=np.int32).reshape(-1, 4)
    asm.set_qoi_calibration(faces, weights=(1.0, 2.0, 0.5), balance=0.3, coord_idx=1, coord_value=0.0, coord_tol=1e-8,
                            comp=1, dt_over_T=0.5)
    asm.set_active(0, [0, 1, 2, 3])
    d = asm.dev
    du, dp, dz = d(u), d(p), torch.zeros(len(u), dtype=torch.float64, device=asm.device)
    dzp = torch.zeros(len(p), dtype=torch.float64, device=asm.device)
    xi = asm.new_state()
    ls = asm.new_linsys()
    asm.set_scatter("atomic")
    assert asm.forward_jacobian(du, dp, dz, dzp, d(xi_prev), xi, ls) == 0
    asm.set_measured(d(u_meas), -0.7)
    pre = asm.qoi_preprocess(du, dp, dz, dzp, d(xi_prev), xi)
    J = torch.zeros(1, dtype=torch.float64, device=asm.device)
    asm.eval_qoi(du, dp, J, xi_prev=d(xi_prev), xi=xi, u_prev=dz, p_prev=dzp)
    g = torch.full((asm.nelems, asm.npts, asm.nloc), 0.01, dtype=torch.float64, device=asm.device)
    f = torch.full((asm.nelems, asm.npts, asm.ndofs), 0.02, dtype=torch.float64, device=asm.device)
    ls.zero()
    asm.adjoint_jacobian(du, dp, dz, dzp, d(xi_prev), xi, g, f, ls)
    phi = torch.zeros_like(g)
    asm.solve_adjoint_local(du, dp, dz, dzp, d(xi_prev), xi, d(z_u), d(z_p), phi, g, f)
    grad = torch.zeros(4, dtype=torch.float64, device=asm.device)
    asm.qoi_gradient(du, dp, dz, dzp, d(xi_prev), xi, d(z_u), d(z_p), phi, grad)
    torch.cuda.synchronize()
    return np.array(pre), float(J.item()), grad.cpu().numpy()


def calibration_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from calibr8_amd import Assembler
        n = (6, 4, 2)
        S = setup_part(rank, world, n, (2, 1, 1), jig=0.0)
        c, conn, u, p, part, plan, asm, comm = (S[k] for k in ("c", "conn", "u", "p", "part", "plan", "asm", "comm"))
        c = c * np.array([1.0, 1.5 / 0.8, 0.5 / 0.7])  # the brick of the single-part reference below
        rng = np.random.default_rng(11)
        z_u, z_p = rng.standard_normal(len(u)) * 1e-3, rng.standard_normal(len(p)) * 1e-3
        u_meas = u + 1e-4 * rng.standard_normal(len(u))
        gid = plan.node_gid
        sl3 = lambda v: np.ascontiguousarray(v.reshape(-1, 3)[gid].ravel())
        # a fresh assembler on the stretched coordinates; the objective's sums go through the halo's communicator
        from calibr8_amd import distributed as D
        asm2 = Assembler(8, c[gid], part.conn, "small_J2", J2, extra_pairs=plan.extra_pairs)
        halo2 = D.Halo(plan, asm2.rowptr[1][1], asm2.colidx[1][1], asm2, comm)
        xi0 = np.zeros((len(part.conn), 8, 7))
        pre, J, grad = evaluate(asm2, c[gid], part.conn, sl3(u), np.ascontiguousarray(p[gid]), xi0, sl3(z_u),
                                np.ascontiguousarray(z_p[gid]), sl3(u_meas))
        t = comm.allreduce(np.array([J] + list(grad)))  # J and dJ/dp are summed over the parts by the caller (adjoint_objective.cpp:37, :109)
        res = {"pre": pre, "J": float(t[0]), "grad": t[1:]}
        if rank == 0:  # single-part reference on the whole mesh
            ref = Assembler(8, c, conn, "small_J2", J2)
            res["ref"] = evaluate(ref, c, conn, u, p, np.zeros((len(conn), 8, 7)), z_u, z_p, u_meas)
        halo2.close()
        S["halo"].close()
        comm.close()
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_calibration_objective_over_two_parts():
    out = spawn(calibration_worker, 2)
    pre_ref, J_ref, grad_ref = out[0]["ref"]
    assert abs(pre_ref[2]) > 1e-3
    for r in range(2):
        assert np.abs(out[r]["pre"] - pre_ref).max() < 1e-12 * np.abs(pre_ref).max(), (r, out[r]["pre"], pre_ref)
        assert abs(out[r]["J"] - J_ref) < 1e-12 * abs(J_ref), (r, out[r]["J"], J_ref)
        assert np.abs(out[r]["grad"] - grad_ref).max() < 1e-11 * np.abs(grad_ref).max(), (r, out[r]["grad"], grad_ref)


# ---- the calibration loop over two parts (BASELINE config 5 in small): InverseProblem with the parts' communicator ---------
def inverse_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from calibr8_amd import Assembler, InverseProblem, PrimalDriver
        from calibr8_amd import distributed as D
        from calibr8_amd.primal import distributed_scipy_solver
        from meshes import brick
        n = (4, 4, 2)
        c, conn, sets = brick(n[0], n[1], n[2], 1.0, 1.5, 1.0)
        ep = (c[conn].mean(axis=1)[:, 0] > 0.5).astype(np.int32)
        part = D.part_from_global(c, conn, ep, rank, world)
        plan = D.HaloPlan(part, dist)
        gid, no = plan.node_gid, part.nowned
        lc = c[gid]
        comm = D.Comm.host(dist, rank, world)
        nsteps, truth, active = 2, np.array(J2), [2, 3]
        kw = dict(weights=(1.0, 1.0, 1.0), balance=1e-2, coord_idx=1, coord_value=0.0, coord_tol=1e-8, comp=1, dt_over_T=1.0 / nsteps)

        def bcs(coords, nlim):
            of = lambda ax, v: np.nonzero(np.abs(coords[:nlim, ax] - v) < 1e-9)[0].astype(np.int32)
            zero = lambda x, y, z, t: 0.0
            return [(0, 0, of(0, 0.0), zero), (0, 1, of(1, 0.0), zero), (0, 2, of(2, 0.0), zero),
                    (0, 1, of(1, 1.5), lambda x, y, z, t: 0.0035 * t)]

        def faces_of(coords, cn):
            return np.array([[int(e[k]) for k in f] for e in cn for f in LOC if all(abs(coords[e[k], 0] - 1.0) < 1e-9 for k in f)],
                            dtype=np.int32).reshape(-1, 4)

        # single-part truth run: the measurements (every rank runs it: small)
        def whole(params, measured=None):
            asm = Assembler(8, c, conn, "small_J2", list(params))
            asm.set_qoi_calibration(faces_of(c, conn), **kw)
            pr = PrimalDriver(asm, bcs(c, len(c)), max_iters=20, abs_tol=1e-12, rel_tol=1e-12).solve(nsteps)
            if measured is not None:
                pr.set_measured(*measured)
            return pr

        pt = whole(truth)
        loads, zm = [0.0], torch.zeros_like(pt.u[1])
        for s_ in range(1, nsteps + 1):
            pt.asm.set_measured(zm, 0.0)
            loads.append(pt.asm.qoi_preprocess(pt.u[s_], pt.p[s_], pt.u[s_ - 1], pt.p[s_ - 1], pt.xi[s_ - 1], pt.xi[s_])[1])
        meas_whole = ([None] + [u.clone() for u in pt.u[1:]], loads)
        gidx = torch.as_tensor((gid[:, None] * 3 + np.arange(3)).ravel(), device=pt.u[1].device)
        meas_part = ([None] + [u[gidx].contiguous() for u in pt.u[1:]], loads)
        keep = []

        def part_primal(params):
            asm = Assembler(8, lc, part.conn, "small_J2", list(params), extra_pairs=plan.extra_pairs)
            halo = D.Halo(plan, asm.rowptr[1][1], asm.colidx[1][1], asm, comm)
            keep.append(halo)
            asm.set_qoi_calibration(faces_of(lc, part.conn), **kw)
            pr = PrimalDriver(asm, bcs(lc, len(lc)), max_iters=20, abs_tol=1e-12, rel_tol=1e-12,
                              solver=distributed_scipy_solver(asm, plan, dist)).solve(nsteps)
            pr.set_measured(*meas_part)
            return pr

        bounds, start = [[50.0, 200.0], [1.0, 4.0]], np.array([150.0, 3.0])
        inv = InverseProblem(part_primal, truth, active, bounds, comm=comm)
        found, info = inv.solve(start, max_iters=6, grad_tol=1e-14, step_tol=1e-12, max_ls_evals=8)
        ref = InverseProblem(lambda prm: whole(prm, meas_whole), truth, active, bounds)
        rfound, rinfo = ref.solve(start, max_iters=6, grad_tol=1e-14, step_tol=1e-12, max_ls_evals=8)
        out[rank] = dict(found=found, rfound=rfound, f=info["f"], rf=rinfo["f"], evals=info["evals"], revals=rinfo["evals"],
                         info={k: str(v) for k, v in info.items()}, rinfo={k: str(v) for k, v in rinfo.items()},
                         hist=[(list(p_), j) for p_, j in inv.history], rhist=[(list(p_), j) for p_, j in ref.history])
        for h in keep:
            h.close()
        comm.close()
    finally:
        dist.destroy_process_group()


def test_inverse_problem_over_two_parts_takes_the_single_part_iterates():
    out = spawn(inverse_worker, 2)
    for r in range(2):
        res = out[r]
        assert res["evals"] == res["revals"] and len(res["hist"]) == len(res["rhist"]) > 3, (r, res["info"], res["rinfo"])
        for (p1, j1), (p2, j2) in zip(res["hist"], res["rhist"]):  # the same trial parameters and objective values, step by step
            assert np.abs(np.array(p1) / np.array(p2) - 1.0).max() < 1e-6 and abs(j1 - j2) < 1e-6 * max(abs(j2), 1e-30), (r, p1, p2, j1, j2)
        assert res["f"] < 1e-3 * res["hist"][0][1]  # the objective went down by three orders in six iterations
    assert out[0]["hist"] == out[1]["hist"]  # both ranks took identical steps


# ---- the RCCL transport itself, one rank (RCCL admits one rank per card) ---------------------------------------------
def test_rccl_transport_single_rank_self_exchange():
    """librccl through dlopen, ncclCommInitRank, grouped ncclSend / ncclRecv on the comm stream ordered against the
    context's stream by events, the small all-reduce -- with the one rank RCCL admits on one card, as a halo whose
    messages go to the rank itself: the upper half of the nodes of a brick plays the ghost rows, row half + k is sent and
    added onto row k (whose graph row is widened with the sender's columns, as extra_pairs do on a real part)."""
    from calibr8_amd import Assembler
    from calibr8_amd import distributed as D
    from meshes import brick, prescribed_fields
    c, conn, _ = brick(3, 2, 2)
    nn = len(c)
    half = nn // 2
    nsend = nn - half
    adj = [set() for _ in range(nn)]
    for e in conn:
        for a in e:
            adj[a].update(int(b) for b in e)
    extra = np.array([(k, col) for k in range(nsend) for col in sorted(adj[half + k])], dtype=np.int32)
    asm = Assembler(8, c, conn, "small_J2", J2, extra_pairs=extra)
    rp, ci = asm.rowptr[1][1], asm.colidx[1][1]
    send_nodes = np.arange(half, nn, dtype=np.int32)
    recv_nodes = np.arange(0, nsend, dtype=np.int32)
    deg = np.array([rp[n + 1] - rp[n] for n in send_nodes])

    class P:  # the fields HaloPlan.desc() reads
        pass
    plan = P()
    plan.part = P()
    plan.part.nowned, plan.part.ntouched, plan.part.rank = half, nn, 0
    plan.world, plan.nnodes = 1, nn
    plan.send_ptr, plan.send_nodes = np.array([0, nsend]), send_nodes
    plan.recv_ptr, plan.recv_nodes = np.array([0, nsend]), recv_nodes
    plan.recv_col_ptr = np.concatenate([[0], np.cumsum(deg)])
    plan.recv_cols = np.concatenate([ci[rp[n]:rp[n + 1]] for n in send_nodes])
    plan.import_ptr, plan.import_nodes = np.array([0, nsend]), send_nodes
    plan.export_ptr, plan.export_nodes = np.array([0, nsend]), recv_nodes
    plan.desc = lambda ndims=3, nres=2: D.HaloPlan.desc(plan, ndims, nres)
    comm = D.Comm.rccl(None, 0, 1)
    v = comm.allreduce(np.array([1.5, -2.0, 3.25]))
    assert np.array_equal(v, [1.5, -2.0, 3.25])
    halo = D.Halo(plan, rp, ci, asm, comm)
    u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
    du, dp = asm.dev(u), asm.dev(p)
    ls = asm.new_linsys()
    assert asm.forward_jacobian(du, dp, torch.zeros_like(du), torch.zeros_like(dp), asm.new_state(), asm.new_state(), ls) == 0
    torch.cuda.synchronize()
    import scipy.sparse as sp
    before = [[sp.csr_matrix((ls.A[i][j].cpu().numpy(), asm.colidx[i][j], asm.rowptr[i][j]), shape=(nn * NEQ[i], nn * NEQ[j]))
               for j in range(2)] for i in range(2)]
    b_before = [ls.b[i].cpu().numpy().copy() for i in range(2)]
    for rep in range(2):  # twice: buffer reuse across exchanges
        halo.gather(ls)
    torch.cuda.synchronize()
    for i in range(2):
        rows_k = (np.repeat(recv_nodes, NEQ[i]) * NEQ[i] + np.tile(np.arange(NEQ[i]), nsend))
        rows_s = (np.repeat(send_nodes, NEQ[i]) * NEQ[i] + np.tile(np.arange(NEQ[i]), nsend))
        # second exchange adds the (unchanged) ghost rows once more
        assert np.allclose(ls.b[i].cpu().numpy()[rows_k], b_before[i][rows_k] + 2 * b_before[i][rows_s], rtol=0, atol=1e-13 * np.abs(b_before[i]).max())
        for j in range(2):
            after = sp.csr_matrix((ls.A[i][j].cpu().numpy(), asm.colidx[i][j], asm.rowptr[i][j]), shape=(nn * NEQ[i], nn * NEQ[j]))
            want = before[i][j][rows_k] + 2 * before[i][j][rows_s]
            assert abs(after[rows_k] - want).max() < 1e-13 * abs(before[i][j]).max()
    x = [du.clone(), dp.clone()]
    halo.scatter_x(x)
    torch.cuda.synchronize()
    xu = x[0].cpu().numpy().reshape(-1, 3)
    assert np.array_equal(xu[half:], u.reshape(-1, 3)[:nsend]) and np.array_equal(x[1].cpu().numpy()[half:], p[:nsend])
    assert halo.send_bytes(1) == nsend * 4 * 8 and halo.send_bytes(0) == nsend * 4 * 8
    halo.close()
    comm.close()
