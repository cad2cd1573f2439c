"""Two ranks sharing the GPU of the box (gloo rendezvous, no RCCL needed): the multi-part pieces that need device
code.  The halo exchange itself is covered on the CPU in test_distributed_gloo.py; here the Calibration objective's
cross-part sums (side-set area, reaction load, load term of J: PCU_Add_Double in calibration.cpp:138, :351, :375-378)
and the parameter gradient summed over the parts must equal the single-part values."""
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
LOC = ([0, 1, 2, 3], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7], [4, 5, 6, 7])


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def evaluate(asm, c, conn, u, p, xi_prev, z_u, z_p, u_meas):
    """objective value and parameter gradient of one (part of a) mesh at a prescribed state"""
    xmax = c[:, 0].max()
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


def worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        from calibr8_amd import Assembler
        from calibr8_amd import distributed as D
        from calibr8_amd.lib import load_library
        from meshes import brick, prescribed_fields
        n = (6, 4, 2)
        c, conn, sets = brick(*n, 1.0, 1.5, 0.5)
        u, p = prescribed_fields(c, 0.004, ramp=True, perturb=5e-2)
        rng = np.random.default_rng(11)
        z_u, z_p = rng.standard_normal(len(u)) * 1e-3, rng.standard_normal(len(p)) * 1e-3
        u_meas = u + 1e-4 * rng.standard_normal(len(u))
        L = load_library()
        ep = np.zeros(len(conn), dtype=np.int32)
        L.c8_brick_partition(n[0], n[1], n[2], 2, 1, 1, ep.ctypes.data_as(C.POINTER(C.c_int32)))
        part = D.part_from_global(c, conn, ep, rank, world)
        plan = D.HaloPlan(part, dist)
        gid = plan.node_gid
        sl3 = lambda v: np.ascontiguousarray(v.reshape(-1, 3)[gid].ravel())
        asm = Assembler(8, c[gid], part.conn, "small_J2", J2, extra_pairs=plan.extra_pairs)
        asm.set_allreduce(dist, world)
        xi0 = np.zeros((len(part.conn), 8, 7))
        pre, J, grad = evaluate(asm, c[gid], part.conn, sl3(u), np.ascontiguousarray(p[gid]), xi0, sl3(z_u),
                                np.ascontiguousarray(z_p[gid]), sl3(u_meas))
        t = torch.tensor([J] + list(grad), dtype=torch.float64)
        dist.all_reduce(t)  # J and dJ/dp are summed over the parts by the caller (adjoint_objective.cpp:37, :109)
        res = {"pre": pre, "J": float(t[0]), "grad": t[1:].numpy()}
        if rank == 0:  # single-part reference on the whole mesh
            ref = Assembler(8, c, conn, "small_J2", J2)
            res["ref"] = evaluate(ref, c, conn, u, p, np.zeros((len(conn), 8, 7)), z_u, z_p, u_meas)
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_calibration_objective_over_two_parts():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker, args=(2, free_port(), out), nprocs=2, join=True)
    pre_ref, J_ref, grad_ref = out[0]["ref"]
    assert abs(pre_ref[2]) > 1e-3
    for r in range(2):
        assert np.abs(out[r]["pre"] - pre_ref).max() < 1e-12 * np.abs(pre_ref).max(), (r, out[r]["pre"], pre_ref)
        assert abs(out[r]["J"] - J_ref) < 1e-12 * abs(J_ref), (r, out[r]["J"], J_ref)
        assert np.abs(out[r]["grad"] - grad_ref).max() < 1e-11 * np.abs(grad_ref).max(), (r, out[r]["grad"], grad_ref)
