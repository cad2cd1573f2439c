"""Device-resident primal driver: the host mirror of the reference's Primal (primal.cpp) for one part.

Each load step calls `c8_primal_solve_step` (C++ Newton + line search around the HIP assembly, boundary
conditions applied on the device).  The sparse linear solve is the caller's callback -- the reference
uses Belos/Teko/MueLu, out of scope here; `scipy_solver` below is the direct solve the tests use.
"""
import ctypes as C

import numpy as np

from . import lib as _l

NEQ = (3, 1)


def scipy_solver(asm):
    """Linear-solve callback: copies the device system to the host, SciPy sparse direct solve, copies dx back."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    n = asm.nnodes
    rp, ci = asm.rowptr, asm.colidx
    nnz = asm.nnz
    NEQ = asm.neq

    nres = getattr(asm, "nres", 2)  # 1 under mechanics_plane_stress: the system is the u block alone

    def solve(user, sys_p, dx_p):
        sys = sys_p.contents
        blocks = [[None] * nres for _ in range(nres)]
        for i in range(nres):
            for j in range(nres):
                vals = _DevView(sys.A[i][j], nnz[i][j], asm.device).to_numpy()
                blocks[i][j] = sp.csr_matrix((vals, ci[i][j], rp[i][j]), shape=(n * NEQ[i], n * NEQ[j]))
        b = np.concatenate([_DevView(sys.b[i], n * NEQ[i], asm.device).to_numpy() for i in range(nres)])
        x = spla.spsolve(sp.bmat(blocks, format="csc"), b)
        _DevView(dx_p[0], n * NEQ[0], asm.device).from_numpy(x[: n * NEQ[0]])
        if nres == 2:
            _DevView(dx_p[1], n, asm.device).from_numpy(x[n * NEQ[0]:])
        return 0

    return _l.LINEAR_SOLVE_FN(solve)


def distributed_scipy_solver(asm, plan, dist):
    """Linear-solve callback for a multi-part mesh (stand-in for the reference's distributed Belos solve, out of
    scope): every rank contributes the OWNED rows of its system (the first plan.part.nowned node rows, columns mapped
    to global ids), all ranks gather them, solve the global system with SciPy and keep the owned part of dx.  The
    driver imports dx to the ghost and phantom copies afterwards (c8_halo_scatter_x)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    n, no = asm.nnodes, plan.part.nowned
    rp, ci, nnz = asm.rowptr, asm.colidx, asm.nnz
    gid = plan.node_gid
    N = plan.part.num_global_nodes
    world = plan.world
    NEQ = asm.neq                    # (3, 1), or (2, 1) on tri3 meshes
    nres = getattr(asm, "nres", 2)   # 1 under mechanics_plane_stress: the u block alone

    def solve(user, sys_p, dx_p):
        sys = sys_p.contents
        mine = {"b": [], "A": {}}
        for i in range(nres):
            nrows = no * NEQ[i]
            mine["b"].append((np.repeat(gid[:no], NEQ[i]) * NEQ[i] + np.tile(np.arange(NEQ[i]), no),
                              _DevView(sys.b[i], n * NEQ[i], asm.device).to_numpy()[:nrows]))
            for j in range(nres):
                vals = _DevView(sys.A[i][j], nnz[i][j], asm.device).to_numpy()[: rp[i][j][nrows]]
                cols = ci[i][j][: rp[i][j][nrows]]
                gcol = gid[cols // NEQ[j]] * NEQ[j] + cols % NEQ[j]
                grow = np.repeat(mine["b"][i][0], np.diff(rp[i][j][: nrows + 1]))
                mine["A"][(i, j)] = (grow, gcol, vals)
        allp = [None] * world
        if world > 1:
            dist.all_gather_object(allp, mine)
        else:
            allp = [mine]
        blocks = [[None] * nres for _ in range(nres)]
        for i in range(nres):
            for j in range(nres):
                r = np.concatenate([q["A"][(i, j)][0] for q in allp])
                c = np.concatenate([q["A"][(i, j)][1] for q in allp])
                v = np.concatenate([q["A"][(i, j)][2] for q in allp])
                blocks[i][j] = sp.csr_matrix((v, (r, c)), shape=(N * NEQ[i], N * NEQ[j]))
        b = [np.zeros(N * NEQ[i]) for i in range(nres)]
        for q in allp:
            for i in range(nres):
                b[i][q["b"][i][0]] = q["b"][i][1]
        x = spla.spsolve(sp.bmat(blocks, format="csc"), np.concatenate(b))
        xs = [x[: N * NEQ[0]], x[N * NEQ[0]:]]
        for i in range(nres):
            loc = np.zeros(n * NEQ[i])
            loc[: no * NEQ[i]] = xs[i][mine["b"][i][0]]
            _DevView(dx_p[i], n * NEQ[i], asm.device).from_numpy(loc)
        return 0

    return _l.LINEAR_SOLVE_FN(solve)


_hip = None


def _hip_rt():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.restype = C.c_int
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hip


class _DevView:
    """Copy helper between a raw device pointer handed to a callback and numpy (hipMemcpy)."""

    def __init__(self, ptr, count, device):
        self.ptr, self.count, self.device = int(ptr), int(count), device

    def to_numpy(self):
        out = np.empty(self.count, dtype=np.float64)
        err = _hip_rt().hipMemcpy(out.ctypes.data, self.ptr, self.count * 8, 2)  # device -> host (synchronous)
        assert err == 0, "hipMemcpy D2H failed: %d" % err
        return out

    def from_numpy(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        err = _hip_rt().hipMemcpy(self.ptr, a.ctypes.data, self.count * 8, 1)  # host -> device
        assert err == 0, "hipMemcpy H2D failed: %d" % err


class PrimalDriver:
    """dbcs: list of (resid, eq, node ids, fn(x, y, z, t)); tbcs: list of (resid, faces [n][3|4], fn(x, y, z, t) -> 3-vector)."""

    def __init__(self, asm, dbcs, tbcs=(), max_iters=15, abs_tol=1e-8, rel_tol=1e-8, step_size=1.0, line_search=True,
                 solver=None):
        import torch
        self.torch, self.asm = torch, asm
        self.dbcs, self.tbcs = list(dbcs), list(tbcs)
        self.step_size = step_size
        self.opts = _l.NewtonOpts(max_iters, abs_tol, rel_tol, int(line_search), 1e-4, 0.5, 0.9, 4)
        self.solver = solver if solver is not None else scipy_solver(asm)
        dev = asm.device
        self.u = [torch.zeros(asm.nnodes * asm.ndims, dtype=torch.float64, device=dev)]
        self.p = [torch.zeros(asm.nnodes, dtype=torch.float64, device=dev)]
        self.xi = [asm.new_state()]
        self.ls = asm.new_linsys()
        self.newton_iters = []
        L = asm.L
        # static device tables of the boundary conditions
        self._dbc_nodes = [torch.as_tensor(np.ascontiguousarray(nodes, dtype=np.int32), device=dev) for _, _, nodes, _ in self.dbcs]
        self._tbc_faces, self._tbc_pts = [], []
        for _, faces, _ in self.tbcs:
            f = np.ascontiguousarray(faces, dtype=np.int32)
            npf = f.shape[1]
            pts = np.zeros((len(f), 4 if npf == 4 else 1, 3))
            _l.check(L.c8_face_points(npf, len(f), asm.coords.ctypes.data_as(_l.dp), f.ctypes.data_as(_l.i32p),
                                      pts.ctypes.data_as(_l.dp)))
            self._tbc_faces.append(torch.as_tensor(f, device=dev))
            self._tbc_pts.append(pts)

    def _bc_structs(self, t):
        torch, asm = self.torch, self.asm
        keep = []
        d = (_l.Dbc * max(1, len(self.dbcs)))()
        for k, (resid, eq, nodes, fn) in enumerate(self.dbcs):
            c = asm.coords[np.asarray(nodes)]
            vals = asm.dev(np.array([fn(x, y, z, t) for x, y, z in c], dtype=np.float64))
            keep.append(vals)
            d[k] = _l.Dbc(resid, eq, len(nodes), self._dbc_nodes[k].data_ptr(), vals.data_ptr())
        tb = (_l.Tbc * max(1, len(self.tbcs)))()
        for k, (resid, faces, fn) in enumerate(self.tbcs):
            pts = self._tbc_pts[k]
            tr = np.array([[fn(x, y, z, t) for x, y, z in fp] for fp in pts], dtype=np.float64)
            tv = asm.dev(tr.ravel())
            keep.append(tv)
            tb[k] = _l.Tbc(resid, len(faces), self._tbc_faces[k].shape[1], self._tbc_faces[k].data_ptr(), tv.data_ptr())
        return d, tb, keep

    def solve_at_step(self, step):
        asm = self.asm
        assert len(self.u) == step
        u, p = self.u[step - 1].clone(), self.p[step - 1].clone()
        xi = self.xi[step - 1].clone()
        st = asm._state(u, p, self.u[step - 1], self.p[step - 1], self.xi[step - 1], xi)
        sy = self.ls.c_struct()
        d, tb, keep = self._bc_structs(step * self.step_size)
        iters = C.c_int32(0)
        rc = asm.L.c8_primal_solve_step(asm.h, C.byref(st), C.byref(sy), len(self.dbcs), d, len(self.tbcs), tb,
                                        C.byref(self.opts), C.cast(self.solver, C.c_void_p), None, C.byref(iters))
        _l.check(rc)
        self.newton_iters.append(iters.value)
        self.u.append(u)
        self.p.append(p)
        self.xi.append(xi)

    def solve(self, nsteps):
        for s in range(1, nsteps + 1):
            self.solve_at_step(s)
        return self

    def set_measured(self, u_meas, loads):
        """Calibration objective: measured displacements (device tensors) and loads per step, index 0 unused."""
        self.measured = (u_meas, loads)

    def begin_qoi_step(self, s):
        if getattr(self, "measured", None) is not None:
            self.asm.set_measured(self.measured[0][s], self.measured[1][s])

    def qoi(self):
        J = self.torch.zeros(1, dtype=self.torch.float64, device=self.asm.device)
        for s in range(1, len(self.u)):
            self.begin_qoi_step(s)
            self.asm.eval_qoi(self.u[s], self.p[s], J, xi_prev=self.xi[s - 1], xi=self.xi[s], u_prev=self.u[s - 1],
                              p_prev=self.p[s - 1])
        self.torch.cuda.synchronize()
        return float(J.item())


def adjoint_gradient(primal, nparams):
    """Adjoint_Objective::gradient (adjoint_objective.cpp:83-109) for one part on the device: march the load
    steps backwards through c8_adjoint_solve_step and return dJ/dp (host array, physical parameters).
    `primal` is a solved PrimalDriver; the active parameters are those set with Assembler.set_active."""
    torch, asm = primal.torch, primal.asm
    dev = asm.device
    nsteps = len(primal.u) - 1
    g = torch.zeros(asm.nelems, asm.npts, asm.nloc, dtype=torch.float64, device=dev)
    f = torch.zeros(asm.nelems, asm.npts, asm.ndofs, dtype=torch.float64, device=dev)
    phi = torch.zeros_like(g)
    grad = torch.zeros(nparams, dtype=torch.float64, device=dev)
    z_u = torch.zeros(asm.nnodes * asm.ndims, dtype=torch.float64, device=dev)
    z_p = torch.zeros(asm.nnodes, dtype=torch.float64, device=dev)
    sy = primal.ls.c_struct()
    zero_vals = [asm.dev(np.zeros(len(nodes))) for _, _, nodes, _ in primal.dbcs]
    d = (_l.Dbc * max(1, len(primal.dbcs)))()
    for k, (resid, eq, nodes, _) in enumerate(primal.dbcs):
        d[k] = _l.Dbc(resid, eq, len(nodes), primal._dbc_nodes[k].data_ptr(), zero_vals[k].data_ptr())
    for step in range(nsteps, 0, -1):
        primal.begin_qoi_step(step)
        st = asm._state(primal.u[step], primal.p[step], primal.u[step - 1], primal.p[step - 1], primal.xi[step - 1],
                        primal.xi[step])
        z = (C.c_void_p * 2)(z_u.data_ptr(), z_p.data_ptr())
        rc = asm.L.c8_adjoint_solve_step(asm.h, C.byref(st), C.byref(sy), len(primal.dbcs), d,
                                         C.cast(primal.solver, C.c_void_p), None, z, C.c_void_p(phi.data_ptr()),
                                         C.c_void_p(g.data_ptr()), C.c_void_p(f.data_ptr()), C.c_void_p(grad.data_ptr()))
        _l.check(rc)
    torch.cuda.synchronize()
    return grad.cpu().numpy()
