"""Host-side mirror of the reference's evaluation interface for the hot path.

`Assembler` plays the role of the (State, Disc) pair the reference's `eval_*` functions
take (evaluations.hpp:23-84): it owns the discretisation tables through a c8_ctx and
exposes one method per entry point with the same meaning and error behaviour
(0 / -1 for a failed local solve; everything else raises).  All field arrays are torch
float64 tensors resident in HBM; nothing here computes.
"""
import ctypes as C

import numpy as np

from . import lib as _l

NUM_PARAMS = {"elastic": 4, "small_J2": 6, "hyper_J2": 8, "small_hill": 11, "hypo_hill": 11, "isotropic_elastic": 2}
NEQ = (3, 1)


def brick_mesh(nx, ny, nz, lx=1.0, ly=1.0, lz=1.0):
    """Structured hex8 brick (SURVEY.md section 8d synthetic meshes): coords [N,3], conn [E,8]."""
    L = _l.load_library()
    coords = np.zeros(((nx + 1) * (ny + 1) * (nz + 1), 3))
    conn = np.zeros((nx * ny * nz, 8), dtype=np.int32)
    _l.check(L.c8_brick_mesh(nx, ny, nz, lx, ly, lz, coords.ctypes.data_as(_l.dp), conn.ctypes.data_as(_l.i32p)))
    return coords, conn


def brick_partition(nx, ny, nz, px, py, pz):
    L = _l.load_library()
    part = np.zeros(nx * ny * nz, dtype=np.int32)
    _l.check(L.c8_brick_partition(nx, ny, nz, px, py, pz, part.ctypes.data_as(_l.i32p)))
    return part


class LinearSystem:
    """A[i][j] CSR values and b[i] in GHOST distribution, as device tensors (la->A, la->b)."""

    def __init__(self, asm):
        import torch
        # one allocation, the six arrays are views of it: zero_all is one fill and the halo exchange packs and
        # unpacks all of them with one gather / one index_add (distributed.Halo.start_gather)
        sizes = [asm.nnz[i][j] for i in range(2) for j in range(2)] + [asm.nnodes * asm.neq[i] for i in range(2)]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)  # A00 A01 A10 A11 b0 b1
        self.flat = torch.zeros(int(self.offsets[-1]), dtype=torch.float64, device=asm.device)
        v = [self.flat[int(self.offsets[k]):int(self.offsets[k + 1])] for k in range(6)]
        self.A = [[v[0], v[1]], [v[2], v[3]]]
        self.b = [v[4], v[5]]

    def zero(self):  # la->zero_all(), primal.cpp:98
        self.flat.zero_()

    def c_struct(self):
        s = _l.System()
        for i in range(2):
            s.b[i] = self.b[i].data_ptr()
            for j in range(2):
                s.A[i][j] = self.A[i][j].data_ptr()
        return s


class Assembler:
    def __init__(self, elem_type, coords, conn, local_type, params, elem_set=None, stab_mult=1.0, max_iters=500,
                 abs_tol=1e-12, rel_tol=1e-12, device="cuda:0", scatter=None, extra_pairs=None, global_type=None,
                 thickness=0.0, line_search=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("calibr8_amd needs a HIP device: there is no CPU execution path")
        self.L = _l.load_library()
        self.torch = torch
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.elem_type = int(elem_type)
        self.nnodes, self.nelems, self.nn = self.coords.shape[0], self.conn.shape[0], self.conn.shape[1]
        self.local_type = local_type
        self.params = np.ascontiguousarray(np.atleast_2d(np.asarray(params, dtype=np.float64)))
        self.nsets = self.params.shape[0]
        self._es = None if elem_set is None else np.ascontiguousarray(elem_set, dtype=np.int32)
        self._xp = None if extra_pairs is None or len(extra_pairs) == 0 else \
            np.ascontiguousarray(extra_pairs, dtype=np.int32).reshape(-1, 2)
        md = _l.MeshDesc(self.elem_type, self.nnodes, self.nelems, self.nsets, self.coords.ctypes.data_as(_l.dp),
                         self.conn.ctypes.data_as(_l.i32p),
                         self._es.ctypes.data_as(_l.i32p) if self._es is not None else None,
                         0 if self._xp is None else len(self._xp),
                         None if self._xp is None else self._xp.ctypes.data_as(_l.i32p))
        if global_type is None:  # the pairing of the reference's decks
            global_type = "mechanics_plane_stress" if local_type.endswith("_plane_stress") else "mechanics"
        mo = _l.ModelDesc(global_type.encode(), local_type.encode(), stab_mult, max_iters, abs_tol, rel_tol,
                          self.params.shape[1], self.params.ctypes.data_as(_l.dp), float(thickness),
                          *((0.0, 0.0, 0.0, 0) if line_search is None else  # (c1, min factor, max factor, max evals)
                            (float(line_search[0]), float(line_search[1]), float(line_search[2]), int(line_search[3]))))
        h = C.c_void_p()
        _l.check(self.L.c8_create(C.byref(md), C.byref(mo), C.byref(h)))
        self.h = h
        self.nloc = self.L.c8_num_local_dofs(h)
        self.npts = self.L.c8_num_local_points(h)
        self.ncolors = self.L.c8_num_colors(h)
        self.ndims = self.L.c8_num_dims(h)  # 3, or 2 on tri3 meshes (u arrays are [nnodes * ndims])
        self.nres = self.L.c8_num_residuals(h)  # 2, or 1 under mechanics_plane_stress: the p arrays / blocks are not used
        self.neq = (self.ndims, 1)
        self.ndofs = (self.ndims + (1 if self.nres == 2 else 0)) * self.nn
        self.nnz = [[int(self.L.c8_graph_nnz(h, i, j)) for j in range(2)] for i in range(2)]
        self._graph = None
        if scatter is not None:  # None: the library's default (staged assembly, see c8_set_scatter_mode)
            self.set_scatter(scatter)
        self.use_current_stream()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.c8_destroy(self.h)
            self.h = None

    # ---- discretisation ------------------------------------------------------------------
    @property
    def graph(self):
        """(rowptr[i][j], colidx[i][j]) host arrays of the four CSR blocks."""
        if self._graph is None:
            rp = [[None, None], [None, None]]
            ci = [[None, None], [None, None]]
            for i in range(2):
                for j in range(2):
                    r = np.zeros(self.nnodes * self.neq[i] + 1, dtype=np.int64)
                    c = np.zeros(self.nnz[i][j], dtype=np.int32)
                    _l.check(self.L.c8_graph(self.h, i, j, r.ctypes.data_as(_l.i64p), c.ctypes.data_as(_l.i32p)))
                    rp[i][j], ci[i][j] = r, c
            self._graph = (rp, ci)
        return self._graph

    @property
    def rowptr(self):
        return self.graph[0]

    @property
    def colidx(self):
        return self.graph[1]

    def new_state(self):
        xi = np.zeros((self.nelems, self.npts, self.nloc))
        _l.check(self.L.c8_init_variables(self.h, xi.ctypes.data_as(_l.dp)))
        return self.torch.from_numpy(xi).to(self.device)

    def new_linsys(self):
        return LinearSystem(self)

    def dev(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    # ---- settings ----------------------------------------------------------------------------
    def set_params(self, params):
        self.params = np.ascontiguousarray(np.atleast_2d(np.asarray(params, dtype=np.float64)))
        _l.check(self.L.c8_set_params(self.h, self.params.ctypes.data_as(_l.dp)))

    def set_active(self, es, idx):
        a = np.ascontiguousarray(idx, dtype=np.int32)
        _l.check(self.L.c8_set_active_params(self.h, es, len(a), a.ctypes.data_as(_l.i32p)))

    def set_scatter(self, mode):
        m = {"colored": _l.C8_SCATTER_COLORED, "atomic": _l.C8_SCATTER_ATOMIC, "gather": _l.C8_SCATTER_GATHER}[mode]
        _l.check(self.L.c8_set_scatter_mode(self.h, m))

    @property
    def scatter(self):
        return {_l.C8_SCATTER_COLORED: "colored", _l.C8_SCATTER_ATOMIC: "atomic", _l.C8_SCATTER_GATHER: "gather"}[self.L.c8_get_scatter_mode(self.h)]

    def set_assign_mode(self, on):
        """scatter='gather': Jacobian assemblies assign A and b (zero_all + assembly in one call) instead of adding"""
        _l.check(self.L.c8_set_assign_mode(self.h, int(bool(on))))

    def set_gather_early_nodes(self, node_begin, node_end):
        """scatter='gather' in two parts: Jacobian assemblies sum the rows of nodes [node_begin, node_end) only (the
        ghost rows of a mesh part); gather_finish() sums the rest, e.g. while those rows are being exchanged."""
        _l.check(self.L.c8_set_gather_early_nodes(self.h, int(node_begin), int(node_end)))

    def gather_finish(self):
        return self._rc(self.L.c8_gather_finish(self.h))

    def set_shape_cache(self, on):
        """cached shape tables of the hex8 wave kernels (default on; 1.7 KB per element)"""
        _l.check(self.L.c8_set_shape_cache(self.h, int(bool(on))))

    def set_stage_chunk(self, min_chunk):
        """scatter='gather': smallest chunk of elements staged at a time (default 8192)"""
        _l.check(self.L.c8_set_stage_chunk(self.h, int(min_chunk)))

    def set_stage_overlap(self, on):
        """scatter='gather' in several chunks: the row sums of a chunk beside the assembly of the next one (second stream)"""
        _l.check(self.L.c8_set_stage_overlap(self.h, int(bool(on))))

    def set_kernel(self, variant):
        """'auto' | 'slot' (one lane group per element) | 'wave' (one wavefront per hex8 element) | 'wave_ad' (the same with the
        iterated, automatically differentiated local solve also where a model has a closed form)"""
        v = {"auto": _l.C8_KERNEL_AUTO, "slot": _l.C8_KERNEL_SLOT, "wave": _l.C8_KERNEL_WAVE, "wave_ad": _l.C8_KERNEL_WAVE_AD, "node": _l.C8_KERNEL_NODE}[variant]
        _l.check(self.L.c8_set_kernel_variant(self.h, v))

    def set_async(self, flag):
        _l.check(self.L.c8_set_async(self.h, int(flag)))

    def use_current_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        _l.check(self.L.c8_set_stream(self.h, C.c_void_p(s)))

    def status(self):
        rc = self.L.c8_status(self.h)
        if rc < -1:
            _l.check(rc)
        return rc

    # ---- the hot path --------------------------------------------------------------------------
    @staticmethod
    def _state(u, p, u_prev, p_prev, xi_prev, xi):
        for t in (u, p, u_prev, p_prev, xi_prev, xi):
            assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch.float64"
        s = _l.State()
        s.x[0], s.x[1] = u.data_ptr(), p.data_ptr()
        s.x_prev[0], s.x_prev[1] = u_prev.data_ptr(), p_prev.data_ptr()
        s.xi_prev, s.xi = xi_prev.data_ptr(), xi.data_ptr()
        return s

    def _rc(self, rc):
        if rc < -1:
            _l.check(rc)
        return rc

    def forward_jacobian(self, u, p, u_prev, p_prev, xi_prev, xi, ls):
        """eval_forward_jacobian: returns 0, or -1 if a local Newton solve failed."""
        st, sy = self._state(u, p, u_prev, p_prev, xi_prev, xi), ls.c_struct()
        return self._rc(self.L.c8_assemble_forward_jacobian(self.h, C.byref(st), C.byref(sy)))

    def forward_jacobian_subset(self, u, p, u_prev, p_prev, xi_prev, xi, ls, elems):
        """eval_forward_jacobian over the elements listed in `elems` (int32 device tensor), atomic adds."""
        st, sy = self._state(u, p, u_prev, p_prev, xi_prev, xi), ls.c_struct()
        return self._rc(self.L.c8_assemble_forward_jacobian_subset(self.h, C.byref(st), C.byref(sy),
                                                                    C.c_void_p(elems.data_ptr()), int(elems.numel())))

    def global_residual(self, u, p, u_prev, p_prev, xi_prev, xi, ls):
        st, sy = self._state(u, p, u_prev, p_prev, xi_prev, xi), ls.c_struct()
        return self._rc(self.L.c8_assemble_residual(self.h, C.byref(st), C.byref(sy)))

    def adjoint_jacobian(self, u, p, u_prev, p_prev, xi_prev, xi, g, f, ls):
        st, sy = self._state(u, p, u_prev, p_prev, xi_prev, xi), ls.c_struct()
        return self._rc(self.L.c8_assemble_adjoint_jacobian(self.h, C.byref(st), C.c_void_p(g.data_ptr()),
                                                            C.c_void_p(f.data_ptr()), C.byref(sy)))

    def solve_adjoint_local(self, u, p, u_prev, p_prev, xi_prev, xi, z_u, z_p, phi, g, f):
        st = self._state(u, p, u_prev, p_prev, xi_prev, xi)
        z = (C.c_void_p * 2)(z_u.data_ptr(), z_p.data_ptr())
        return self._rc(self.L.c8_solve_adjoint_local(self.h, C.byref(st), z, C.c_void_p(phi.data_ptr()),
                                                      C.c_void_p(g.data_ptr()), C.c_void_p(f.data_ptr())))

    def qoi_gradient(self, u, p, u_prev, p_prev, xi_prev, xi, z_u, z_p, phi, grad):
        st = self._state(u, p, u_prev, p_prev, xi_prev, xi)
        z = (C.c_void_p * 2)(z_u.data_ptr(), z_p.data_ptr())
        return self._rc(self.L.c8_param_gradient(self.h, C.byref(st), z, C.c_void_p(phi.data_ptr()),
                                                 C.c_void_p(grad.data_ptr())))

    # ---- next to the hot path: boundary conditions on the device system, y = A x ----------------------
    def apply_dirichlet(self, dbcs, x_u, x_p, ls, is_adjoint=False):
        """dbcs: list of (resid, eq, nodes int32 device tensor, values float64 device tensor)."""
        d = (_l.Dbc * max(1, len(dbcs)))()
        for k, (resid, eq, nodes, vals) in enumerate(dbcs):
            d[k] = _l.Dbc(resid, eq, nodes.numel(), nodes.data_ptr(), vals.data_ptr())
        x = (C.c_void_p * 2)(x_u.data_ptr(), x_p.data_ptr())
        sy = ls.c_struct()
        return self._rc(self.L.c8_apply_dirichlet(self.h, len(dbcs), d, x, C.byref(sy), int(is_adjoint)))

    def apply_traction(self, tbcs, ls):
        """tbcs: list of (resid, faces int32 device tensor [n][3|4], traction float64 device tensor [n][pts][3])."""
        t = (_l.Tbc * max(1, len(tbcs)))()
        for k, (resid, faces, tr) in enumerate(tbcs):
            t[k] = _l.Tbc(resid, faces.shape[0], faces.shape[1], faces.data_ptr(), tr.data_ptr())
        sy = ls.c_struct()
        return self._rc(self.L.c8_apply_traction(self.h, len(tbcs), t, C.byref(sy)))

    def apply_A(self, ls, x_u, x_p, y_u, y_p):
        x = (C.c_void_p * 2)(x_u.data_ptr(), x_p.data_ptr())
        y = (C.c_void_p * 2)(y_u.data_ptr(), y_p.data_ptr())
        sy = ls.c_struct()
        return self._rc(self.L.c8_apply_A(self.h, C.byref(sy), x, y))

    # ---- objective ----
    def set_qoi_avg_disp(self):
        _l.check(self.L.c8_set_qoi_avg_disp(self.h))

    def set_qoi_calibration(self, faces, weights=(1.0, 1.0, 1.0), balance=1.0, coord_idx=1, coord_value=0.0,
                            coord_tol=1e-8, comp=1, dt_over_T=1.0):
        """Calibration objective (calibration.cpp): faces = node ids of the displacement side set [n][3|4] (host),
        load plane = nodes with |x[coord_idx] - coord_value| < coord_tol, reaction component comp."""
        f = np.ascontiguousarray(faces if faces is not None and len(faces) else np.zeros((0, 1)), dtype=np.int32)
        if f.ndim == 1:  # tri3 meshes: element ids of the displacement term (none = every element)
            f = f.reshape(-1, 1)
        d = _l.CalibrationDesc()
        d.num_faces, d.nodes_per_face, d.faces = f.shape[0], f.shape[1], f.ctypes.data
        for k in range(3):
            d.weights[k] = float(weights[k])
        d.balance_factor, d.coord_idx, d.coord_value, d.coord_tol = float(balance), int(coord_idx), float(coord_value), float(coord_tol)
        d.reaction_comp, d.dt_over_total_time = int(comp), float(dt_over_T)
        _l.check(self.L.c8_set_qoi_calibration(self.h, C.byref(d)))

    def set_allreduce(self, dist, num_parts):
        """Multi-part objective sums: `dist` = an initialised torch.distributed module (SUM all-reduce of host doubles)."""
        import torch

        def cb(_user, vals, n):
            t = torch.from_numpy(np.ctypeslib.as_array(vals, shape=(n,)))
            if dist.get_backend() == "nccl":
                d = t.to(self.device)
                dist.all_reduce(d, op=dist.ReduceOp.SUM)
                t.copy_(d.cpu())
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)

        self._allreduce_cb = _l.ALLREDUCE_FN(cb)  # keep the callback alive
        _l.check(self.L.c8_set_allreduce(self.h, self._allreduce_cb, None, int(num_parts)))

    def set_measured(self, u_meas, load_meas):
        """measured nodal displacements of the step (device tensor, kept by reference) and measured load"""
        self._u_meas = u_meas
        _l.check(self.L.c8_set_measured(self.h, C.c_void_p(u_meas.data_ptr()), float(load_meas)))

    def qoi_preprocess(self, u, p, u_prev, p_prev, xi_prev, xi):
        """preprocess_qoi: (side-set area, total load, load mismatch)"""
        st = self._state(u, p, u_prev, p_prev, xi_prev, xi)
        out = (C.c_double * 3)()
        _l.check(self.L.c8_qoi_preprocess(self.h, C.byref(st), out))
        return tuple(out)

    def eval_qoi(self, u, p, J, xi_prev=None, xi=None, u_prev=None, p_prev=None):
        """eval_qoi: J (1-element device tensor) += QoI value.  The local state is only needed by QoIs
        that read it ("average displacement" does not)."""
        s = _l.State()
        s.x[0], s.x[1] = u.data_ptr(), p.data_ptr()
        s.x_prev[0], s.x_prev[1] = (u_prev if u_prev is not None else u).data_ptr(), (p_prev if p_prev is not None else p).data_ptr()
        s.xi_prev = xi_prev.data_ptr() if xi_prev is not None else None
        s.xi = xi.data_ptr() if xi is not None else None
        return self._rc(self.L.c8_eval_qoi(self.h, C.byref(s), C.c_void_p(J.data_ptr())))
