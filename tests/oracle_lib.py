"""ctypes binding for oracle/libc8oracle.so (TEST INFRASTRUCTURE).

The oracle is the CPU restatement of the reference algorithm; it is the checker
for the HIP path and is never imported by the product package `calibr8_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libc8oracle.so")

TRI3, TET4, HEX8 = 3, 4, 8
NUM_PARAMS = {"elastic": 4, "small_J2": 6, "hyper_J2": 8}

_lib = None
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def build():
    src = os.path.join(ORACLE_DIR, "c8_oracle.cpp")
    if (not os.path.exists(LIB_PATH)) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB_PATH)
    ):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.c8o_create.restype = C.c_void_p
        L.c8o_create.argtypes = [C.c_int, C.c_int, C.c_int, dp, ip, ip, C.c_int, C.c_char_p, C.c_double,
                                 C.c_int, C.c_double, C.c_double, dp, C.c_int, C.c_int, ip]
        L.c8o_destroy.argtypes = [C.c_void_p]
        L.c8o_nloc.argtypes = [C.c_void_p]
        L.c8o_ndims.argtypes = [C.c_void_p]
        L.c8o_nres.argtypes = [C.c_void_p]
        L.c8o_set_thickness.argtypes = [C.c_void_p, C.c_double]
        L.c8o_set_local_line_search.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int]
        L.c8o_npts.argtypes = [C.c_void_p]
        L.c8o_set_params.argtypes = [C.c_void_p, dp]
        L.c8o_set_active.argtypes = [C.c_void_p, C.c_int, C.c_int, ip]
        L.c8o_init_variables.argtypes = [C.c_void_p, dp]
        L.c8o_graph_nnz.restype = C.c_int64
        L.c8o_graph_nnz.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.c8o_graph.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), ip]
        L.c8o_forward_jacobian.restype = C.c_int
        L.c8o_forward_jacobian.argtypes = [C.c_void_p] + [dp] * 12
        L.c8o_forward_jacobian_mt.restype = C.c_int
        L.c8o_forward_jacobian_mt.argtypes = [C.c_void_p, C.c_int] + [dp] * 12
        L.c8o_global_residual.argtypes = [C.c_void_p] + [dp] * 8
        L.c8o_adjoint_jacobian.argtypes = [C.c_void_p] + [dp] * 14
        L.c8o_solve_adjoint_local.argtypes = [C.c_void_p] + [dp] * 11
        L.c8o_eval_qoi.restype = C.c_double
        L.c8o_eval_qoi.argtypes = [C.c_void_p, dp, dp]
        L.c8o_qoi_gradient.argtypes = [C.c_void_p] + [dp] * 10
        L.c8o_qoi_gradient_abs.argtypes = [C.c_void_p] + [dp] * 11
        L.c8o_set_calibration.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, dp, C.c_double, C.c_int, C.c_double,
                                          C.c_double, C.c_int, C.c_double]
        L.c8o_set_avg_disp.argtypes = [C.c_void_p]
        L.c8o_set_measured.argtypes = [C.c_void_p, dp, C.c_double]
        L.c8o_qoi_preprocess.restype = C.c_double
        L.c8o_qoi_preprocess.argtypes = [C.c_void_p] + [dp] * 7
        L.c8o_kit_npts.restype = C.c_int
        L.c8o_kit_npts.argtypes = [C.c_int, C.c_int]
        L.c8o_kit_point.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp]
        L.c8o_shape.restype = C.c_double
        L.c8o_shape.argtypes = [C.c_int, dp, dp, dp, dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(dp)


def _i(a):
    return a.ctypes.data_as(ip)


class LinSys:
    """Four CSR value arrays + two residual vectors over the oracle's graphs."""

    def __init__(self, oracle):
        self.A = [[np.zeros(oracle.nnz[i][j]) for j in range(2)] for i in range(2)]
        self.b = [np.zeros(oracle.nnodes * getattr(oracle, "ndims", 3)), np.zeros(oracle.nnodes)]

    def zero(self):
        for i in range(2):
            self.b[i][:] = 0.0
            for j in range(2):
                self.A[i][j][:] = 0.0


class Oracle:
    def __init__(self, elem_type, coords, conn, local_type, params, elem_set=None, stab_mult=1.0,
                 max_iters=500, abs_tol=1e-12, rel_tol=1e-12, extra_pairs=None):
        L = lib()
        self.L = L
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.nnodes = self.coords.shape[0]
        self.nelems = self.conn.shape[0]
        self.elem_type = elem_type
        self.local_abs_tol = abs_tol
        params = np.atleast_2d(np.asarray(params, dtype=np.float64))
        self.nsets = params.shape[0]
        self.params = np.ascontiguousarray(params)
        es = None
        if elem_set is not None:
            es = np.ascontiguousarray(elem_set, dtype=np.int32)
        self._es = es
        self.h = L.c8o_create(elem_type, self.nnodes, self.nelems, _d(self.coords), _i(self.conn),
                              _i(es) if es is not None else None, self.nsets, local_type.encode(), stab_mult,
                              max_iters, abs_tol, rel_tol, _d(self.params), self.params.shape[1],
                              0 if extra_pairs is None else len(extra_pairs),
                              None if extra_pairs is None or len(extra_pairs) == 0 else
                              _i(np.ascontiguousarray(extra_pairs, dtype=np.int32)))
        if not self.h:
            raise RuntimeError("c8o_create failed")
        self.nloc = L.c8o_nloc(self.h)
        self.npts = L.c8o_npts(self.h)
        self.ndims = L.c8o_ndims(self.h)  # 3, or 2 for tri3 (u has ndims equations per node; coords stay [n][3])
        self.nres = L.c8o_nres(self.h)    # 2 (`mechanics`), or 1 for the *_plane_stress models (`mechanics_plane_stress`)
        self.nn = self.conn.shape[1]
        self.ndofs = (self.ndims + (1 if self.nres == 2 else 0)) * self.nn
        self.nnz = [[L.c8o_graph_nnz(self.h, i, j) for j in range(2)] for i in range(2)]
        self.rowptr = [[None, None], [None, None]]
        self.colidx = [[None, None], [None, None]]
        neq = [self.ndims, 1]
        for i in range(2):
            for j in range(2):
                rp = np.zeros(self.nnodes * neq[i] + 1, dtype=np.int64)
                ci = np.zeros(self.nnz[i][j], dtype=np.int32)
                L.c8o_graph(self.h, i, j, rp.ctypes.data_as(C.POINTER(C.c_int64)), _i(ci))
                self.rowptr[i][j], self.colidx[i][j] = rp, ci

    def __del__(self):
        if getattr(self, "h", None):
            self.L.c8o_destroy(self.h)
            self.h = None

    def set_local_line_search(self, c1=1e-4, bmin=0.5, bmax=0.9, max_evals=4):
        """the `line search:` sublist of the local residual (Hosford / Barlat models); defaults of line_search.hpp:28-35"""
        self.L.c8o_set_local_line_search(self.h, c1, bmin, bmax, int(max_evals))

    def set_thickness(self, t):
        self.L.c8o_set_thickness(self.h, float(t))

    def set_params(self, params):
        self.params = np.ascontiguousarray(np.atleast_2d(np.asarray(params, dtype=np.float64)))
        self.L.c8o_set_params(self.h, _d(self.params))

    def set_active(self, es, idx):
        a = np.ascontiguousarray(idx, dtype=np.int32)
        self.L.c8o_set_active(self.h, es, len(a), _i(a))

    def new_state(self):
        xi = np.zeros((self.nelems, self.npts, self.nloc))
        self.L.c8o_init_variables(self.h, _d(xi))
        return xi

    def new_linsys(self):
        return LinSys(self)

    def forward_jacobian(self, u, p, u_prev, p_prev, xi_prev, xi, ls, nthreads=1):
        a = [_d(u), _d(p), _d(u_prev), _d(p_prev), _d(xi_prev), _d(xi), _d(ls.A[0][0]), _d(ls.A[0][1]),
             _d(ls.A[1][0]), _d(ls.A[1][1]), _d(ls.b[0]), _d(ls.b[1])]
        if nthreads > 1:
            return self.L.c8o_forward_jacobian_mt(self.h, nthreads, *a)
        return self.L.c8o_forward_jacobian(self.h, *a)

    def global_residual(self, u, p, u_prev, p_prev, xi_prev, xi, ls):
        self.L.c8o_global_residual(self.h, _d(u), _d(p), _d(u_prev), _d(p_prev), _d(xi_prev), _d(xi),
                                   _d(ls.b[0]), _d(ls.b[1]))

    def adjoint_jacobian(self, u, p, u_prev, p_prev, xi_prev, xi, g, f, ls):
        self.L.c8o_adjoint_jacobian(self.h, _d(u), _d(p), _d(u_prev), _d(p_prev), _d(xi_prev), _d(xi), _d(g),
                                    _d(f), _d(ls.A[0][0]), _d(ls.A[0][1]), _d(ls.A[1][0]), _d(ls.A[1][1]),
                                    _d(ls.b[0]), _d(ls.b[1]))

    def solve_adjoint_local(self, u, p, u_prev, p_prev, xi_prev, xi, z_u, z_p, phi, g, f):
        self.L.c8o_solve_adjoint_local(self.h, _d(u), _d(p), _d(u_prev), _d(p_prev), _d(xi_prev), _d(xi),
                                       _d(z_u), _d(z_p), _d(phi), _d(g), _d(f))

    def eval_qoi(self, u, p):
        return self.L.c8o_eval_qoi(self.h, _d(u), _d(p))

    # ---- Calibration QoI (calibration.cpp) ----
    def set_calibration(self, faces, weights=(1.0, 1.0, 1.0), balance=1.0, coord_idx=1, coord_value=0.0,
                        coord_tol=1e-8, comp=1, dt_over_T=1.0):
        """faces: [nfaces][3 or 4] global node ids of the displacement side set; load plane: x[coord_idx] = value."""
        f = np.ascontiguousarray(faces if faces is not None and len(faces) else np.zeros((0, 1)), dtype=np.int32)
        if f.ndim == 1:  # tri3 meshes: a list of element ids
            f = f.reshape(-1, 1)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        self.L.c8o_set_calibration(self.h, f.shape[0], f.shape[1], f.ctypes.data_as(ip), _d(w), balance, coord_idx,
                                   coord_value, coord_tol, comp, dt_over_T)
        self.calibration = True

    def set_measured(self, u_meas, load_meas):
        self.L.c8o_set_measured(self.h, _d(np.ascontiguousarray(u_meas, dtype=np.float64)), float(load_meas))

    def qoi_preprocess(self, u, p, u_prev, p_prev, xi_prev, xi):
        """preprocess_qoi: returns (area, total load, load mismatch) of the step."""
        out = np.zeros(3)
        self.L.c8o_qoi_preprocess(self.h, _d(u), _d(p), _d(u_prev), _d(p_prev), _d(xi_prev), _d(xi), _d(out))
        return out

    def qoi_gradient(self, u, p, u_prev, p_prev, xi_prev, xi, z_u, z_p, phi, nparams):
        grad = np.zeros(nparams)
        self.L.c8o_qoi_gradient(self.h, _d(u), _d(p), _d(u_prev), _d(p_prev), _d(xi_prev), _d(xi), _d(z_u),
                                _d(z_p), _d(phi), _d(grad))
        return grad


def _oracle_qoi_gradient_with_scale(self, u, p, u_prev, p_prev, xi_prev, xi, z_u, z_p, phi, nparams):
    """(grad, grad_abs): grad_abs[i] = sum of the magnitudes of all products summed into grad[i] -- the scale for
    judging another evaluation order of the same sums (a component may be a cancelled sum)."""
    grad, gabs = np.zeros(nparams), np.zeros(nparams)
    self.L.c8o_qoi_gradient_abs(self.h, _d(u), _d(p), _d(u_prev), _d(p_prev), _d(xi_prev), _d(xi), _d(z_u),
                                _d(z_p), _d(phi), _d(grad), _d(gabs))
    return grad, gabs


Oracle.qoi_gradient_with_scale = _oracle_qoi_gradient_with_scale


def kit_points(elem_type, ip_set):
    L = lib()
    n = L.c8o_kit_npts(elem_type, ip_set)
    pts, wts = np.zeros((n, 3)), np.zeros(n)
    for k in range(n):
        x, w = np.zeros(3), C.c_double(0.0)
        L.c8o_kit_point(elem_type, ip_set, k, _d(x), C.byref(w))
        pts[k], wts[k] = x, w.value
    return pts, wts


def shape(elem_type, X, xi):
    L = lib()
    X = np.ascontiguousarray(X, dtype=np.float64)
    nn = X.shape[0]
    N, dN = np.zeros(nn), np.zeros((nn, 3))
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    dv = L.c8o_shape(elem_type, _d(X), _d(xi), _d(N), _d(dN))
    return N, dN, dv
