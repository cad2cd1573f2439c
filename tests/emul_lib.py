"""ctypes binding of tests/emul/libc8emul.so: the kernel source run on the CPU (TEST INFRASTRUCTURE).

`Emul` has the same call shape as oracle_lib.Oracle so that the same tests and drivers run on the
oracle, on the emulated kernels, and (gpu_backend.GpuBackend) on the HIP library."""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL_DIR = os.path.join(HERE, "emul")
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
_lib = None
K_FORWARD, K_RESIDUAL, K_ADJ_JAC, K_ADJ_LOCAL, K_GRAD, K_QOI, K_FORWARD_WAVE, K_ADJ_JAC_WAVE, K_ADJ_LOCAL_WAVE, K_GRAD_WAVE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
K_RESIDUAL_WAVE, K_QOI_WAVE, K_QOI_PREPROCESS, K_FORWARD_NODE, K_ADJ_JAC_NODE, K_ADJ_LOCAL_CLOSED, K_GRAD_CLOSED = 11, 12, 13, 14, 15, 16, 17


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", EMUL_DIR, "-s"])
        L = C.CDLL(os.path.join(EMUL_DIR, "libc8emul.so"))
        L.c8emu_call.restype = C.c_int
        L.c8emu_call.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, ip, ip, C.c_int, C.c_char_p, C.c_double,
                                 C.c_int, C.c_double, C.c_double, dp, ip, C.POINTER(dp)]
        _lib = L
    return _lib


class Emul:
    def __init__(self, elem_type, coords, conn, local_type, params, elem_set=None, stab_mult=1.0, max_iters=500,
                 abs_tol=1e-12, rel_tol=1e-12):
        # discretisation queries (graph, sizes, initial state) come from an Oracle object: test-side only
        self.orc = ol.Oracle(elem_type, coords, conn, local_type, params, elem_set, stab_mult, max_iters, abs_tol,
                             rel_tol)
        o = self.orc
        self.local_type, self.stab_mult = local_type, stab_mult
        self.max_iters, self.abs_tol, self.rel_tol = max_iters, abs_tol, rel_tol
        self.nnodes, self.nelems, self.nn, self.npts, self.nloc = o.nnodes, o.nelems, o.nn, o.npts, o.nloc
        self.rowptr, self.colidx = o.rowptr, o.colidx
        self.params = o.params
        self.active = np.zeros((o.nsets, 10), dtype=np.int32)
        self.set_active(0, [0])
        self.wave = False  # True: K1 through the wave-per-element kernel (hex8 only)
        self.staged = False  # staged (gather) assembly of the two Jacobian kernels (hex8 slot K3 excepted)
        self.assign = False  # staged assembly: assign A and b instead of adding (c8_set_assign_mode)
        self.node = False    # True: K1 through the row-per-node kernel (hex8 models with a closed form; staged = its two-part form)
        self.closed = True   # wave K1 of a model with a closed form (small_J2): that form, as the library's default; False: AD / Newton
        self.ls = (1e-4, 0.5, 0.9, 4)  # local line search (Hosford / Barlat models): defaults of line_search.hpp:28-35

    def set_local_line_search(self, c1=1e-4, bmin=0.5, bmax=0.9, max_evals=4):
        self.ls = (c1, bmin, bmax, int(max_evals))
        self.orc.set_local_line_search(*self.ls)

    def new_state(self):
        return self.orc.new_state()

    def new_linsys(self):
        return self.orc.new_linsys()

    def set_params(self, p):
        self.params = np.ascontiguousarray(np.atleast_2d(np.asarray(p, dtype=np.float64)))

    def set_active(self, es, idx):
        self.active[es, 1] = len(idx)
        self.active[es, 2:2 + len(idx)] = idx
        ofs = 0
        for s in range(self.active.shape[0]):
            self.active[s, 0] = ofs
            ofs += self.active[s, 1]

    def _call(self, what, ptrs):
        o = self.orc
        self._push_objective()
        lib().c8emu_set_local_line_search.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
        lib().c8emu_set_local_line_search(*self.ls)
        arr = (dp * 18)()
        for k, a in ptrs.items():
            arr[k] = a.ctypes.data_as(dp)
        es = o._es.ctypes.data_as(ip) if o._es is not None else None
        return lib().c8emu_call(what, o.elem_type, o.nnodes, o.nelems, o.coords.ctypes.data_as(dp),
                                o.conn.ctypes.data_as(ip), es, o.nsets, self.local_type.encode(), self.stab_mult,
                                self.max_iters, self.abs_tol, self.rel_tol, self.params.ctypes.data_as(dp),
                                self.active.ctypes.data_as(ip), arr)

    @staticmethod
    def _fields(u, p, up, pp, xip, xi):
        return {0: u, 1: p, 2: up, 3: pp, 4: xip, 5: xi}

    @staticmethod
    def _sys(ls):
        return {6: ls.A[0][0], 7: ls.A[0][1], 8: ls.A[1][0], 9: ls.A[1][1], 10: ls.b[0], 11: ls.b[1]}

    def forward_jacobian(self, u, p, up, pp, xip, xi, ls):
        what = (K_FORWARD_NODE if self.node else K_FORWARD_WAVE if self.wave else K_FORWARD) | (256 if self.staged else 0) | (512 if self.assign else 0) | \
            (1024 if self.closed and self.max_iters >= 8 else 0)
        return self._call(what, {**self._fields(u, p, up, pp, xip, xi), **self._sys(ls)})

    def global_residual(self, u, p, up, pp, xip, xi, ls):
        return self._call(K_RESIDUAL_WAVE if self.wave else K_RESIDUAL, {**self._fields(u, p, up, pp, xip, xi), **self._sys(ls)})

    def adjoint_jacobian(self, u, p, up, pp, xip, xi, g, f, ls):
        what = (K_ADJ_JAC_NODE if self.node else K_ADJ_JAC_WAVE if self.wave else K_ADJ_JAC) | (256 if self.staged else 0) | (512 if self.assign else 0)
        return self._call(what, {**self._fields(u, p, up, pp, xip, xi), **self._sys(ls), 12: g, 13: f})

    def solve_adjoint_local(self, u, p, up, pp, xip, xi, z_u, z_p, phi, g, f):
        # node: the library's choice for a model with a closed form on hex8 (K4 in closed form beside the row-per-node assemblies)
        return self._call(K_ADJ_LOCAL_CLOSED if self.node else K_ADJ_LOCAL_WAVE if self.wave else K_ADJ_LOCAL, {**self._fields(u, p, up, pp, xip, xi), 12: g, 13: f, 14: z_u, 15: z_p,
                                        16: phi})

    def qoi_gradient(self, u, p, up, pp, xip, xi, z_u, z_p, phi, nparams):
        grad = np.zeros(nparams)
        self._call(K_GRAD_CLOSED if self.node else K_GRAD_WAVE if self.wave else K_GRAD, {**self._fields(u, p, up, pp, xip, xi), 14: z_u, 15: z_p, 16: phi, 17: grad})
        return grad

    # ---- Calibration objective (the emulator keeps one objective configuration, like a context) ----
    def set_calibration(self, faces, weights=(1.0, 1.0, 1.0), balance=1.0, coord_idx=1, coord_value=0.0, coord_tol=1e-8,
                        comp=1, dt_over_T=1.0):
        f = np.ascontiguousarray(faces if faces is not None and len(faces) else np.zeros((0, 1)), dtype=np.int32)
        self._cal = (f.reshape(-1, 1) if f.ndim == 1 else f, np.ascontiguousarray(weights, dtype=np.float64), balance,
                     coord_idx, coord_value, coord_tol, comp, dt_over_T)
        self._meas = None
        self.calibration = True

    def set_measured(self, u_meas, load_meas):
        self._meas = (np.ascontiguousarray(u_meas, dtype=np.float64), float(load_meas))

    def _push_objective(self):
        """the library keeps ONE objective configuration: install this object's before every call"""
        L = lib()
        cal = getattr(self, "_cal", None)
        if cal is None:
            L.c8emu_set_qoi_avg_disp()
            return
        f, w, balance, coord_idx, coord_value, coord_tol, comp, dt_over_T = cal
        L.c8emu_set_qoi_calibration.argtypes = [C.c_int, C.c_int, ip, dp, C.c_double, C.c_int, C.c_double, C.c_double,
                                                C.c_int, C.c_double]
        L.c8emu_set_qoi_calibration(f.shape[0], f.shape[1], f.ctypes.data_as(ip), w.ctypes.data_as(dp), balance, coord_idx,
                                    coord_value, coord_tol, comp, dt_over_T)
        if self._meas is not None:
            L.c8emu_set_measured.argtypes = [C.c_int, dp, C.c_double]
            L.c8emu_set_measured(len(self._meas[0]), self._meas[0].ctypes.data_as(dp), self._meas[1])

    def qoi_preprocess(self, u, p, up, pp, xip, xi):
        self._qoi_state = (u, p, up, pp, xip, xi)
        J = np.zeros(1)
        self._call(K_QOI_PREPROCESS | (0 if not self.wave else 0), {**self._fields(u, p, up, pp, xip, xi), 17: J})
        out = np.zeros(3)
        lib().c8emu_qoi_info.argtypes = [dp]
        lib().c8emu_qoi_info(out.ctypes.data_as(dp))
        return out

    def eval_qoi(self, u, p):
        J = np.zeros(1)
        st = getattr(self, "_qoi_state", None)
        if st is None:
            xi = self.new_state()
            self._call(K_QOI_WAVE if self.wave else K_QOI, {**self._fields(u, p, u, p, xi, xi), 17: J})
        else:  # calibration: the state of the step given to qoi_preprocess
            self._call(K_QOI_WAVE if self.wave else K_QOI, {**self._fields(u, p, st[2], st[3], st[4], st[5]), 17: J})
        return float(J[0])
