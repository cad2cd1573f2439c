"""ctypes binding of tests/emul/libc8emul.so: the kernel source run on the CPU (TEST INFRASTRUCTURE)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL_DIR = os.path.join(HERE, "emul")
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", EMUL_DIR, "-s"])
        L = C.CDLL(os.path.join(EMUL_DIR, "libc8emul.so"))
        L.c8emu_forward_jacobian.restype = C.c_int
        L.c8emu_forward_jacobian.argtypes = [C.c_int, C.c_int, C.c_int, dp, ip, ip, C.c_int, C.c_char_p, C.c_double,
                                             C.c_int, C.c_double, C.c_double, dp] + [dp] * 12
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(dp)


def forward_jacobian(orc, u, p, u_prev, p_prev, xi_prev, xi, ls, local_type, stab_mult=1.0, max_iters=500,
                     abs_tol=1e-12, rel_tol=1e-12):
    """Same call shape as Oracle.forward_jacobian; mesh/params are taken from the Oracle object."""
    L = lib()
    es = orc._es.ctypes.data_as(ip) if orc._es is not None else None
    return L.c8emu_forward_jacobian(orc.elem_type, orc.nnodes, orc.nelems, _d(orc.coords),
                                    orc.conn.ctypes.data_as(ip), es, orc.nsets, local_type.encode(), stab_mult,
                                    max_iters, abs_tol, rel_tol, _d(orc.params), _d(u), _d(p), _d(u_prev),
                                    _d(p_prev), _d(xi_prev), _d(xi), _d(ls.A[0][0]), _d(ls.A[0][1]),
                                    _d(ls.A[1][0]), _d(ls.A[1][1]), _d(ls.b[0]), _d(ls.b[1]))
