"""G5 of SURVEY.md 8c: every dual-number intrinsic of calibr8_amd/csrc/c8_math.hpp (the AD type of the kernels), one
operation at a time, against complex-step / central-difference derivatives.  Runs the header on the CPU through the
test emulator library; no GPU."""
import ctypes as C

import numpy as np
import pytest

import emul_lib as em

OPS = {
    0: lambda a, b: a + b, 1: lambda a, b: a - b, 2: lambda a, b: a * b, 3: lambda a, b: a / b,
    4: lambda a, b: np.sqrt(a), 5: lambda a, b: a ** (1.0 / 3.0), 6: lambda a, b: np.exp(a),
    7: lambda a, b: a ** b, 8: lambda a, b: b.real / a if isinstance(b, complex) else b / a, 9: lambda a, b: a / (b.real if isinstance(b, complex) else b),
}


def tens10(a, b):
    t = np.array([[a, a * 0.5, b], [b * 2.0, a + 1.0, a - b], [b, a * b, a + 2.0]])
    return np.linalg.det(t) + np.sqrt((t * t).sum())


def tens11(a, b):
    t = np.array([[a + 3.0, b, a * 0.1], [b * 0.2, a + 4.0, b], [a * 0.3, b * 0.1, a + 5.0]])
    return np.trace(np.linalg.inv(t))


def dual(op, a, da, b, db):
    L = em.lib()
    L.c8emu_dual_op.argtypes = [C.c_int] + [C.c_double] * 4 + [C.POINTER(C.c_double)]
    out = (C.c_double * 2)()
    assert L.c8emu_dual_op(op, a, da, b, db, out) == 0
    return out[0], out[1]


@pytest.mark.parametrize("op", sorted(OPS))
def test_scalar_intrinsics_against_complex_step(op):
    rng = np.random.default_rng(op)
    for _ in range(20):
        a, b = rng.uniform(0.3, 2.5, 2)
        da, db = rng.standard_normal(2)
        if op in (8, 9):
            db = 0.0  # the double argument carries no tangent
        v, d = dual(op, a, da, b, db)
        h = 1e-30
        f = OPS[op]
        ref = f(a, b)
        cs = f(complex(a, h * da), complex(b, h * db)).imag / h  # complex step: exact to rounding
        assert abs(v - ref) <= 1e-15 * abs(ref)
        assert abs(d - cs) <= 1e-14 * max(1.0, abs(cs)), (op, a, b, d, cs)


def test_pow_has_zero_derivative_at_zero_base():
    v, d = dual(7, 0.0, 1.0, 2.0, 0.5)  # Sacado's rule, used by the power-law hardening offset (hyper_J2.cpp:261)
    assert v == 0.0 and d == 0.0


@pytest.mark.parametrize("op,fn", [(10, tens10), (11, tens11)])
def test_tensor_helpers_against_central_differences(op, fn):
    rng = np.random.default_rng(7)
    for _ in range(10):
        a, b = rng.uniform(0.5, 2.0, 2)
        da, db = rng.standard_normal(2)
        v, d = dual(op, a, da, b, db)
        h = 1e-6
        fd = (fn(a + h * da, b + h * db) - fn(a - h * da, b - h * db)) / (2 * h)
        assert abs(v - fn(a, b)) < 1e-13 * abs(fn(a, b))
        assert abs(d - fd) < 1e-7 * max(1.0, abs(fd))
