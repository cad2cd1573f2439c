"""Canonical parameter scaling against golden vectors produced by the reference's own
Python module (tests/golden/make_transform_fixture.py), plus the C++ formulas of objective.cpp:41-61,125-137."""
import json
import os

import numpy as np

from calibr8_amd import lib

HERE = os.path.dirname(os.path.abspath(__file__))


def encode(scales):
    kind, a, b = [], [], []
    for s in scales:
        if s is None:
            kind.append(lib.C8_SCALE_NONE); a.append(0.0); b.append(0.0)
        elif isinstance(s, float):
            kind.append(lib.C8_SCALE_LOG); a.append(s); b.append(0.0)
        else:
            kind.append(lib.C8_SCALE_BOUNDS); a.append(float(s[0])); b.append(float(s[1]))
    return np.array(kind, dtype=np.int32), np.array(a), np.array(b)


def transform(values, scales, from_canonical):
    L = lib.load_library()
    kind, a, b = encode(scales)
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = np.zeros_like(v)
    lib.check(L.c8_transform_params(len(v), v.ctypes.data_as(lib.dp), kind.ctypes.data_as(lib.i32p), a.ctypes.data_as(lib.dp),
                                    b.ctypes.data_as(lib.dp), int(from_canonical), out.ctypes.data_as(lib.dp)))
    return out


def test_transforms_match_reference_golden_vectors():
    L = lib.load_library()
    d = json.load(open(os.path.join(HERE, "golden", "parameter_transforms.json")))
    assert len(d["cases"]) >= 10
    for c in d["cases"]:
        scales = c["scales"]
        assert np.allclose(transform(c["canonical"], scales, True), c["physical"], rtol=1e-15, atol=0)
        if "to_canonical_of_physical" in c:
            assert np.allclose(transform(c["physical"], scales, False), c["to_canonical_of_physical"], rtol=1e-14, atol=1e-15)
            assert np.allclose(transform(c["physical_outside"], scales, False), c["to_canonical_of_outside"], rtol=1e-14, atol=1e-15)
            kind, a, b = encode(scales)
            g, v = np.array(c["grad"]), np.array(c["canonical"])
            out = np.zeros_like(g)
            lib.check(L.c8_transform_gradient(len(g), g.ctypes.data_as(lib.dp), v.ctypes.data_as(lib.dp), kind.ctypes.data_as(lib.i32p),
                                              a.ctypes.data_as(lib.dp), b.ctypes.data_as(lib.dp), out.ctypes.data_as(lib.dp)))
            assert np.allclose(out, c["grad_transformed"], rtol=1e-15, atol=0)


def test_bounds_scaling_is_the_cpp_formula():
    # objective.cpp:41-61: span = (hi-lo)/2, mean = (hi+lo)/2, clipped to [-1, 1]; gradient scaled by span (:125-137)
    lo, hi = np.array([800.0, 0.2, 90.0, 1.0]), np.array([1000.0, 0.3, 110.0, 3.0])
    scales = [[l, h] for l, h in zip(lo, hi)]
    p = np.array([1000.0, 0.25, 100.0, 2.0])
    canon = transform(p, scales, False)
    assert np.allclose(canon, (p - 0.5 * (hi + lo)) / (0.5 * (hi - lo)))
    assert np.allclose(transform(canon, scales, True), p)
    assert np.allclose(transform([2000.0, 0.0, 100.0, 2.0], scales, False)[:2], [1.0, -1.0])
