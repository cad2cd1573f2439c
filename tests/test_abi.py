"""The C-ABI library loads on a machine without a GPU and exports every symbol include/c8.h
declares; calls that need a device fail loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "c8.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(c8_[A-Za-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from calibr8_amd import lib
    L = lib.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libc8.so does not export %s" % n
    assert sorted(s[0] for s in lib.SYMBOLS) == names  # the ctypes table is complete too


def test_library_is_a_product_build():
    """The library the tests (and the GPU box) load carries no tuning or diagnostic switch: calibr8_amd/build.py records
    its flag set in the binary, and a stale object built from other flags is rebuilt (build._stale)."""
    from calibr8_amd import build, lib
    info = lib.load_library().c8_build_info().decode()
    assert info.startswith("id=") and "flags=" in info
    assert "C8_TUNE" not in info and "C8_STAMPS" not in info and "C8_EXPERIMENT" not in info, info
    if not (os.environ.get("C8_EXTRA_FLAGS") or os.environ.get("C8_STAMPS")):
        assert info.split()[0] == "id=" + build.build_id(), "libc8.so was built from other sources or flags: python -m calibr8_amd.build"


def test_host_helpers_work_without_gpu():
    import calibr8_amd
    c, conn = calibr8_amd.brick_mesh(3, 2, 2, 3.0, 2.0, 2.0)
    assert c.shape == (36, 3) and conn.shape == (12, 8)
    assert np.allclose(c.max(0), [3, 2, 2]) and conn.max() == 35
    from meshes import brick
    c2, conn2, _ = brick(3, 2, 2, 3.0, 2.0, 2.0)
    assert np.allclose(c, c2) and np.array_equal(conn, conn2)
    part = calibr8_amd.brick_partition(4, 4, 4, 2, 2, 2)
    assert sorted(np.unique(part)) == list(range(8)) and (np.bincount(part) == 8).all()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import calibr8_amd
    from calibr8_amd import lib
    c, conn = calibr8_amd.brick_mesh(2, 2, 2)
    with pytest.raises(RuntimeError):
        calibr8_amd.Assembler(8, c, conn, "small_J2", [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0])
    # and straight through the ABI: c8_create reports a device error, it does not compute on the host
    L = lib.load_library()
    params = np.array([1000.0, 0.25, 100.0, 2.0, 0.0, 0.0])
    md = lib.MeshDesc(8, len(c), len(conn), 1, c.ctypes.data_as(lib.dp), conn.ctypes.data_as(lib.i32p), None, 0, None)
    mo = lib.ModelDesc(b"mechanics", b"small_J2", 1.0, 10, 1e-12, 1e-12, 6, params.ctypes.data_as(lib.dp))
    h = C.c_void_p()
    rc = L.c8_create(C.byref(md), C.byref(mo), C.byref(h))
    assert rc == lib.C8_ERR_DEVICE and not h.value
    assert b"no HIP device" in L.c8_last_error()
    # argument checking
    mo2 = lib.ModelDesc(b"mechanics", b"no_such_model", 1.0, 10, 1e-12, 1e-12, 6, params.ctypes.data_as(lib.dp))
    assert L.c8_create(C.byref(md), C.byref(mo2), C.byref(h)) == lib.C8_ERR_UNSUPPORTED
    assert b"unknown local residual name" in L.c8_last_error()


def test_product_does_not_reference_the_oracle_or_emulator():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "calibr8_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in text and "libc8oracle" not in text and "c8o_" not in text, f
                assert "libc8emul" not in text and "emul_lib" not in text, f
