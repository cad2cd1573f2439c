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


# ---- compiled clients of include/c8.h (tests/abi_client): the boundary without ctypes -------------------------------------
ABI_OUT = "/tmp/c8_abi_client"


def _build_abi_clients():
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "abi_client"), "-s", "OUT=" + ABI_OUT])


def test_header_compiles_as_c_and_cpp_and_matches_the_ctypes_mirror():
    """include/c8.h compiled by gcc -std=c11 and g++ -std=c++17 (-pedantic -Werror): sizeof / offsetof of every struct and
    the enum values, compared with calibr8_amd/lib.py -- the layouts every GPU test relies on through ctypes."""
    import json
    import subprocess
    from calibr8_amd import lib
    _build_abi_clients()
    c_layout = json.loads(subprocess.check_output([os.path.join(ABI_OUT, "layout_c")]))
    cpp_layout = json.loads(subprocess.check_output([os.path.join(ABI_OUT, "layout_cpp")]))
    assert c_layout == cpp_layout
    mirror = {"c8_mesh_desc": lib.MeshDesc, "c8_model_desc": lib.ModelDesc, "c8_state": lib.State, "c8_system": lib.System,
              "c8_calibration_desc": lib.CalibrationDesc, "c8_dbc": lib.Dbc, "c8_tbc": lib.Tbc, "c8_newton_opts": lib.NewtonOpts,
              "c8_halo_desc": lib.HaloDesc, "c8_lbfgs_opts": lib.LbfgsOpts, "c8_lbfgs_result": lib.LbfgsResult}
    structs = {k: v for k, v in c_layout.items() if k != "enums"}
    assert sorted(structs) == sorted(mirror)  # every struct of the header has a mirror and the other way round
    for name, cls in mirror.items():
        lay = structs[name]
        assert C.sizeof(cls) == lay["sizeof"], name
        fields = [f for f in lay if f != "sizeof"]
        assert fields == [f[0] for f in cls._fields_], (name, fields)  # same members, same order
        for f in fields:
            assert getattr(cls, f).offset == lay[f], (name, f)
    for k, v in c_layout["enums"].items():
        assert getattr(lib, k) == v, k
    # the clients that run on the GPU box link against libc8.so here (no device needed to link)
    assert os.path.exists(os.path.join(ABI_OUT, "client_c")) and os.path.exists(os.path.join(ABI_OUT, "client_cpp"))


@pytest.mark.gpu
@pytest.mark.parametrize("binary", ["client_c", "client_cpp"])
def test_compiled_client_matches_the_python_path_bitwise(binary, tmp_path):
    """tests/abi_client/abi_client.c (C11 and C++17 builds): c8_create -> c8_assemble_forward_jacobian (twice) ->
    c8_assemble_residual -> c8_halo_build / attach / gather / scatter_x -> c8_comm_allreduce_sum, straight through the C ABI;
    the same inputs through calibr8_amd.Assembler give the same bits."""
    import subprocess
    import torch
    import calibr8_amd
    _build_abi_clients()
    out = str(tmp_path / "abi.bin")
    msg = subprocess.check_output([os.path.join(ABI_OUT, binary), out]).decode()
    assert msg.startswith("abi_client ok")
    raw = np.fromfile(out)
    nnodes, nelems = int(raw[0]), int(raw[1])
    nnz = [int(v) for v in raw[2:6]]
    nx, ny = int(raw[6]), int(raw[7])
    nz = nelems // (nx * ny)
    ofs = [8]
    for n in [3 * nnodes, nnodes] + nnz + [3 * nnodes, nnodes, nelems * 8 * 7]:
        ofs.append(ofs[-1] + n)
    assert ofs[-1] == len(raw)
    part = [raw[ofs[k]:ofs[k + 1]] for k in range(len(ofs) - 1)]
    u_h, p_h, A, b, xi_c = part[0], part[1], part[2:6], part[6:8], part[8]
    c, conn = calibr8_amd.brick_mesh(nx, ny, nz, 1.0, 0.8, 0.6)
    asm = calibr8_amd.Assembler(8, c, conn, "small_J2", [1000.0, 0.25, 100.0, 2.0, 0.0, 0.0])
    assert [asm.nnz[i][j] for i in range(2) for j in range(2)] == nnz
    u, p = asm.dev(u_h), asm.dev(p_h)
    ls, xi = asm.new_linsys(), asm.new_state()
    z, zp = torch.zeros_like(u), torch.zeros_like(p)
    for _ in range(2):
        assert asm.forward_jacobian(u, p, z, zp, asm.new_state(), xi, ls) == 0
    assert asm.global_residual(u, p, z, zp, asm.new_state(), xi, ls) == 0
    torch.cuda.synchronize()
    for k, (i, j) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        assert np.array_equal(ls.A[i][j].cpu().numpy(), A[k]) and np.abs(A[k]).max() > 0
    for i in range(2):  # (the residual-only assembly adds with atomics: its share of b agrees to rounding, not bit for bit)
        assert np.abs(ls.b[i].cpu().numpy() - b[i]).max() < 1e-13 * np.abs(b[i]).max() and np.abs(b[i]).max() > 0
    assert np.array_equal(xi.cpu().numpy().ravel(), xi_c) and (xi_c.reshape(-1, 7)[:, 6] > 0).any()
