"""Builds calibr8_amd/libc8.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libc8.so")
SOURCES = ["c8_kernels.hip", "c8_api.hip", "c8_primal.hip", "c8_qoi.hip", "c8_host.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]
if os.environ.get("C8_STAMPS"):  # diagnostic build for tools/stamp_phases.py; its outputs are timing shares only
    FLAGS.append("-DC8_STAMPS")
FLAGS += os.environ.get("C8_EXTRA_FLAGS", "").split()  # kernel experiments (-D switches), never for shipped builds


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "c8.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libc8.so.  hipcc cross-compiles without a GPU."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
