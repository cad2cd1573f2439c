"""Builds calibr8_amd/libc8.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libc8.so")
SOURCES = ["c8_kernels.hip", "c8_api.hip", "c8_primal.hip", "c8_qoi.hip", "c8_host.cpp", "c8_lbfgs.cpp"]
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
    """Compile every HIP source for gfx950 and link libc8.so.  hipcc cross-compiles without a GPU.
    An object is recompiled when its source, any header, or the flag set changed (force: all of them)."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    stamp = os.path.join(CSRC, ".flags")
    flags_now = " ".join(FLAGS)
    flags_same = os.path.exists(stamp) and open(stamp).read() == flags_now
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(HERE, "..", "include", "c8.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        srcp = os.path.join(CSRC, src)
        fresh = flags_same and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), newest_header)
        if force or not fresh:
            cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", srcp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    open(stamp, "w").write(flags_now)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    # python -m calibr8_amd.build [--all]: recompile what changed (--all: everything) and relink
    import sys
    if os.path.exists(LIB):
        os.remove(LIB)
    print(build(force="--all" in sys.argv, verbose=True))
