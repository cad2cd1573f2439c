"""Builds calibr8_amd/libc8.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libc8.so")
KERNEL_PARTS = 6  # c8_kernels.hip is compiled once per group of template instantiations (-DC8_KERNEL_PART=n), in parallel
SOURCES = ["c8_kernels.hip:%d" % k for k in range(KERNEL_PARTS)] + ["c8_api.hip", "c8_primal.hip", "c8_qoi.hip", "c8_halo.hip",
                                                                      "c8_host.cpp", "c8_lbfgs.cpp"]
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]


def flags():
    """Compiler flags of this build.  C8_STAMPS=1 is the diagnostic build of tools/stamp_phases.py (timing shares
    only); C8_EXTRA_FLAGS holds tuning switches (-DC8_TUNE_*: same results, different timing).  Both are recorded in
    the library (c8_build_info) and tests/test_abi.py refuses a library built with either."""
    f = list(BASE_FLAGS)
    if os.environ.get("C8_STAMPS"):
        f.append("-DC8_STAMPS")
    return f + os.environ.get("C8_EXTRA_FLAGS", "").split()


def _inputs():
    return sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp"))] +
                  [os.path.join(HERE, "..", "include", "c8.h")])


def build_id(fl=None):
    """Identity of a build: sha256 over the flag set and every source and header.  Baked into the library
    (c8_build_info) and stamped into the PMC profiles, so that bench.py can tell whether a committed traffic figure
    belongs to the library it is timing."""
    h = hashlib.sha256(" ".join(fl if fl is not None else flags()).encode())
    for p in _inputs():
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def _stamp_path():
    return os.path.join(CSRC, ".flags")


def _stale():
    """True when the library is missing, older than a source, or was built from another flag set or source state
    (the stamp holds the flags and the build id of the last link)."""
    if not os.path.exists(LIB) or not os.path.exists(_stamp_path()):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(d) > t for d in _inputs()):
        return True
    return open(_stamp_path()).read() != " ".join(flags()) + "\n" + build_id()


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libc8.so.  hipcc cross-compiles without a GPU.
    An object is recompiled when its source, any header, or the flag set changed (force: all of them)."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    fl = flags()
    flags_now = " ".join(fl)
    bid = build_id(fl)
    stamp = _stamp_path()
    old = open(stamp).read().split("\n") if os.path.exists(stamp) else [""]
    flags_same = old[0] == flags_now
    headers = [p for p in _inputs() if p.endswith((".hpp", ".h"))]
    newest_header = max(os.path.getmtime(h) for h in headers)
    objs, jobs = [], []
    for entry in SOURCES:
        src, _, part = entry.partition(":")
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ("_p" + part if part else "") + ".o")
        srcp = os.path.join(CSRC, src)
        fresh = flags_same and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), newest_header)
        if src == "c8_api.hip":  # carries the build id
            fresh = fresh and len(old) > 1 and old[1] == bid
        if force or not fresh:
            extra = ['-DC8_BUILD_ID="%s"' % bid, '-DC8_BUILD_FLAGS="%s"' % flags_now] if src == "c8_api.hip" else []
            if part:
                extra.append("-DC8_KERNEL_PART=" + part)
            jobs.append([hipcc] + fl + extra + ["-x", "hip", "-c", srcp, "-o", obj])
        objs.append(obj)
    if jobs:  # independent translation units: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)

        workers = max(1, min(len(jobs), int(os.environ.get("C8_BUILD_JOBS", "0")) or (os.cpu_count() or 4)))
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(run, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    open(stamp, "w").write(flags_now + "\n" + bid)
    return LIB


if __name__ == "__main__":
    # python -m calibr8_amd.build [--all]: recompile what changed (--all: everything) and relink
    import sys
    if os.path.exists(LIB):
        os.remove(LIB)
    print(build(force="--all" in sys.argv, verbose=True))
