"""Build the in-tree C-ABI library semantic-icp_amd/libsicp.so for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the tree.  Every source is compiled to an
object of its own (in parallel, only when it or a header is newer), then linked with a version script that exports
exactly the sicp_* entry points of include/sicp.h."""
from __future__ import annotations

import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libsicp.so")
OBJDIR = os.path.join(PKG, "build")  # git-ignored and gpurun-ignored: the .so is what travels
DEVICE_SOURCES = ["knn_kernels.hip", "feature_kernels.hip", "solve_kernels.hip", "build_tree.hip"]
HOST_SOURCES = ["memory.cpp", "clouds.cpp", "stages.cpp", "solve.cpp", "streams.cpp", "sicp_api.cpp"]
SOURCES = DEVICE_SOURCES + HOST_SOURCES
HEADERS = ["kernels.h", "device_geometry.hpp", "lm.hpp", "se3.hpp", "bvh.hpp", "build_tree.h", "fast_log.hpp", "log_table.inc",
           "engine.hpp", "abi_barrier.hpp"]
EXPORTS = os.path.join(CSRC, "exports.map")
ARCH = "gfx950"


def _deps():
    return [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(ROOT, "include", "sicp.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES] + _deps() + [EXPORTS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False, out: str | None = None, extra_flags=()) -> str:
    """`out` / `extra_flags` build an experimental variant next to the product library (tuning aid:
    SICP_LIB=<path> makes the Python binding load it); variants keep their objects in a directory of their own."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *extra_flags,
             "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    tag = hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8] if extra_flags else "product"
    objdir = os.path.join(OBJDIR, tag)
    os.makedirs(objdir, exist_ok=True)
    newest_header = max(os.path.getmtime(d) for d in _deps())

    def compile_one(src: str):
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            return obj, None
        cmd = [hipcc, *flags, "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or r.returncode != 0:
            print(" ".join(cmd))
            print(r.stdout)
            print(r.stderr)
        return obj, (r.stderr if r.returncode != 0 else None)

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as pool:
        results = list(pool.map(compile_one, SOURCES))
    errors = [e for _, e in results if e]
    if errors:
        raise RuntimeError("hipcc failed:\n" + "\n".join(errors))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *[o for o, _ in results], "-o", out or LIB,
           f"-Wl,--version-script={EXPORTS}", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(" ".join(cmd))
        print(r.stdout)
        print(r.stderr)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return out or LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
