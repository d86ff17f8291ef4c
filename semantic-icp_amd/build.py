"""Build the in-tree C-ABI library semantic-icp_amd/libsicp.so for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the tree."""
from __future__ import annotations

import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libsicp.so")
SOURCES = ["knn_kernels.hip", "feature_kernels.hip", "solve_kernels.hip", "build_tree.hip", "sicp_api.cpp"]
HEADERS = ["kernels.h", "device_geometry.hpp", "lm.hpp", "se3.hpp", "bvh.hpp", "build_tree.h"]
ARCH = "gfx950"


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "sicp.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False, out: str | None = None, extra_flags=()) -> str:
    """`out` / `extra_flags` build an experimental variant next to the product library (tuning aid:
    SICP_LIB=<path> makes the Python binding load it)."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [
        hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", *extra_flags,
        "-I", os.path.join(ROOT, "include"), "-I", CSRC,
        *[os.path.join(CSRC, s) for s in SOURCES],
        "-o", out or LIB, "-Wl,-rpath,/opt/rocm/lib",
    ]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(" ".join(cmd))
        print(r.stdout)
        print(r.stderr)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr)
    return out or LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
