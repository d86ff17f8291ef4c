"""Recipe: compile the reference's OWN driver programs (exec/*.cc), unchanged and where they lie under
/root/reference, against the product's class shims (semantic-icp_amd/host) and link them with
semantic-icp_amd/libsicp.so.  Outputs go to oracle/_ref/drivers/ only (git-ignored, not gpurun-ignored: the
binaries travel to the GPU box like the built .so files; no reference SOURCE is copied anywhere).

This is the "link unchanged" boundary check of BASELINE.json's north_star / SURVEY.md section 8(b): the main() of
kitti_eval / nyu_eval / scenenet_eval / roc_eval / test_icp and the three small utilities is the reference's, the
registration classes they instantiate are this repository's.  It is NOT an oracle: these programs run the product
engine.  (The reference's algorithm itself -- its header-only classes over PCL / Ceres / Sophus / Eigen -- stays
unbuildable here: none of those libraries exist in the image)  exec/test_gradient.cc is not
in the list: it instantiates the reference's own Ceres cost function under ceres::GradientChecker, i.e. it tests the
code this engine replaces, not the boundary.

Flags: the reference's own (CMakeLists.txt:5: -std=c++11 -O3).  TEST INFRASTRUCTURE: run by tests/ and
__graft_entry__.build() only."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_EXEC = "/root/reference/exec"
OUT = os.path.join(ROOT, "oracle", "_ref", "drivers")
HOST = os.path.join(ROOT, "semantic-icp_amd", "host")
PKG = os.path.join(ROOT, "semantic-icp_amd")
DRIVERS = ["test_icp", "kitti_eval", "nyu_eval", "scenenet_eval", "roc_eval", "make_semantic", "pcd_read", "pcd_write"]


def command(name: str, out: str) -> list[str]:
    # rpath relative to the binary: oracle/_ref/drivers/ -> semantic-icp_amd/ (the tree moves to the GPU box as a whole)
    return ["g++", "-std=c++11", "-O3", "-I", os.path.join(HOST, "compat", "include"), "-I", HOST, "-I", os.path.join(ROOT, "include"),
            os.path.join(REF_EXEC, name + ".cc"), "-L", PKG, "-lsicp", "-Wl,-rpath,$ORIGIN/../../../semantic-icp_amd",
            "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", out]


def available() -> bool:
    return os.path.isdir(REF_EXEC)


def build(names=None, out_dir: str = OUT, verbose: bool = False) -> dict:
    """Returns {name: path}.  Raises RuntimeError with the compiler's output if a driver does not compile or link."""
    if not available():
        raise RuntimeError("the reference tree is not present (it only exists in the build container)")
    names = list(names or DRIVERS)
    os.makedirs(out_dir, exist_ok=True)

    def one(name):
        out = os.path.join(out_dir, name)
        r = subprocess.run(command(name, out), capture_output=True, text=True)
        if verbose:
            print(" ".join(command(name, out)))
            print(r.stderr)
        return name, out, r

    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 2)) as pool:
        results = list(pool.map(one, names))
    bad = [(n, r.stderr) for n, _, r in results if r.returncode != 0]
    if bad:
        raise RuntimeError("\n".join(f"{n}:\n{e[-3000:]}" for n, e in bad))
    return {n: o for n, o, _ in results}


if __name__ == "__main__":
    for n, p in build(sys.argv[1:] or None, verbose=False).items():
        print(n, "->", p)
