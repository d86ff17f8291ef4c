"""How often does the trust-region machine compiled for the host (g++, glibc sin / cos) differ from the device build (ocml sincos)
on sequences with steps of ANY size?  (tests/test_gpu_surface.py asserts equality on small rotations only.)  usage (GPU box): lm_host_vs_device.py"""
import importlib, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_surface as T
sicp = importlib.import_module("semantic-icp_amd")
rng = np.random.default_rng(7)
wild = T.lm_sequences(rng, 4000, True)
src = open(os.path.join(ROOT, "tests", "test_gpu_surface.py")).read()
code = src[src.index('#include <cstdio>\n#include <vector>\n#define SICP_HD'):src.index('""")\n    exe = tmp_path / "lm_seq"')]
d = tempfile.mkdtemp()
open(d + "/a.cc", "w").write(code)
subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "semantic-icp_amd", "csrc"), d + "/a.cc", "-o", d + "/a"], check=True)
with open(d + "/in", "wb") as f:
    f.write(np.int32(wild.shape[0]).tobytes()); f.write(wild.tobytes())
subprocess.run([d + "/a", d + "/in", d + "/out"], check=True)
host = np.fromfile(d + "/out").reshape(wild.shape[0], sicp.LM_SEQUENCE_OUT)
p = sicp.default_params(sicp.MODE_GICP)
with sicp.Engine(0, p) as e:
    dev = e.se3_device(sicp.LM_SEQUENCE, wild)
same = np.all((host == dev) | (np.isnan(host) & np.isnan(dev)), axis=1)
dec = np.all(host[:, 31:] == dev[:, 31:], axis=1)
rel = np.nanmax(np.abs(host[:, :14] - dev[:, :14]) / (1e-300 + np.abs(host[:, :14])), axis=1)
print({"sequences": int(same.size), "bit_equal": int(same.sum()), "same_decisions": int(dec.sum()),
       "max_rel_pose_difference_where_decisions_agree": float(rel[dec].max())})
