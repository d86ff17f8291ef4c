"""CPU test of csrc/fast_log.hpp (the table-driven logarithm of the accumulate kernel, host + device from one
source): compiled for the host with g++ and compared with a 60-digit logarithm."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
from mpmath import log as mplog, mp, mpf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semantic-icp_amd", "csrc")

SRC = r'''
#include "fast_log.hpp"
static const double T[] = {
#include "log_table.inc"
};
extern "C" void fast_log(int n, const double* x, double* y) {
  for (int i = 0; i < n; ++i) {
    const unsigned o = sicp::log_entry_offset(x[i]) / 8;
    y[i] = sicp::log_from_entry(x[i], T[o], T[o + 1]);
  }
}
'''


def test_fast_log_is_accurate_to_an_ulp_on_its_whole_domain():
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(SRC)
        so = os.path.join(d, "t.so")
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I", CSRC, os.path.join(d, "t.cpp"), "-o", so], check=True)
        lib = ctypes.CDLL(so)
        rng = np.random.default_rng(0)
        x = np.concatenate([
            1.0 + rng.random(20000) * 1e-12, 1.0 + rng.random(20000) * 1e-6, 1.0 + rng.random(50000),       # what tiny residuals give
            np.exp(rng.uniform(0, 30, 50000)), 2.0 ** rng.integers(0, 1000, 2000) * (1 + rng.random(2000)),  # up to huge residuals
            np.array([1.0, np.nextafter(1.0, 2.0), 2.0, np.nextafter(2.0, 1.0), 1.5, 1e300]),
            1.0 + (np.arange(0, 257) / 256.0), np.nextafter(1.0 + (np.arange(1, 257) / 256.0), 0.0),         # both edges of every table interval
        ])
        y = np.empty_like(x)
        lib.fast_log(len(x), x.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), y.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    mp.dps = 60
    worst = 0.0
    for xi, yi in zip(x, y):
        ref = mplog(mpf(float(xi)))
        if ref == 0:
            assert yi == 0.0
            continue
        worst = max(worst, float(abs(mpf(float(yi)) - ref) / ref))
    assert worst < 3e-16, worst        # ~1 ulp (2.2e-16) everywhere, including x -> 1 (no cancellation: every term is >= 0)
    assert y[np.argmax(x == 1.0)] == 0.0
