"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/sicp.h declares, and fails loudly (no fallback) when there is no GPU."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sicp = importlib.import_module("semantic-icp_amd")


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sicp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sicp_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    path = sicp.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sicp.h but not exported"


def test_struct_layouts_match_the_header():
    # sizes computed by the C compiler for the same declarations
    import subprocess, tempfile, textwrap

    code = textwrap.dedent(
        """
        #include <stdio.h>
        #include "sicp.h"
        int main(void) { printf("%zu %zu\\n", sizeof(sicp_params), sizeof(sicp_stats)); return 0; }
        """
    )
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(code)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        a, b = map(int, subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split())
    assert ctypes.sizeof(sicp.SicpParams) == a
    assert ctypes.sizeof(sicp.SicpStats) == b


def test_default_params_reproduce_reference_literals():
    em = sicp.default_params(sicp.MODE_EM)
    assert (em.knn, em.k_cov, em.epsilon, em.gate_sq, em.cauchy_a, em.use_sqloss) == (4, 20, 1e-3, 250.0, 3.0, 1)
    assert (em.outer_tol, em.max_outer, em.max_lm_iterations) == (1e-5, 50, 400)
    assert em.gradient_tolerance == em.function_tolerance == 0.1 * 1e-10
    g = sicp.default_params(sicp.MODE_GICP)
    assert (g.knn, g.cauchy_a, g.use_sqloss, g.outer_tol, g.max_outer) == (1, 3.0, 1, 1e-5, 50)
    s = sicp.default_params(sicp.MODE_SEMANTIC)
    assert (s.knn, s.cauchy_a, s.use_sqloss, s.outer_tol, s.max_outer, s.min_class_pts) == (1, 1.5, 0, 1e-3, 35, 400)
    with pytest.raises(sicp.SicpError):
        sicp.default_params(7)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_gpu_means_loud_failure_not_fallback():
    assert sicp.device_count() == 0
    with pytest.raises(sicp.SicpError) as e:
        sicp.Engine(0)
    assert e.value.status == sicp.ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """Only tests/, bench.py's cpu_baseline leg and smoke() may use oracle/."""
    pkg = os.path.join(ROOT, "semantic-icp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), (dirpath, f)
    assert "oracle" not in open(os.path.join(ROOT, "include", "sicp.h")).read().lower()
