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


def test_library_exports_exactly_the_declared_symbols():
    """The version script (csrc/exports.map) keeps everything but the C ABI out of the dynamic symbol table."""
    import subprocess

    path = sicp.build()
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == declared_symbols()


def _entry_points(path):
    """(name, body) of every function defined at column 0 inside the extern "C" blocks of a source file"""
    src = open(path).read()
    out = []
    for m in re.finditer(r"^(?:int|const char\*) (sicp_[a-z0-9_]+)\(", src, flags=re.M):
        end = src.index("\n}\n", m.start())
        out.append((m.group(1), src[m.start():end]))
    return out


def test_every_entry_point_runs_inside_the_exception_barrier():
    """include/sicp.h promises that no exception crosses the ABI: every status-returning entry point is one
    `return abi_guard(...)` statement, and the stream's worker thread catches everything too."""
    csrc = os.path.join(ROOT, "semantic-icp_amd", "csrc")
    seen = set()
    for f in ("sicp_api.cpp", "streams.cpp"):
        for name, body in _entry_points(os.path.join(csrc, f)):
            seen.add(name)
            if body.startswith("const char*"):
                continue  # sicp_version / sicp_strerror / sicp_last_error return stored strings; sicp_stream_last_error has its own try
            first_stmt = body[body.index("{") + 1:].strip()
            assert first_stmt.startswith("return abi_guard("), name
    assert seen == set(declared_symbols())
    streams = open(os.path.join(csrc, "streams.cpp")).read()
    worker = streams[streams.index("void stream_worker(sicp_stream_ctx* S)"):]
    assert "catch (const std::bad_alloc&)" in worker and "catch (...)" in worker
    assert "std::thread(stream_worker" in streams
    # developer logs are behind the SICP_DEBUG gate
    for f in ("solve.cpp", "stages.cpp", "streams.cpp"):
        text = open(os.path.join(csrc, f)).read()
        for m in re.finditer(r'getenv\("(SICP_[A-Z_]*(?:LOG|STATS))"\)', text):
            line = text[text.rfind("\n", 0, m.start()):m.end()]
            assert "debug_enabled()" in line, (f, m.group(1))


def test_exception_barrier_maps_exceptions_to_statuses(tmp_path):
    """csrc/abi_barrier.hpp compiled on its own: bad_alloc -> OUT_OF_MEMORY, anything else -> INTERNAL, the
    description reaches the note, a throwing note is swallowed, a regular return passes through."""
    import subprocess, textwrap

    code = textwrap.dedent(
        r"""
        #include <cstdio>
        #include <stdexcept>
        #include <string>
        #include <vector>
        #include "abi_barrier.hpp"
        using namespace sicp::host;
        int main() {
          std::string said;
          auto note = [&](const char* w) { said = w ? w : ""; };
          int ok = 1;
          ok &= abi_guard([]() -> int { return 42; }) == 42;
          ok &= abi_guard([]() -> int { throw std::bad_alloc(); }) == SICP_ERR_OUT_OF_MEMORY;
          ok &= abi_guard([]() -> int { std::vector<int> v; v.at(3) = 1; return 0; }) == SICP_ERR_INTERNAL;
          ok &= abi_guard([]() -> int { throw 7; }) == SICP_ERR_INTERNAL;
          ok &= abi_guard_note([]() -> int { throw std::runtime_error("boom"); }, note) == SICP_ERR_INTERNAL && said == "boom";
          ok &= abi_guard_note([]() -> int { throw std::bad_alloc(); }, note) == SICP_ERR_OUT_OF_MEMORY && said.find("bad_alloc") != std::string::npos;
          ok &= abi_guard_note([]() -> int { throw 7; }, [](const char*) { throw std::runtime_error("note throws"); }) == SICP_ERR_INTERNAL;
          std::printf("%d\n", ok);
          return ok ? 0 : 1;
        }
        """
    )
    c = tmp_path / "t.cpp"
    c.write_text(code)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "semantic-icp_amd", "csrc"),
                    str(c), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "1"


def test_new_status_codes_have_messages():
    assert "internal" in sicp._strerror(sicp.ERR_INTERNAL).lower()
    assert "memory" in sicp._strerror(sicp.ERR_OUT_OF_MEMORY).lower()
