#!/bin/bash
# probe_ref_deps.sh -- does this machine hold the third-party libraries the reference's own
# build needs (CMakeLists.txt:7-14: PCL >= 1.8, Eigen >= 3.3, Sophus, Ceres; PCL pulls FLANN)?
# If all of them are present, oracle/_ref could be built from the reference's headers;
# if not, the reference is unbuildable here and the oracle stays pinned by tests/golden only.
# Run once in the build container and once on the GPU box (through gpurun); the outputs are kept
# under profiles/.  Nothing is installed, nothing is written outside the given output file.
out=${1:-/dev/stdout}
{
  echo "# probe_ref_deps: $(date -u +%Y-%m-%dT%H:%M:%SZ) host=$(hostname) gpu=$(ls /dev/kfd >/dev/null 2>&1 && echo yes || echo no)"
  prefixes="/usr/include /usr/local/include /opt /usr/lib /usr/local/lib /usr/share /root /home"
  found_all=1
  for hdr in Eigen/Core sophus/se3.hpp ceres/ceres.h pcl/point_types.h flann/flann.hpp; do
    hit=$(find $prefixes -path "*/$hdr" 2>/dev/null | head -n 3 | tr '\n' ' ')
    if [ -z "$hit" ]; then echo "header $hdr: absent"; found_all=0; else echo "header $hdr: $hit"; fi
  done
  for lib in libceres libflann libpcl_common libpcl_kdtree; do
    hit=$(find /usr/lib /usr/local/lib /opt /lib -name "$lib*" 2>/dev/null | head -n 3 | tr '\n' ' ')
    if [ -z "$hit" ]; then echo "library $lib: absent"; else echo "library $lib: $hit"; fi
  done
  for pkg in eigen3 ceres-solver flann pcl_common; do
    if command -v pkg-config >/dev/null 2>&1 && pkg-config --exists "$pkg" 2>/dev/null; then
      echo "pkg-config $pkg: $(pkg-config --modversion $pkg)"
    else
      echo "pkg-config $pkg: absent"
    fi
  done
  python3 - <<'EOF'
import importlib
for m in ("pcl", "open3d", "pyceres", "sophuspy", "pyflann"):
    try:
        importlib.import_module(m)
        print(f"python module {m}: present")
    except Exception:
        print(f"python module {m}: absent")
EOF
  if [ $found_all = 1 ]; then echo "verdict: all headers present -- oracle/_ref is buildable"; else echo "verdict: reference unbuildable here (headers missing); no stand-ins are written"; fi
} > "$out" 2>&1
