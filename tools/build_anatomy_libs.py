#!/usr/bin/env python3
"""Developer builds of libsicp.so with one ingredient of the accumulate kernel removed each (timing
only: their sums are wrong by construction) -> build_dbg/, for tools/accumulate_anatomy.sh."""
import importlib.util, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "semantic-icp_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
os.makedirs(os.path.join(ROOT, "build_dbg"), exist_ok=True)
VARIANTS = {
    "nocompute": ["-DSICP_DEBUG_NOCOMPUTE"],                        # every loaded value consumed once, no arithmetic
    "nogather": ["-DSICP_DEBUG_NOGATHER"],                          # every target gather reads record 0
    "nostream": ["-DSICP_DEBUG_NOSTREAM", "-DSICP_DEBUG_NOGATHER"],  # all loads hit the cache: arithmetic + reduction only
    "nostream_nocompute": ["-DSICP_DEBUG_NOSTREAM", "-DSICP_DEBUG_NOGATHER", "-DSICP_DEBUG_NOCOMPUTE"],  # the loop's skeleton
    "noreduce": ["-DSICP_DEBUG_NOREDUCE"],                          # no chunk-end reduction (all 28 sums stay live)
}
for name, flags in VARIANTS.items():
    print(b.build_lib(out=os.path.join(ROOT, "build_dbg", f"libsicp_{name}.so"), extra_flags=flags))
