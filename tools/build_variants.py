"""Experimental builds of libsicp.so next to the product library (build_dbg/libsicp_<name>.so; SICP_LIB=<path> makes the
Python binding load one).  usage: build_variants.py name=flag,flag ..."""
import importlib.util, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "semantic-icp_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
os.makedirs(os.path.join(ROOT, "build_dbg"), exist_ok=True)
for a in sys.argv[1:]:
    name, flags = a.split("=", 1)
    print(b.build_lib(out=os.path.join(ROOT, "build_dbg", f"libsicp_{name}.so"), extra_flags=[f for f in flags.split(",") if f]))
