#!/usr/bin/env python3
"""Instruction histogram of a kernel's basic blocks from hipcc -save-temps output (development aid).
usage: asm_hist.py <file.s> <substring of the kernel's mangled name> [min block size]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 150
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if key in l and not l.startswith((".", "\t")) and ":" in l and l.split(":")[0].strip().startswith("_Z"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].strip() == "s_endpgm")
blocks, cur, name = [], [], lines[start]
for l in lines[start + 1:end + 1]:
    t = l.strip()
    t = t.split(";")[0].strip()
    if not t or t.startswith("."):
        if t.startswith(".LBB") and t.endswith(":"):
            blocks.append((name, cur)); cur, name = [], t
        continue
    if t.endswith(":"):
        blocks.append((name, cur)); cur, name = [], t
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
print("kernel", lines[start], "blocks", len(blocks), "instructions", sum(len(b) for _, b in blocks))
for n, b in blocks:
    if len(b) >= minsz:
        h = collections.Counter(b)
        cls = collections.Counter()
        for op, c in h.items():
            k = ("valu_f64" if re.match(r"v_(fma|mul|add|fmac|rcp|rsq|sqrt|frexp|ldexp|min|max|cvt_f64|trig|div|cmp|cndmask_b64|mov_b64).*f64", op) or "f64" in op
                 else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_load") and not op.startswith("s_barrier")
                 else "wait" if op.startswith("s_waitcnt") else "smem" if op.startswith("s_load") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
            cls[k] += c
        print(f"\n{n} {len(b)} instr  classes {dict(cls)}")
        print("  " + "  ".join(f"{op}:{c}" for op, c in h.most_common(40)))
