#!/usr/bin/env python3
"""Anatomy of ONE pair alone (sicp_align, the reference's call pattern): rocprofv3 kernel trace of
bench.py --pairs-in-flight 1 --timed-only -> per-kernel durations and the gaps between consecutive kernels of
the solve.  usage (GPU box): tools/trace_single.py <prefix>"""
import csv, glob, json, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
os.environ["TMPDIR"] = "/tmp"
pre = sys.argv[1]
d = "/tmp/ts_prof"
subprocess.run(["rm", "-rf", d])
r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--pairs-in-flight", "1",
                    "--timed-only", "--steps", "20", "--warmup", "3"], capture_output=True, text=True, timeout=900)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
bench = json.loads(line[-1]) if line else {}
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda x: int(x["Start_Timestamp"]))
rows = rows[len(rows) // 4:]    # skip set-up and warm-up
agg = collections.defaultdict(list)
for x in rows:
    agg[x["Kernel_Name"].split("(")[0][:60]].append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
gaps = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    ka, kb = a["Kernel_Name"].split("(")[0][:40], b["Kernel_Name"].split("(")[0][:40]
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if -50 < g < 200:
        gaps[ka + " -> " + kb].append(g)
out = {"ms_per_align": bench.get("ms_per_step"), "outer_iters_per_align": bench.get("outer_iters_per_align"),
       "evals_per_outer": bench.get("accumulate_passes_per_outer_iter"),
       "kernels_us": {k: {"n": len(v), "mean": sum(v) / len(v), "total_ms": sum(v) / 1e3} for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))},
       "gaps_us": {k: {"n": len(v), "mean": sum(v) / len(v)} for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]}}
json.dump(out, open(f"gpurun_out/{pre}_single_pair_anatomy.json", "w"), indent=1)
print(json.dumps(out, indent=1))
