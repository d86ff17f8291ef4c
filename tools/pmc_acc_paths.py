#!/usr/bin/env python3
"""Which on-chip path does the accumulate launch saturate?  rocprofv3 --pmc passes (counters only) over tools/bench_acc_batch.py:
texture addresser / L1 / L2 activity next to the issue counters.  Counter names differ between rocprofiler builds: the list of
this box is saved first, unknown names just fail their pass.  usage (GPU box): pmc_acc_paths.py <out.json> [pairs] [env=val ...]"""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT); os.environ["TMPDIR"] = "/tmp"
out_path = sys.argv[1]
S = sys.argv[2] if len(sys.argv) > 2 else "32"
env = dict(os.environ)
for a in sys.argv[3:]:
    k, v = a.split("=", 1); env[k] = v
groups = [  # (a pass with TA_ADDR_STALLED_BY_* never returned on this pool: left out; every pass has a short timeout)
    "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE",
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum",
    "TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum",
    "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum",
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES",
]
res, errors = {}, {}
for gi, g in enumerate(groups):
    d = f"/tmp/accpath_{gi}"
    subprocess.run(["rm", "-rf", d])
    try:
      r = subprocess.run(["rocprofv3", "--pmc", *g.split(), "--output-format", "csv", "-d", d, "--", "python3", "tools/bench_acc_batch.py", S],
                         capture_output=True, text=True, timeout=360, env=env)
    except subprocess.TimeoutExpired:
        errors[g] = "timed out"
        json.dump({"pairs_per_launch": int(S), "env": sys.argv[3:], "counters": res, "failed_passes": errors}, open(out_path, "w"), indent=1)
        continue
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        errors[g] = (r.stderr or r.stdout)[-400:]
        continue
    acc = {}
    for x in csv.DictReader(open(f[0])):
        if "accumulate_staged_kernel" in x["Kernel_Name"]:
            acc.setdefault(x["Counter_Name"], []).append(float(x["Counter_Value"]))
    for c, v in acc.items():
        v.sort()
        big = [t for t in v if t > 0.5 * v[-1]] or v      # the S-pair launches (a single-pair launch is also in the run)
        res[c] = {"mean_per_launch": sum(big) / len(big), "dispatches": len(big)}
    json.dump({"pairs_per_launch": int(S), "env": sys.argv[3:], "counters": res, "failed_passes": errors}, open(out_path, "w"), indent=1)
json.dump({"pairs_per_launch": int(S), "env": sys.argv[3:], "counters": res, "failed_passes": errors}, open(out_path, "w"), indent=1)
print(json.dumps({k: round(v["mean_per_launch"], 1) for k, v in res.items()}, indent=0))
print("failed:", list(errors))
