#!/usr/bin/env python3
"""HBM traffic of the accumulate launch from rocprofv3 PMC counters (GPU box) -> gpurun_out/<prefix>_pmc_hbm_traffic.json
(copy into profiles/<round>/pmc_hbm_traffic.json: bench.py reports it as roofline.traffic).

FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), each over tools/bench_acc_batch.py <pairs>; FETCH_SIZE x 2 as
calibrated on this access pattern (profiles/r02/r02_fetch_calibration.txt: the gfx950 counter tallies 128-byte requests
as 64), both in KB.  usage: pmc_accumulate.py <prefix> [pairs ...]"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
os.environ["TMPDIR"] = "/tmp"
pre = sys.argv[1]
shapes = [int(a) for a in sys.argv[2:]] or [256, 32]
n, K = 100000, 4
out = {}
for S in shapes:
    rec = {"pairs": S, "points": n, "K": K, "algorithmic_bytes_per_launch": S * (24 * n + 32 * K * n)}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = f"/tmp/acc_pmc_{c}_{S}"
        subprocess.run(["rm", "-rf", d])
        r = subprocess.run(["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", d, "--", "python3", "tools/bench_acc_batch.py", str(S)],
                           capture_output=True, text=True, timeout=240)
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not f:
            rec[c] = {"error": (r.stderr or r.stdout)[-300:]}
            continue
        vals = sorted(float(x["Counter_Value"]) for x in csv.DictReader(open(f[0]))
                      if x["Counter_Name"] == c and "accumulate_staged_kernel" in x["Kernel_Name"])
        big = [v for v in vals if v > 0.5 * vals[-1]]   # the S-pair launches (a single-pair launch is also in the run)
        rec[c] = {"dispatches": len(big), "mean_raw_KB": sum(big) / len(big)}
    if "mean_raw_KB" in rec.get("FETCH_SIZE", {}) and "mean_raw_KB" in rec.get("WRITE_SIZE", {}):
        rec["bytes_per_launch"] = (2.0 * rec["FETCH_SIZE"]["mean_raw_KB"] + rec["WRITE_SIZE"]["mean_raw_KB"]) * 1024.0
        rec["over_algorithmic"] = rec["bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    out[f"accumulate_batch_K{K}_pairs{S}_n{n}"] = rec
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/{pre}_pmc_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
