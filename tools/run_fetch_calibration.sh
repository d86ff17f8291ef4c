#!/bin/bash
# usage (GPU box): tools/run_fetch_calibration.sh <prefix>
# 1. FETCH_SIZE calibration on known byte counts (tools/fetch_calibration.hip)
# 2. FETCH_SIZE / WRITE_SIZE of the batched accumulate launch (32 pairs, 100K x 100K, K = 4), separate passes
# -> gpurun_out/<prefix>_fetch_calibration.txt, gpurun_out/<prefix>_pmc_hbm_traffic.json
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
pre=$1
out=gpurun_out/${pre}_fetch_calibration.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_calibration tools/fetch_calibration.hip || exit 1
{
  echo "# unprofiled run (HIP-event times)"
  /tmp/fetch_calibration
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/fc_pmc
    rocprofv3 --pmc $c --output-format csv -d /tmp/fc_pmc -- /tmp/fetch_calibration > /dev/null 2>&1
    f=$(find /tmp/fc_pmc -name '*counter_collection.csv' | head -1)
    echo "# rocprofv3 --pmc $c (raw counter value per dispatch, in dispatch order; every kernel runs twice)"
    python3 - "$f" $c <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[2]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
for r in rows:
    print(f'{sys.argv[2]} dispatch {r["Dispatch_Id"]:>3} {r["Kernel_Name"].split("(")[0]:<10} {float(r["Counter_Value"]):.1f}')
PY
  done
} > $out 2>&1
cat $out
python3 - $pre <<'PY'
import csv, glob, json, os, subprocess, sys
pre = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = f"/tmp/acc_pmc_{c}"
    subprocess.run(["rm", "-rf", d])
    subprocess.run(["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", d, "--", "python3", "tools/bench_acc_batch.py", "32"],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    vals = []
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c and "accumulate_staged_kernel" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
    vals.sort()
    big = [v for v in vals if v > 0.5 * vals[-1]]  # the 32-pair launches (one single-pair launch is also in the run)
    res[c] = {"dispatches": len(big), "mean_raw": sum(big) / len(big)}
json.dump(res, open(f"gpurun_out/{pre}_pmc_raw.json", "w"), indent=1)
print(res)
PY
