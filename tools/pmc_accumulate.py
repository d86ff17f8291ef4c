#!/usr/bin/env python3
"""HBM traffic of the accumulate launch from rocprofv3 PMC counters (GPU box) -> gpurun_out/<prefix>_pmc_hbm_traffic.json
(copy into profiles/<round>/pmc_hbm_traffic.json: bench.py reports it as roofline.traffic).

FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), each over tools/bench_acc_batch.py <pairs>; FETCH_SIZE x 2 as
calibrated on this access pattern (profiles/r02/r02_fetch_calibration.txt: the gfx950 counter tallies 128-byte requests
as 64), both in KB.  A third and fourth pass count the launch's instructions (SQ_INSTS_*, SQ_ACTIVE_INST_VALU,
GRBM_GUI_ACTIVE): the kernel's SECOND roof, the vector issue rate -- a wave64 VALU instruction holds its 16-lane SIMD
for 4 cycles, 1024 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-instructions/s (an FP64 FMA per lane and cycle is the
78.6 TFLOP/s vector peak); bench.py reports the block as roofline.valu.  A pass that times out is recorded as such and the
others are kept.  usage: pmc_accumulate.py <prefix> [pairs ...]"""
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
    def one_pass(counters, tag):
        d = f"/tmp/acc_pmc_{tag}_{S}"
        subprocess.run(["rm", "-rf", d])
        try:
            r = subprocess.run(["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", "python3", "tools/bench_acc_batch.py", str(S)],
                               capture_output=True, text=True, timeout=240)
        except subprocess.TimeoutExpired:
            return None, "timed out"
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not f:
            return None, (r.stderr or r.stdout)[-300:]
        return f[0], None

    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f, err = one_pass([c], c)
        if not f:
            rec[c] = {"error": err}
            continue
        f = [f]
        vals = sorted(float(x["Counter_Value"]) for x in csv.DictReader(open(f[0]))
                      if x["Counter_Name"] == c and "accumulate_staged_kernel" in x["Kernel_Name"])
        big = [v for v in vals if v > 0.5 * vals[-1]]   # the S-pair launches (a single-pair launch is also in the run)
        rec[c] = {"dispatches": len(big), "mean_raw_KB": sum(big) / len(big)}
    if "mean_raw_KB" in rec.get("FETCH_SIZE", {}) and "mean_raw_KB" in rec.get("WRITE_SIZE", {}):
        rec["bytes_per_launch"] = (2.0 * rec["FETCH_SIZE"]["mean_raw_KB"] + rec["WRITE_SIZE"]["mean_raw_KB"]) * 1024.0
        rec["over_algorithmic"] = rec["bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    # ---- the instruction roof
    inst = {}
    for cs in (["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],
               ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "SQ_WAIT_INST_ANY"]):
        f, err = one_pass(cs, cs[0])
        if not f:
            inst["_error_" + cs[0]] = err
            continue
        acc = {}
        for x in csv.DictReader(open(f)):
            if "accumulate_staged_kernel" in x["Kernel_Name"]:
                acc.setdefault(x["Counter_Name"], []).append(float(x["Counter_Value"]))
        for c, v in acc.items():
            v.sort()
            big = [b for b in v if b > 0.5 * v[-1]]
            inst[c] = sum(big) / len(big)
    if inst.get("SQ_INSTS_VALU"):
        slots = S * n * K
        inst["slots_per_launch"] = slots
        # one lane evaluates one slot at a time: wave-instructions x 64 lanes / slots = lane-instructions per slot
        inst["valu_lane_instructions_per_slot"] = inst["SQ_INSTS_VALU"] * 64.0 / slots
        if inst.get("GRBM_GUI_ACTIVE"):
            cycles = inst["GRBM_GUI_ACTIVE"] / 8.0      # summed over the 8 XCDs
            inst["gpu_cycles"] = cycles
            inst["valu_issue_rounds_used_frac"] = inst.get("SQ_ACTIVE_INST_VALU", inst["SQ_INSTS_VALU"]) / (1024.0 * cycles / 4.0)
    rec["instructions"] = inst
    out[f"accumulate_batch_K{K}_pairs{S}_n{n}"] = rec
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/{pre}_pmc_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
