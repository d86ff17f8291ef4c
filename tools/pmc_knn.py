#!/usr/bin/env python3
"""rocprofv3 evidence for the search kernels (GPU box): kernel-trace durations and PMC passes of
tools/bench_knn_jobs.py, one phase per run -> gpurun_out/<prefix>_pmc_knn.json (copy into profiles/<round>/).

Per kernel and phase: mean dispatch duration, HBM bytes per dispatch from FETCH_SIZE (x2: the gfx950 rocprofv3
counter tallies 128-byte requests as 64, MI355X_MICROARCH.md / profiles/r02/r02_fetch_calibration.txt) and
WRITE_SIZE in separate passes -> counter HBM GB/s; instruction counts per dispatch -> the issue-bound
fraction = wave-instructions / (1024 SIMDs x 2.4 GHz x duration), i.e. of one instruction per SIMD and cycle.
usage: pmc_knn.py <prefix> [pairs] [points]"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
os.environ["TMPDIR"] = "/tmp"
pre = sys.argv[1]
S = sys.argv[2] if len(sys.argv) > 2 else "16"
N = sys.argv[3] if len(sys.argv) > 3 else "100000"
REPS = "10"
PHASES = {"k4_first": "bvh_knn_packet_jobs_kernel<4", "k4_hinted": "bvh_knn_packet_jobs_kernel<4", "k20": "bvh_knn_packet_jobs_kernel<20"}
PASSES = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"],
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "GRBM_GUI_ACTIVE"],
]


def prof(args, phase):
    d = f"/tmp/pk_{phase}"
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", *args, "--output-format", "csv", "-d", d, "--", "python3", "tools/bench_knn_jobs.py", phase, S, N, REPS]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired as e:   # one hung counter pass must not lose the passes already collected
        r = subprocess.CompletedProcess(cmd, 124, stdout=(e.stdout or b"").decode(errors="ignore") if isinstance(e.stdout, bytes) else (e.stdout or ""),
                                        stderr="timed out after 240 s")
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    info = json.loads(line[-1]) if line else None
    return d, info, r


res = {}
for phase, kname in PHASES.items():
    d, info, r = prof(["--kernel-trace"], phase)
    if not info:
        res[phase] = {"error": (r.stderr or r.stdout)[-400:]}
        continue
    timed = info["phases"][phase]["timed_dispatches"]
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = [x for x in csv.DictReader(open(f[0])) if kname in x["Kernel_Name"]]
    rows.sort(key=lambda x: int(x["Start_Timestamp"]))
    rows = rows[-timed:]
    dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) * 1e-9 for x in rows]
    jobs_per_dispatch = int(S) / ((int(S) + 7) // 8)
    ent = {"kernel": rows[0]["Kernel_Name"].split("(")[0], "phase": phase, "pairs": int(S), "points": int(N),
           "dispatches": len(rows), "searches_per_dispatch": jobs_per_dispatch,
           "avg_dispatch_us": 1e6 * sum(dur) / len(dur), "us_per_search_kernel_trace": 1e6 * sum(dur) / len(dur) / jobs_per_dispatch,
           "us_per_search_hip_events_unprofiled_order": info["phases"][phase]["us_per_search"],
           "vgpr": rows[0].get("VGPR_Count"), "sgpr": rows[0].get("SGPR_Count"), "lds": rows[0].get("LDS_Block_Size"),
           "counters_per_dispatch": {}}
    for cs in PASSES:
        d, info2, r2 = prof(["--pmc", *cs], phase)
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not f:
            ent["counters_per_dispatch"]["_error_" + cs[0]] = (r2.stderr or "")[-300:]
            continue
        acc = {}
        for x in csv.DictReader(open(f[0])):
            if kname in x["Kernel_Name"]:
                acc.setdefault(x["Counter_Name"], []).append((int(x["Dispatch_Id"]), float(x["Counter_Value"])))
        for c, v in acc.items():
            v.sort()
            v = [b for _, b in v][-timed:]
            ent["counters_per_dispatch"][c] = sum(v) / len(v)
    c = ent["counters_per_dispatch"]
    t = ent["avg_dispatch_us"] * 1e-6
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        ent["hbm_bytes_per_dispatch"] = hbm
        ent["hbm_GBps"] = hbm / t / 1e9
        ent["hbm_frac_of_8TBps"] = hbm / t / 8e12
    alg = jobs_per_dispatch * (12 * int(N) + 12 * int(N) + 8 * (4 if "<4" in kname else 20) * int(N))
    ent["algorithmic_bytes_per_dispatch"] = alg
    insts = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
    if insts:
        ent["wave_instructions_per_dispatch"] = insts
        # A wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles, and a SIMD issues to each pipe (vector,
        # scalar, memory ...) at most once per 4-cycle round: the SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* counters
        # are in those 4-cycle units (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU).  GRBM_GUI_ACTIVE is summed over the 8 XCDs.
        rounds = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / 4.0 if c.get("GRBM_GUI_ACTIVE") else 2.4e9 * t / 4.0
        simds = 1024.0
        ent["issue_rounds_per_simd"] = rounds
        ent["valu_pipe_busy"] = c.get("SQ_ACTIVE_INST_VALU", c.get("SQ_INSTS_VALU", 0.0)) / (simds * rounds)
        ent["scalar_pipe_busy"] = c.get("SQ_ACTIVE_INST_SCA", c.get("SQ_INSTS_SALU", 0.0)) / (simds * rounds)
        if c.get("SQ_WAVE_CYCLES"):
            ent["waves_per_simd_resident"] = c["SQ_WAVE_CYCLES"] / (simds * rounds)
            ent["wave_time_waiting_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            ent["wave_time_issuing_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        ent["bound"] = "issue"
        if "SQ_WAVES" in c and c["SQ_WAVES"]:
            ent["instructions_per_wave"] = insts / c["SQ_WAVES"]
            ent["valu_per_wave"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_WAVES"]
            ent["salu_per_wave"] = c.get("SQ_INSTS_SALU", 0.0) / c["SQ_WAVES"]
    res[phase] = ent
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/{pre}_pmc_knn.json", "w"), indent=1)
print(json.dumps(res, indent=1))
