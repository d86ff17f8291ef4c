#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests_6.txt 2>&1; tail -6 $O/gpu_tests_6.txt
timeout 900 python3 bench.py --steps 5 > $O/bench_full_6.json 2> $O/bench_full_6.err; echo "bench exit $?"
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_full_6.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_align_alone", "ms_per_icp_iter_alone")})
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_us", "traffic")})
print("step_roofline", json.dumps(d["step_roofline"], indent=0)[:3000])
print("cpu", d.get("cpu_baseline", {}).get("value"))
for w in d["other_workloads"]:
    print(round(w["value"] / 1e9, 3), w.get("ms_per_step"), w.get("ms_per_align"), w.get("pairs_per_s_end_to_end"), w["workload"][:100])
PY
bash tools/probe_ref_deps.sh > $O/probe_ref_deps_gpubox.txt 2>&1; tail -3 $O/probe_ref_deps_gpubox.txt
