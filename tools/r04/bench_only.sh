#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_final; mkdir -p $O
SECONDS=0; timeout 900 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench exit $? after $SECONDS s"
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04_final/bench_final.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_align_alone")}, "frac", d["roofline"]["frac"], "step frac", d["step_roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
for w in d["other_workloads"]:
    print(round(w["value"] / 1e9, 3), w.get("ms_per_step"), w.get("ms_per_pair"), w.get("accumulate_launch"), w["workload"][:80])
PY
