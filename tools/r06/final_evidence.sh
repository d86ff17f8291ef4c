#!/bin/bash
# round 6: the evidence of the final build (GPU box) -> gpurun_out/r06_final/ (copied into profiles/r06/ afterwards)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_final; mkdir -p $O
bash tools/probe_ref_deps.sh $O/probe_ref_deps_gpubox.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.txt 2>&1; tail -4 $O/gpu_tests_final.txt
timeout 900 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench exit $?"
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_final/bench_final.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_align_alone", "ms_per_icp_iter_alone", "ms_per_icp_iter_amortised")})
r = d["roofline"]
print("roofline", {k: r.get(k) for k in ("achieved", "frac", "avg_launch_us", "traffic", "frac_by_counter_traffic")}, "valu", r.get("valu"), "step frac", d["step_roofline"]["frac"])
print("cpu", d.get("cpu_baseline", {}).get("value"), "pose delta", d.get("pose_delta_vs_cpu"))
for w in d["other_workloads"]:
    print(round(w.get("value", 0) / 1e9, 3), w.get("ms_per_step"), w.get("ms_per_align"), w.get("pairs_per_s_end_to_end"), w.get("pairs_per_s_align_only"), w["workload"][:90])
PY
# the whole bench under rocprofv3 (ticks as plain launches) + the accumulate launches of its roofline legs
bash tools/profile_bench.sh r06_final/r06 > $O/profile_bench.log 2>&1; tail -12 $O/profile_bench.log | cut -c1-220
# the timed region alone
rm -rf /tmp/tl_prof
SICP_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl_prof -- python3 bench.py --timed-only --steps 4 --warmup 1 > $O/bench_timed_only_under_rocprof.json 2> $O/bench_timed_only_under_rocprof.err
echo "timed-only under rocprofv3: exit $?"
f=$(find /tmp/tl_prof -name '*kernel_trace.csv' | head -1); s=$(find /tmp/tl_prof -name '*kernel_stats.csv' | head -1)
[ -n "$s" ] && cp "$s" $O/kernel_stats_timed_region.csv
[ -n "$f" ] && python3 tools/trace_timeline.py "$f" 10 700 > $O/timeline_timed_region.txt
# counters: HBM traffic + instruction counts of the accumulate launch, the search kernels
timeout 1500 python3 tools/pmc_accumulate.py r06_final/r06 256 32 > $O/pmc_accumulate.log 2>&1; tail -3 $O/pmc_accumulate.log
timeout 1500 python3 tools/pmc_knn.py r06_final/r06 > $O/pmc_knn.log 2>&1; tail -3 $O/pmc_knn.log | cut -c1-300
timeout 300 python3 tools/one_pair_latency.py > $O/one_pair_latency.txt 2>&1; tail -1 $O/one_pair_latency.txt
ls $O
