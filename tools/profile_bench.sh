#!/bin/bash
# usage (GPU box): tools/profile_bench.sh <prefix>
# rocprofv3 --kernel-trace --stats of the default bench command with the CPU leg off and SICP_NO_GRAPH=1 (rocprofv3 of
# ROCm 7.2 dies with a segmentation fault on hipGraphLaunch -- tools/r04/two_thread_dispatch.hip reproduces it without this
# library; with the ticks as plain launches the whole bench, stream legs included, profiles), the per-kernel statistics csv,
# and a summary of the accumulate launches that separates the roofline legs of bench.py -- the LAST 60 launches
# of accumulate_staged_kernel<4, true, 256> are its 256-pair leg (6 x 10 launches), the 300 before them its
# 32-pair leg (6 x 50): the launches bench.py brackets with HIP events -- from the launches of the timed
# region and the other workloads.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
pre=$1
export SICP_NO_GRAPH=1
for attempt in 1 2; do
  rm -rf /tmp/pb_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_prof -- python3 bench.py --no-cpu-baseline --no-dropin \
    > gpurun_out/${pre}_bench_under_rocprof.json 2> /tmp/pb_prof.err
  [ -n "$(find /tmp/pb_prof -name '*kernel_trace.csv' 2>/dev/null | head -1)" ] && break
  echo "attempt $attempt: no kernel trace (rocprofv3 crashed?)"
done
f=$(find /tmp/pb_prof -name '*kernel_stats.csv' | head -1)
t=$(find /tmp/pb_prof -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${pre}_kernel_stats_bench.csv
python3 - "$t" gpurun_out/${pre}_bench_under_rocprof.json > gpurun_out/${pre}_accumulate_launches.json <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "accumulate_staged_kernel<4, true, 256>" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
leg256, leg32, rest = dur[-60:], dur[-360:-60], dur[:-360]
bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
def st(v): return {"n": len(v), "mean_us": sum(v) / max(1, len(v)), "min_us": min(v) if v else None, "max_us": max(v) if v else None}
print(json.dumps({
    "kernel": "accumulate_staged_kernel<4, true, 256> (this instantiation only)",
    "roofline_leg_256_pairs_last_60_launches": dict(st(leg256), what="sicp_accumulate_batch over 256 pairs, 6 x 10 launches: bracketed with HIP events by bench.py (its first 2 x 10 are warm-up there)"),
    "roofline_leg_32_pairs_300_launches_before": dict(st(leg32), what="the same over 32 pairs, 6 x 50 launches"),
    "other_launches": dict(st(rest), what="timed region + warm-up + other workloads: 1..256 pairs inside a solve per launch, plus no-op launches at the tail of a tick"),
    "bench_roofline_same_run": {"avg_launch_us": bench["roofline"]["avg_launch_us"], "pairs_per_launch": bench["roofline"]["pairs_per_launch"],
                                "other_launch_shapes": bench["roofline"]["other_launch_shapes"]},
}, indent=1))
PY
cat gpurun_out/${pre}_accumulate_launches.json
head -14 gpurun_out/${pre}_kernel_stats_bench.csv | cut -c1-200
