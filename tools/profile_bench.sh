#!/bin/bash
# usage (GPU box): tools/profile_bench.sh <prefix>
# rocprofv3 --kernel-trace --stats of the default bench command (sequence / CPU legs off), the per-kernel
# statistics csv, and a summary of the accumulate launches that separates the roofline leg of bench.py
# (the last 400 launches: 8 x 50 launches over all 32 pairs, the ones bench.py times with HIP events)
# from the launches of the timed region (1..32 pairs inside a solve each).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
pre=$1
rm -rf /tmp/pb_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_prof -- python3 bench.py --no-cpu-baseline --sequence-pairs 0 \
  > gpurun_out/${pre}_bench_under_rocprof.json 2> /tmp/pb_prof.err
f=$(find /tmp/pb_prof -name '*kernel_stats.csv' | head -1)
t=$(find /tmp/pb_prof -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${pre}_kernel_stats_bench.csv
python3 - "$t" gpurun_out/${pre}_bench_under_rocprof.json > gpurun_out/${pre}_accumulate_launches.json <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "accumulate_staged_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
leg = dur[-400:]
rest = dur[:-400]
bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(json.dumps({
    "kernel": rows[0]["Kernel_Name"].split("(")[0] if rows else None,
    "all_launches": {"n": len(dur), "mean_us": sum(dur) / len(dur), "min_us": min(dur), "max_us": max(dur)},
    "roofline_leg_last_400_launches": {"n": len(leg), "mean_us": sum(leg) / len(leg), "min_us": min(leg), "max_us": max(leg),
                                       "what": "sicp_accumulate_batch over all 32 pairs, 8 x 50 launches: the launches bench.py brackets with HIP events"},
    "other_launches": {"n": len(rest), "mean_us": sum(rest) / max(1, len(rest)),
                       "what": "timed region + warm-up + other workloads: 1..128 pairs inside a solve per launch, plus no-op launches at the tail of a tick"},
    "bench_roofline_avg_launch_us_same_run": bench["roofline"]["avg_launch_us"],
}, indent=1))
PY
cat gpurun_out/${pre}_accumulate_launches.json
head -12 gpurun_out/${pre}_kernel_stats_bench.csv
