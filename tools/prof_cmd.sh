#!/bin/bash
# usage (on the GPU box): tools/prof_cmd.sh <tag> <python script> [args...]  -> top kernels by time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; shift
rm -rf gpurun_out/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 "$@" > gpurun_out/${tag}.log 2>&1
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no kernel stats"; tail -5 gpurun_out/${tag}.log; exit 1; fi
cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:8.2f} max_us={float(r['MaxNs'])/1e3:8.2f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
PY
rm -rf gpurun_out/prof_$tag
