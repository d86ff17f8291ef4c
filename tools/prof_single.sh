#!/bin/bash
# usage (on the GPU box): tools/prof_single.sh <tag> [bench args...]   -> gpurun_out/<tag>_kernel_stats.csv
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; shift
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_err.log
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
rm -rf gpurun_out/prof_$tag
