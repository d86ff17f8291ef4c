#!/bin/bash
# usage (GPU box): tools/trace_run.sh <tag> [bench args]  -> concurrency analysis of the kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; shift
rm -rf gpurun_out/tr_$tag
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_$tag -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/tr_${tag}.json 2> gpurun_out/tr_${tag}.err
f=$(find gpurun_out/tr_$tag -name '*kernel_trace.csv' 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no trace"; tail -3 gpurun_out/tr_${tag}.err; exit 1; fi
python3 tools/trace_concurrency.py "$f"
rm -rf gpurun_out/tr_$tag
