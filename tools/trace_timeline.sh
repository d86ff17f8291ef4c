#!/bin/bash
# usage (GPU box): tools/trace_timeline.sh <tag> <bin_ms> <last_ms> [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; bin=$2; last=$3; shift 3
rm -rf /tmp/tl_$tag
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python3 bench.py --timed-only "$@" > gpurun_out/tl_${tag}.json 2> gpurun_out/tl_${tag}.err
f=$(find /tmp/tl_$tag -name '*kernel_trace.csv' 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no trace"; tail -3 gpurun_out/tl_${tag}.err; exit 1; fi
tail -1 gpurun_out/tl_${tag}.json | cut -c1-200
python3 tools/trace_timeline.py "$f" $bin $last $ZOOM0 $ZOOM1 $ZOOMBIN
