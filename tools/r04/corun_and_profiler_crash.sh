#!/bin/bash
# round 4, GPU experiment 3: (a) is the rocprofv3 crash the tool's?  two host threads dispatching, no libsicp; (b) what co-resides
# with a resident accumulate kernel: product build (2 x 79 KB LDS per CU), the LDS-lean build, one workgroup per CU;
# (c) the stream test again; (d) kernel trace + timeline of the timed region (open stream)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for mode in 1 2 3; do
  rm -rf /tmp/tt_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt_prof -- ./build_dbg/two_thread_dispatch 6 $mode > $O/two_thread_mode$mode.log 2>&1
  echo "two_thread_dispatch mode $mode under rocprofv3: exit $? | $(grep -h '^ok' $O/two_thread_mode$mode.log)"
done
./build_dbg/two_thread_dispatch 6 3; echo "two_thread_dispatch mode 3 plain: exit $?"
timeout 600 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "fused_labels" > $O/gpu_tests_3.txt 2>&1; tail -3 $O/gpu_tests_3.txt
export SICP_ACC_INNER_REPEAT=30
python3 tools/corun_probe.py 64 16 24 > $O/corun_product.json 2> $O/corun_product.err; cat $O/corun_product.json
SICP_LIB=build_dbg/libsicp_v1.so python3 tools/corun_probe.py 64 16 24 > $O/corun_v1.json 2> $O/corun_v1.err; cat $O/corun_v1.json
SICP_ACC_GRID=256 python3 tools/corun_probe.py 64 16 24 > $O/corun_grid256.json 2> $O/corun_grid256.err; cat $O/corun_grid256.json
unset SICP_ACC_INNER_REPEAT
rm -rf /tmp/tl_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl_prof -- python3 bench.py --timed-only --steps 3 --warmup 1 > $O/bench_timed_only_under_rocprof.json 2> $O/bench_timed_only_under_rocprof.err
echo "timed-only under rocprofv3: exit $?"
f=$(find /tmp/tl_prof -name '*kernel_trace.csv' | head -1); s=$(find /tmp/tl_prof -name '*kernel_stats.csv' | head -1)
[ -n "$s" ] && cp "$s" $O/kernel_stats_timed_region_stream.csv
[ -n "$f" ] && python3 tools/trace_timeline.py "$f" 10 600 > $O/timeline_timed_region_stream.txt && tail -45 $O/timeline_timed_region_stream.txt
