#!/bin/bash
# round 4, GPU experiment 1: launch-shape sweep of the accumulate kernel, LDS-lean variants of it with / without the tick
# queued ahead of the searches, rocprofv3 on the open stream alone, the GPU test suite.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
python3 tools/bench_acc_shapes.py 100000 50 > $O/acc_shapes_product.json 2> $O/acc_shapes_product.log
SICP_LIB=build_dbg/libsicp_v1.so python3 tools/bench_acc_shapes.py 100000 50 > $O/acc_shapes_v1.json 2> $O/acc_shapes_v1.log
for v in product v1 v2 v3; do
  for tf in 0 1; do
    lib=semantic-icp_amd/libsicp.so; [ $v != product ] && lib=build_dbg/libsicp_$v.so
    if [ $tf = 1 ]; then export SICP_TICK_FIRST=1; else unset SICP_TICK_FIRST; fi
    SICP_LIB=$lib timeout 600 python3 bench.py --timed-only --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${v}_tf${tf}.json 2> $O/bench_${v}_tf${tf}.err
    echo "bench $v tick_first=$tf: $(python3 -c "import json,sys; d=json.loads([l for l in open('$O/bench_${v}_tf${tf}.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" 2>&1)"
  done
done
unset SICP_TICK_FIRST
# the stream leg alone under rocprofv3 (round 3: a segmentation fault with the worker thread alive)
for attempt in 1 2; do
  rm -rf /tmp/sp_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_prof -- python3 tools/stream_probe.py 256 > $O/stream_probe_rocprof_$attempt.log 2>&1
  echo "stream_probe under rocprofv3, attempt $attempt: exit $?; trace: $(find /tmp/sp_prof -name '*kernel_trace.csv' | head -1)"
  f=$(find /tmp/sp_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/stream_probe_kernel_stats_$attempt.csv
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests_1.txt 2>&1
tail -5 $O/gpu_tests_1.txt
cat $O/acc_shapes_product.json | python3 -c "import json,sys; [print(r) for r in json.load(sys.stdin)['shapes']]"
cat $O/acc_shapes_v1.json | python3 -c "import json,sys; [print(r) for r in json.load(sys.stdin)['shapes']]"
