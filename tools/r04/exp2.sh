#!/bin/bash
# round 4, GPU experiment 2: the timed region through the open stream (ticks of 4 / 8), the new GPU tests, and the full
# default bench under rocprofv3 with python's faulthandler on (round 3: a segmentation fault on about every other run)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_validation.py -m gpu -x -q > $O/gpu_tests_2.txt 2>&1; tail -4 $O/gpu_tests_2.txt
for tick in 4 8; do
  timeout 600 python3 bench.py --timed-only --steps 4 --warmup 1 --tick $tick > $O/bench_stream_tick$tick.json 2> $O/bench_stream_tick$tick.err
  echo "stream tick $tick: $(python3 -c "import json; d=json.loads([l for l in open('$O/bench_stream_tick$tick.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['lockstep']['busy_fraction'])" 2>&1)"
done
timeout 600 python3 bench.py --timed-only --steps 4 --warmup 1 --closed-batch > $O/bench_closed.json 2> $O/bench_closed.err
echo "closed: $(python3 -c "import json; d=json.loads([l for l in open('$O/bench_closed.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['lockstep']['busy_fraction'])" 2>&1)"
timeout 900 python3 bench.py --no-cpu-baseline --steps 3 > $O/bench_full_a.json 2> $O/bench_full_a.err; echo "full bench exit $?"; tail -c 1500 $O/bench_full_a.json | head -c 600; echo
for attempt in 1 2 3; do
  rm -rf /tmp/pb_prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_prof -- python3 -X faulthandler bench.py --no-cpu-baseline --steps 3 > $O/bench_under_rocprof_$attempt.json 2> $O/bench_under_rocprof_$attempt.err
  echo "bench under rocprofv3 attempt $attempt: exit $?; trace: $(find /tmp/pb_prof -name '*kernel_trace.csv' | head -1)"
  grep -n "Fatal\|Segmentation\|File \"\|Thread 0x\|Current thread" $O/bench_under_rocprof_$attempt.err | head -30
done
