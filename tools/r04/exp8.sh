#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py tests/test_gpu_stream.py -m gpu -x -q > $O/gpu_tests_8.txt 2>&1; tail -5 $O/gpu_tests_8.txt
for v in hist proj hist proj; do
  if [ $v = proj ]; then export SICP_WEIGHTS_FROM_PROJ=1; else unset SICP_WEIGHTS_FROM_PROJ; fi
  timeout 600 python3 bench.py --timed-only --steps 5 --warmup 1 > $O/bench_w_$v.json 2> $O/bench_w_$v.err
  echo "weights from $v: $(python3 -c "import json; d=json.loads([l for l in open('$O/bench_w_$v.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" 2>&1)"
done
unset SICP_WEIGHTS_FROM_PROJ
