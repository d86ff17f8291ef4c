#!/bin/bash
# EM weights from the label histograms (SICP_WEIGHTS_FROM_HIST) against the default kernel over the projection arrays:
# the kernel alone, and the timed region of the bench, alternating on one box -> profiles/r04/weights_from_histograms.json
cd "$GRAFT_REPO_ROOT" || exit 1
python3 tools/bench_weights.py; SICP_WEIGHTS_FROM_HIST=1 python3 tools/bench_weights.py
for v in proj hist proj hist; do
  if [ $v = hist ]; then export SICP_WEIGHTS_FROM_HIST=1; else unset SICP_WEIGHTS_FROM_HIST; fi
  timeout 600 python3 bench.py --timed-only --steps 5 --warmup 1 > gpurun_out/bench_w_$v.json 2> /dev/null
  echo "weights from $v: $(python3 -c "import json; d=json.loads([l for l in open('gpurun_out/bench_w_$v.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" 2>&1)"
done
