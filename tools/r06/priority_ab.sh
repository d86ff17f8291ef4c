#!/bin/bash
# round 6: the side stream (searches, weights, features) at the lowest / highest queue priority against the default, timed region
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_prio; mkdir -p $O
for rep in 1 2; do
  for v in default low high; do
    if [ $v = default ]; then unset SICP_SIDE_PRIORITY; else export SICP_SIDE_PRIORITY=$v; fi
    timeout 600 python3 bench.py --no-cpu-baseline --timed-only --steps 10 --warmup 3 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python3 -c "
import json; d=json.load(open('$O/bench_${v}_$rep.json')); print('$v $rep', round(d['value']/1e9,4), 'G corr/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done
