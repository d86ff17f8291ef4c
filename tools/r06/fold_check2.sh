#!/bin/bash
# round 6: the LM step inside the accumulate launch on the OTHER workloads (small batches, bigger clouds): whole bench both ways
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_fold2; mkdir -p $O
for v in fold kernel; do
  if [ $v = fold ]; then export SICP_LM_STEP_IN_LAUNCH=1 SICP_LIB=$GRAFT_REPO_ROOT/gpurun_out/libsicp_fold.so; else unset SICP_LM_STEP_IN_LAUNCH SICP_LIB; fi
  timeout 900 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_$v.json 2> $O/bench_$v.err
  python3 - <<PY
import json
d=json.load(open('$O/bench_$v.json'))
print('$v', round(d['value']/1e9,4), round(d['ms_per_step'],2), 'alone', round(d['ms_per_align_alone'],3))
for w in d['other_workloads']:
    print('   ', w['workload'][:100], '|', {k: (round(v,4) if isinstance(v,float) else v) for k,v in w.items() if k in ('value','ms_per_step','ms_per_align','ms_per_pair','pairs_per_s','end_to_end_pairs_per_s','resident_pairs_per_s')})
PY
done
