#!/bin/bash
# round 6: dynamic chunk runs (product) against static ranges (-DSICP_ACC_STATIC_RANGES) -- GPU tests, then the WHOLE bench both ways
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_dyn2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt
for rep in 1 2; do
for v in product static; do
  if [ $v = product ]; then unset SICP_LIB; else export SICP_LIB=$GRAFT_REPO_ROOT/semantic-icp_amd/variants/libsicp_$v.so; fi
  timeout 900 python3 bench.py --no-cpu-baseline --no-dropin --steps 8 --warmup 2 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  python3 - <<PY
import json
d=json.load(open('$O/bench_${v}_$rep.json'))
print('$v $rep', round(d['value']/1e9,4), round(d['ms_per_step'],2), 'alone', round(d['ms_per_align_alone'],3), 'acc us', round(d['roofline']['avg_launch_us'],1), '|', ' '.join(str(round(w.get('ms_per_step', w.get('pairs_per_s_end_to_end', 0)),2)) for w in d['other_workloads'][1:]))
PY
done
done
