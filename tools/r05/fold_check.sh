#!/bin/bash
# round 5: EM weights in the search epilogue + ordered feature rewrites in the stream: whole GPU suite, search times, bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_fold; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -8 $O/gpu_tests.txt
timeout 600 python tools/bench_knn_jobs.py all 16 100000 20 | tail -1 | tee $O/knn_new.jsonl
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05_fold/bench.json'))
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'roofline', d['roofline']['frac'])
for w in d['other_workloads'][:3]: print({k:v for k,v in w.items() if not isinstance(v,(dict,list))})
print(d['step_roofline']['ms_per_step_by_unit_cost'])
PY
SICP_NO_WEIGHT_FOLD=1 timeout 600 python bench.py --timed-only > $O/bench_nofold.json 2> $O/bench_nofold.err; python3 -c "
import json; d=json.load(open('gpurun_out/r05_fold/bench_nofold.json')); print('no fold: value', d['value'], 'ms_per_step', d['ms_per_step'])"
timeout 600 python bench.py --timed-only > $O/bench_fold2.json 2> $O/bench_fold2.err; python3 -c "
import json; d=json.load(open('gpurun_out/r05_fold/bench_fold2.json')); print('fold again: value', d['value'], 'ms_per_step', d['ms_per_step'])"
