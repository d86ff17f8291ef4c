#!/bin/bash
# round 5: wait-then-order admission + uncontracted corr_eval: the tests that changed, fold vs no fold, one pair alone
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_fold2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -m gpu -q -x > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
for i in 1 2; do
timeout 600 python bench.py --timed-only > $O/bench_fold_$i.json 2> $O/bench_fold_$i.err; python3 -c "
import json; d=json.load(open('$O/bench_fold_$i.json')); print('fold: value', d['value'], 'ms_per_step', d['ms_per_step'])"
SICP_NO_WEIGHT_FOLD=1 timeout 600 python bench.py --timed-only > $O/bench_nofold_$i.json 2> $O/bench_nofold_$i.err; python3 -c "
import json; d=json.load(open('$O/bench_nofold_$i.json')); print('no fold: value', d['value'], 'ms_per_step', d['ms_per_step'])"
done
timeout 300 python tools/one_pair_latency.py 2>&1 | tail -1
SICP_KNN_WPB=2 SICP_KNN_WPB20=1 timeout 300 python tools/one_pair_latency.py 2>&1 | tail -1
SICP_NO_WEIGHT_FOLD=1 timeout 300 python tools/one_pair_latency.py 2>&1 | tail -1
timeout 900 python tools/trace_single.py r05 > $O/trace_single.log 2>&1; tail -60 $O/trace_single.log | head -80
