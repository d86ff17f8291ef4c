#!/bin/bash
# round 5: one pair alone with the weights in the search epilogue + the polled state (fix: the turn collectors carry the fold policy)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_lone; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "epilogue or batch_equals or metric_size" > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
for i in 1 2; do
timeout 300 python tools/one_pair_latency.py 2>&1 | tail -1
SICP_NO_WEIGHT_FOLD=1 timeout 300 python tools/one_pair_latency.py 2>&1 | tail -1
SICP_NO_WEIGHT_FOLD=1 SICP_SOLO_NO_HOST_POLL=1 timeout 300 python tools/one_pair_latency.py 2>&1 | tail -1
done
