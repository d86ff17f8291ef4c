#!/bin/bash
# round 5: the persistent solve's state read by polling pinned memory instead of a read-back copy: tests + one pair alone, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_solo; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_validation.py tests/test_gpu_parity.py tests/test_gpu_stream.py -m gpu -q -x > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
for i in 1 2; do
timeout 300 python tools/one_pair_latency.py 2>&1 | tail -1
SICP_SOLO_NO_HOST_POLL=1 timeout 300 python tools/one_pair_latency.py 2>&1 | tail -1
done
timeout 600 python tools/soak_persistent.py 2>&1 | tail -3
