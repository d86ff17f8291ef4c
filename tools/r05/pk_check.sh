#!/bin/bash
# round 5: packed float32 distances in the leaf scans: bit-exactness + times against the previous build of this round
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_pk; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_validation.py tests/test_gpu_fullsize.py -m gpu -x -q -k "knn or nn or neighbour or cov or ties or correspond or fullsize or config" > $O/knn_tests.txt 2>&1; tail -3 $O/knn_tests.txt
for rep in 1 2; do
  timeout 600 python tools/bench_knn_jobs.py all 16 100000 20 | tail -1 | tee -a $O/knn_new.jsonl | cut -c1-600
done
KNN_MODE=gicp timeout 600 python tools/bench_knn_jobs.py k4_hinted 16 100000 20 | tail -1 | cut -c1-300
