#!/bin/bash
# round 5: the search-kernel diet against round 4's kernel (build_dbg/libsicp_knn_r04.so = HEAD~'s knn_kernels.hip):
# bit-exactness first, then per-search times in a 16-job launch, then the counters of the new kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_knn; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_validation.py -m gpu -x -q -k "knn or nn or neighbour or cov or ties or correspond" > $O/knn_tests.txt 2>&1; tail -3 $O/knn_tests.txt
for rep in 1 2; do
  echo "new:"; timeout 600 python tools/bench_knn_jobs.py all 16 100000 20 | tail -1 | tee -a $O/knn_new.jsonl
  echo "r04:"; SICP_LIB=build_dbg/libsicp_knn_r04.so timeout 600 python tools/bench_knn_jobs.py all 16 100000 20 | tail -1 | tee -a $O/knn_r04.jsonl
done
echo "gicp K=1 new:"; KNN_MODE=gicp timeout 600 python tools/bench_knn_jobs.py k4_hinted 16 100000 20 | tail -1 | tee -a $O/knn_new.jsonl
echo "gicp K=1 r04:"; KNN_MODE=gicp SICP_LIB=build_dbg/libsicp_knn_r04.so timeout 600 python tools/bench_knn_jobs.py k4_hinted 16 100000 20 | tail -1 | tee -a $O/knn_r04.jsonl
timeout 1500 python tools/pmc_knn.py r05 > $O/pmc_knn.log 2>&1; tail -5 $O/pmc_knn.log | cut -c1-300
timeout 600 python bench.py --timed-only > $O/bench_timed_only.json 2> $O/bench_timed_only.err; tail -c 600 $O/bench_timed_only.json
timeout 2400 python -m pytest tests/test_reference_drivers.py tests/test_gpu_stream.py tests/test_host_shims.py -m gpu -q > $O/new_tests.txt 2>&1; tail -15 $O/new_tests.txt
