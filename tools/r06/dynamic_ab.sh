#!/bin/bash
# round 6: accumulate chunks handed out dynamically in runs (-DSICP_ACC_DYNAMIC=<chunks per run>) against equal static ranges, in the
# timed region (where the accumulate workgroups start staggered beside the search kernels); interleaved on one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_dyn; mkdir -p $O
unset SICP_LIB
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch" 2>&1 | tail -1
for v in dyn16; do SICP_LIB=$GRAFT_REPO_ROOT/semantic-icp_amd/variants/libsicp_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q -m gpu -k "batch or stream" 2>&1 | tail -1; done
for rep in 1 2; do
  for v in product dyn8 dyn16 dyn32; do
    if [ $v = product ]; then unset SICP_LIB; else export SICP_LIB=$GRAFT_REPO_ROOT/semantic-icp_amd/variants/libsicp_$v.so; fi
    timeout 600 python3 bench.py --no-cpu-baseline --timed-only --steps 10 --warmup 3 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python3 -c "
import json; d=json.load(open('$O/bench_${v}_$rep.json')); print('$v $rep', round(d['value']/1e9,4), 'G corr/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done
unset SICP_LIB
for v in product dyn16; do if [ $v = product ]; then unset SICP_LIB; else export SICP_LIB=$GRAFT_REPO_ROOT/semantic-icp_amd/variants/libsicp_$v.so; fi; echo -n "$v acc256: "; python3 tools/bench_acc_batch.py 256 | tail -1; done
