#!/bin/bash
# round 6: the persistent solve's columns as tagged granules (product) against plain columns behind one flag per worker
# (-DSICP_SOLO_FLAG_COLUMNS: rounds 3-5); parity of the persistent path first, then ms per align() of one pair alone, interleaved
# on one box, then cycles per phase (-DSICP_SOLO_TIMING builds)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_tagcols; mkdir -p $O
V=$GRAFT_REPO_ROOT/semantic-icp_amd/variants
unset SICP_LIB
timeout 900 python -m pytest tests/test_gpu_validation.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
  for v in product flagcols; do
    if [ $v = product ]; then unset SICP_LIB; else export SICP_LIB=$V/libsicp_$v.so; fi
    echo -n "$v $rep: "; timeout 300 python3 tools/one_pair_latency.py 2>&1 | tail -1
  done
done | tee $O/one_pair_ab.txt
for v in timing timing_flag; do
  export SICP_LIB=$V/libsicp_$v.so
  echo "== $v"; SICP_DEBUG=1 timeout 300 python3 tools/one_pair_latency.py 2>&1 | grep "solo timing" | tail -4
done | tee $O/phase_cycles.txt
