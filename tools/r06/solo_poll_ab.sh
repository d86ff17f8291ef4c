#!/bin/bash
# round 6: the persistent solve with its polling loads in flight one behind the other (product: depth 4) against one load at a
# time (-DSICP_SOLO_POLL_DEPTH=1: rounds 3-5) and depth 8; the finite test of the 28 sums on 28 lanes is in all of them.
# Parity of the persistent path first, then ms per align() of one pair alone, interleaved on one box, then cycles per phase.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_solopoll; mkdir -p $O
V=$GRAFT_REPO_ROOT/semantic-icp_amd/variants
unset SICP_LIB
timeout 900 python -m pytest tests/test_gpu_validation.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
  for v in product d1 d8; do
    if [ $v = product ]; then unset SICP_LIB; else export SICP_LIB=$V/libsicp_$v.so; fi
    echo -n "$v $rep: "; timeout 300 python3 tools/one_pair_latency.py 2>&1 | tail -1
  done
done | tee $O/one_pair_ab.txt
for v in timing timing_d1; do
  export SICP_LIB=$V/libsicp_$v.so
  echo "== $v"; SICP_DEBUG=1 timeout 300 python3 tools/one_pair_latency.py 2>&1 | grep "solo timing" | tail -3
done | tee $O/phase_cycles.txt
