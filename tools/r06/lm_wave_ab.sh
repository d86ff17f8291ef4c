#!/bin/bash
# round 6: the LM step by a whole wave (six sqrt(diag / radius) and the two sincos of a Plus in different lanes) against the step
# of one lane (prev = the commit before): parity of everything first, then ms per align() of one pair alone, interleaved on one
# box, then cycles per phase (-DSICP_SOLO_TIMING), then the 256-pair step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_lmwave; mkdir -p $O
V=$GRAFT_REPO_ROOT/semantic-icp_amd/variants
unset SICP_LIB
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
  for v in product prev; do
    if [ $v = product ]; then unset SICP_LIB; else export SICP_LIB=$V/libsicp_$v.so; fi
    echo -n "$v $rep: "; timeout 300 python3 tools/one_pair_latency.py 2>&1 | tail -1
  done
done | tee $O/one_pair_ab.txt
export SICP_LIB=$V/libsicp_timing.so
SICP_DEBUG=1 timeout 300 python3 tools/one_pair_latency.py 2>&1 | grep "solo timing" | tail -3 | tee $O/phase_cycles.txt
for rep in 1 2; do
  for v in product prev; do
    if [ $v = product ]; then unset SICP_LIB; else export SICP_LIB=$V/libsicp_$v.so; fi
    timeout 600 python3 bench.py --no-cpu-baseline --no-dropin --timed-only --steps 10 --warmup 3 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python3 -c "
import json; d=json.loads([l for l in open('$O/bench_${v}_$rep.json') if l.startswith('{')][-1]); print('$v $rep', round(d['value']/1e9,4), 'G corr/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done | tee $O/step_ab.txt
