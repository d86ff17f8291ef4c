#!/bin/bash
# round 6: the LM step inside the accumulate launch -- GPU parity tests, then an interleaved A/B of the timed region
# (SICP_LM_STEP_KERNEL=1 = lm_step_batch_kernel after every accumulate launch, the tick of rounds 1-5; same library)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_fold; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
python3 tools/bench_acc_batch.py 256 | tee $O/acc256.txt
for rep in 1 2 3; do
  for v in fold kernel; do
    if [ $v = fold ]; then export SICP_LM_STEP_IN_LAUNCH=1 SICP_LIB=$GRAFT_REPO_ROOT/gpurun_out/libsicp_fold.so; else unset SICP_LM_STEP_IN_LAUNCH SICP_LIB; fi
    timeout 600 python3 bench.py --no-cpu-baseline --timed-only --steps 10 --warmup 3 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python3 -c "
import json; d=json.load(open('$O/bench_${v}_$rep.json')); print('$v $rep', round(d['value']/1e9,4), 'G corr/s', round(d['ms_per_step'],2), 'ms/step busy', round(d['lockstep']['busy_fraction'],3))"
  done
done
unset SICP_LM_STEP_KERNEL
