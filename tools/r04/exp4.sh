#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for mode in 4 8 2; do
  rm -rf /tmp/tt_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt_prof -- ./build_dbg/two_thread_dispatch 5 $mode > $O/two_thread_mode$mode.log 2>&1
  echo "two_thread_dispatch mode $mode under rocprofv3: exit $? | $(grep -h '^ok' $O/two_thread_mode$mode.log)"
done
rm -rf /tmp/tt_prof; rocprofv3 --kernel-trace --output-format csv -d /tmp/tt_prof -- ./build_dbg/two_thread_dispatch 5 2 > $O/two_thread_mode2_nostats.log 2>&1; echo "mode 2, --kernel-trace without --stats: exit $?"
python3 tools/pmc_acc_paths.py $O/pmc_acc_paths_32.json 32 > $O/pmc_acc_paths_32.log 2>&1; tail -80 $O/pmc_acc_paths_32.log
SICP_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_prof -- python3 bench.py --no-cpu-baseline --steps 3 > $O/bench_under_rocprof_nograph.json 2> $O/bench_under_rocprof_nograph.err
echo "full bench under rocprofv3 with SICP_NO_GRAPH=1: exit $?"
s=$(find /tmp/pb_prof -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp "$s" $O/kernel_stats_bench_nograph.csv && head -8 "$s" | cut -c1-150
