#!/bin/bash
# usage (GPU box): tools/collect_profiles.sh <prefix>   -> gpurun_out/<prefix>_*  (copy into profiles/)
# rocprofv3 kernel statistics of the default bench (lock-step batch) and of one pair alone, and the
# HBM traffic counters of the batched accumulate launch (FETCH_SIZE / WRITE_SIZE in separate passes).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
pre=$1
stats() {  # tag, bench args...
  local tag=$1; shift
  rm -rf gpurun_out/cp_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cp_$tag -- python3 bench.py --no-cpu-baseline "$@" \
    > gpurun_out/${pre}_bench_${tag}_under_rocprof.json 2> gpurun_out/cp_$tag.err
  local f=$(find gpurun_out/cp_$tag -name '*kernel_stats.csv' 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${pre}_kernel_stats_${tag}.csv
  rm -rf gpurun_out/cp_$tag gpurun_out/cp_$tag.err
}
stats lockstep32
stats single_pair --pairs-in-flight 1
pmc() {  # counter
  local c=$1
  rm -rf gpurun_out/cp_pmc
  timeout 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/cp_pmc -- python3 tools/bench_acc_batch.py 32 > gpurun_out/cp_pmc.log 2>&1
  local f=$(find gpurun_out/cp_pmc -name '*counter_collection.csv' 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" $c <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "accumulate" in k or "finalize" in k:
        print(f"{sys.argv[2]},{k.split('(')[0]},{len(v)},{sum(v)/len(v):.1f}")
PY
  rm -rf gpurun_out/cp_pmc gpurun_out/cp_pmc.log
}
{
  echo "# rocprofv3 --pmc <counter> -- python3 tools/bench_acc_batch.py 32   (one counter per pass; values in KB per dispatch)"
  echo "counter,kernel,dispatches,mean_per_dispatch"
  pmc FETCH_SIZE
  pmc WRITE_SIZE
} > gpurun_out/${pre}_pmc_hbm_traffic.csv
python3 bench.py > gpurun_out/${pre}_bench.json 2> gpurun_out/${pre}_bench.err
tail -c 600 gpurun_out/${pre}_bench.json
cat gpurun_out/${pre}_pmc_hbm_traffic.csv
