#!/bin/bash
# usage (GPU box): tools/accumulate_anatomy.sh <prefix>
# What the accumulate launch (32 pairs x 100K x 100K, K = 4: tools/bench_acc_batch.py 32) is made of:
#  1. HIP-event time of the product build and of the developer builds that remove one ingredient each
#     (build_dbg/libsicp_<name>.so, made on the build host by tools/build_anatomy_libs.py:
#     -DSICP_DEBUG_NOCOMPUTE / NOGATHER / NOSTREAM / NOREDUCE; their sums are wrong by construction)
#  2. rocprofv3 --pmc passes over the product build (counters only): instruction counts and the split of
#     the waves' time into issuing / waiting to issue / parked on s_waitcnt
# -> gpurun_out/<prefix>_accumulate_anatomy.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/$1_accumulate_anatomy.txt
{
  echo "# HIP-event time per launch, 32 pairs x 100000 x 100000, K = 4"
  for n in product nocompute nogather nostream nostream_nocompute noreduce; do
    if [ $n = product ]; then unset SICP_LIB; else export SICP_LIB=$PWD/build_dbg/libsicp_$n.so; fi
    [ $n = product ] || [ -f "$SICP_LIB" ] || { echo "$n: library missing"; continue; }
    printf "%-20s " $n; python3 tools/bench_acc_batch.py 32 | sed 's/^pairs 32 points 100000: //' | cut -c1-70
  done
  unset SICP_LIB
  echo "# rocprofv3 --pmc (product build; mean per launch over the 32-pair launches)"
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD"; do
    bash tools/pmc_pass.sh anat "$set" tools/bench_acc_batch.py 32 | grep accumulate | sed 's/^void sicp:://'
  done
} > $out 2>&1
cat $out
