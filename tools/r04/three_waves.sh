#!/bin/bash
# the accumulate kernel at THREE waves per SIMD: 2-slot groups (SICP_SG4=2: <= 168 VGPRs, half the staging LDS), optionally a
# 7-row reduction tile, three workgroups per CU -- against the product (4-slot groups, two workgroups per CU)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
for v in product v4 v5 v6; do
  lib=semantic-icp_amd/libsicp.so; [ $v != product ] && lib=build_dbg/libsicp_$v.so
  echo "== $v: $(SICP_LIB=$lib python3 tools/bench_acc_batch.py 256 2>&1 | tail -1)"
  echo "== $v: $(SICP_LIB=$lib python3 tools/bench_acc_batch.py 32 2>&1 | tail -1)"
done
for v in product v4 v6 product; do
  lib=semantic-icp_amd/libsicp.so; [ $v != product ] && lib=build_dbg/libsicp_$v.so
  SICP_LIB=$lib timeout 600 python3 bench.py --timed-only --steps 4 --warmup 1 > $O/bench_3w_$v.json 2> /dev/null
  echo "bench $v: $(python3 -c "import json; d=json.loads([l for l in open('$O/bench_3w_$v.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" 2>&1)"
done
