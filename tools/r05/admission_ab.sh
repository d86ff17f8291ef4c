#!/bin/bash
# round 5 A/B on ONE box: the stream's ordered admission (product) against round 4's unordered one (build_dbg/libsicp_racy_admission.so:
# today's kernels, round 4's streams.cpp -- measurement only) and against round 4's whole library
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_admission; mkdir -p $O
run() { tag=$1; shift; timeout 600 env "$@" python bench.py --timed-only --cloud-sets ${SETS:-1} > $O/$tag.json 2> $O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag: value', d['value'], 'ms_per_step', d['ms_per_step'])"; }
for i in 1 2; do
SETS=1 run new_sets1_$i X=1
SETS=1 run racy_sets1_$i SICP_LIB=build_dbg/libsicp_racy_admission.so
SETS=3 run new_sets3_$i X=1
SETS=3 run racy_sets3_$i SICP_LIB=build_dbg/libsicp_racy_admission.so
SETS=1 run r04all_sets1_$i SICP_LIB=build_dbg/libsicp_r04_all.so
done
