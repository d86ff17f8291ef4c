#!/bin/bash
# round 5: does a repeated pair's wait for its own previous registration (the stream's rewrite rule) cost the bench? cloud sets 1 / 3 / 5
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_sets; mkdir -p $O
for sets in 3 1 5 3; do
timeout 600 python bench.py --timed-only --cloud-sets $sets > $O/bench_sets$sets.json 2> $O/bench_sets$sets.err; python3 -c "
import json; d=json.load(open('$O/bench_sets$sets.json')); print('sets $sets fold: value', d['value'], 'ms_per_step', d['ms_per_step'])"
done
SICP_NO_WEIGHT_FOLD=1 timeout 600 python bench.py --timed-only --cloud-sets 3 > $O/bench_sets3_nofold.json 2> $O/bench_sets3_nofold.err; python3 -c "
import json; d=json.load(open('$O/bench_sets3_nofold.json')); print('sets 3 no fold: value', d['value'], 'ms_per_step', d['ms_per_step'])"
SICP_LIB=build_dbg/libsicp_knn_r04.so timeout 600 python bench.py --timed-only --cloud-sets 1 > $O/bench_r04lib.json 2> $O/bench_r04lib.err; python3 -c "
import json; d=json.load(open('$O/bench_r04lib.json')); print('r04 knn lib (older streams.cpp too), sets 1: value', d['value'], 'ms_per_step', d['ms_per_step'])"
