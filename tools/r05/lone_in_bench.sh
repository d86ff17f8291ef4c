#!/bin/bash
# why is "one pair alone" slower inside bench.py than in tools/one_pair_latency.py on the same box?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_lone2; mkdir -p $O
run() { tag=$1; shift; timeout 900 env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --full-size-pairs 0 --million-pairs 0 --sequence-pairs 0 > $O/$tag.json 2> $O/$tag.err; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag: ms_per_align_alone', d.get('ms_per_align_alone'), 'ms_per_step', d['ms_per_step'])"; }
timeout 300 python tools/one_pair_latency.py 2>&1 | tail -1
run sets3 X=1 
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --full-size-pairs 0 --million-pairs 0 --sequence-pairs 0 --cloud-sets 1 > $O/sets1.json 2> $O/sets1.err; python3 -c "import json; d=json.load(open('$O/sets1.json')); print('sets1: ms_per_align_alone', d.get('ms_per_align_alone'))"
run sets3_nopoll SICP_SOLO_NO_HOST_POLL=1
run sets3_nofold SICP_NO_WEIGHT_FOLD=1
timeout 300 python tools/one_pair_latency.py 2>&1 | tail -1
