#!/bin/bash
# round 6, first GPU call: baseline bench of HEAD on this box + the counter evidence VERDICT r05 asked for
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_ev1; mkdir -p $O
bash tools/probe_ref_deps.sh > $O/probe_ref_deps_gpubox.txt 2>&1
timeout 600 python3 bench.py > $O/bench_baseline.json 2> $O/bench_baseline.err
python3 -c "
import json; d=json.load(open('$O/bench_baseline.json')); print('baseline', d['value'], d['ms_per_step'], d.get('ms_per_align_alone'), d['roofline']['frac'])"
timeout 1200 python3 tools/pmc_acc_paths.py $O/pmc_accumulate_paths_256.json 256 SICP_NO_GRAPH=1 2>&1 | tail -30
timeout 2400 python3 tools/r06/pmc_features.py $O/pmc_features.json 2>&1 | tail -80
