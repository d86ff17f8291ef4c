#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_final_check; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.txt 2>&1; tail -3 $O/gpu_tests_final.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d.get('ms_per_align_alone'), d['roofline']['frac'])"
