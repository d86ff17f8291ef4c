#!/bin/bash
# round 6: HEAD on a fresh box -- smoke, the whole GPU suite, the default bench line, one pair alone
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_final_check; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2700 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.txt 2>&1; tail -3 $O/gpu_tests_final.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python3 -c "
import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_align_alone'), d['roofline']['frac'], d.get('pose_delta_vs_cpu'))"
timeout 300 python3 tools/one_pair_latency.py > $O/one_pair_latency.txt 2>&1; tail -1 $O/one_pair_latency.txt
