#!/bin/bash
# round 5: LM evaluations per tick of the timed region's stream (2 / 3 / 4 / 6), one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_tick; mkdir -p $O
for t in 4 3 2 6 4; do
timeout 600 python bench.py --timed-only --tick $t > $O/tick$t.json 2> $O/tick$t.err; python3 -c "
import json; d=json.load(open('$O/tick$t.json')); print('tick $t: value', d['value'], 'ms_per_step', d['ms_per_step'], 'busy', d['lockstep']['busy_fraction'])"
done
