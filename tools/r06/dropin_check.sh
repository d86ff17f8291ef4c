#!/bin/bash
# round 6: handle pool + device-resident SemanticPointCloud -- whole GPU suite, then the drop-in timing again
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_dropin; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
timeout 1500 python3 tools/r06/dropin_timing.py $O/dropin_after.json 13 > $O/dropin_after.log 2>&1; tail -3 $O/dropin_after.log
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06_dropin/dropin_after.json'))
for c in d['calls']:
    print(c['sequence'], 'total', c['us_per_pair'])
    for k,v in c['us_per_call_median'].items(): print('   %-55s %10.1f'%(k,v))
for e in d['end_to_end']: print(e['program'][:80], round(e['wall_s'],3), 's', round(e['pairs_per_s'],2), 'pairs/s')
PY
