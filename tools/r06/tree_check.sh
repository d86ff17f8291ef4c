#!/bin/bash
# round 6: fused upper tree levels + O(n) label segmentation -- whole GPU suite, then the upload probes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_tree; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
timeout 300 python3 tools/r06/semantic_upload_probe.py 2>&1 | tail -3 | tee $O/semantic_upload_probe.txt
timeout 900 python3 tools/r06/dropin_timing.py $O/dropin.json 13 > $O/dropin.log 2>&1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06_tree/dropin.json'))
for c in d['calls']:
    print(c['sequence'], 'total', c['us_per_pair'])
    for k,v in c['us_per_call_median'].items(): print('   %-55s %10.1f'%(k,v))
for e in d['end_to_end']: print(e['program'][:80], round(e['wall_s'],3), 's', round(e['pairs_per_s'],2), 'pairs/s')
PY
timeout 600 python3 tools/stream_probe.py 2>&1 | tail -4 | tee $O/stream_probe.txt
