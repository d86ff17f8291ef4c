#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_tree2; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout 300 python3 tools/r06/semantic_upload_probe.py 2>&1 | tail -2 | tee $O/semantic_upload_probe.txt
rm -rf /tmp/up_prof
SICP_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/up_prof -- python3 tools/stream_probe.py 256 > $O/stream_probe_under_rocprof.txt 2>&1
s=$(find /tmp/up_prof -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && grep -E "upper_levels|level_box|leaf_box|codes_kernel|gather_kernel|rocprim" "$s" | cut -c1-60,150-400 | head -12
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_tree2/bench.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_align_alone")}, d['verification']['ok'])
for w in d["other_workloads"]:
    if 'open stream' in w['workload']: print({k: w.get(k) for k in ('value','pairs_per_s_end_to_end','pairs_per_s_align_only')})
PY
