#!/bin/bash
# usage (GPU box): tools/trace_seq.sh <kernel substring> [bench args]: durations of matching launches in time order
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
pat=$1; shift
rm -rf gpurun_out/ts
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ts -- python3 bench.py --no-cpu-baseline --timed-only "$@" > /dev/null 2> gpurun_out/ts.err
f=$(find gpurun_out/ts -name '*kernel_trace.csv' 2>/dev/null | head -1)
[ -z "$f" ] && { echo "no trace"; exit 1; }
python3 - "$f" "$pat" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if sys.argv[2] in r["Kernel_Name"]:
        out.append(f"{(e - s) / 1e3:.0f}")
print(" ".join(out))
PY
rm -rf gpurun_out/ts gpurun_out/ts.err
