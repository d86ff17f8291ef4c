#!/bin/bash
# usage (on the GPU box): tools/pmc_pass.sh <tag> "<COUNTER ...>" <python script> [args...]
# one rocprofv3 --pmc pass (counters only, no trace domains), mean per kernel name
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; ctrs=$2; shift 2
rm -rf gpurun_out/pmc_$tag
timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -- python3 "$@" > gpurun_out/pmc_${tag}.log 2>&1
f=$(find gpurun_out/pmc_$tag -name '*counter_collection.csv' 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no counter csv"; tail -5 gpurun_out/pmc_${tag}.log; exit 1; fi
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    n = max(len(v) for v in d.values())
    if not any(s in k for s in ("knn", "accumulate", "lm_step", "cov_kernel", "weight")):
        continue
    print(k, "dispatches", n, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
PY
rm -rf gpurun_out/pmc_$tag
