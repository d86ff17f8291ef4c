#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [extra hipcc flags...]  -- registers / LDS / scratch / occupancy of every kernel in the file
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -I "$(dirname "$0")/../include" -I "$(dirname "$0")/../semantic-icp_amd/csrc" \
  -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur = t.split(":", 1)[1].strip()[:90].ljust(92)
    elif cur and re.match(r"(VGPRs|AGPRs|ScratchSize|Occupancy|LDS Size|TotalSGPRs)", t):
        cur += " " + t.replace(" [bytes/lane]", "").replace(" [waves/SIMD]", "").replace(" [bytes/block]", "")
if cur: print(cur)
'
