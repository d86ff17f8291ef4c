#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 1000 python3 tools/pmc_acc_paths.py $O/pmc_acc_paths_32.json 32 > $O/pmc_acc_paths_32.log 2>&1; tail -30 $O/pmc_acc_paths_32.log
