#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_final; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.txt 2>&1; tail -4 $O/gpu_tests_final.txt
