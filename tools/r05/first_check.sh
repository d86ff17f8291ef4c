#!/bin/bash
# round 5, first GPU call: smoke, the whole GPU suite (with the reference's own drivers run on the engine), the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_first; mkdir -p $O
bash tools/probe_ref_deps.sh $O/probe_ref_deps_gpubox.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
