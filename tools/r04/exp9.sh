#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python3 tools/bench_weights.py; SICP_WEIGHTS_FROM_PROJ=1 python3 tools/bench_weights.py
timeout 600 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "fused_labels" 2>&1 | tail -3
