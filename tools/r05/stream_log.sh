#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_tick; mkdir -p $O
SICP_DEBUG=1 SICP_STREAM_LOG=1 timeout 600 python bench.py --timed-only --steps 6 --warmup 1 > $O/log_run.json 2> $O/log_run.err; grep "^\[stream\]" $O/log_run.err | tail -25
