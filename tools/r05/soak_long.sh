#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python3 tools/soak_stream.py 420 4 8000 2>&1 | tail -2; echo "soak_stream exit $?"
timeout 900 python3 tools/soak_persistent.py 2>&1 | tail -1
