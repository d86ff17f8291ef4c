#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python3 tools/soak_stream.py 90 3 8000 2>&1 | tail -4; echo "soak_stream exit $?"
