#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python3 tools/soak_teardown.py 80; echo "soak_teardown exit $?"
