#!/bin/bash
# round 6: handle pool, feature-mark restore on failed admissions, fallback copy area, fused tree levels under stress
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python3 tools/soak_stream.py 90 3 8000 2>&1 | tail -3; echo "soak_stream exit $?"
timeout 600 python3 tools/stress_batch.py 2>&1 | tail -3; echo "stress_batch exit $?"
timeout 600 python3 tools/soak_persistent.py 2>&1 | tail -2; echo "soak_persistent exit $?"
timeout 600 python3 tools/soak_teardown.py 80 2>&1 | tail -3; echo "soak_teardown exit $?"
