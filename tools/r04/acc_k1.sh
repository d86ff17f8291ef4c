cd "$GRAFT_REPO_ROOT" || exit 1
ACC_MODE=gicp python3 tools/bench_acc_batch.py 256 | tail -1
ACC_MODE=gicp python3 tools/bench_acc_batch.py 32 | tail -1
