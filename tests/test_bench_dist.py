"""CPU test of the N > 1 path of bench.py: two gloo ranks, pairs sharded (one per rank), no
data-path collective; rank 0 reports the aggregate over the max-over-ranks time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_world_size_2_gloo_dry_run():
    one = run([sys.executable, "bench.py", "--dry-run", "--steps", "2", "--warmup", "1", "--points", "20000"])
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--dry-run", "--points", "20000"])
    for out, n in ((one, 1), (two, 2)):
        assert out["n_gpus"] == n and out["steps"] == 2 and out["warmup"] == 1
        assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["data"] == "dry-run"
        assert out["metric"] == "correspondences/sec" and out["unit"] == "correspondences/s"
        assert out["vs_baseline"] is None and "workload" in out["config"]
    # dry-run step of rank r sleeps 10*(1+r) ms and reports 3 outer iterations of 20000*4 slots for each of
    # its 256 pairs (the default): the aggregate counts both ranks' correspondences over the slower rank's time
    corr_per_rank = 2 * 3 * 20000 * 4 * 256
    assert abs(one["value"] * one["ms_per_step"] * 2e-3 - corr_per_rank) < 1e-6 * corr_per_rank
    assert abs(two["value"] * two["ms_per_step"] * 2e-3 - 2 * corr_per_rank) < 1e-6 * corr_per_rank
    assert two["ms_per_step"] > 1.5 * one["ms_per_step"]  # max over ranks, rank 1 is slower
    # every rank's own figures: the straggler (rank 1 sleeps twice as long) is visible in the line
    assert [r["rank"] for r in two["per_rank"]] == [0, 1] and len(one["per_rank"]) == 1
    assert two["per_rank"][1]["elapsed_ms"] > 1.5 * two["per_rank"][0]["elapsed_ms"]
    assert two["rank_time_spread_max_over_min"] > 1.5 and one["rank_time_spread_max_over_min"] == 1.0
    for r in two["per_rank"]:
        assert abs(r["value"] * r["elapsed_ms"] * 1e-3 - corr_per_rank) < 1e-6 * corr_per_rank


def test_gpu_numa_lookup_is_harmless_without_a_gpu():
    """bench.py binds a rank to the CPUs next to its GPU from sysfs alone; where the box says nothing (this container:
    no kfd) it changes nothing."""
    sys.path.insert(0, ROOT)
    import importlib

    bench = importlib.import_module("bench")
    before = os.sched_getaffinity(0)
    got = bench.bind_to_gpu_numa_node(0)
    assert got is None or set(os.sched_getaffinity(0)) <= set(before)
    if not os.path.exists("/sys/class/kfd/kfd/topology/nodes"):
        assert got is None and os.sched_getaffinity(0) == before


def test_bare_gpus_flag_fans_out_by_itself():
    """`python bench.py --gpus 2` with no launcher starts its own two ranks (child processes) and
    reports what they did; a rank count that contradicts --gpus is refused."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--points", "20000"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "pairs-sharded x2"
    assert abs(out["value"] * out["ms_per_step"] * 2e-3 - 2 * 2 * 3 * 20000 * 4 * 256) < 1.0
    # one rank launched for --gpus 2: mismatch, loud failure
    bad = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run"], cwd=ROOT, env=dict(env, WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
