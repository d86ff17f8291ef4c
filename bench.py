#!/usr/bin/env python3
"""bench.py -- headline benchmark of the semantic-ICP hot path on MI355X.

Metric (BASELINE.json): correspondences/sec (+ ms per outer ICP iteration) for EM-ICP
(K = 4 correspondences per source point, C = 11 classes) on synthetic KITTI-like scan pairs
subsampled to exactly 100 000 x 100 000 points -- the metric point of configs[1].

A "step" registers one batch of independent scan pairs on each GPU: `--pairs-in-flight` S pairs
(default 256 = ~30 GB of the 288 GB: the stragglers of a batch -- a pair that needs 500 evaluations
when the median needs 150 -- weigh less the larger it is: 1.39 G corr/s at 32 pairs, 1.54 at 64, 1.69 at
128, 1.84-1.87 at 256, 1.84 at 512), every one a DIFFERENT pair (its own street, ego-motion and noise: seeds
2 + S*rank + k) and a complete align() (covariances of both clouds + every outer ICP iteration:
transform -> kNN -> EM weights -> inner LM solve) on its own handle, with all clouds already
resident in HBM when the timed region starts.  Independent pairs are the reference's unit of work
(exec/kitti_eval.cc:124-249 loops over them) and the north star shards them across GPUs.  The S
pairs of a GPU go through one sicp_align_batch call: every launch of the solve evaluates the pairs
that are inside an inner solve (up to 256 per launch), the searches of the pairs between two solves
are queued on a second stream and those pairs rejoin at the next tick (continuous batching: no pair
waits for another pair's solve or outer loop); per pair the result
is bit-identical to a lone align().  `lockstep.busy_fraction` = the pairs' own LM evaluations / the
evaluation launches they took part in.

One correspondence = one (source, target) slot that went through kNN + weighting + accumulation
in one outer iteration.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher (WORLD_SIZE unset) the script starts `python -m torch.distributed.run
--nproc-per-node N` on itself as a child process before anything touches the GPU; under a launcher
it is one rank of N (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  One process per GPU,
independent scan pairs per rank, no collective on the solve path: the process group is gloo and
only carries the barrier and the max-over-ranks of the timings.

At N = 1 the JSON line also carries, outside the timed region: `other_workloads` (SE3-GICP K = 1
on the same pairs; one pair alone = the reference's own call pattern; a stride-1 scan sequence
registered end to end INCLUDING cloud upload and search-tree build), the `roofline` of the
dominant kernel from HIP events, and `cpu_baseline` (the CPU restatement of the reference's
PCL-KdTree + Ceres path on the rank-0 pair).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
N_POINTS = 100_000
K_CORR = 4
N_CLASSES = 11
PROFILE_DIR = os.path.join(ROOT, "profiles", "r03")   # committed rocprofv3 summaries of this round's build


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--mode", choices=["em", "gicp"], default="em",
                    help="workload of the timed region: EM-ICP K=4 C=11 (the metric) or SE3-GICP K=1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nn-method", type=int, default=None, help="0 brute force, 1 box tree (default: library default)")
    ap.add_argument("--lm-on-device", type=int, default=None, help="0 host LM loop, 1 device-resident (default: library default)")
    ap.add_argument("--lm-batch", type=int, default=None)
    ap.add_argument("--pairs-in-flight", type=int, default=256,
                    help="independent, distinct scan pairs registered together on each GPU (one handle each; round 1 used 32)")
    ap.add_argument("--same-pair", action="store_true", help="r01 behaviour: every handle of a GPU gets the rank's first pair")
    ap.add_argument("--profile", type=int, default=0, help="SICP_PROFILE_* mask applied inside the timed region")
    ap.add_argument("--sequence-pairs", type=int, default=1024, help="registrations of the open-stream leg (0 = skip)")
    ap.add_argument("--stream-in-flight", type=int, default=256, help="registrations that share the GPU in the open-stream leg")
    ap.add_argument("--stream-lm-batch", type=int, default=4, help="LM evaluations per tick in the open-stream leg")
    ap.add_argument("--timed-only", action="store_true", help="skip the other workloads / roofline / CPU legs (for tracing the timed region)")
    ap.add_argument("--dry-run", action="store_true", help="exercise the multi-process plumbing without a GPU")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than visible devices (rank r uses device local_rank %% n_devices): the real N > 1 code path on a "
                         "one-GPU box; the ranks then share the GPU, so the value is no scaling number")
    ap.add_argument("--full-size-pairs", type=int, default=16, help="pairs of the full-size config-2 / config-3 batches (0 = skip)")
    return ap.parse_args()


# ---- fan-out --------------------------------------------------------------------------------------
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def fan_out_if_needed(args) -> None:
    """`python bench.py --gpus N` without a launcher: become the parent of N ranks.  Nothing in this
    process has touched the GPU (no torch, no libsicp), and the ranks are fresh child processes --
    never an exec of a process that initialised HIP."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.run(cmd, env=env).returncode)


class Dist:
    """torch.distributed (gloo) only when launched with WORLD_SIZE > 1."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.pg = None
        if self.world > 1:
            # gloo on CPU tensors only: torch never initialises HIP in this process (no .cuda(), no device query), so
            # libsicp is the only user of the GPU and it does not matter which of the two is loaded first
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.pg = dist

    def barrier(self):
        if self.pg:
            self.pg.barrier()

    def reduce(self, value: float, op: str) -> float:
        if not self.pg:
            return value
        import torch

        t = torch.tensor([value], dtype=torch.float64)
        self.pg.all_reduce(t, op={"max": self.pg.ReduceOp.MAX, "min": self.pg.ReduceOp.MIN, "sum": self.pg.ReduceOp.SUM}[op])
        return float(t.item())

    def close(self):
        if self.pg:
            self.pg.destroy_process_group()


# ---- synthetic data ----------------------------------------------------------------------------------
# ALL of it is generated at the top of main(): worker processes are forked while this process is still
# single-threaded -- before torch.distributed starts its gloo threads and before libsicp loads the HIP
# runtime (a fork of a process that holds a live, multi-threaded HIP runtime inherits its locks).
def pair_motion(seed: int):
    """Ego-motion of the pair with this seed: seed 2 is the r01 pair (1 m, 2 deg); the others vary in
    forward step and yaw, so the pairs of a batch need different numbers of iterations."""
    if seed == 2:
        return (1.0, 2.0)
    return (0.5 + 0.11 * (seed % 11), -2.6 + 0.65 * (seed % 9))


def gen_pair(job):
    import synth

    seed, n = job
    src, sl, tgt, tl, T_gt, cm = synth.lidar_pair(seed=seed, n_points=n, motion=pair_motion(seed))
    return src, sl, tgt, tl


def gen_scan(job):
    import synth

    seed, i, n = job
    p, l, pose = synth.lidar_sequence_scan(seed, i, n_points=n, period=128)   # 128 steps up the street, 128 back, ...
    return p, l


def gen_full_scan_pair(seed):
    """config 2: a full KITTI-like scan pair (~142K x 142K points, not subsampled)"""
    import synth

    src, sl, tgt, tl, T_gt, cm = synth.lidar_pair(seed=seed, n_points=None, motion=pair_motion(seed))
    return src, sl, tgt, tl


def gen_rgbd_pair(seed):
    """config 3: a 640x480 RGB-D frame pair (307 200 x 307 200 points, 13 classes)"""
    import synth

    src, sl, tgt, tl, T_gt = synth.rgbd_pair(seed=seed)[:5]
    return src, sl, tgt, tl


def pool_map(fn, jobs, world):
    import concurrent.futures as cf
    import multiprocessing as mp

    workers = max(1, min(len(jobs), (os.cpu_count() or 1) // max(1, world), 64))
    if workers == 1:
        return [fn(j) for j in jobs]
    with cf.ProcessPoolExecutor(workers, mp_context=mp.get_context("fork")) as ex:
        return list(ex.map(fn, jobs))


# ---- legs outside the timed region ------------------------------------------------------------------
def cpu_baseline(src, sl, tgt, tl, cm, repeats=3):
    """The CPU restatement of the reference PCL-KdTree + Ceres path (the oracle), timed on this
    host on the rank-0 pair: kNN + problem build on 1 thread (em_icp.hpp:57-156), residual /
    Jacobian evaluation on 8 threads (em_icp.hpp:166).  `repeats` runs, the median is reported."""
    import numpy as np

    import oracle_lib as O

    p = O.default_params(O.MODE_EM)
    p.num_classes = N_CLASSES
    p.use_kdtree = 1
    threads = min(8, os.cpu_count() or 1)
    p.num_threads = threads
    runs = []
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        qt, st = O.align(p, src, sl, tgt, tl, cm, np.array([0, 0, 0, 1, 0, 0, 0.0]))
        runs.append((time.perf_counter() - t0, st))
    runs.sort(key=lambda r: r[0])
    dt, st = runs[len(runs) // 2]
    return {
        "value": st["total_corr"] / dt,
        "unit": "correspondences/s",
        "cores": threads,
        "kind": "port",
        "sample": f"median of {len(runs)} full align() calls of the rank-0 pair of the batch ({len(src)}x{len(tgt)}): {st['outer_iters']} outer iterations, "
                  f"{st['total_evals']} residual sweeps, {dt:.1f} s each ({', '.join(f'{r[0]:.2f}' for r in runs)} s); kd-tree kNN + problem build on 1 thread, "
                  f"residual/Jacobian sweeps on {threads} threads (host has {os.cpu_count()} cores)",
        "runs_s": [r[0] for r in runs],
        "ms_per_icp_iter": 1e3 * (dt - st["t_cov_s"]) / max(1, st["outer_iters"]),
        "cov_ms": 1e3 * st["t_cov_s"],
        "pose": [float(v) for v in qt],
    }, qt


def run_batch_steps(sicp, engines, steps, warmup, sync):
    """`steps` sicp_align_batch calls over `engines`; returns wall time and the summed counters."""
    for _ in range(warmup):
        sicp.align_batch(engines)
    sync()
    keys = ("total_corr", "outer_iters", "total_evals", "total_lm_iters", "lockstep_slots", "graph_builds")
    agg = {k: 0 for k in keys}
    per_pair_outer, per_pair_evals = [], []
    t0 = time.perf_counter()
    qts = None
    for _ in range(steps):
        res = sicp.align_batch(engines)
        qts = [q for q, _ in res]
        for _, st in res:
            for k in keys:
                agg[k] += st[k]
        per_pair_outer = [st["outer_iters"] for _, st in res]
        per_pair_evals = [st["total_evals"] for _, st in res]
    sync()
    return time.perf_counter() - t0, agg, per_pair_outer, per_pair_evals, qts


def stream_leg(sicp, device, params, cm, scans, in_flight, lm_batch):
    """Config-5 stand-in on one GPU: an OPEN stream of len(scans) - 1 consecutive registrations (scan p+1 onto
    scan p, the loop of exec/kitti_eval.cc:124-249) through sicp_stream_*: every scan is uploaded once by
    this thread (sicp_stream_add_cloud: copy into pinned memory, H2D and search-tree build queued on the
    stream's own HIP stream), shared as the source of one registration and the target of the next
    (setSourceCloud(cloud, kdtree, covs), gicp.h:48-56), its normals / histograms are computed once, and
    up to `in_flight` registrations share the GPU while the library's worker thread runs the ticks.
    Timed twice: END TO END (uploads inside the timed region, clouds released as soon as both their
    registrations are submitted) and ALIGN ONLY (every cloud resident and indexed before the clock starts)."""
    import numpy as np

    n_pairs = len(scans) - 1
    p = sicp.SicpParams.from_buffer_copy(params)
    p.lm_batch = lm_batch
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])

    def run(resident):
        with sicp.Stream(device, p, max_in_flight=in_flight, confusion=cm) as S:
            res = []
            if resident:
                ids = [S.add_cloud(*sc) for sc in scans]
                # (a registration of the first and the last scan: when it is back every upload before it has landed)
                S.submit(ids[-1], ids[0], ident)
                S.drain()
                t0 = time.perf_counter()
                for k in range(n_pairs):
                    S.submit(ids[k + 1], ids[k], ident)
                    if k % 64 == 0:
                        res += S.poll(wait=0)
            else:
                t0 = time.perf_counter()
                ids = [S.add_cloud(*scans[0])]
                for k in range(n_pairs):
                    ids.append(S.add_cloud(*scans[k + 1]))
                    S.submit(ids[k + 1], ids[k], ident)
                    if k >= 1:
                        S.release_cloud(ids[k - 1])
                    if k % 64 == 0:
                        res += S.poll(wait=0)
            res += S.drain()
            dt = time.perf_counter() - t0
            c = S.counters()
        assert len(res) == n_pairs and all(st == 0 for _, st, _, _ in res)
        res.sort(key=lambda r: r[0])
        return dt, res, c

    run(False)                               # warm-up: pools, graphs, allocations
    t_e2e, res, c_e2e = run(False)
    t_align, res2, c_al = run(True)
    for (_, _, qa, _), (_, _, qb, _) in zip(res, res2):
        assert np.array_equal(qa, qb)       # the same poses whatever a pair shared the GPU with
    corr = sum(st["total_corr"] for _, _, _, st in res)
    return {
        "workload": f"open stream (sicp_stream_*): {n_pairs} consecutive registrations (scan p+1 onto scan p) of {len(scans[0][0])}-point scans, "
                    f"EM-ICP K={K_CORR} C={N_CLASSES}, {in_flight} in flight, ticks of {lm_batch} LM evaluations; every scan uploaded and indexed once "
                    "(inside the timed region for the end-to-end figure), shared between its two registrations, features computed once",
        "pairs": n_pairs,
        "value": corr / t_e2e, "unit": "correspondences/s",
        "pairs_per_s_end_to_end": n_pairs / t_e2e,
        "pairs_per_s_align_only": n_pairs / t_align,
        "end_to_end_over_align_only": t_e2e / t_align,
        "busy_fraction": c_e2e["busy_fraction"],
        "busy_fraction_align_only": c_al["busy_fraction"],
        "ms_per_pair_end_to_end": 1e3 * t_e2e / n_pairs,
        "ms_per_pair_align_only": 1e3 * t_align / n_pairs,
        "outer_iters_per_pair": sum(st["outer_iters"] for _, _, _, st in res) / n_pairs,
    }


def full_size_leg(sicp, device, mode, pairs, C, cm, epsilon, label):
    """A closed batch of full-size pairs (config 2 / config 3 of BASELINE.json) through one sicp_align_batch call."""
    p = sicp.default_params(mode)
    p.num_classes = C if mode == sicp.MODE_EM else 0
    if epsilon is not None:
        p.epsilon = epsilon
    E = []
    for src, sl, tgt, tl in pairs:
        e = sicp.Engine(device, p)
        if mode == sicp.MODE_EM:
            e.set_confusion(cm)
        lab = mode != sicp.MODE_GICP
        e.set_source(src, sl if lab else None)
        e.set_target(tgt, tl if lab else None)
        E.append(e)
    dt, a, o, ev, _ = run_batch_steps(sicp, E, 2, 1, lambda: [e.synchronize() for e in E])
    for e in E:
        e.close()
    npts = sum(len(pr[0]) for pr in pairs) / len(pairs)
    return {"workload": f"{label}: {len(pairs)} different pairs of ~{int(npts)} points through one sicp_align_batch call",
            "value": a["total_corr"] / dt, "unit": "correspondences/s", "ms_per_step": 1e3 * dt / 2, "pairs_per_s": 2 * len(pairs) / dt,
            "ms_per_pair": 1e3 * dt / (2 * len(pairs)), "outer_iters_per_align": a["outer_iters"] / (2 * len(pairs)),
            "busy_fraction": a["total_evals"] / max(1, a["lockstep_slots"])}


def load_profile_json(name):
    path = os.path.join(PROFILE_DIR, name)
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def main():
    args = parse_args()
    fan_out_if_needed(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N")
    import numpy as np

    import synth

    n = args.points
    S = max(1, args.pairs_in_flight)
    em = args.mode == "em"
    K = K_CORR if em else 1
    cm = synth.confusion_matrix(N_CLASSES)
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    # weak scaling: every rank registers its own S pairs; all pairs of the job are different
    seeds = [2 + S * rank + (0 if args.same_pair else k) for k in range(S)]
    extras = rank == 0 and world == 1 and not args.timed_only and not args.dry_run
    # ---- every worker pool runs HERE: nothing has started a thread or touched the GPU yet -----------
    pairs = scans = full_pairs = rgbd_pairs = None
    if not args.dry_run:
        pairs = [gen_pair((seeds[0], n))] * S if args.same_pair else pool_map(gen_pair, [(sd, n) for sd in seeds], world)
        if extras and em:
            if args.sequence_pairs > 0:
                scans = pool_map(gen_scan, [(5, i, n) for i in range(args.sequence_pairs + 1)], 1)
            if args.full_size_pairs > 0:
                full_pairs = pool_map(gen_full_scan_pair, [1000 + k for k in range(args.full_size_pairs)], 1)
                rgbd_pairs = pool_map(gen_rgbd_pair, [3 + k for k in range(args.full_size_pairs)], 1)
    dist = Dist()   # gloo (threads) only from here on

    engines, sicp, device = [], None, 0
    if args.dry_run:
        def timed(steps):
            t0 = time.perf_counter()
            time.sleep(0.01 * (1 + dist.rank) * steps)
            agg = dict(total_corr=3 * n * K * S * steps, outer_iters=3 * S * steps, total_evals=90 * S * steps, total_lm_iters=80 * S * steps,
                       lockstep_slots=120 * S * steps, graph_builds=0)
            return time.perf_counter() - t0, agg, [3] * S, [90] * S, [ident] * S
        sync = lambda: None  # noqa: E731
    else:
        sicp = importlib.import_module("semantic-icp_amd")
        ndev = sicp.device_count()
        if ndev < 1:
            raise SystemExit("bench.py: no HIP device visible (there is no CPU fallback)")
        if dist.world > ndev and not args.oversubscribe:
            raise SystemExit(f"bench.py: --gpus {dist.world} but only {ndev} HIP device(s) visible (--oversubscribe shares them)")
        device = dist.local_rank % ndev
        p = sicp.default_params(sicp.MODE_EM if em else sicp.MODE_GICP)
        p.num_classes = N_CLASSES if em else 0
        p.profile = args.profile
        if args.nn_method is not None:
            p.nn_method = args.nn_method
        if args.lm_on_device is not None:
            p.lm_on_device = args.lm_on_device
        if args.lm_batch is not None:
            p.lm_batch = args.lm_batch
        for src, sl, tgt, tl in pairs:
            e = sicp.Engine(device, p)
            if em:
                e.set_confusion(cm)
            e.set_source(src, sl if em else None)   # clouds resident in HBM before the timed region
            e.set_target(tgt, tl if em else None)
            engines.append(e)

        def sync():
            for e in engines:
                e.synchronize()

        def timed(steps):
            return run_batch_steps(sicp, engines, steps, 0, sync)

    if args.warmup > 0:
        timed(args.warmup)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    elapsed_local, agg, outer_pp, evals_pp, qts = timed(args.steps)
    sync()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_max = dist.reduce(elapsed, "max")
    corr_all = dist.reduce(float(agg["total_corr"]), "sum")
    evals_all = dist.reduce(float(agg["total_evals"]), "sum")
    slots_all = dist.reduce(float(agg["lockstep_slots"]), "sum")
    outer_all = dist.reduce(float(agg["outer_iters"]), "sum")
    seed_lo = dist.reduce(float(min(seeds)), "min")
    seed_hi = dist.reduce(float(max(seeds)), "max")

    out = None
    if dist.rank == 0:
        steps = max(1, args.steps)
        wl = (f"EM-ICP align(), K={K_CORR}, C={N_CLASSES}" if em else "SE3-GICP align(), K=1") + \
             f", synthetic KITTI-like scan pairs of {n}x{n} points (metric point of BASELINE configs[1]); " + \
             (f"{S} copies of one pair per GPU (--same-pair)" if args.same_pair else f"{S} DIFFERENT pairs per GPU (seeds 2+{S}*rank+k: own street, ego-motion, noise)")
        out = {
            "metric": "correspondences/sec",
            "value": corr_all / elapsed_max,
            "unit": "correspondences/s",
            "n_gpus": dist.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (kNN) / f64 (residuals, Jacobians, solve)",
            "data": "dry-run" if args.dry_run else "synthetic",
            "config": {
                "workload": wl,
                "points": n, "K": K, "classes": N_CLASSES if em else 0, "parallelism": f"pairs-sharded x{dist.world}",
                "pairs_in_flight_per_gpu": S,
                "seeds_min_max": [int(seed_lo), int(seed_hi)],
                "oversubscribed": bool(args.oversubscribe),
                "step": f"{S} full align() calls per GPU (covariances of both clouds + all outer ICP iterations each), "
                        "registered together by one sicp_align_batch call (continuous batching)",
            },
            "pairs_per_s": S * dist.world * steps / elapsed_max,
            # three meanings of "ms per outer ICP iteration" (BASELINE.json's second metric), spelled out:
            #   in_batch_wall : wall time between two outer iterations of ONE pair while it shares the GPU with the other S - 1
            #   amortised     : GPU wall time per outer iteration completed on that GPU (all pairs) = 1 / throughput
            #   alone         : one pair alone on the GPU, the reference's call pattern (filled in below at N = 1)
            "ms_per_icp_iter_in_batch_wall": 1e3 * elapsed_max * dist.world * S / max(1.0, outer_all),
            "ms_per_icp_iter_amortised": 1e3 * elapsed_max * dist.world / max(1.0, outer_all),
            "ms_per_icp_iter_alone": None,
            "outer_iters_per_align": outer_all / (steps * S * dist.world),
            "accumulate_passes_per_outer_iter": evals_all / max(1.0, outer_all),
            "lockstep": {
                # sum over pairs of their own LM evaluations / sum over pairs of the evaluation launches they sat through
                "busy_fraction": evals_all / max(1.0, slots_all),
                "outer_iters_min_max": [int(min(outer_pp)), int(max(outer_pp))],
                "evals_per_align_min_max": [int(min(evals_pp)), int(max(evals_pp))],
                "graph_builds_in_timed_region": int(agg["graph_builds"]),
                "note": "rank 0's batch, last step; a pair only takes part in the ticks (graph launches of lm_batch evaluations) it is inside an "
                        "inner solve for, so the idle part is the tail of the tick in which its solve ends",
            },
        }

    single = others = None
    if engines and extras:
        engine = engines[0]
        others = []
        # --- one pair alone on the GPU: the reference's own call pattern (latency) ---------------------
        reps = 5
        engine.align(ident)
        engine.synchronize()
        t1 = time.perf_counter()
        sc = so = 0
        for _ in range(reps):
            _, st1 = engine.align(ident)
            sc += st1["total_corr"]; so += st1["outer_iters"]
        dt1 = time.perf_counter() - t1
        single = {"workload": f"one pair alone (sicp_align, the reference's call pattern), {'EM-ICP K=4' if em else 'SE3-GICP K=1'}, {n}x{n}",
                  "value": sc / dt1, "unit": "correspondences/s", "ms_per_align": 1e3 * dt1 / reps,
                  "ms_per_icp_iter": 1e3 * dt1 / max(1, so), "outer_iters": so / reps}
        out["ms_per_icp_iter_alone"] = single["ms_per_icp_iter"]
        others.append(single)
        # --- the other registration class of exec/kitti_eval.cc on the same pairs ----------------------
        alt = sicp.default_params(sicp.MODE_GICP if em else sicp.MODE_EM)
        alt.num_classes = 0 if em else N_CLASSES
        alt_engines = []
        for src, sl, tgt, tl in pairs:
            e = sicp.Engine(device, alt)
            if not em:
                e.set_confusion(cm)
            e.set_source(src, None if em else sl)
            e.set_target(tgt, None if em else tl)
            alt_engines.append(e)
        dt, a2, o2, e2, _ = run_batch_steps(sicp, alt_engines, 3, 1, lambda: [e.synchronize() for e in alt_engines])
        others.append({"workload": ("SE3-GICP align(), K=1" if em else f"EM-ICP align(), K={K_CORR}, C={N_CLASSES}") + f", the same {S} pairs in lock step",
                       "value": a2["total_corr"] / dt, "unit": "correspondences/s", "ms_per_step": 1e3 * dt / 3,
                       "pairs_per_s": 3 * S / dt, "outer_iters_per_align": a2["outer_iters"] / (3 * S),
                       "busy_fraction": a2["total_evals"] / max(1, a2["lockstep_slots"])})
        for e in alt_engines:
            e.close()
        # --- the full-size configs of BASELINE.json as batches ------------------------------------------
        if full_pairs:
            others.append(full_size_leg(sicp, device, sicp.MODE_EM, full_pairs, N_CLASSES, cm, None,
                                        f"config 2 at full size: EM-ICP K={K_CORR} C={N_CLASSES} on whole KITTI-like scans (not subsampled)"))
        if rgbd_pairs:
            others.append(full_size_leg(sicp, device, sicp.MODE_SEMANTIC, rgbd_pairs, 13, None, None,
                                        "config 3: SemanticICP (per-class K=1 search, Cauchy 1.5; exec/nyu_eval.cc:139) on 640x480 RGB-D frame pairs, 13 classes"))
            others.append(full_size_leg(sicp, device, sicp.MODE_EM, rgbd_pairs, 13, synth.confusion_matrix(13), 1e-6,
                                        "config 3: EM-ICP K=4 C=13 eps=1e-6 (exec/scenenet_eval.cc:174) on the same RGB-D frame pairs"))
        # --- end-to-end sequence (cloud upload + tree build inside the timed region) ------------------
        if scans:
            others.append(stream_leg(sicp, device, engine.get_params(), cm, scans, args.stream_in_flight, args.stream_lm_batch))
        out["other_workloads"] = others

        # --- rooflines (SURVEY.md 8d per-unit bytes x units per launch / HIP-event duration) ----------
        # accumulate: the path's HBM-model kernel (24 B per source point + 32 B per correspondence and
        # pass).  The timed region runs it once per LM evaluation over all pairs that still iterate (up to
        # 256 per launch); here the launch is timed alone with HIP events on its stream
        # (sicp_accumulate_batch: 50 launches back to back between two events) at BOTH shapes: over
        # min(S, 256) pairs -- the launch of the timed region, the headline `roofline` -- and over 32
        # pairs, the shape of rounds 1 and 2.
        for e, q in zip(engines[:min(S, 256)], qts[:min(S, 256)]):
            e.correspondences(q)
        acc = {}
        for L in sorted({min(S, 256), min(S, 32)}):
            ms_l = []
            for _ in range(6):
                _, ms = sicp.accumulate_batch(engines[:L], np.array(qts[:L]), repeat=50 if L <= 32 else 10)
                ms_l.append(ms)
            us = 1e3 * float(np.mean(ms_l[2:]))
            b = L * (24 * n + 32 * K * n)
            acc[L] = {"pairs_per_launch": L, "avg_launch_us": us, "algorithmic_bytes_per_launch": b, "achieved": b / (us * 1e-6) / 1e9,
                      "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "launches_timed": (50 if L <= 32 else 10) * len(ms_l[2:])}
        L = max(acc)
        traffic, traffic_src = None, None
        t = load_profile_json("pmc_hbm_traffic.json")
        key = f"accumulate_batch_K{K}_pairs{L}_n{n}"
        if t and key in t:
            traffic = t[key]["bytes_per_launch"]
            traffic_src = (f"{os.path.relpath(PROFILE_DIR, ROOT)}/pmc_hbm_traffic.json:{key}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same "
                           "launch of this build (tools/pmc_accumulate.sh), FETCH_SIZE x2 as calibrated on this access pattern; from the committed "
                           "profile, not measured in this run")
        out["roofline"] = {
            "kernel": f"accumulate_staged_kernel<K={K}> (Mahalanobis residual + 6-DoF Jacobian + robust loss -> 28 sums per pair; persistent "
                      f"workgroups over contiguous chunk ranges, LDS-staged gathers, wave-private reductions; one launch per LM evaluation, timed here "
                      f"over {L} pairs: the launch shape of the timed region)",
            "bound": "hbm", "achieved": acc[L]["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": acc[L]["frac"],
            "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_us": acc[L]["avg_launch_us"], "launches_timed": acc[L]["launches_timed"],
            "algorithmic_bytes_per_launch": acc[L]["algorithmic_bytes_per_launch"], "pairs_per_launch": L,
            "other_launch_shapes": [acc[k] for k in sorted(acc) if k != L],
        }
        # the search kernels: HIP-event durations measured here (one pair alone), counter-based figures
        # (HBM bytes, instruction counts) from the committed PMC passes of this build
        pp = engine.get_params()
        pp.profile = 1 | 2  # SICP_PROFILE_NN | SICP_PROFILE_COV: the search kernels alone
        engine.set_params(pp)
        b0 = engine.stats()
        for _ in range(20):
            engine.correspondences(qts[0])
        b1 = engine.stats()
        pp.profile = 0
        engine.set_params(pp)
        avg_ms = (b1["nn_kernel_ms"] - b0["nn_kernel_ms"]) / max(1, b1["nn_launches"] - b0["nn_launches"])
        alg_bytes = 12 * n + 12 * n + 8 * K * n     # src+tgt xyz once, idx+dist^2 out
        knn_pmc = load_profile_json("pmc_knn.json") or {}
        ok = []
        ent = {"kernel": f"bvh_knn_packet_kernel<{K}> (exact box-tree search, 16 queries per wave share one walk), one pair alone, HIP events",
               "bound": "issue", "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes,
               "algorithmic_GBps": alg_bytes / (avg_ms * 1e-3) / 1e9}
        ok.append(ent)
        for name, rec in knn_pmc.items():   # per kernel: counter HBM GB/s and the issue-bound fraction, as measured by tools/pmc_knn.sh
            ok.append(dict(rec, kernel=name, source=f"{os.path.relpath(PROFILE_DIR, ROOT)}/pmc_knn.json (committed profile of this build)"))
        out["other_kernels"] = ok
        if not args.no_cpu_baseline and em:
            src, sl, tgt, tl = pairs[0]
            base, oq = cpu_baseline(src, sl, tgt, tl, cm)
            from scipy.spatial.transform import Rotation

            import oracle_lib as O

            D = np.linalg.inv(O.se3_matrix(oq)) @ O.se3_matrix(qts[0])
            out["pose_delta_vs_cpu"] = {"rot_rad": float(np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec())),
                                        "trans_m": float(np.linalg.norm(D[:3, 3]))}
            base.pop("pose")
            out["cpu_baseline"] = base
    dist.barrier()
    for e in engines:
        e.close()
    dist.close()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
