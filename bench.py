#!/usr/bin/env python3
"""bench.py -- headline benchmark of the semantic-ICP hot path on MI355X.

Metric (BASELINE.json): correspondences/sec (+ ms per outer ICP iteration) for EM-ICP
(K = 4 correspondences per source point, C = 11 classes) on a synthetic KITTI-like scan pair
subsampled to exactly 100 000 x 100 000 points -- the metric point of configs[1].

A "step" registers one batch of independent scan pairs on each GPU: `--pairs-in-flight` S pairs
(default 32), each a complete align() (covariances of both clouds + every outer ICP iteration:
transform -> kNN -> EM weights -> inner LM solve) on its own handle, with all clouds already
resident in HBM when the timed region starts.  Independent pairs are the reference's unit of work
(exec/kitti_eval.cc loops over them) and the north star shards them across GPUs.  By default the S
pairs of a GPU advance in lock step through one sicp_align_batch call: every kernel launch of the
path (searches, weights, LM evaluations, LM steps) covers all S pairs, per pair bit-identical to a
lone align().  `--concurrency threads` runs them as S host threads + streams instead.
Single-pair latency (S = 1) is measured after the timed region and reported in "single_pair".
One correspondence = one (source, target) slot that went through kNN + weighting + accumulation
in one outer iteration.

  python bench.py --gpus N --steps K --warmup W
For N > 1 launch with:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...
(one process per GPU; independent scan pairs per rank, no collective on the solve path, so the
process group is gloo and only carries the barrier and the max-over-ranks of the timings).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# (points, pairs per launch) -> FETCH_SIZE + WRITE_SIZE bytes per accumulate_batch launch, raw counter values from
# profiles/r01_final_pmc_hbm_traffic.csv (16 pairs: 117805.8 KB + 2773.6 KB; 32 pairs: 235500 KB + 5545 KB)
PMC_TRAFFIC_BYTES = {(100_000, 16): (117805.8 + 2773.6) * 1024, (100_000, 32): (235500.0 + 5545.0) * 1024}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
VALU_PAIR_PEAK = 9.8e12        # 78.6e12 FP32 lane-ops/s / 8 lane-ops per pair
N_POINTS = 100_000
K_CORR = 4
N_CLASSES = 11


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nn-method", type=int, default=None, help="0 brute force, 1 box tree (default: library default)")
    ap.add_argument("--lm-on-device", type=int, default=None, help="0 host LM loop, 1 device-resident (default: library default)")
    ap.add_argument("--lm-batch", type=int, default=None)
    ap.add_argument("--pairs-in-flight", type=int, default=32,
                    help="independent scan pairs registered concurrently on each GPU (one handle + host thread each)")
    ap.add_argument("--profile", type=int, default=0,
                    help="SICP_PROFILE_* mask applied inside the timed region (default 0: the roofline kernels are timed with "
                         "HIP events right after it, on the same data and streams)")
    ap.add_argument("--concurrency", choices=["lockstep", "threads"], default="lockstep",
                    help="how the pairs in flight share the GPU: one sicp_align_batch call (lock step, batched launches) or "
                         "one host thread + stream per pair")
    ap.add_argument("--timed-only", action="store_true", help="skip the single-pair / roofline / CPU legs (for tracing the timed region)")
    ap.add_argument("--dry-run", action="store_true", help="exercise the multi-process plumbing without a GPU")
    return ap.parse_args()


class Dist:
    """torch.distributed (gloo) only when launched with WORLD_SIZE > 1."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.pg = None
        if self.world > 1:
            import torch  # noqa: F401  (imported before libsicp so that both share one HIP runtime)
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
        self.pg.all_reduce(t, op=self.pg.ReduceOp.MAX if op == "max" else self.pg.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.pg:
            self.pg.destroy_process_group()


def cpu_baseline(src, sl, tgt, tl, cm):
    """The CPU restatement of the reference PCL-KdTree + Ceres path (the oracle), timed on this
    host on the same pair: kNN + problem build on 1 thread (em_icp.hpp:57-156), residual /
    Jacobian evaluation on 8 threads (em_icp.hpp:166)."""
    import numpy as np

    import oracle_lib as O

    p = O.default_params(O.MODE_EM)
    p.num_classes = N_CLASSES
    p.use_kdtree = 1
    threads = min(8, os.cpu_count() or 1)
    p.num_threads = threads
    t0 = time.perf_counter()
    qt, st = O.align(p, src, sl, tgt, tl, cm, np.array([0, 0, 0, 1, 0, 0, 0.0]))
    dt = time.perf_counter() - t0
    return {
        "value": st["total_corr"] / dt,
        "unit": "correspondences/s",
        "cores": threads,
        "kind": "port",
        "sample": f"1 full align() of the same {len(src)}x{len(tgt)} pair (rank-0 pair): {st['outer_iters']} outer iterations, "
                  f"{st['total_evals']} residual sweeps, {dt:.1f} s; kd-tree kNN + problem build on 1 thread, "
                  f"residual/Jacobian sweeps on {threads} threads (host has {os.cpu_count()} cores)",
        "ms_per_icp_iter": 1e3 * (dt - st["t_cov_s"]) / max(1, st["outer_iters"]),
        "cov_ms": 1e3 * st["t_cov_s"],
        "pose": [float(v) for v in qt],
    }, qt


def main():
    args = parse_args()
    dist = Dist()
    import numpy as np

    import synth

    n = args.points
    # weak scaling: every rank registers its own, differently seeded, pair
    src, sl, tgt, tl, T_gt, cm = synth.lidar_pair(seed=2 + dist.rank, n_points=n)
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])

    if args.dry_run:
        engine = None

        def step():
            time.sleep(0.01 * (1 + dist.rank))
            return ident, dict(total_corr=3 * n * K_CORR, outer_iters=3, total_evals=30, nn_kernel_ms=0.0, nn_launches=0,
                               t_cov_ms=0.0, total_lm_iters=0)
    else:
        sicp = importlib.import_module("semantic-icp_amd")
        ndev = sicp.device_count()
        if ndev < 1:
            raise SystemExit("bench.py: no HIP device visible (there is no CPU fallback)")
        p = sicp.default_params(sicp.MODE_EM)
        p.num_classes = N_CLASSES
        p.profile = args.profile
        if args.nn_method is not None:
            p.nn_method = args.nn_method
        if args.lm_on_device is not None:
            p.lm_on_device = args.lm_on_device
        if args.lm_batch is not None:
            p.lm_batch = args.lm_batch
        nn_method = p.nn_method
        S = max(1, args.pairs_in_flight)
        engines = []
        for k in range(S):
            e = sicp.Engine(dist.local_rank % ndev, p)
            e.set_confusion(cm)
            e.set_source(src, sl)   # clouds resident in HBM before the timed region
            e.set_target(tgt, tl)
            engines.append(e)
        engine = engines[0]

        def step():
            if S == 1:
                return engine.align(ident)
            if args.concurrency == "lockstep":
                # S independent registrations advanced in lock step by one call (batched launches)
                res = sicp.align_batch(engines)
            else:
                # S independent registrations in flight: one host thread per handle (ctypes drops the GIL)
                import concurrent.futures as cf

                with cf.ThreadPoolExecutor(S) as ex:
                    res = list(ex.map(lambda e: e.align(ident), engines))
            qt0, st0 = res[0]
            agg = dict(st0)
            for _, st in res[1:]:
                for key in ("total_corr", "outer_iters", "total_evals", "total_lm_iters", "nn_kernel_ms", "nn_launches",
                            "t_cov_ms", "acc_kernel_ms", "acc_launches"):
                    agg[key] += st[key]
            return qt0, agg

    for _ in range(args.warmup):
        step()
    if engine:
        engine.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    corr = outer = evals = nn_launches = lm_iters = acc_launches = 0
    nn_ms = cov_ms = acc_ms = 0.0
    qt = ident
    for _ in range(args.steps):
        qt, st = step()
        corr += st["total_corr"]; outer += st["outer_iters"]; evals += st["total_evals"]
        nn_ms += st["nn_kernel_ms"]; nn_launches += st["nn_launches"]; cov_ms += st["t_cov_ms"]
        lm_iters += st["total_lm_iters"]
        acc_ms += st.get("acc_kernel_ms", 0.0); acc_launches += st.get("acc_launches", 0)
    if engine:
        engine.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_max = dist.reduce(elapsed, "max")
    corr_all = dist.reduce(float(corr), "sum")
    single = None
    if engine and dist.rank == 0 and not args.timed_only:
        # latency of one pair alone on the GPU (outside the timed region)
        reps = 3
        engine.align(ident)
        t1 = time.perf_counter()
        sc = so = 0
        scov = 0.0
        for _ in range(reps):
            _, st1 = engine.align(ident)
            sc += st1["total_corr"]; so += st1["outer_iters"]; scov += st1["t_cov_ms"]
        dt1 = time.perf_counter() - t1
        single = {"value": sc / dt1, "unit": "correspondences/s", "ms_per_align": 1e3 * dt1 / reps,
                  "ms_per_icp_iter": (1e3 * dt1 - scov) / max(1, so), "cov_ms_per_align": scov / reps}

    out = None
    if dist.rank == 0:
        steps = max(1, args.steps)
        ms_per_step = 1e3 * elapsed_max / steps
        out = {
            "metric": "correspondences/sec",
            "value": corr_all / elapsed_max,
            "unit": "correspondences/s",
            "n_gpus": dist.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (kNN) / f64 (residuals, Jacobians, solve)",
            "data": "dry-run" if args.dry_run else "synthetic",
            "config": {
                "workload": f"EM-ICP align() on a synthetic KITTI-like scan pair, {n}x{n} points, K={K_CORR}, C={N_CLASSES} "
                            "(metric point of BASELINE configs[1]); every pair of a GPU's batch is its own registration",
                "points": n, "K": K_CORR, "classes": N_CLASSES, "parallelism": f"pairs-sharded x{dist.world}",
                "pairs_in_flight_per_gpu": max(1, args.pairs_in_flight),
                "concurrency": args.concurrency,
                "step": f"{max(1, args.pairs_in_flight)} full align() calls per GPU (covariances of both clouds + all outer ICP iterations each), "
                        + ("advanced in lock step by one sicp_align_batch call" if args.concurrency == "lockstep" else "one host thread + stream each"),
            },
            "ms_per_icp_iter": 1e3 * (elapsed * max(1, args.pairs_in_flight) - 1e-3 * cov_ms) / max(1, outer),
            "cov_ms_per_align": cov_ms / (steps * max(1, args.pairs_in_flight)),
            "outer_iters_per_align": outer / (steps * max(1, args.pairs_in_flight)),
            "accumulate_passes_per_outer_iter": evals / max(1, outer),
            "lm_iters_per_outer_iter": lm_iters / max(1, outer),
        }
        if acc_ms > 0:
            out["accumulate_kernel_us_per_launch"] = 1e3 * acc_ms / max(1, acc_launches)
        if single:
            out["single_pair"] = single
        if not args.dry_run and not args.timed_only:
            # --- rooflines (SURVEY.md 8d per-unit bytes x units per launch / HIP-event duration) -------
            # accumulate: the path's HBM-model kernel (38 B per correspondence per pass); timed alone
            # with HIP events on the handle's stream, right after the timed region, on the
            # correspondences the last align() left in HBM
            pp = engine.get_params()
            pp.profile = 1  # SICP_PROFILE_NN: the correspondence-search kernel, 20 launches at the final pose
            engine.set_params(pp)
            b0 = engine.stats()
            for _ in range(20):
                engine.correspondences(qt)
            b1 = engine.stats()
            nn_ms = b1["nn_kernel_ms"] - b0["nn_kernel_ms"]
            nn_launches = b1["nn_launches"] - b0["nn_launches"]
            pp.profile = 0
            engine.set_params(pp)
            # accumulate: the path's HBM-model kernel (38 B per correspondence per pass).  The timed
            # region runs it as accumulate_batch_kernel, one launch per LM evaluation for all S pairs of
            # the lock-step batch; here the same launch is timed alone with HIP events on its stream
            # (sicp_accumulate_batch: 50 launches back to back between two events, 6 rounds)
            # (with SICP_BATCH_CHAINS=2 the batched solve graph has two chains -- the halves of the batch -- and one
            # launch of the timed region covers S/2 pairs; the launch shape of the timed region is what is timed here)
            chains = 2 if (S >= 4 and args.concurrency == "lockstep" and os.environ.get("SICP_BATCH_CHAINS", "1") == "2") else 1
            L = S // chains if args.concurrency == "lockstep" else 1
            for e in engines[:L]:
                e.correspondences(qt)
            qts = np.tile(qt, (L, 1))
            acc_ms_l = []
            for _ in range(8):
                _, ms = sicp.accumulate_batch(engines[:L], qts, repeat=50)
                acc_ms_l.append(ms)
            acc_us = 1e3 * float(np.mean(acc_ms_l[2:]))
            n_acc = 50 * len(acc_ms_l[2:])
            acc_bytes = L * (24 * n + 32 * K_CORR * n)      # pairs per launch x (24*N_s + 32*K*N_s)
            acc_gbs = acc_bytes / (acc_us * 1e-6) / 1e9
            out["roofline"] = {
                "kernel": f"accumulate_batch_kernel<K=4> (Mahalanobis residual + 6-DoF Jacobian -> 28 doubles per pair; one launch per LM "
                          f"evaluation covers {L} pairs ({chains} concurrent chain(s) of the {S}-pair batch), {evals / max(1, outer):.1f} launches per chain and outer iteration)",
                "bound": "hbm", "achieved": acc_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": acc_gbs / HBM_PEAK_GBS,
                # PMC passes (profiles/): FETCH_SIZE + WRITE_SIZE per launch, raw (the guide's x2 FETCH correction is
                # calibrated for 16-B streams only); filled from the committed profile for the default configuration
                "traffic": PMC_TRAFFIC_BYTES.get((n, L)),
                "avg_launch_us": acc_us, "launches_timed": n_acc, "algorithmic_bytes_per_launch": acc_bytes,
                "pairs_per_launch": L, "concurrent_chains": chains,
                "note": "FP64 issue first (PMC: 245 VALU instructions per correspondence incl. reductions "
                        f"-> {L * 4e5 * 245 / (1024 * 16 * 2.4e9) * 1e6:.0f} us on 1024 SIMDs), HBM second (what the launch really moves, f64 normals and "
                        f"weights included: ~12 MB per pair -> {L * 12e6 / 6.3e12 * 1e6:.0f} us at the 6.3 TB/s achievable); two waves per SIMD "
                        "(180 VGPRs); DESIGN.md section 3",
            }
            avg_ms = nn_ms / nn_launches
            alg_bytes = 12 * n + 12 * n + 8 * K_CORR * n     # src+tgt xyz once, idx+dist^2 out
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            kname = {0: "nn_partial_kernel<K=4,Q=2> (LDS-tiled brute force)", 1: "bvh_knn_packet_kernel<K=4> (exact box-tree search, 16 queries per wave share one walk)",
                     2: "bvh_knn_quad_kernel<K=4> (exact box-tree search, 4 lanes per query)"}[nn_method]
            out["other_kernels"] = [{
                "kernel": kname + ", one launch per outer iteration; timed with HIP events right after the timed region",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "avg_launch_ms": avg_ms, "launches": nn_launches, "algorithmic_bytes_per_launch": alg_bytes,
                "note": "latency bound tree walk over an L2-resident cloud (one pair alone; the batch runs all pairs' searches in one launch)" if nn_method >= 1 else
                        "FP32-VALU bound: 1e10 pair evaluations per launch",
                "pair_evals_per_s_if_brute_force": float(n) * n / (avg_ms * 1e-3),
            }]
        if not args.dry_run and not args.no_cpu_baseline and not args.timed_only:
            base, oq = cpu_baseline(src, sl, tgt, tl, cm)
            from scipy.spatial.transform import Rotation

            import oracle_lib as O

            D = np.linalg.inv(O.se3_matrix(oq)) @ O.se3_matrix(qt)
            out["pose_delta_vs_cpu"] = {"rot_rad": float(np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec())),
                                        "trans_m": float(np.linalg.norm(D[:3, 3]))}
            base.pop("pose")
            out["cpu_baseline"] = base
    dist.barrier()
    if engine:
        for e in engines:
            e.close()
    dist.close()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
