#!/usr/bin/env python3
"""bench.py -- NN queries/s of the obstacle-cloud engine on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic queries: the batched 1-NN
(pct_nn_batch_dev, include/pct_engine.h) of Q queries against the cloud resident in HBM.
Workload at N=1 (config C3-throughput of SURVEY.md section 8(d)): 10,000,000 uniform points in
[0,100)^3 (seed 3), Q = 1,048,576 uniform queries (seed 5), cell-pruned kernel (device-side query
binning + nn_grid_coop_kernel).
At N>1 (weak scaling): the cloud grows to N x 10M points at constant
density, rank r owns the contiguous index range [r*10M, (r+1)*10M) in its own HBM, the query
batch is replicated, every rank runs the same kernel on its shard and ONE exchange step
(RCCL all_reduce(min) on fp64 d2, then all_reduce(min) on the matching indices) merges them.

Reported `value` = ranks x Q / t: (query, 10M-point shard) evaluations per second, i.e. queries/s
in units of the metric's 10M-point cloud; `config.answered_queries_per_s` = Q / t is the rate of
merged answers against the whole N x 10M cloud.  At N=1 both are the same number.

One JSON line on rank 0; see the field notes in DESIGN.md section 6.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def measured_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this same command
    (scripts/prof_bench.sh -> profiles/*_pmc.json; 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, separate passes).
    None when no summary is committed (PMC counters cannot be collected inside the timed run)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for k, v in d.items():
            if kernel in k and "<true>" not in k and "derived_hbm_traffic_bytes_per_launch" in v:   # <true> = instrumented twin
                best = v["derived_hbm_traffic_bytes_per_launch"]
    return best


def replan_probe(E, synth, window=5_000_000, frame=50_000, ticks=60):
    """Config C5: rolling window of 5,000,000 points fed 50,000 per sensor frame
    (uniform in a 60 m cube around a drone moving +0.1 m per frame along x, seed 8), oldest frame evicted.
    Per tick, through the host-buffer entry points (PCIe + launch latency included), on the un-indexed
    rolling cloud (brute-force kernels; the cloud changes every tick):
      ingest   pct_cloud_append_aos of the new frame
      corridor pct_inflate_batch of 64 corridor nodes (SafeRegionEvaluate's re-check, corridor_finder.cpp:829-835)
      bezier   pct_bezier_check: 3 segments of order 6, dt 0.02 s over a 2.0 s horizon = 99 samples
               (checkSafeTrajectory, sim_planning_demo.cpp:729-781)
    Reported: p50 / p99 milliseconds per tick and per part."""
    import numpy as np
    cloud = E.Cloud(window)
    pos = 0.0

    def frame_pts(k):
        p = synth.uniform_points(8, frame, -30.0, 30.0, offset=k * frame)
        p[:, 0] += np.float32(0.1 * k)
        p[:, 2] = np.abs(p[:, 2]) * np.float32(0.2)
        return p

    nfill = window // frame
    for k in range(nfill - 2):
        cloud.append(frame_pts(k))
    orders = np.int32([6, 6, 6])
    seg_time = np.float64([1.0, 1.0, 1.0])
    t_ing, t_cor, t_bez = [], [], []
    import gc
    gc.collect()
    gc.disable()                                   # a generation-2 collection of the interpreter (40 ms) used to land in one tick
    for k in range(nfill - 2, nfill + ticks):      # two untimed warm-up ticks (workspaces, first launches)
        x0 = 0.1 * k
        new_frame = frame_pts(k)                  # the sensor's output: produced outside the timed region
        t0 = time.perf_counter()
        cloud.append(new_frame)
        t1 = time.perf_counter()
        nodes = (synth.uniform_points(9, 64, -1.0, 1.0, offset=k * 64).astype(np.float64) * [8.0, 3.0, 1.0] + [x0 + 6.0, 0.0, 2.5])
        prm = E.inflate_params((x0, 0.0, 2.5), 30.0, 0.25, 1.5)
        cloud.inflate(prm, nodes)
        t2 = time.perf_counter()
        coef = np.zeros((3, 21))
        ctrl = synth.uniform_points(9, 21, -0.3, 0.3, offset=1_000_000 + k * 21).astype(np.float64)
        for sgm in range(3):
            for d in range(3):
                for j in range(7):
                    w = (sgm + j / 6.0) / 3.0
                    base = [x0 + 12.0 * w, 0.0, 2.5][d]
                    coef[sgm, d * 7 + j] = (base + (ctrl[sgm * 7 + j, d] if 0 < j < 6 else 0.0)) / seg_time[sgm]
        cloud.bezier_check(prm, coef, seg_time, orders, 0.0, 2.0, cap=128)
        t3 = time.perf_counter()
        if k >= nfill:
            t_ing.append(1e3 * (t1 - t0)); t_cor.append(1e3 * (t2 - t1)); t_bez.append(1e3 * (t3 - t2))
    gc.enable()
    tot = np.asarray(t_ing) + np.asarray(t_cor) + np.asarray(t_bez)
    cloud.close()
    pct = lambda a, q: float(np.percentile(a, q))
    return {"what": "C5: 5,000,000-point rolling cloud, +50,000 points per tick, 64 corridor-node inflations + 99-sample Bezier check per tick, host buffers",
            "ticks": ticks, "ms_per_tick_p50": pct(tot, 50), "ms_per_tick_p99": pct(tot, 99),
            "ingest_ms_p50": pct(t_ing, 50), "corridor_inflate_ms_p50": pct(t_cor, 50), "bezier_check_ms_p50": pct(t_bez, 50),
            "worst_tick": {"index": int(np.argmax(tot)), "ingest_ms": float(np.asarray(t_ing)[np.argmax(tot)]),
                           "corridor_inflate_ms": float(np.asarray(t_cor)[np.argmax(tot)]), "bezier_check_ms": float(np.asarray(t_bez)[np.argmax(tot)])},
            "budget_ms_at_20Hz": 50.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=10_000_000, help="points per GPU")
    ap.add_argument("--queries", type=int, default=1 << 20)
    ap.add_argument("--algo", choices=["grid", "stream"], default="grid")
    ap.add_argument("--cell", type=float, default=0.0, help="grid cell size (<=0: automatic)")
    ap.add_argument("--cpu-queries", type=int, default=20000, help="queries timed on the host kd-tree (0 = skip)")
    ap.add_argument("--cpu-points", type=int, default=0, help="points in the host kd-tree (0 = same as --points)")
    ap.add_argument("--stream-probe", type=int, default=1, help="also time the streaming / brute-force / corridor probes (0 = skip)")
    ap.add_argument("--replan-probe", type=int, default=1, help="config C5: rolling 5M-point cloud, 20 Hz replan ticks (0 = skip)")
    return ap.parse_args()


def cpu_baseline(points_fn, n_points, queries, nq):
    """The reference's own kdtree.c (oracle/_ref, kind 'reference') or, when that library did not
    travel, our port of it (kind 'port'); one host thread, the loop the reference's callers run
    (kd_nearestf -> kd_res_item_data -> kd_res_free, corridor_finder.cpp:428-437)."""
    from oracle import oracle as O
    O.build()
    kd = O.RefKD() if O.have_ref() else O.PortKD()
    from pointcloudtraj_amd import synth
    order = synth.shuffled_order(1234, n_points)       # shuffled insertion, as the survey's calibration did
    t0 = time.perf_counter()
    pts = points_fn()
    kd.insert(pts[order])
    t_build = time.perf_counter() - t0
    q = np.ascontiguousarray(queries[:nq])
    if hasattr(kd, "nearest_timed"):
        secs, idx = kd.nearest_timed(q)
    else:
        t0 = time.perf_counter()
        idx, _ = kd.nearest(q)
        secs = time.perf_counter() - t0
    out = {"value": nq / secs, "unit": "queries/s", "cores": 1, "kind": kd.kind,
           "sample": f"{nq} of the batch's queries against a host kd-tree of {n_points} points "
                     f"(shuffled kd_insertf build {t_build:.1f} s, not included)",
           "build_s": round(t_build, 2)}
    if hasattr(kd, "nearest_timed_mt"):
        # the same tree shared read-only by every host core this process may use (the reference itself is single-threaded)
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        rep = max(1, min(64, threads // 4))                  # a few thousand queries per thread, so thread start-up does not dominate
        qq = np.ascontiguousarray(np.concatenate([q] * rep))
        secs_mt, idx_mt = kd.nearest_timed_mt(qq, threads)
        if secs_mt > 0:
            out["all_host_cores"] = {"value": len(qq) / secs_mt, "cores": threads, "same_ids_as_one_thread": bool(np.array_equal(idx_mt[:nq], idx))}
    return out, order[idx.astype(np.int64)], q


def main():
    a = parse()
    import torch   # before the engine: one HIP runtime per process (engine._preload_hip_runtime)
    from pointcloudtraj_amd import dist as D, engine as E, synth
    import torch.distributed as tdist

    rank, local, world = D.init_process_group_from_env()
    if world != a.gpus and world > 1:
        a.gpus = world
    # one GPU per rank; PCT_DIST_BACKEND=gloo rehearses the multi-rank path with several ranks on one card
    dev = (local % max(torch.cuda.device_count(), 1)) if world > 1 else 0
    n_total = a.points * world
    side = 100.0 * (world ** (1.0 / 3.0))      # constant density as the cloud grows
    Q = a.queries
    algo = E.ALGO_GRID if a.algo == "grid" else E.ALGO_STREAM

    sc = D.ShardedCloud(n_total, rank, world, dev)
    local_pts = synth.uniform_points(3, sc.end - sc.begin, 0.0, side, offset=sc.begin)
    sc.set_input_local(local_pts)
    t0 = time.perf_counter()
    if algo == E.ALGO_GRID:
        sc.build_grid(a.cell)
    E.sync()
    t_grid = time.perf_counter() - t0
    q_host = synth.uniform_points(5, Q, 0.0, side)
    sc.reserve(Q)
    q = torch.from_numpy(q_host).to(sc.device)

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    kern_ms = []
    for _ in range(a.warmup):
        sc.nn_submit(q, algo)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        # at N > 1 the exchange step of batch k runs on a side stream under the kernels of batch k+1 (dist.nn_submit);
        # the closing barrier + synchronize waits for every batch's merged answer
        d2, idx, done = sc.nn_submit(q, algo)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=sc.device if tdist.get_backend() == "nccl" else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / a.steps

    # dominant-kernel time: HIP events the engine recorded on the launch stream around every launch of the TIMED region
    # (the engine keeps the last 64 pairs; with more steps than that, the most recent 64 of them)
    kern_ms = sc.cloud.kernel_ms_history(min(a.steps, 64))
    k_ms = float(np.mean(kern_ms))
    # all kernels of one batch (sort + search), from a few extra untimed passes
    batch_ms = []
    sc.cloud.set_timing(2)                      # whole-batch events are opt-in (they cost ~9 us per batch)
    for _ in range(3):
        sc.nn_local(q, algo)
        batch_ms.append(sc.cloud.last_batch_ms())
    sc.cloud.set_timing(1)
    b_ms = float(np.mean(batch_ms))
    # algorithmic work of one launch (separate instrumented pass)
    sc.cloud.set_work_counters(True)
    sc.nn_local(q, algo)
    torch.cuda.synchronize()
    pts_scanned, runs = sc.cloud.last_work()
    sc.cloud.set_work_counters(False)
    bytes_alg = 16 * pts_scanned + 8 * runs + 24 * Q if algo == E.ALGO_GRID else 12 * len(local_pts) * ((Q + 7) // 8) + 24 * Q
    achieved = bytes_alg / (k_ms * 1e-3) / 1e9

    if rank != 0:
        if world > 1:
            tdist.destroy_process_group()
        return

    out = {
        "metric": "nn_queries_per_sec_10M_point_cloud",
        "value": world * Q / elapsed * a.steps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"C3-throughput: {a.points} uniform fp32 points per GPU in [0,{side:.1f})^3 (seed 3), "
                        f"{Q} uniform NN queries per step (seed 5), {a.algo} kernel, inputs resident in HBM",
            "points_per_gpu": a.points, "total_points": n_total, "queries_per_step": Q, "algo": a.algo,
            "answered_queries_per_s": Q / elapsed * a.steps,
            "parallelism": f"cloud sharded by contiguous index range over {world} GPU(s), queries replicated, "
                           "all_reduce(min) merge" if world > 1 else "single GPU",
            "grid": sc.cloud.grid_info() if sc.cloud.has_grid else None,
            "grid_build_s": round(t_grid, 4),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": measured_traffic("nn_grid_coop_kernel" if algo == E.ALGO_GRID else "nn_tile_candidates_kernel"),
            "kernel": "nn_grid_coop_kernel" if algo == E.ALGO_GRID else "nn_tile_candidates_kernel",
            "kernel_ms": k_ms, "kernel_launches_timed": len(kern_ms), "batch_kernels_ms": b_ms, "algorithmic_bytes": int(bytes_alg), "points_scanned": int(pts_scanned),
            "cell_runs": int(runs), "pair_evals_per_s": pts_scanned / (k_ms * 1e-3),
        },
    }

    if a.stream_probe and world == 1:
        cs = torch.cuda.current_stream().cuda_stream

        def timed(fn, reps):
            for _ in range(2):
                fn()
            ms = []
            for _ in range(reps):
                fn()
                ms.append(sc.cloud.last_kernel_ms())
            return float(np.median(ms))

        # (a) streaming kernel at its HBM-bound operating points: the SoA cloud is read once per pass
        probes = []
        for qn in (1, 2, 4):
            ms = timed(lambda: sc.cloud.nn_device(q.data_ptr(), qn, sc._idx32.data_ptr(), sc._d2.data_ptr(), cs, E.ALGO_STREAM), 20)
            sb = 12 * len(local_pts) + 24 * qn
            probes.append({"queries": qn, "kernel_ms": ms, "algorithmic_bytes": sb, "achieved_GBs": sb / (ms * 1e-3) / 1e9,
                           "frac_of_hbm_peak": sb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        out["stream_probe"] = {"kernel": "nn_stream_kernel<QT> + nn_reduce_partials_kernel (all-fp64, one pass over the SoA cloud)",
                               "points": probes}
        # (b) brute force at config C2's batch size: LDS-tiled packed-fp32 filter + exact fp64 recheck
        qn = 4096
        ms = timed(lambda: sc.cloud.nn_device(q.data_ptr(), qn, sc._idx32.data_ptr(), sc._d2.data_ptr(), cs, E.ALGO_STREAM), 5)
        out["brute_force_probe"] = {"kernel": "nn_sample_bounds_kernel + nn_tile_candidates_kernel + nn_reduce_candidates_kernel",
                                    "queries": qn, "kernel_ms": ms, "queries_per_s": qn / (ms * 1e-3),
                                    "pair_evals_per_s": qn * len(local_pts) / (ms * 1e-3),
                                    "flops_per_s": 8 * qn * len(local_pts) / (ms * 1e-3)}
        # (c) corridor side (config C3): sphere inflation of 200 seeds against the 10M-point cloud through the
        # host-buffer entry point (PCIe and launch latency included) -- ms per pass
        seeds = synth.uniform_points(4, 200, 10.0, 90.0).astype(np.float64)
        prm = E.inflate_params((50.0, 50.0, 50.0), 1.0e9, 0.25, 1.5)
        for _ in range(3):
            sc.cloud.inflate(prm, seeds)
        ts = []
        for _ in range(20):
            t1 = time.perf_counter()
            sc.cloud.inflate(prm, seeds)
            ts.append(1e3 * (time.perf_counter() - t1))
        out["corridor_probe"] = {"what": "pct_inflate_batch, 200 seeds (seed 4), search_margin 0.25, max_radius 1.5, host buffers",
                                 "ms_per_pass_median": float(np.median(ts)), "ms_per_pass_p99": float(np.percentile(ts, 99))}

    if a.stream_probe and world == 1 and sc.cloud.has_grid:
        # radius count through the index (kd_nearest_range + kd_res_size semantics, d2 <= r*r), same cloud and query batch
        r1 = torch.full((Q,), 1.0, dtype=torch.float32, device=sc.device)
        ts = []
        for k in range(6):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cnt = sc.radius_count(q, r1)
            torch.cuda.synchronize()
            if k:
                ts.append(time.perf_counter() - t1)
        out["radius_count_probe"] = {"what": f"pct_radius_count_batch_dev, {Q} queries, r = 1.0, cell-pruned cooperative kernel",
                                     "ms_per_batch": 1e3 * float(np.median(ts)), "queries_per_s": Q / float(np.median(ts)),
                                     "mean_count": float(cnt.double().mean().item())}
        del r1, cnt

    if a.stream_probe and world == 1:
        # config C2: 1 M uniform points (seed 1), 4096 queries (seed 2), host buffers in and out (PCIe + launch latency included)
        p2 = synth.uniform_points(1, 1_000_000, 0.0, 100.0)
        q2 = synth.uniform_points(2, 4096, 0.0, 100.0)
        with E.Cloud(len(p2)) as c2:
            c2.set_input(p2)
            def med(fn, n=7):
                ts = []
                for k in range(n):
                    t1 = time.perf_counter()
                    fn()
                    if k:
                        ts.append(1e3 * (time.perf_counter() - t1))
                return float(np.median(ts))
            brute_ms = med(lambda: c2.nn(q2, E.ALGO_STREAM))
            build_ms = med(lambda: c2.build_grid(), 4)
            grid_ms = med(lambda: c2.nn(q2, E.ALGO_GRID))
        out["c2_probe"] = {"what": "C2: 1,000,000 uniform points, 4096 NN queries, host buffers (pct_nn_batch_algo)",
                           "brute_force_ms": brute_ms, "pair_evals_per_s": 4096 * 1e6 / (brute_ms * 1e-3),
                           "index_build_ms": build_ms, "indexed_ms": grid_ms, "indexed_queries_per_s": 4096 / (grid_ms * 1e-3)}

    if a.replan_probe and world == 1:
        out["replan_probe"] = replan_probe(E, synth)
        # corridor generation per replan (config C1: seed-6 pillar map seen from the start pose, clean_demo.launch constants,
        # fixed iteration counts 1500 / 400 / 200): safe-region RRT* on the engine, speculative batches of 64 samples, one fused launch per batch
        from pointcloudtraj_amd import corridor, scenarios
        cloud1 = scenarios.sensed_cloud(12.0)
        scenarios.timed_scenario(corridor.SafeRegionRrtStar(80000), cloud1)          # warm-up (first launches, allocations)
        out["corridor_replan_probe"] = dict(scenarios.timed_scenario(corridor.SafeRegionRrtStar(80000), cloud1),
                                            what="C1 corridor scenario: setInput + SafeRegionExpansion(1500) + Refine(400) + new frame + Evaluate + Refine(200)",
                                            cloud_points=int(len(cloud1)))

    if a.replan_probe and world == 1:
        # ingest stage in front of the cloud: voxel de-duplication of the 10 M-point cloud at res 0.25
        from pointcloudtraj_amd import voxel
        d_pts = torch.from_numpy(local_pts).to(sc.device)
        vm = voxel.VoxelMap(0.25, len(local_pts))
        ms = []
        for _ in range(4):
            vm.clear()
            n_vox = vm.add_device(d_pts.data_ptr(), len(local_pts), 12)
            ms.append(vm.last_ms())
        t1 = time.perf_counter()
        again = vm.add_device(d_pts.data_ptr(), len(local_pts), 12)      # second pass: every point hits an existing voxel
        out["ingest_probe"] = {"what": "pct_voxel_map_add_dev: 10 M fp32 points resident in HBM -> first-seen voxel cloud, res 0.25",
                               "points": int(len(local_pts)), "voxels": int(n_vox), "kernels_ms": float(np.median(ms[1:])),
                               "points_per_s": len(local_pts) / (float(np.median(ms[1:])) * 1e-3),
                               "all_duplicates_pass_ms": vm.last_ms(), "all_duplicates_new_voxels": int(again)}
        vm.close()
        del d_pts

    if a.replan_probe and world == 1:
        # config C4 on ONE card: the whole 100 M-point cloud (seed 6, [0,200)^3) resident, Q = 4096 (seed 7), host buffers
        p4 = synth.uniform_points(6, 100_000_000, 0.0, 200.0)
        q4 = synth.uniform_points(7, 4096, 0.0, 200.0)
        with E.Cloud(len(p4)) as c4:
            t1 = time.perf_counter()
            c4.set_input(p4)
            t2 = time.perf_counter()
            c4.build_grid()
            E.sync()
            t3 = time.perf_counter()
            c4.nn(q4, E.ALGO_GRID)
            ts = []
            for _ in range(5):
                t4 = time.perf_counter()
                i4, d4 = c4.nn(q4, E.ALGO_GRID)
                ts.append(1e3 * (time.perf_counter() - t4))
            t5 = time.perf_counter()
            ib, db = c4.nn(q4[:512], E.ALGO_STREAM)
            t6 = time.perf_counter()
        out["c4_probe"] = {"what": "C4 on one card: 100,000,000 uniform points resident (1.2 GB SoA + 1.6 GB cell-sorted), 4096 NN queries, host buffers",
                           "upload_ms": 1e3 * (t2 - t1), "index_build_ms": 1e3 * (t3 - t2), "indexed_batch_ms": float(np.median(ts)),
                           "brute_force_512_queries_ms": 1e3 * (t6 - t5), "brute_force_pair_evals_per_s": 512 * 1e8 / (t6 - t5),
                           "indexed_equals_brute_force": bool(np.array_equal(ib, i4[:512]) and np.array_equal(db, d4[:512]))}
        del p4

    if a.cpu_queries > 0 and world == 1:
        ncpu = a.cpu_points or a.points
        base, cpu_idx, cq = cpu_baseline(lambda: local_pts[:ncpu], ncpu, q_host, min(a.cpu_queries, Q))
        out["cpu_baseline"] = base
        if ncpu == a.points:     # same cloud: the GPU answers must equal the host kd-tree's (parity in the bench run itself)
            gi = idx[:len(cq)].cpu().numpy()
            out["cpu_baseline"]["gpu_matches_cpu_indices"] = bool(np.array_equal(gi, cpu_idx))
        if "corridor_replan_probe" in out:   # the same corridor scenario on the CPU restatement (oracle/rrt_port.c + kd-tree port), one core
            from oracle import oracle as O
            from pointcloudtraj_amd import scenarios
            cpu_cor = scenarios.timed_scenario(O.PortCorridor(), scenarios.sensed_cloud(12.0))
            out["cpu_baseline"]["corridor_replan_ms"] = cpu_cor["total_ms"]
            out["cpu_baseline"]["corridor_phases_ms"] = {k: v for k, v in cpu_cor.items() if k.endswith("_ms")}
            if "ingest_probe" in out:            # the sequential container restated (oracle/voxel_port.c) on the first 2 M points
                om = O.PortVoxelMap(0.25)
                t1 = time.perf_counter()
                om.add(local_pts[:2_000_000])
                out["cpu_baseline"]["voxel_ingest_points_per_s"] = 2_000_000 / (time.perf_counter() - t1)
            out["cpu_baseline"]["corridor_same_path_as_gpu"] = bool(cpu_cor["status"] == out["corridor_replan_probe"]["status"]
                                                                     and cpu_cor["path_len"] == out["corridor_replan_probe"]["path_len"])
    print(json.dumps(out))
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
