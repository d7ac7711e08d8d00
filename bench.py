#!/usr/bin/env python3
"""bench.py -- NN queries/s of the obstacle-cloud engine on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic queries: the batched 1-NN (pct_nn_batch_dev,
include/pct_engine.h) of Q = 1,048,576 queries against the cloud resident in HBM, cell-pruned kernel (device-side query
binning + nn_grid_coop_kernel).  FOUR distinct query batches (seeds 5, 1005, 2005, 3005) are resident and step k answers
batch k mod 4, so a step is never a replay of the previous one.  W warmup steps, then exactly K timed steps (barrier +
synchronize on both sides); there is no other untimed repetition of the step -- the secondary probes of the line (streaming
kernel, brute force, radius count, index build: GPU work the program does anyway) run BEFORE the warmup steps instead of after
the timed region, so the card is not measured straight out of the idle state the host-side input generation leaves it in.

N = 1 (config C3-throughput of SURVEY.md section 8(d), the configuration the metric is quoted on): 10,000,000 uniform points in
[0,100)^3 (seed 3).
N > 1 (config C4, STRONG scaling): ONE cloud of 100,000,000 uniform points in [0,200)^3 (seed 6); rank r owns the contiguous
index range [r*1e8/N, (r+1)*1e8/N) in its own HBM; the query batch (seed 5, scaled to the box) is replicated; every rank
answers it on its shard and ONE exchange step (RCCL all_reduce(min) on fp64 d2, then all_reduce(min) on the matching global
indices) merges them.  `value` = merged answers per second = Q * steps / elapsed (NOT multiplied by the rank count);
`config.query_shard_evaluations_per_s` = N * value is the secondary figure.  The line also carries config C4's own batch
(Q = 4096, seed 7) through the brute-force streaming kernels (the form whose per-rank work shrinks with the shard) and through
the index, and -- on rank 0's card -- the same 1 M-query batch against the WHOLE 100 M-point cloud on one GPU, so the strong-
scaling ratio can be read from one line.

`python bench.py --gpus N` started without RANK / WORLD_SIZE in the environment launches its N ranks itself (child processes,
one per GPU, started before the parent touches the GPU; a card count below N turns the run into a gloo rehearsal with several
ranks per card).  Under `python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.

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


def measured_counters(kernel: str, pattern: str = "*_bench_pmc.json", threads: int = 0):
    """mean per-launch PMC counters of `kernel` from the newest committed summary matching `pattern` (rocprofv3 --pmc passes of this
    same command, scripts/prof_bench.sh; one entry per kernel and grid size) -- ({counter: value}, file) or ({}, None).
    threads: the launch's grid size in threads (8 x the batch for the cell-pruned kernels); 0 = the largest grid on file."""
    import glob
    import re
    best, src, best_grid = {}, None, -1
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        found = None
        for k, v in d.items():
            if kernel not in k or "kernel<true" in k:                # <true, ...> = the instrumented twin
                continue
            m = re.search(r"@grid=(\d+)", k)
            g = int(m.group(1)) if m else 0
            if (threads and g == threads) or (not threads and (found is None or g > found[0])):      # a named launch shape: that shape or nothing
                found = (g, v)
        if found:
            best = {c: (x["mean_per_launch"] if isinstance(x, dict) else x) for c, x in found[1].items()}
            src, best_grid = "profiles/" + os.path.basename(f), found[0]
    return best, src


def bound_from_counters(cnt, kernel_ms=None):
    """what limits the kernel by the committed counters: 'hbm' when the fabric traffic it causes (2 x FETCH_SIZE + WRITE_SIZE) moves at
    half the HBM peak or more during the kernel's time -- the waves then wait on bandwidth; 'valu' when the vector ALUs issue during
    60 % or more of the kernel's cycles (a wave64 instruction occupies its SIMD's 16 lanes for 4 cycles: busy = 4 x SQ_ACTIVE_INST_VALU /
    (1024 SIMDs x cycles), cycles = GRBM_GUI_ACTIVE / 8 XCDs) -- the waves' own waiting (SQ_WAIT_ANY) is then hidden behind the other
    waves of the SIMD; 'latency' when neither holds and the waves wait more than half of their cycles: dependent round trips.
    (None, None) without counters."""
    if not cnt or "SQ_WAVE_CYCLES" not in cnt:
        return None, None
    wait = cnt.get("SQ_WAIT_ANY", 0.0) / max(cnt["SQ_WAVE_CYCLES"], 1.0)
    traffic = cnt.get("derived_hbm_traffic_bytes_per_launch")
    tfrac = (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and kernel_ms else None
    cycles = cnt.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    valu_busy = (4.0 * cnt.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024.0 * cycles)) if cycles > 0 else None
    detail = {"wait_fraction_of_wave_cycles": wait, "traffic_fraction_of_hbm_peak": tfrac, "l2_hit_rate": cnt.get("derived_l2_hit_rate"),
              "valu_instructions_per_wave": (cnt.get("SQ_INSTS_VALU", 0.0) / cnt["SQ_WAVES"]) if cnt.get("SQ_WAVES") else None,
              "valu_issue_busy_fraction": valu_busy}
    if tfrac is not None and tfrac >= 0.5:
        return "hbm", detail
    if valu_busy is not None and valu_busy >= 0.6:
        return "valu", detail
    return ("latency" if wait > 0.5 else "valu"), detail


def clustered_probe(E, synth, torch, device, Q, steps=10):
    """"all clouds are also run in a clustered variant (points on 0.1-grid pillar surfaces a la map_generator)".
    The reference's own world (seed 6, 182,332 points) and the same generator on a 7.4 times wider square (10.2 M points), uniform
    queries over the bounding box, two distinct batches alternating.  Such clouds carry the bounding-box pyramid (pyramid.hpp); the
    small one is also run with the plain shell walk for comparison."""
    out = {}
    for name, make in (("pillar_map_seed6", synth.pillar_map), ("pillar_map_x7.4_10M", lambda: synth.pillar_map_scaled(7.4))):
        pts = make()
        lo, hi = pts.min(0), pts.max(0)
        qs = [torch.from_numpy((lo + synth.uniform01_f32(77 + 1000 * b, 3 * Q).reshape(Q, 3) * (hi - lo)).astype(np.float32)).to(device) for b in range(2)]
        oi = torch.empty(Q, dtype=torch.int32, device=device)
        od = torch.empty(Q, dtype=torch.float64, device=device)
        cs = torch.cuda.current_stream().cuda_stream
        legs = {}
        with E.Cloud(len(pts)) as c:
            c.set_input(pts)
            c.reserve_queries(Q)
            for mode in (("pyramid", None),) + ((("shell_walk", "0"),) if len(pts) < 1_000_000 else ()):
                if mode[1] is not None:
                    os.environ["PCT_PYRAMID"] = mode[1]
                c.build_grid()                                    # a sensor cloud is rebuilt every frame (corridor_finder.cpp:93-99): the timed
                t0 = time.perf_counter()                           # build is the steady-state one (a sparse cloud's second build halves the cells)
                c.build_grid()
                E.sync()
                t_build = 1e3 * (time.perf_counter() - t0)
                os.environ.pop("PCT_PYRAMID", None)
                for k in range(3):
                    c.nn_device(qs[k % 2].data_ptr(), Q, oi.data_ptr(), od.data_ptr(), cs, E.ALGO_GRID)
                torch.cuda.synchronize()
                c.set_timing_stride(1)
                t0 = time.perf_counter()
                for k in range(steps):
                    c.nn_device(qs[k % 2].data_ptr(), Q, oi.data_ptr(), od.data_ptr(), cs, E.ALGO_GRID)
                torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - t0) / steps
                km = float(np.mean(c.kernel_ms_history(steps)))
                c.set_work_counters(True)
                w = np.zeros(3)
                for b in range(2):
                    c.nn_device(qs[b].data_ptr(), Q, oi.data_ptr(), od.data_ptr(), cs, E.ALGO_GRID)
                    torch.cuda.synchronize()
                    w += np.asarray(c.last_work_ex(), np.float64) / 2
                c.set_work_counters(False)
                # parity inside the run: a slice of the batch against the all-fp64 streaming kernel
                bi, bd = c.nn(qs[1][:2048].cpu().numpy(), E.ALGO_STREAM)
                same = bool(np.array_equal(bi, oi[:2048].cpu().numpy().view(np.uint32)) and np.array_equal(bd, od[:2048].cpu().numpy()))
                alg = 12 * w[0] + 8 * w[1] + 192 * w[2] + 24 * Q
                pi = c.pyramid_info()
                cntp, srcp = measured_counters("nn_grid_pyr_kernel", "*_pyr_pmc.json", threads=8 * Q) if (pi["levels"] and len(pts) > 1_000_000) else ({}, None)
                boundp, detailp = bound_from_counters(cntp, km)      # scripts/probe_pyr.py pillar10m under rocprofv3 (profiles/r03_pyr_*)
                legs[mode[0]] = {"index_build_ms": t_build, "ms_per_step": ms, "queries_per_s": Q / (ms * 1e-3), "kernel_ms": km,
                                 "kernel": "nn_grid_pyr_kernel (+ nn_grid_pyr_todo_kernel)" if pi["levels"] else "nn_grid_coop_kernel",
                                 "points_per_query": w[0] / Q, "cell_runs_per_query": w[1] / Q, "pyramid_node_visits_per_query": w[2] / Q,
                                 "algorithmic_bytes": int(alg), "algorithmic_bytes_rule": "12 B x points scanned + 8 B x cell runs + 192 B x node visits (8 boxes of 6 floats) + 24 B x Q",
                                 "achieved_GBs": alg / (km * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (km * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "bound": boundp, "bound_evidence": detailp, "traffic": cntp.get("derived_hbm_traffic_bytes_per_launch"), "traffic_source": srcp,
                                 "pyramid_levels": pi["levels"], "empty_cell_fraction": pi["empty_fraction"], "grid": c.grid_info(),
                                 "equals_streaming_kernel_on_2048_queries": same}
        out[name] = {"points": int(len(pts)), "queries_per_step": Q, **legs}
        del qs, oi, od
    return out


def c1_stated_probe(E, oracle_mod):
    """Config C1 exactly as SURVEY 8(d) states it: the seed-6 map cropped to the 9,383 points within 5 m of the start pose,
    clean_demo.launch constants, corridor RNG seed 0, 2,000 SafeRegionExpansion iterations -- corridor-generation ms on the engine
    (speculative batches, fused launches) and on the CPU restatement over the kd-tree port, same corridor required."""
    from pointcloudtraj_amd import corridor, scenarios, synth
    p = scenarios.PARAMS
    cloud = synth.crop_ball(synth.pillar_map(), scenarios.START, 5.0)

    def run(f):
        f.setParam(p["safety_margin"], p["search_margin"], p["max_radius"], p["sensing_range"])
        t0 = time.perf_counter()
        f.setInput(cloud)
        t1 = time.perf_counter()
        f.reset()
        f.setPt(scenarios.START, scenarios.GOAL, *scenarios.BOUNDS, p["sensing_range"], p["max_samples"], p["sample_portion"], p["goal_portion"])
        f.SafeRegionExpansion(2000)
        t2 = time.perf_counter()
        return 1e3 * (t1 - t0), 1e3 * (t2 - t1), f.getPath(), f.status()

    if oracle_mod is None:                                          # the engine's side (run while the card is busy with the other probes)
        run(corridor.SafeRegionRrtStar(20000))                      # warm-up: first launches, allocations
        gs = [run(corridor.SafeRegionRrtStar(20000)) for _ in range(5)]
        g = sorted(gs, key=lambda r: r[1])[len(gs) // 2]
        return {"what": "C1 as stated: 9,383-point crop (5 m around the start) of the seed-6 map, clean_demo.launch constants, RNG seed 0, SafeRegionExpansion(2000 iterations); median of 5 runs",
                "cloud_points": int(len(cloud)), "set_input_ms": g[0], "expansion_2000_ms": g[1], "status": g[3], "path_len": int(len(g[2][0])),
                "_path": g[2]}
    c = run(oracle_mod.PortCorridor())                              # the CPU restatement's side
    return {"set_input_ms": c[0], "expansion_2000_ms": c[1], "cores": 1, "_path": c[2], "status": c[3]}


def replan_probe(E, synth, ticks=200, clustered=False):
    """Config C5: rolling window of 5,000,000 points fed 50,000 per sensor frame (uniform in a 60 m
    cube around a drone moving +0.1 m per frame along x, seed 8), oldest frame evicted; the cloud keeps the rolling-map index
    (pct_cloud_ring_index: appends update it in place).  Per tick, through the host-buffer entry points (PCIe + launch
    latency included):
      ingest   pct_cloud_append_aos of the new frame (H2D + evict + file)
      replan   pct_plan_replan_run: ONE captured hipGraph = inflation of 64 corridor nodes (SafeRegionEvaluate's re-check,
               corridor_finder.cpp:829-835) + the sampled Bezier check, 3 segments of order 6, dt 0.02 s over a 2.0 s horizon
               = 99 samples (checkSafeTrajectory, sim_planning_demo.cpp:729-781) + the 21 control points
    Reported: p50 / p99 milliseconds per tick and per part over `ticks` ticks; the same queries through the un-captured
    brute-force entry points on an un-indexed copy of the window (round 1's path) for a few ticks beside it."""
    import gc
    import numpy as np
    from pointcloudtraj_amd import scenarios as S
    window, frame = S.C5_WINDOW, S.C5_FRAME
    P = S.C5_PARAMS
    c5_frame = S.c5_frame_clustered if clustered else S.c5_frame      # clustered: points on 0.1-lattice pillar faces, re-sensed frame after frame
    cloud = E.Cloud(window)
    cloud.ring_index()
    nfill = window // frame
    for k in range(nfill - 2):
        cloud.append(c5_frame(k))
    plan = E.ReplanPlan(cloud, S.C5_NODES, 128, S.C5_SEGMENTS)
    t_ing, t_rep, t_lib = [], [], []
    first_hits = 0
    gc.collect()
    gc.disable()                                   # a generation-2 collection of the interpreter (40 ms) used to land in one tick
    for k in range(nfill - 2, nfill + ticks):      # two untimed warm-up ticks (first launches)
        new_frame = c5_frame(k)                  # the sensor's output: produced outside the timed region
        start, nodes, coef, T, od = S.c5_tick_queries(k)
        prm = E.inflate_params(start, P["sample_range"], P["search_margin"], P["max_radius"])
        t0 = time.perf_counter()
        cloud.append(new_frame)
        t1 = time.perf_counter()
        r = plan.run(prm, nodes, coef, T, od, 0.0, 2.0, 0.02, want_nn=False, copy=False)
        t2 = time.perf_counter()
        if k >= nfill:
            t_ing.append(1e3 * (t1 - t0)); t_rep.append(1e3 * (t2 - t1)); t_lib.append(plan.last_run_us())
            first_hits += int(r["first_hit_sample"] >= 0)
    # the same ticks with zero-copy ingest: the producer (here: a copy outside the timed region, standing for the sensor driver or
    # the PointCloud2 deserialiser) writes the frame into the cloud's host-mapped staging buffer, pct_cloud_append_frame files it
    t_zc, t_zc_rep = [], []
    buf = cloud.frame_buffer(frame)
    for k in range(nfill + ticks, nfill + ticks + max(20, ticks // 4)):
        buf[:] = c5_frame(k)
        start, nodes, coef, T, od = S.c5_tick_queries(k)
        prm = E.inflate_params(start, P["sample_range"], P["search_margin"], P["max_radius"])
        t0 = time.perf_counter()
        cloud.append_frame(frame)
        t1 = time.perf_counter()
        plan.run(prm, nodes, coef, T, od, 0.0, 2.0, 0.02, want_nn=False, copy=False)
        t2 = time.perf_counter()
        t_zc.append(1e3 * (t1 - t0)); t_zc_rep.append(1e3 * (t2 - t1))
    ticks_total = nfill + ticks + max(20, ticks // 4)
    gc.enable()
    info = cloud.ring_info()
    plan.close()
    tot = np.asarray(t_ing) + np.asarray(t_rep)
    pct = lambda a, q: float(np.percentile(a, q))
    out = {"what": ("C5, CLUSTERED variant (points on 0.1-lattice pillar faces, the same lattice points re-sensed every frame): " if clustered else "C5: ") +
                   "5,000,000-point rolling cloud with the rolling-map index, +50,000 points per tick; per tick ONE captured hipGraph = "
                   "64 corridor-node inflations + 99-sample Bezier check + 21 control points; host buffers",
           "ticks": ticks, "ms_per_tick_p50": pct(tot, 50), "ms_per_tick_p99": pct(tot, 99),
           "ingest_ms_p50": pct(t_ing, 50), "ingest_ms_p99": pct(t_ing, 99), "replan_graph_ms_p50": pct(t_rep, 50), "replan_graph_ms_p99": pct(t_rep, 99),
           "worst_tick": {"index": int(np.argmax(tot)), "ingest_ms": float(np.asarray(t_ing)[np.argmax(tot)]), "replan_graph_ms": float(np.asarray(t_rep)[np.argmax(tot)])},
           "replan_inside_library_us_p50": dict(zip(("fill", "graph_launch", "wait", "read_out"), [float(v) for v in np.percentile(np.asarray(t_lib), 50, axis=0)])),
           "zero_copy_ingest": {"what": "pct_cloud_frame_buffer + pct_cloud_append_frame: the frame is produced in the cloud's host-mapped staging "
                                        "buffer, the insert kernel reads it over the bus (no host-side copy inside the tick)",
                                "ingest_ms_p50": pct(t_zc, 50), "ingest_ms_p99": pct(t_zc, 99),
                                "ms_per_tick_p50": pct(np.asarray(t_zc) + np.asarray(t_zc_rep), 50), "ms_per_tick_p99": pct(np.asarray(t_zc) + np.asarray(t_zc_rep), 99)},
           "ticks_with_a_colliding_sample": first_hits, "ring_index": info, "budget_ms_at_20Hz": 50.0}
    # round 1's path for comparison: the same window un-indexed, brute-force kernels, three separate calls
    cloud.ring_drop()
    t_old = []
    for k in range(ticks_total, ticks_total + 12):
        new_frame = c5_frame(k)
        start, nodes, coef, T, od = S.c5_tick_queries(k)
        prm = E.inflate_params(start, P["sample_range"], P["search_margin"], P["max_radius"])
        t0 = time.perf_counter()
        cloud.append(new_frame)
        cloud.inflate(prm, nodes)
        cloud.bezier_check(prm, coef, T, od, 0.0, 2.0, cap=128)
        cloud.ctrl_points_check(prm, coef, T, od)
        if k >= ticks_total + 2:
            t_old.append(1e3 * (time.perf_counter() - t0))
    out["unindexed_brute_force_ms_per_tick_p50"] = pct(t_old, 50)
    cloud.close()
    return out


def c4_probe(E, synth, torch, device, Q):
    """Config C4 on ONE card: the whole 100 M-point cloud (seed 6, [0,200)^3) resident -- 1.2 GB SoA + 1.6 GB cell-sorted, far beyond
    the 256 MiB Infinity Cache, so these are the DRAM-resident figures of both kernel families (scripts/probe_c4.py runs it alone,
    under rocprofv3: profiles/r02_c4_*)."""
    out = {}
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
        # (a) streaming kernel, one pass over 1.2 GB of SoA per launch
        cs4 = torch.cuda.current_stream().cuda_stream
        c4.reserve_queries(Q)
        qd = torch.from_numpy(synth.uniform_points(5, Q, 0.0, 200.0)).to(device)
        oi = torch.empty(Q, dtype=torch.int32, device=device)
        od = torch.empty(Q, dtype=torch.float64, device=device)
        sp = []
        for qn in (1, 2, 4):
            ms = []
            for k in range(8):
                c4.nn_device(qd.data_ptr(), qn, oi.data_ptr(), od.data_ptr(), cs4, E.ALGO_STREAM)
                if k >= 2:
                    ms.append(c4.last_kernel_ms())
            m = float(np.median(ms))
            sb = 12 * len(p4) + 24 * qn
            sp.append({"queries": qn, "kernel_ms": m, "algorithmic_bytes": sb, "achieved_GBs": sb / (m * 1e-3) / 1e9,
                       "frac_of_hbm_peak": sb / (m * 1e-3) / 1e9 / HBM_PEAK_GBS})
        # (b) the 1 M-query throughput batch through the index
        for _ in range(2):
            c4.nn_device(qd.data_ptr(), Q, oi.data_ptr(), od.data_ptr(), cs4, E.ALGO_GRID)
        torch.cuda.synchronize()
        t7 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            c4.nn_device(qd.data_ptr(), Q, oi.data_ptr(), od.data_ptr(), cs4, E.ALGO_GRID)
        torch.cuda.synchronize()
        step4 = 1e3 * (time.perf_counter() - t7) / reps
        k4 = float(np.mean(c4.kernel_ms_history(reps)))
        c4.set_work_counters(True)
        c4.nn_device(qd.data_ptr(), Q, oi.data_ptr(), od.data_ptr(), cs4, E.ALGO_GRID)
        torch.cuda.synchronize()
        ps4, runs4 = c4.last_work()
        c4.set_work_counters(False)
        alg4 = 12 * ps4 + 8 * runs4 + 24 * Q
        cnt4, src4 = measured_counters("nn_grid_coop_kernel", "*_c4_pmc.json", threads=8 * Q)       # scripts/probe_c4.py under rocprofv3
        tr4 = cnt4.get("derived_hbm_traffic_bytes_per_launch")
        bound4, detail4 = bound_from_counters(cnt4, k4)
        del qd, oi, od
        # (c) the same cloud with a batch dense enough for neighbouring queries to share cache lines: 8 Q queries (one per two cells)
        Qd = 8 * Q
        c4.reserve_queries(Qd)
        qd = torch.from_numpy(synth.uniform_points(8, Qd, 0.0, 200.0)).to(device)
        oi = torch.empty(Qd, dtype=torch.int32, device=device)
        od = torch.empty(Qd, dtype=torch.float64, device=device)
        for _ in range(2):
            c4.nn_device(qd.data_ptr(), Qd, oi.data_ptr(), od.data_ptr(), cs4, E.ALGO_GRID)
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        for _ in range(5):
            c4.nn_device(qd.data_ptr(), Qd, oi.data_ptr(), od.data_ptr(), cs4, E.ALGO_GRID)
        torch.cuda.synchronize()
        stepd = 1e3 * (time.perf_counter() - t8) / 5
        kd = float(np.mean(c4.kernel_ms_history(5)))
        c4.set_work_counters(True)
        c4.nn_device(qd.data_ptr(), Qd, oi.data_ptr(), od.data_ptr(), cs4, E.ALGO_GRID)
        torch.cuda.synchronize()
        psd, runsd = c4.last_work()
        c4.set_work_counters(False)
        algd = 12 * psd + 8 * runsd + 24 * Qd
        cntd, srcd = measured_counters("nn_grid_coop_kernel", "*_c4_pmc.json", threads=8 * Qd)
        trd = cntd.get("derived_hbm_traffic_bytes_per_launch")
        boundd, detaild = bound_from_counters(cntd, kd)
        dense = {"queries": Qd, "ms_per_step": stepd, "queries_per_s": Qd / (stepd * 1e-3), "kernel": "nn_grid_coop_kernel", "kernel_ms": kd,
                 "algorithmic_bytes": int(algd), "achieved_GBs": algd / (kd * 1e-3) / 1e9, "frac_of_hbm_peak": algd / (kd * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "points_scanned": int(psd), "cell_runs": int(runsd), "bound": boundd or "hbm", "bound_evidence": detaild, "traffic": trd,
                 "traffic_source": srcd, "frac_of_measured_traffic": (trd / (kd * 1e-3) / 1e9 / HBM_PEAK_GBS) if trd else None,
                 "note": "the DRAM-resident cloud with one query per two cells: neighbouring queries of the sorted batch share the lines of their "
                         "runs, so the fabric traffic falls towards the record bytes"}
        del qd, oi, od
    out["c4_probe"] = {"what": "C4 on one card: 100,000,000 uniform points resident (1.2 GB SoA + 1.6 GB cell-sorted, beyond the 256 MiB Infinity Cache)",
                       "upload_ms": 1e3 * (t2 - t1), "index_build_ms": 1e3 * (t3 - t2), "indexed_4096_queries_ms_host_buffers": float(np.median(ts)),
                       "brute_force_512_queries_ms": 1e3 * (t6 - t5), "brute_force_pair_evals_per_s": 512 * 1e8 / (t6 - t5),
                       "indexed_equals_brute_force": bool(np.array_equal(ib, i4[:512]) and np.array_equal(db, d4[:512])),
                       "stream_kernel": {"kernel": "nn_stream_kernel<QT> (all-fp64, one pass over the SoA cloud)", "points": sp},
                       "grid_throughput": {"queries": Q, "ms_per_step": step4, "queries_per_s": Q / (step4 * 1e-3), "kernel": "nn_grid_coop_kernel",
                                           "kernel_ms": k4, "algorithmic_bytes": int(alg4), "achieved_GBs": alg4 / (k4 * 1e-3) / 1e9,
                                           "frac_of_hbm_peak": alg4 / (k4 * 1e-3) / 1e9 / HBM_PEAK_GBS, "points_scanned": int(ps4), "cell_runs": int(runs4),
                                           "bound": bound4 or "hbm", "bound_evidence": detail4, "traffic": tr4, "traffic_source": src4,
                                           "frac_of_measured_traffic": (tr4 / (k4 * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr4 else None,
                                           "note": "DRAM-resident: one query touches ~10 cache lines of 16-byte records (4 x-runs of ~12 records, each "
                                                   "starting anywhere in a line: 1 + 192/128 lines per run) that no other query of the batch shares "
                                                   "(1 M queries over 16.7 M cells), so the fabric traffic is ~1.7x the 12-byte-rule bytes by line "
                                                   "granularity alone; DESIGN.md section 4"},
                       "grid_throughput_dense_queries": dense}
    del p4
    return out["c4_probe"]



def guarded(out, key, fn):
    """a SECONDARY probe must not cost the line: its failure is reported under its own key, the headline and the other probes stand"""
    try:
        out[key] = fn()
    except Exception as ex:      # noqa: BLE001
        out[key] = {"error": f"{type(ex).__name__}: {ex}"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preheat-steps", type=int, default=0,
                    help="extra untimed passes of the step before the W warmup steps (default none; reported in the line).  The same count on "
                         "every rank (each pass holds a collective at N > 1)")
    ap.add_argument("--points", type=int, default=10_000_000, help="points of the N = 1 cloud (config C3)")
    ap.add_argument("--points-total", type=int, default=100_000_000, help="points of the N > 1 cloud (config C4), split over the ranks")
    ap.add_argument("--queries", type=int, default=1 << 20)
    ap.add_argument("--algo", choices=["grid", "stream"], default="grid")
    ap.add_argument("--cell", type=float, default=0.0, help="grid cell size (<=0: automatic)")
    ap.add_argument("--cpu-queries", type=int, default=20000, help="queries timed on the host kd-tree (0 = skip)")
    ap.add_argument("--cpu-points", type=int, default=0, help="points in the host kd-tree (0 = same as --points)")
    ap.add_argument("--stream-probe", type=int, default=1, help="also time the streaming / brute-force / corridor probes (0 = skip)")
    ap.add_argument("--replan-probe", type=int, default=1, help="config C5: rolling 5M-point cloud, 20 Hz replan ticks (0 = skip)")
    ap.add_argument("--clustered-probe", type=int, default=1, help="N = 1: the clustered (pillar-surface) variants of the clouds, SURVEY 8(d) (0 = skip)")
    ap.add_argument("--c4-probe", type=int, default=1, help="N = 1: also run the 100 M-point cloud (config C4) on this one card (0 = skip)")
    ap.add_argument("--probes-directly-before-warmup", type=int, default=1,
                    help="1: the instrumented (work-counting) passes run first and the secondary probes directly before the warmup steps; 0: the other way round")
    ap.add_argument("--routed", type=int, default=1, help="N > 1: the routed form through the C ABI is the headline (0 = index-range shards only)")
    ap.add_argument("--spatial-leg", type=int, default=1, help="N > 1: also time the batch with spatially routed queries (0 = skip)")
    ap.add_argument("--one-gpu-ref", type=int, default=1, help="N > 1: rank 0 also times the batch against the whole cloud on its one card (0 = skip)")
    return ap.parse_args()


def self_launch(a) -> int:
    """`bench.py --gpus N` without a launcher: start the N ranks as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment) BEFORE this process touches the GPU, pass rank 0's JSON line through, fail if any rank fails.  Fewer cards
    than ranks = rehearsal: ranks share cards round-robin and exchange over gloo (RCCL refuses two ranks on one device)."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()                 # counting devices does not initialise the GPU
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if ndev < a.gpus:
            env.setdefault("PCT_DIST_BACKEND", "gloo")
        child = os.environ.get("PCT_BENCH_CHILD") or os.path.abspath(__file__)      # test hook: a stand-in rank program (tests/test_bench_launch.py)
        procs.append(subprocess.Popen([sys.executable, child, *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # read rank 0's line while watching every rank: one that fails takes the others down with it (they would otherwise sit in a
    # collective until its timeout) -- by their exact PIDs
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out0 = buf[0] if buf else b""
    if failed or any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        return 1
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.startswith("{")]
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    print(lines[-1])
    return 0


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


def timed_batches(sc, q, algo, reps, barrier):
    """milliseconds per merged batch (sharded kernels + exchange), barrier + synchronize on both sides"""
    for _ in range(2):
        sc.nn_submit(q, algo)
    barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        sc.nn_submit(q, algo)
    barrier()
    return 1e3 * (time.perf_counter() - t0) / reps


def gpu_probes(a, E, synth, torch, sc, q, local_pts, Q):
    """N = 1 secondary figures on the headline cloud (they run BEFORE the warmup steps: see the module docstring)"""
    out = {}
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
    # (b) brute force at config C2's batch size: packed-fp32 filter with the points in registers + exact fp64 recheck
    qn = 4096
    ms = timed(lambda: sc.cloud.nn_device(q.data_ptr(), qn, sc._idx32.data_ptr(), sc._d2.data_ptr(), cs, E.ALGO_STREAM), 5)
    out["brute_force_probe"] = {"kernel": "nn_sample_bounds_kernel + brute2_prep_kernel + tile_reg_kernel<false> (expanded-form packed-fp32 filter, points in registers, exact fp64 recheck) + nn_reduce_candidates_kernel",
                                "queries": qn, "kernel_ms": ms, "queries_per_s": qn / (ms * 1e-3),
                                "pair_evals_per_s": qn * len(local_pts) / (ms * 1e-3),
                                "algorithmic_flops_per_s": 8 * qn * len(local_pts) / (ms * 1e-3)}
    # the same batch as a radius count (kd_nearest_range + kd_res_size semantics) through the brute-force family
    r4 = torch.full((qn,), 1.0, dtype=torch.float32, device=sc.device)
    cnt4 = torch.empty(qn, dtype=torch.int32, device=sc.device)
    ms = timed(lambda: sc.cloud.radius_count_device(q.data_ptr(), r4.data_ptr(), qn, cnt4.data_ptr(), cs, E.ALGO_STREAM), 5)
    out["brute_force_count_probe"] = {"kernel": "brute2_prep_count_kernel + tile_reg_kernel<true> (same filter, exact fp64 test of what may lie inside the ball)",
                                      "queries": qn, "radius": 1.0, "kernel_ms": ms, "pair_evals_per_s": qn * len(local_pts) / (ms * 1e-3),
                                      "mean_count": float(cnt4.double().mean().item())}
    del r4, cnt4
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
    if sc.cloud.has_grid:
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
    if a.algo == "grid":
        # the per-frame index build on the headline cloud (setInput's rebuild, corridor_finder.cpp:93-99): bounding box + two-level LDS
        # counting sort into cells, points already resident in HBM; wall clock around pct_cloud_build_grid (it ends synchronised)
        tb = []
        for _ in range(6):
            E.sync()
            t1 = time.perf_counter()
            sc.cloud.build_grid(a.cell)
            tb.append(1e3 * (time.perf_counter() - t1))
        bm = float(np.median(tb[1:]))
        out["index_build_probe"] = {"what": f"pct_cloud_build_grid on the {len(local_pts)}-point cloud resident in HBM (bbox + gb_hist + gb_scatter + gb_cells, self-check read back)",
                                    "ms_median": bm, "points_per_s": len(local_pts) / (bm * 1e-3)}
    return out


QUERY_SEEDS = (5, 1005, 2005, 3005)        # the step's batches: step k answers batch k mod 4


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    import torch   # before the engine: one HIP runtime per process (engine._preload_hip_runtime)
    from pointcloudtraj_amd import dist as D, engine as E, synth
    import torch.distributed as tdist

    rank, local, world = D.init_process_group_from_env()
    if world != a.gpus and world > 1:
        a.gpus = world
    # one GPU per rank; PCT_DIST_BACKEND=gloo rehearses the multi-rank path with several ranks on one card
    ncards = max(torch.cuda.device_count(), 1)
    dev = (local % ncards) if world > 1 else 0
    c4 = world > 1
    n_total = a.points_total if c4 else a.points
    side = 200.0 if c4 else 100.0
    cloud_seed = 6 if c4 else 3
    Q = a.queries
    algo = E.ALGO_GRID if a.algo == "grid" else E.ALGO_STREAM

    sc = D.ShardedCloud(n_total, rank, world, dev)
    local_pts = synth.uniform_points(cloud_seed, sc.end - sc.begin, 0.0, side, offset=sc.begin)
    sc.set_input_local(local_pts)
    t0 = time.perf_counter()
    if algo == E.ALGO_GRID:
        sc.build_grid(a.cell)
    E.sync()
    t_grid = time.perf_counter() - t0
    q_hosts = [synth.uniform_points(sd, Q, 0.0, side) for sd in QUERY_SEEDS]
    sc.reserve(Q)
    qs = [torch.from_numpy(h).to(sc.device) for h in q_hosts]
    q = qs[0]

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    # ---- N = 1: the secondary probes on the same cloud, and the instrumented (work-counting) pass of every batch ----
    pre = {}
    if a.stream_probe and world == 1 and not a.probes_directly_before_warmup:
        pre = gpu_probes(a, E, synth, torch, sc, q, local_pts, Q)
    sc.cloud.set_work_counters(True)
    pts_scanned = runs = 0.0
    for b in range(len(qs)):
        sc.nn_local(qs[b], algo)
        torch.cuda.synchronize()
        w = sc.cloud.last_work()
        pts_scanned += w[0] / len(qs)
        runs += w[1] / len(qs)
    sc.cloud.set_work_counters(False)
    if a.stream_probe and world == 1 and a.probes_directly_before_warmup:
        try:
            pre = gpu_probes(a, E, synth, torch, sc, q, local_pts, Q)
        except Exception as ex:      # noqa: BLE001  (secondary figures: the headline must still be measured and printed)
            pre = {"secondary_probes_error": f"{type(ex).__name__}: {ex}"}

    # ---- the measured region: W warmup steps, then exactly K timed steps; the dominant kernel's duration is sampled on every 4th
    # timed launch (the kernel's own begin / end timestamps, read after the closing barrier) ----
    preheat = max(a.preheat_steps, 0)                 # opt-in extra untimed steps (default none); reported in the line
    for k in range(preheat + a.warmup):
        sc.nn_submit(qs[k % len(qs)], algo)
    barrier()
    sc.cloud.set_timing_stride(4)                     # the first timed step is a sampled one
    samples0 = sc.cloud.kernel_ms_samples()
    t0 = time.perf_counter()
    for k in range(a.steps):
        # at N > 1 the exchange step of batch k runs on a side stream under the kernels of batch k+1 (dist.nn_submit);
        # the closing barrier + synchronize waits for every batch's merged answer
        sc.nn_submit(qs[k % len(qs)], algo)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=sc.device if tdist.get_backend() == "nccl" else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / a.steps
    n_samples = sc.cloud.kernel_ms_samples() - samples0
    kern_ms = sc.cloud.kernel_ms_history(max(1, min(n_samples, 64)))
    sc.cloud.set_timing_stride(1)
    k_ms = float(np.mean(kern_ms))
    # answers of batch 0 for the parity checks below (untimed)
    d2, idx, done = sc.nn_submit(q, algo)
    barrier()
    d2_keep, idx_keep = d2.clone(), idx.clone()       # the result slots are reused by the probes below
    # 12 B per point scanned (3 x fp32) + 8 B per examined cell run + 12 B per query in + 12 B out
    bytes_alg = 12 * pts_scanned + 8 * runs + 24 * Q if algo == E.ALGO_GRID else 12 * len(local_pts) * ((Q + 7) // 8) + 24 * Q
    bytes_rec = 16 * pts_scanned + 8 * runs + 24 * Q if algo == E.ALGO_GRID else bytes_alg   # what the 16-byte {x,y,z,index} records move
    achieved = bytes_alg / (k_ms * 1e-3) / 1e9

    # ---- N > 1: config C4's own batch (Q = 4096, seed 7) through both kernel families, merged ------------------------------
    c4_legs = None
    if c4:
        q4 = torch.from_numpy(synth.uniform_points(7, 4096, 0.0, side)).to(sc.device)
        brute_ms = timed_batches(sc, q4, E.ALGO_STREAM, 3, barrier)
        index_ms = timed_batches(sc, q4, E.ALGO_GRID, 10, barrier) if algo == E.ALGO_GRID else None
        c4_legs = {"what": "C4 batch: 4096 NN queries (seed 7) against the sharded 100 M-point cloud, per-shard kernels + all_reduce(min) merge",
                   "brute_force_ms_per_batch": brute_ms, "brute_force_pair_evals_per_s": 4096 * n_total / (brute_ms * 1e-3),
                   "brute_force_merged_answers_per_s": 4096 / (brute_ms * 1e-3),
                   "indexed_ms_per_batch": index_ms, "indexed_merged_answers_per_s": (4096 / (index_ms * 1e-3)) if index_ms else None}
    # ---- N > 1: the same batch with spatially routed queries (slab ownership + halo, dist.SpatialShardedCloud): each rank answers
    # only the queries of its slab, the same all_reduce(min) pair merges.  Secondary leg: a failure here must not cost the line.
    spatial = None
    if c4 and algo == E.ALGO_GRID and a.spatial_leg:
        try:
            sp = D.SpatialShardedCloud(rank, world, dev)
            t1 = time.perf_counter()
            sp.build(local_pts, sc.begin)
            t_build = time.perf_counter() - t1
            for _ in range(2):
                sd2, sidx = sp.nn(q)
            barrier()
            t1 = time.perf_counter()
            reps = max(3, a.steps // 2)
            for _ in range(reps):
                sd2, sidx = sp.nn(q)
            barrier()
            sp_ms = 1e3 * (time.perf_counter() - t1) / reps
            spatial = {"what": "the same batch, queries routed to the rank that owns their slab (halo 4 point spacings, certified answers, second round "
                               "for the rest); same all_reduce(min) exchange; un-pipelined",
                       "ms_per_step": sp_ms, "merged_answers_per_s": Q / (sp_ms * 1e-3), "redistribution_s": round(t_build, 3),
                       "slab_points_rank0": sp.slab_points, "owned_queries_rank0_per_batch": sp.stats["owned"] / max(sp.stats["batches"], 1),
                       "uncertified_rank0_per_batch": sp.stats["uncertified"] / max(sp.stats["batches"], 1),
                       "same_answers_as_index_range_shards": bool(torch.equal(sd2, d2_keep) and torch.equal(sidx.to(torch.int32), idx_keep.to(torch.int32)))}
            sp.close()
        except Exception as ex:      # noqa: BLE001  (report, do not fail the benchmark line)
            spatial = {"error": f"{type(ex).__name__}: {ex}"}
    # ---- N > 1 HEADLINE: the routed form through the C ABI (include/pct_shard.h, libpct_shard.so over RCCL): slab ownership, every
    # query answered by ONE rank, the owned answers exchanged as records.  W warmup steps, K timed steps, batches in rotation, barrier +
    # synchronize on both sides, max over ranks -- the contract's region.  (gloo rehearsal with several ranks per card: RCCL refuses
    # that, the torch.distributed form above stands in.)
    routed = None
    if c4 and algo == E.ALGO_GRID and a.routed:
        try:
            if tdist.get_backend() != "nccl":
                raise RuntimeError("rehearsal over gloo: the C-ABI exchange needs one card per rank")
            from pointcloudtraj_amd import shard as SH
            tok = torch.zeros(SH.ID_BYTES, dtype=torch.uint8, device=sc.device)
            if rank == 0:
                tok.copy_(torch.frombuffer(bytearray(SH.unique_id()), dtype=torch.uint8))
            tdist.broadcast(tok, src=0)
            shd = SH.Shard(bytes(tok.cpu().numpy().tobytes()), rank, world, dev)
            t1 = time.perf_counter()
            route = shd.route(local_pts, sc.begin, 4.0)
            barrier()
            t_route = time.perf_counter() - t1
            r_idx = torch.empty(Q, dtype=torch.int32, device=sc.device)
            r_d2 = torch.empty(Q, dtype=torch.float64, device=sc.device)
            cs_r = torch.cuda.current_stream().cuda_stream
            route.nn_device(q.data_ptr(), Q, r_idx.data_ptr(), r_d2.data_ptr(), cs_r)
            barrier()
            same_r = bool(torch.equal(r_d2, d2_keep) and torch.equal(r_idx, idx_keep.to(torch.int32)))
            for k in range(a.warmup):
                route.nn_device(qs[k % len(qs)].data_ptr(), Q, r_idx.data_ptr(), r_d2.data_ptr(), cs_r)
            barrier()
            t1 = time.perf_counter()
            for k in range(a.steps):
                route.nn_device(qs[k % len(qs)].data_ptr(), Q, r_idx.data_ptr(), r_d2.data_ptr(), cs_r)
            barrier()
            r_elapsed = time.perf_counter() - t1
            t = torch.tensor([r_elapsed], dtype=torch.float64, device=sc.device)
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            r_elapsed = float(t.item())
            # PARTITIONED batches (weak scaling): every rank brings its own batches; the queries travel to their owners, the answers back
            route.nn_partitioned_device(q.data_ptr(), Q, r_idx.data_ptr(), r_d2.data_ptr(), cs_r)        # same batch on every rank: comparable
            barrier()
            same_p = bool(torch.equal(r_d2, d2_keep) and torch.equal(r_idx, idx_keep.to(torch.int32)))
            own = [torch.from_numpy(synth.uniform_points(sd + 10007 * (rank + 1), Q, 0.0, side)).to(sc.device) for sd in QUERY_SEEDS]
            for k in range(a.warmup):
                route.nn_partitioned_device(own[k % len(own)].data_ptr(), Q, r_idx.data_ptr(), r_d2.data_ptr(), cs_r)
            barrier()
            t1 = time.perf_counter()
            for k in range(a.steps):
                route.nn_partitioned_device(own[k % len(own)].data_ptr(), Q, r_idx.data_ptr(), r_d2.data_ptr(), cs_r)
            barrier()
            p_elapsed = time.perf_counter() - t1
            t = torch.tensor([p_elapsed], dtype=torch.float64, device=sc.device)
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            p_elapsed = float(t.item())
            del own
            st_r = route.stats()
            partitioned = {"what": "partitioned batches in the C ABI (pct_shard_route_nn_partitioned_dev): every rank brings its OWN batch of Q queries per step "
                                   "(distinct seeds per rank and step), the queries travel to the rank that owns their slab and the answers travel back (two "
                                   "variable-sized all-to-alls of 16 bytes per query); the job answers world x Q queries per step (weak scaling)",
                           "elapsed_s": p_elapsed, "ms_per_step": 1e3 * p_elapsed / a.steps, "queries_per_step_whole_job": world * Q,
                           "answers_per_s_whole_job": world * Q * a.steps / p_elapsed, "same_answers_as_index_range_shards_on_a_common_batch": same_p}
            routed = {"what": "routed form in the C ABI (pct_shard_route_build / pct_shard_route_nn_dev): slabs of equal point count along the longest axis + halo "
                              "of 4 spacings, every query answered by the rank that owns its slab, owned answers exchanged as 16-byte records (grouped "
                              "ncclSend / ncclRecv), uncertified answers in a second round",
                      "elapsed_s": r_elapsed, "ms_per_step": 1e3 * r_elapsed / a.steps, "merged_answers_per_s": Q * a.steps / r_elapsed,
                      "redistribution_s": round(t_route, 3), "slab_points_rank0": st_r["slab_points"],
                      "owned_queries_rank0_per_batch": st_r["owned"] / max(st_r["batches"], 1), "uncertified_per_batch": st_r["uncertified"] / max(st_r["batches"], 1),
                      "same_answers_as_index_range_shards": same_r, "partitioned": partitioned}
            route.close()
            shd.close()
        except Exception as ex:      # noqa: BLE001  (the index-range figure then stays the headline)
            routed = {"error": f"{type(ex).__name__}: {ex}"}
    one_gpu = None
    if c4 and a.one_gpu_ref:
        # strong-scaling base in the same run: rank 0's card alone holds the WHOLE cloud and answers the same batch
        if rank == 0:
            with E.Cloud(n_total) as whole:
                first = True
                for o, blk in synth.uniform_points_chunked(cloud_seed, n_total, 0.0, side):
                    (whole.set_input if first else whole.append)(blk)
                    first = False
                whole.build_grid(a.cell)
                whole.reserve_queries(Q)
                cs0 = torch.cuda.current_stream().cuda_stream
                oi = torch.empty(Q, dtype=torch.int32, device=sc.device)
                od = torch.empty(Q, dtype=torch.float64, device=sc.device)
                for _ in range(3):
                    whole.nn_device(q.data_ptr(), Q, oi.data_ptr(), od.data_ptr(), cs0, E.ALGO_GRID)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    whole.nn_device(q.data_ptr(), Q, oi.data_ptr(), od.data_ptr(), cs0, E.ALGO_GRID)
                torch.cuda.synchronize()
                one_ms = 1e3 * (time.perf_counter() - t1) / a.steps
                same = bool(torch.equal(od, d2_keep) and torch.equal(oi, idx_keep.to(torch.int32)))
                one_gpu = {"what": "the same batch against the whole cloud resident on ONE card (rank 0), cell-pruned kernel",
                           "ms_per_step": one_ms, "answers_per_s": Q / (one_ms * 1e-3), "same_answers_as_sharded": same}
        barrier()

    if rank != 0:
        if world > 1:
            tdist.destroy_process_group()
        return

    kname = "nn_grid_coop_kernel" if algo == E.ALGO_GRID else "nn_tile_candidates_kernel"
    cnt, cnt_src = measured_counters(kname, threads=8 * Q)
    traffic = cnt.get("derived_hbm_traffic_bytes_per_launch")
    if c4:
        cnt, cnt_src, traffic = {}, None, None       # the committed PMC passes are of the N = 1 command
    bound, bound_detail = bound_from_counters(cnt, k_ms)
    value = Q / elapsed * a.steps                    # merged answers per second (never multiplied by the rank count)
    scaling_kind = "strong" if c4 else "weak"
    index_range = None
    if c4:
        index_range = {"what": "index-range shards (SURVEY 8e's partition): every rank answers the whole replicated batch on its contiguous index range, "
                               "all_reduce(min) on d2 then on the matching indices; the cost of a cell-pruned query does not shrink with the shard, so this "
                               "form stays near one GPU's rate", "ms_per_step": ms_per_step, "merged_answers_per_s": value}
        part = routed.get("partitioned") if routed and "error" not in routed else None
        if part and part["same_answers_as_index_range_shards_on_a_common_batch"]:
            value = part["answers_per_s_whole_job"]      # the headline at N > 1: every rank its own queries (weak scaling), nothing replicated
            ms_per_step = part["ms_per_step"]
            scaling_kind = "weak"
        elif routed and "error" not in routed and routed.get("same_answers_as_index_range_shards"):
            value = routed["merged_answers_per_s"]       # replicated queries, every query answered by ONE rank
            ms_per_step = routed["ms_per_step"]
        elif spatial and "error" not in spatial and spatial.get("same_answers_as_index_range_shards") and spatial["merged_answers_per_s"] > value:
            value = spatial["merged_answers_per_s"]      # rehearsal (gloo): the same routing through torch.distributed
            ms_per_step = spatial["ms_per_step"]
    out = {
        "metric": "nn_queries_per_sec_10M_point_cloud",
        "value": value,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": scaling_kind,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"C4: ONE cloud of {n_total} uniform fp32 points in [0,{side:.0f})^3 (seed 6) sharded by contiguous index range over {world} "
                         f"GPU(s) ({sc.end - sc.begin} points on rank 0), {Q} uniform NN queries per step ({len(qs)} distinct batches in rotation, seeds "
                         f"{list(QUERY_SEEDS)}) per rank; value = answers/s of the whole job in the form config.headline_form names (partitioned batches through the routed C ABI: every "
                         f"rank brings its own batch, weak scaling; the replicated-batch routed form and the index-range form -- {a.algo} kernel per shard + "
                         f"all_reduce(min) merge -- beside it in routed_c_abi / index_range_shards)"
                         if c4 else
                         f"C3-throughput: {a.points} uniform fp32 points in [0,{side:.1f})^3 (seed 3), {Q} uniform NN queries per step "
                         f"({len(qs)} distinct batches in rotation, seeds {list(QUERY_SEEDS)}), {a.algo} kernel, inputs resident in HBM"),
            "points_per_gpu": sc.end - sc.begin, "total_points": n_total, "queries_per_step": Q, "algo": a.algo,
            "query_batches_in_rotation": len(qs), "extra_untimed_steps_before_warmup": preheat,
            "gpu_work_before_the_warmup_steps": ("one instrumented pass per query batch, then the line's secondary probes on the same cloud (streaming kernel, "
                                                 "brute force, radius count, C2, index rebuild)" if pre else "one instrumented pass per query batch"),
            "query_shard_evaluations_per_s": world * value,
            "parallelism": (f"cloud sharded by contiguous index range over {world} GPU(s), queries replicated, all_reduce(min) merge"
                            if world > 1 else "single GPU"),
            "backend": tdist.get_backend() if world > 1 else None,
            "cards": ncards,
            "grid": sc.cloud.grid_info() if sc.cloud.has_grid else None,
            "grid_build_s": round(t_grid, 4),
        },
        "roofline": {
            "bound": bound or "hbm", "roofline": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": cnt_src,
            "frac_of_measured_traffic": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "bound_evidence": bound_detail,
            "kernel": kname,
            "kernel_ms": k_ms, "kernel_launches_timed": len(kern_ms), "other_kernels_and_launch_gaps_ms": max(ms_per_step - k_ms, 0.0),
            "algorithmic_bytes": int(bytes_alg), "algorithmic_bytes_rule": "SURVEY 8(d): 12 B x points scanned + 8 B x cell runs + 24 B x Q (mean over the batches in rotation)",
            "record_bytes": int(bytes_rec), "achieved_with_16B_records": bytes_rec / (k_ms * 1e-3) / 1e9,
            "points_scanned": int(pts_scanned), "cell_runs": int(runs), "pair_evals_per_s": pts_scanned / (k_ms * 1e-3),
            "note": ("gather kernel: `frac` = algorithmic bytes / kernel time / 8 TB/s; the 10 M-point cloud (160 MB cell-sorted) stays in the 256 MiB "
                     "Infinity Cache, so the bytes that reach the fabric are `traffic` (frac_of_measured_traffic) and `bound` says what the SQ counters "
                     "of the committed profile say; the DRAM-resident figures are in c4_probe (100 M-point cloud) and stream_probe"),
        },
    }
    out.update(pre)
    if index_range:
        out["index_range_shards"] = index_range
    if routed:
        out["routed_c_abi"] = routed
        out["config"]["headline_form"] = ("partitioned batches through the routed C ABI over RCCL: every rank its own Q queries per step (weak scaling)"
                                          if scaling_kind == "weak" and c4 else
                                          "routed (slab ownership, C ABI over RCCL), replicated batch" if value == routed.get("merged_answers_per_s") else
                                          "routed through torch.distributed (rehearsal)" if spatial and value == spatial.get("merged_answers_per_s") else "index-range shards")
    if c4_legs:
        out["c4_q4096"] = c4_legs
    if spatial:
        out["spatial_routing"] = spatial
    if one_gpu:
        out["one_gpu_whole_cloud"] = dict(one_gpu, speedup_of_this_run=value / one_gpu["answers_per_s"])

    if a.replan_probe and world == 1:
        guarded(out, "replan_probe", lambda: replan_probe(E, synth))
        guarded(out, "replan_probe_clustered", lambda: replan_probe(E, synth, ticks=100, clustered=True))
        # corridor generation per replan (config C1 scenario: seed-6 pillar map seen from the start pose, clean_demo.launch constants,
        # fixed iteration counts 1500 / 400 / 200): safe-region RRT* on the engine, speculative batches, one fused launch per batch
        from pointcloudtraj_amd import corridor, scenarios

        def corridor_scenario():
            cloud1 = scenarios.sensed_cloud(12.0)
            scenarios.timed_scenario(corridor.SafeRegionRrtStar(80000), cloud1)      # warm-up (first launches, allocations)
            return dict(scenarios.timed_scenario(corridor.SafeRegionRrtStar(80000), cloud1),
                        what="C1 corridor scenario: setInput + SafeRegionExpansion(1500) + Refine(400) + new frame + Evaluate + Refine(200)",
                        cloud_points=int(len(cloud1)))
        guarded(out, "corridor_replan_probe", corridor_scenario)
        guarded(out, "c1_stated_probe", lambda: c1_stated_probe(E, None))    # config C1 exactly as SURVEY 8(d) states it

    if a.replan_probe and world == 1:
        # ingest stage in front of the cloud: voxel de-duplication of the 10 M-point cloud at res 0.25
        def ingest():
            from pointcloudtraj_amd import voxel
            d_pts = torch.from_numpy(local_pts).to(sc.device)
            vm = voxel.VoxelMap(0.25, len(local_pts))
            ms = []
            for _ in range(4):
                vm.clear()
                n_vox = vm.add_device(d_pts.data_ptr(), len(local_pts), 12)
                ms.append(vm.last_ms())
            again = vm.add_device(d_pts.data_ptr(), len(local_pts), 12)      # second pass: every point hits an existing voxel
            res = {"what": "pct_voxel_map_add_dev: 10 M fp32 points resident in HBM -> first-seen voxel cloud, res 0.25",
                   "points": int(len(local_pts)), "voxels": int(n_vox), "kernels_ms": float(np.median(ms[1:])),
                   "points_per_s": len(local_pts) / (float(np.median(ms[1:])) * 1e-3),
                   "all_duplicates_pass_ms": vm.last_ms(), "all_duplicates_new_voxels": int(again)}
            vm.close()
            return res
        guarded(out, "ingest_probe", ingest)

    if a.clustered_probe and world == 1:
        guarded(out, "clustered_probe", lambda: clustered_probe(E, synth, torch, sc.device, Q))

    if a.c4_probe and world == 1:
        guarded(out, "c4_probe", lambda: c4_probe(E, synth, torch, sc.device, Q))

    O = None
    if a.cpu_queries > 0 and world == 1:
        from oracle import oracle as O
        ncpu = a.cpu_points or a.points
        try:
            base, cpu_idx, cq = cpu_baseline(lambda: local_pts[:ncpu], ncpu, q_hosts[0], min(a.cpu_queries, Q))
        except Exception as ex:      # noqa: BLE001  (e.g. oracle/_ref not built on this machine: say so in the line instead of losing it)
            base, cpu_idx, cq = {"value": None, "unit": "queries/s", "cores": 0, "kind": "reference", "sample": "not measured",
                                 "error": f"{type(ex).__name__}: {ex}"}, None, []
        out["cpu_baseline"] = base
        if ncpu == a.points and cpu_idx is not None:     # same cloud: the GPU answers must equal the host kd-tree's (parity in the bench run itself)
            gi = idx_keep[:len(cq)].cpu().numpy()
            out["cpu_baseline"]["gpu_matches_cpu_indices"] = bool(np.array_equal(gi, cpu_idx))
        if "corridor_replan_probe" in out and "error" not in out["corridor_replan_probe"]:   # the same corridor scenario on the CPU restatement (oracle/rrt_port.c + kd-tree port), one core
            from pointcloudtraj_amd import scenarios
            cpu_cor = scenarios.timed_scenario(O.PortCorridor(), scenarios.sensed_cloud(12.0))
            out["cpu_baseline"]["corridor_replan_ms"] = cpu_cor["total_ms"]
            out["cpu_baseline"]["corridor_phases_ms"] = {k: v for k, v in cpu_cor.items() if k.endswith("_ms")}
            if "ingest_probe" in out and "error" not in out["ingest_probe"]:            # the sequential container restated (oracle/voxel_port.c) on the first 2 M points
                om = O.PortVoxelMap(0.25)
                t1 = time.perf_counter()
                om.add(local_pts[:2_000_000])
                out["cpu_baseline"]["voxel_ingest_points_per_s"] = 2_000_000 / (time.perf_counter() - t1)
            out["cpu_baseline"]["corridor_same_path_as_gpu"] = bool(cpu_cor["status"] == out["corridor_replan_probe"]["status"]
                                                                     and cpu_cor["path_len"] == out["corridor_replan_probe"]["path_len"])
    if "c1_stated_probe" in out and "error" not in out["c1_stated_probe"]:
        g = out["c1_stated_probe"]
        gpath = g.pop("_path")
        if O is not None:                                    # the CPU side only with the baseline leg
            try:
                c = c1_stated_probe(E, O)
                cpath = c.pop("_path")
                c["same_corridor"] = bool(np.array_equal(cpath[0], gpath[0]) and np.array_equal(cpath[1], gpath[1]) and c.pop("status") == g["status"])
                g["cpu_restatement"] = c
            except Exception as ex:      # noqa: BLE001
                g["cpu_restatement"] = {"error": f"{type(ex).__name__}: {ex}"}
    print(json.dumps(out))
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
