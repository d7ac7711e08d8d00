#!/usr/bin/env python3
"""Kernel timing probe (GPU box): sweeps the tuning knobs and prints one line per configuration.
Not part of the product or of bench.py; used to choose defaults recorded in DESIGN.md."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (first: one HIP runtime per process)
from pointcloudtraj_amd import engine as E, synth  # noqa: E402

N = int(os.environ.get("PROBE_N", 10_000_000))
what = sys.argv[1] if len(sys.argv) > 1 else "all"
E.init(0)
pts = synth.uniform_points(3, N, 0, 100)
c = E.Cloud(N)
c.set_input(pts)
qh = synth.uniform_points(5, 1 << 20, 0, 100)
q = torch.from_numpy(qh).cuda()
Qmax = len(qh)
c.reserve_queries(Qmax)
c.set_timing(2)
idx = torch.empty(Qmax, dtype=torch.int32, device="cuda")
d2 = torch.empty(Qmax, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream


def timeit(Q, algo, reps=10, batch=True):
    for _ in range(2):
        c.nn_device(q.data_ptr(), Q, idx.data_ptr(), d2.data_ptr(), s, algo)
    ms = []
    for _ in range(reps):
        c.nn_device(q.data_ptr(), Q, idx.data_ptr(), d2.data_ptr(), s, algo)
        ms.append(c.last_batch_ms() if batch else c.last_kernel_ms())
    torch.cuda.synchronize()
    return float(np.median(ms)), float(np.min(ms))


if what in ("all", "stream"):
    for name, algo, tile in (("exact ", E.ALGO_STREAM_EXACT, 8), ("filter", E.ALGO_STREAM, 16)):
        for blocks in (1024, 2048):
            os.environ["PCT_STREAM_BLOCKS"] = str(blocks)
            for Q in (1, 2, 4, 8, 16, 64, 200, 4096):
                med, mn = timeit(Q, algo, reps=5 if Q > 100 else 10)
                passes = (Q + tile - 1) // tile
                gbs = 12 * N * passes / (med * 1e-3) / 1e9
                print(f"stream {name} blocks={blocks:5d} Q={Q:4d} median={med*1e3:9.1f}us min={mn*1e3:9.1f}us  {gbs:7.0f} GB/s ({gbs/80:.1f}% of 8TB/s)  pairs/s={Q*N/(med*1e-3):.3e}", flush=True)
    os.environ.pop("PCT_STREAM_BLOCKS", None)

if what in ("all", "grid"):
    for ppc, shift in ((0.5, 1), (0.5, 2), (1.0, 1), (1.0, 2), (2.0, 0), (2.0, 1), (2.0, 2), (4.0, 1)):
        os.environ["PCT_GRID_PPC"] = str(ppc)
        os.environ["PCT_BIN_SHIFT"] = str(shift)
        t0 = time.perf_counter()
        c.build_grid()
        E.sync()
        tb = time.perf_counter() - t0
        for Q in (4096, 1 << 16, 1 << 20):
            med, mn = timeit(Q, E.ALGO_GRID)
            kmed, _ = timeit(Q, E.ALGO_GRID, batch=False)
            print(f"[kernel only {kmed*1e3:7.1f}us] grid ppc={ppc:4.1f} shift={shift} dims={c.grid_info()['dims']} build={tb*1e3:6.2f}ms Q={Q:8d} median={med*1e3:9.1f}us  {Q/(med*1e-3):.3e} q/s", flush=True)
        c.set_work_counters(True)
        c.nn_device(q.data_ptr(), 1 << 20, idx.data_ptr(), d2.data_ptr(), s, E.ALGO_GRID)
        torch.cuda.synchronize()
        print("    work (points, runs) per query:", [w / (1 << 20) for w in c.last_work()], flush=True)
        c.set_work_counters(False)
    # spatially coherent queries (sorted by grid cell) -- how much does locality buy?
    os.environ["PCT_GRID_PPC"] = "2.0"
    c.build_grid()
    gi = c.grid_info()
    h = gi["cell_size"]
    cell = np.floor(qh / h).astype(np.int64)
    key = (cell[:, 2] * gi["dims"][1] + cell[:, 1]) * gi["dims"][0] + cell[:, 0]
    qs = torch.from_numpy(qh[np.argsort(key, kind="stable")]).cuda()
    q_save = q
    q = qs
    med, mn = timeit(1 << 20, E.ALGO_GRID)
    print(f"grid ppc=2.0 SORTED queries Q=1M median={med*1e3:9.1f}us  {(1<<20)/(med*1e-3):.3e} q/s", flush=True)
    q = q_save

if what == "sort":   # batch vs kernel-only time with the default grid (env knobs are read once per process: one run per setting)
    c.build_grid()
    E.sync()
    for Q in (1 << 16, 1 << 18, 1 << 20):
        med, mn = timeit(Q, E.ALGO_GRID, reps=20)
        kmed, kmn = timeit(Q, E.ALGO_GRID, reps=20, batch=False)
        print(f"sort probe fine={os.environ.get('PCT_SORT_FINE', '1')} shift={os.environ.get('PCT_BIN_SHIFT', '1')} Q={Q:8d} batch median={med*1e3:7.1f}us min={mn*1e3:7.1f}us  kernel median={kmed*1e3:7.1f}us min={kmn*1e3:7.1f}us", flush=True)

if what == "octant":   # 2x2x2-block-first search vs cube-first, over cell sizes
    for ppc in (1.0, 2.0, 3.0, 4.0, 6.0, 8.0):
        for octant in (0, 1):
            os.environ["PCT_GRID_PPC"] = str(ppc)
            os.environ["PCT_OCTANT_FIRST"] = str(octant)
            c.build_grid()
            E.sync()
            for Q in (1 << 20,):
                med, mn = timeit(Q, E.ALGO_GRID, reps=15)
                kmed, kmn = timeit(Q, E.ALGO_GRID, reps=15, batch=False)
                c.set_work_counters(True)
                c.nn_device(q.data_ptr(), Q, idx.data_ptr(), d2.data_ptr(), s, E.ALGO_GRID)
                torch.cuda.synchronize()
                w = [x / Q for x in c.last_work()]
                c.set_work_counters(False)
                print(f"octant={octant} ppc={ppc:3.1f} dims={c.grid_info()['dims']} Q={Q} batch median={med*1e3:7.1f}us kernel median={kmed*1e3:7.1f}us min={kmn*1e3:7.1f}us  points/query={w[0]:6.1f} runs/query={w[1]:5.2f}", flush=True)

if what == "octant2":
    os.environ["PCT_OCTANT_FIRST"] = "1"
    for ppc in (5.0, 6.0, 7.0):
        for shift in (0, 1):
            os.environ["PCT_GRID_PPC"] = str(ppc)
            os.environ["PCT_BIN_SHIFT"] = str(shift)
            c.build_grid()
            E.sync()
            for Q in (1 << 16, 1 << 20):
                med, mn = timeit(Q, E.ALGO_GRID, reps=15)
                kmed, kmn = timeit(Q, E.ALGO_GRID, reps=15, batch=False)
                print(f"octant=1 ppc={ppc:3.1f} shift={shift} dims={c.grid_info()['dims']} Q={Q} batch median={med*1e3:7.1f}us kernel median={kmed*1e3:7.1f}us min={kmn*1e3:7.1f}us", flush=True)

if what == "count":   # radius count through the grid and brute force
    c.build_grid()
    E.sync()
    cnt = torch.empty(Qmax, dtype=torch.int32, device="cuda")
    for rad in (0.5, 1.0, 2.0):
        r = torch.full((Qmax,), rad, dtype=torch.float32, device="cuda")
        for Q in (4096, 1 << 16, 1 << 20):
            for _ in range(2):
                c.radius_count_device(q.data_ptr(), r.data_ptr(), Q, cnt.data_ptr(), s, E.ALGO_GRID)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                c.radius_count_device(q.data_ptr(), r.data_ptr(), Q, cnt.data_ptr(), s, E.ALGO_GRID)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print(f"count grid r={rad} Q={Q:8d}: {dt*1e3:8.3f} ms  {Q/dt:.3e} q/s  mean count {float(cnt[:Q].float().mean()):.1f}", flush=True)
    r = torch.full((Qmax,), 1.0, dtype=torch.float32, device="cuda")
    for Q in (64, 4096):
        for _ in range(2):
            c.radius_count_device(q.data_ptr(), r.data_ptr(), Q, cnt.data_ptr(), s, E.ALGO_STREAM)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            c.radius_count_device(q.data_ptr(), r.data_ptr(), Q, cnt.data_ptr(), s, E.ALGO_STREAM)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"count stream r=1.0 Q={Q:8d}: {dt*1e3:8.3f} ms  {Q/dt:.3e} q/s  pairs/s {Q*N/dt:.3e}", flush=True)
