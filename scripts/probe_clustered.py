"""Surface / clustered clouds (SURVEY 8d's clustered variants): NN throughput through the cell index with the shell walk
(PCT_PYRAMID=0) and with the bounding-box pyramid (default), points / runs / node visits per query, parity with brute force.
usage: probe_clustered.py [Q] [which ...]      which in {pillar, clustered2m, pillar10m}"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1_048_576
which = sys.argv[2:] or ["pillar", "clustered2m", "pillar10m"]
clouds = {"pillar": lambda: synth.pillar_map(), "clustered2m": lambda: synth.clustered_points(62, 2_000_000, 0, 100),
          "pillar10m": lambda: synth.pillar_map_scaled(7.4)}
for name in which:
    pts = clouds[name]()
    lo, hi = pts.min(0), pts.max(0)
    u = synth.uniform01_f32(77, 3 * Q).reshape(Q, 3)
    q = (lo + u * (hi - lo)).astype(np.float32)
    c = E.Cloud(len(pts)); c.set_input(pts); c.reserve_queries(Q)
    dq = torch.from_numpy(q).cuda()
    di = torch.empty(Q, dtype=torch.int32, device="cuda"); dd = torch.empty(Q, dtype=torch.float64, device="cuda")
    ns = min(Q, 4096)
    i1, d1 = c.nn(q[:ns], E.ALGO_STREAM)
    for mode, ppc in (("0", "6"), ("1", "6"), ("1", "2"), ("1", "16"), ("1", "48")):
        os.environ["PCT_PYRAMID"] = mode
        os.environ["PCT_GRID_PPC"] = ppc
        t0 = time.perf_counter(); c.build_grid(); tb = time.perf_counter() - t0
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3): c.nn_device(dq.data_ptr(), Q, di.data_ptr(), dd.data_ptr(), s, E.ALGO_GRID)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): c.nn_device(dq.data_ptr(), Q, di.data_ptr(), dd.data_ptr(), s, E.ALGO_GRID)
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
        km = c.last_kernel_ms()
        ok = np.array_equal(i1, di[:ns].cpu().numpy().view(np.uint32)) and np.array_equal(d1, dd[:ns].cpu().numpy())
        c.set_work_counters(True); c.nn_device(dq.data_ptr(), Q, di.data_ptr(), dd.data_ptr(), s, E.ALGO_GRID); torch.cuda.synchronize()
        w = c.last_work_ex(); c.set_work_counters(False)
        pi = c.pyramid_info()
        print(f"{name}: N={len(pts)} dims={c.grid_info()['dims']} ppc={ppc} pyramid={mode} levels={pi['levels']} empty={pi['empty_fraction']:.3f} "
              f"build {tb*1e3:.2f} ms; {Q} queries: {tg*1e3:.3f} ms/step = {Q/tg:.3e} q/s, kernel {km:.3f} ms; per query: points {w[0]/Q:.1f} runs {w[1]/Q:.1f} "
              f"nodes {w[2]/Q:.1f}; matches_brute={ok}", flush=True)
    c.close()
