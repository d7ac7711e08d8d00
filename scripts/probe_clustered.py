import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
for name, pts in (("pillar_map_182k", synth.pillar_map()), ("clustered_2M", synth.clustered_points(62, 2_000_000, 0, 100))):
    lo, hi = pts.min(0), pts.max(0)
    Q = 200_000
    u = synth.uniform01_f32(77, 3 * Q).reshape(Q, 3)
    q = (lo + u * (hi - lo)).astype(np.float32)
    c = E.Cloud(len(pts)); c.set_input(pts)
    t0 = time.perf_counter(); i1, d1 = c.nn(q[:20000], E.ALGO_STREAM); ts = time.perf_counter() - t0
    for ppc in (2.0, 8.0, 32.0):
        os.environ["PCT_GRID_PPC"] = str(ppc)
        c.build_grid()
        c.nn(q[:1000], E.ALGO_GRID)
        t0 = time.perf_counter(); i2, d2 = c.nn(q, E.ALGO_GRID); tg = time.perf_counter() - t0
        c.set_work_counters(True); c.nn(q, E.ALGO_GRID); w = c.last_work(); c.set_work_counters(False)
        ok = np.array_equal(i1, i2[:20000]) and np.array_equal(d1, d2[:20000])
        print(f"{name}: N={len(pts)} grid dims={c.grid_info()['dims']} ppc={ppc}: {Q} queries in {tg*1e3:.2f} ms ({Q/tg:.3e} q/s) "
              f"points/query={w[0]/Q:.0f} runs/query={w[1]/Q:.0f} kernel={c.last_kernel_ms():.3f}ms matches_stream={ok}; stream 20000 q: {ts*1e3:.1f} ms", flush=True)
    c.close()
