"""Brute-force family probe: 10 M x 4096 NN (kernel + wall), C2 (1 M x 4096, host buffers), radius count 10 M x 4096."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
def med(fn, n=6):
    ts = []
    for k in range(n):
        t0 = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(ts[1:]))
pts = synth.uniform_points(3, 10_000_000, 0, 100)
c = E.Cloud(len(pts)); c.set_input(pts)
q = synth.uniform_points(5, 4096, 0, 100)
i0, d0 = c.nn(q, E.ALGO_STREAM)
ms = med(lambda: c.nn(q, E.ALGO_STREAM))
print(f"NN 10M x 4096 brute force: wall {ms:.3f} ms  dominant kernel {c.last_kernel_ms():.3f} ms  = {4096 * 1e7 / (c.last_kernel_ms() * 1e-3):.3e} pairs/s", flush=True)
r = np.full(4096, 1.0, np.float32)
cnt = c.radius_count(q, r, E.ALGO_STREAM)
ms = med(lambda: c.radius_count(q, r, E.ALGO_STREAM), 4)
print(f"radius count 10M x 4096 (r = 1) brute force: wall {ms:.3f} ms = {4096 * 1e7 / (ms * 1e-3):.3e} pairs/s   mean count {cnt.mean():.2f}", flush=True)
c.build_grid()
i1, d1 = c.nn(q, E.ALGO_GRID)
cg = c.radius_count(q, r, E.ALGO_GRID)
print("brute == indexed:", bool(np.array_equal(i0, i1) and np.array_equal(d0, d1)), " counts equal:", bool(np.array_equal(cnt, cg)), flush=True)
c.close()
p2 = synth.uniform_points(1, 1_000_000, 0, 100); q2 = synth.uniform_points(2, 4096, 0, 100)
c2 = E.Cloud(len(p2)); c2.set_input(p2)
print(f"C2 1M x 4096 brute force, host buffers: {med(lambda: c2.nn(q2, E.ALGO_STREAM), 8):.3f} ms", flush=True)
