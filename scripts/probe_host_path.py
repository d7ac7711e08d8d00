"""Host-buffer entry point (pct_nn_batch: pageable numpy arrays in and out) at the bench's batch size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
pts = synth.uniform_points(3, 10_000_000, 0, 100)
c = E.Cloud(len(pts)); c.set_input(pts); c.build_grid()
for Q in (1 << 16, 1 << 20):
    q = synth.uniform_points(5, Q, 0, 100)
    c.nn(q, E.ALGO_GRID)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); c.nn(q, E.ALGO_GRID); ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    print(f"host buffers Q={Q}: {dt*1e3:.3f} ms  {Q/dt:.3e} q/s  ({24*Q/dt/1e9:.1f} GB/s of query+result traffic)")
