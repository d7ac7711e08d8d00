"""Index build time (pct_cloud_build_grid: bounding box + counting sort into cells) for the C2 / C3 / C4 clouds, wall clock around the
call (it ends with a stream synchronise).  PCT_LDS_GRID_BUILD=0 selects the per-point-atomic build for comparison."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000, 100_000_000]
for n in sizes:
    side = 200.0 if n > 50_000_000 else 100.0
    c = E.Cloud(n)
    first = True
    for o, blk in synth.uniform_points_chunked(6 if n > 50_000_000 else 3, n, 0.0, side):
        (c.set_input if first else c.append)(blk)
        first = False
    ts = []
    for rep in range(6):
        E.sync(); t0 = time.perf_counter()
        c.build_grid()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"n={n}: build_grid ms first {ts[0]:.3f}, then median {np.median(ts[1:]):.3f} min {min(ts[1:]):.3f}  ({n / np.median(ts[1:]) / 1e6:.2f} G points/s)", flush=True)
    c.close()
