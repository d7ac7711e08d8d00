"""Host-buffer batches (pageable numpy arrays in, numpy arrays out) through pct_nn_batch_algo / pct_radius_count_batch: config C2's
shape (1 M points, 4096 queries, brute force) and the indexed path at several batch sizes.  PCT_MAPPED_IO=0 = DMA copies + stream
synchronise (the round-1 form) for comparison."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudtraj_amd import engine as E, synth
E.init(0)
pts = synth.uniform_points(1, 1_000_000, 0.0, 100.0)
c = E.Cloud(len(pts)); c.set_input(pts)
def timed(fn, reps=30):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(ts)), float(np.percentile(ts, 95))
q = synth.uniform_points(2, 4096, 0.0, 100.0)
print("C2 brute force, 4096 queries, host buffers: median %.3f ms  p95 %.3f" % timed(lambda: c.nn(q, E.ALGO_STREAM)), flush=True)
print("C2 radius count r=1, 4096 queries, brute force: median %.3f ms  p95 %.3f" % timed(lambda: c.radius_count(q, 1.0, E.ALGO_STREAM)), flush=True)
c.build_grid()
for Q in (2048, 4096, 16384, 65536, 262144):
    qq = synth.uniform_points(2, Q, 0.0, 100.0)
    print("indexed NN, %6d queries, host buffers: median %.3f ms  p95 %.3f" % ((Q,) + timed(lambda: c.nn(qq, E.ALGO_GRID))), flush=True)
    print("indexed radius count r=1, %6d queries: median %.3f ms  p95 %.3f" % ((Q,) + timed(lambda: c.radius_count(qq, 1.0, E.ALGO_GRID))), flush=True)
