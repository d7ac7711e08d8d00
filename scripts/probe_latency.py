import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth, kdtree as K
E.init(0)
pts = synth.uniform_points(3, 10_000_000, 0, 100)
c = E.Cloud(len(pts)); c.set_input(pts); c.build_grid()
prm = E.inflate_params((50, 50, 50), 1e9, 0.25, 1.5)
def lat(fn, n=200):
    for _ in range(10): fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(1e6 * (time.perf_counter() - t0))
    return np.percentile(ts, 50), np.percentile(ts, 99)
for Q in (1, 8, 64, 200, 4096):
    q64 = synth.uniform_points(4, Q, 10, 90).astype(np.float64)
    q32 = q64.astype(np.float32)
    print(f"Q={Q:5d} inflate(grid) p50/p99 = %.1f / %.1f us   nn(grid) = %.1f / %.1f us   nn(stream) = %.1f / %.1f us" % (
        *lat(lambda: c.inflate(prm, q64)), *lat(lambda: c.nn(q32, E.ALGO_GRID)), *lat(lambda: c.nn(q32, E.ALGO_STREAM), 30)), flush=True)
plan = E.NNPlan(c, 64, E.ALGO_GRID)
q32 = synth.uniform_points(4, 64, 10, 90)
print("graph plan nn(grid) Q=64: %.1f / %.1f us" % lat(lambda: plan.run(q32)))
t = K.KDTree(); t.insert(synth.uniform_points(5, 1000, 0, 10))
import ctypes as C
L = K.lib(); qq = (C.c_float * 3)(5, 5, 5)
def kdq():
    r = L.kd_nearestf(t.h, qq); L.kd_res_item_data(r); L.kd_res_free(r)
print("kd_nearestf on a 1000-node tree: %.1f / %.1f us" % lat(kdq))
def kdr():
    r = L.kd_nearest_rangef(t.h, qq, C.c_float(1.5)); L.kd_res_free(r)
print("kd_nearest_rangef(1.5) on a 1000-node tree: %.1f / %.1f us" % lat(kdr))
# the planner's checkSafeTrajectory (99 samples over a 2 s horizon) on the indexed cloud: one launch; and the fused RRT* step
orders = np.int32([6, 6, 6]); seg_time = np.float64([1.0, 1.0, 1.0])
coef = np.zeros((3, 21))
for sgm in range(3):
    for d in range(3):
        for j in range(7):
            coef[sgm, d * 7 + j] = (40.0 + 4.0 * (sgm + j / 6.0) + (0.3 if d == 1 else 0.0))
print("bezier_check 99 samples (indexed cloud, one launch): %.1f / %.1f us" % lat(lambda: c.bezier_check(prm, coef, seg_time, orders, 0.0, 2.0, cap=128)))
c2 = E.Cloud(len(pts)); c2.set_input(pts[:5_000_000])
print("bezier_check 99 samples (un-indexed 5 M cloud, brute force, mapped I/O): %.1f / %.1f us" % lat(lambda: c2.bezier_check(prm, coef, seg_time, orders, 0.0, 2.0, cap=128), 50))
