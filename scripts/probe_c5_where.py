"""Where the clustered C5 tick spends its search time: per-query kernel time against the distance to the nearest surface"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudtraj_amd import engine as E, synth, scenarios as S
E.init(0)
window, frame = S.C5_WINDOW, S.C5_FRAME
c = E.Cloud(window); c.ring_index()
for k in range(window // frame + 8): c.append(S.c5_frame_clustered(k))
print(c.ring_info())
P = S.C5_PARAMS
k = window // frame + 8
start, nodes, coef, T, od = S.c5_tick_queries(k)
for name, pts in (("nodes", nodes), ("axis", np.stack([np.linspace(start[0], start[0] + 12, 64), np.zeros(64), np.full(64, 2.5)], 1)),
                  ("y=2", np.stack([np.linspace(start[0], start[0] + 12, 64), np.full(64, 2.0), np.full(64, 2.5)], 1)),
                  ("y=3.1", np.stack([np.linspace(start[0], start[0] + 12, 64), np.full(64, 3.1), np.full(64, 2.5)], 1)),
                  ("y=5", np.stack([np.linspace(start[0], start[0] + 12, 64), np.full(64, 5.0), np.full(64, 2.5)], 1))):
    for far in (True, False, True, False):
        prm = E.inflate_params(start, P["sample_range"], P["search_margin"], 1e9 if far else P["max_radius"])
        for _ in range(3): rad, idx, d2 = c.inflate(prm, pts)
        t0 = time.perf_counter()
        for _ in range(10): rad, idx, d2 = c.inflate(prm, pts)
        dt = (time.perf_counter() - t0) / 10
        print(f"{name:6s} max_radius={'inf' if far else P['max_radius']}: {dt*1e6:8.1f} us per 64-point call; radius min/median/max {rad.min():.2f} {np.median(rad):.2f} {rad.max():.2f}")
