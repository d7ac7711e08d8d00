"""Per-tick timing of the C5 probe around the slow tick (diagnostic)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
window, frame = 5_000_000, 50_000
cloud = E.Cloud(window)
def frame_pts(k):
    p = synth.uniform_points(8, frame, -30.0, 30.0, offset=k * frame)
    p[:, 0] += np.float32(0.1 * k); p[:, 2] = np.abs(p[:, 2]) * np.float32(0.2)
    return p
nfill = window // frame
for k in range(nfill - 2): cloud.append(frame_pts(k))
orders = np.int32([6, 6, 6]); seg_time = np.float64([1.0, 1.0, 1.0])
for k in range(nfill - 2, nfill + 30):
    x0 = 0.1 * k
    f = frame_pts(k)
    t0 = time.perf_counter(); cloud.append(f); t1 = time.perf_counter()
    nodes = (synth.uniform_points(9, 64, -1.0, 1.0, offset=k * 64).astype(np.float64) * [8.0, 3.0, 1.0] + [x0 + 6.0, 0.0, 2.5])
    prm = E.inflate_params((x0, 0.0, 2.5), 30.0, 0.25, 1.5)
    cloud.inflate(prm, nodes); t2 = time.perf_counter()
    coef = np.zeros((3, 21))
    ctrl = synth.uniform_points(9, 21, -0.3, 0.3, offset=1_000_000 + k * 21).astype(np.float64)
    for sgm in range(3):
        for d in range(3):
            for j in range(7):
                w = (sgm + j / 6.0) / 3.0
                base = [x0 + 12.0 * w, 0.0, 2.5][d]
                coef[sgm, d * 7 + j] = (base + (ctrl[sgm * 7 + j, d] if 0 < j < 6 else 0.0)) / seg_time[sgm]
    r = cloud.bezier_check(prm, coef, seg_time, orders, 0.0, 2.0, cap=128); t3 = time.perf_counter()
    print(f"tick {k - nfill:3d}: ingest {1e3*(t1-t0):7.3f} inflate {1e3*(t2-t1):7.3f} bezier {1e3*(t3-t2):7.3f} ms  n={r['n']} first_hit={r['first_hit']} min_d2={r['d2'].min():.4g}", flush=True)
