"""Wall-clock milliseconds per 1M-query batch on the C3 cloud (no event readback): used by scripts/ab.sh to compare builds."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
pts = synth.uniform_points(3, 10_000_000, 0, 100)
c = E.Cloud(len(pts)); c.set_input(pts); c.build_grid()
Q = 1 << 20
q = torch.from_numpy(synth.uniform_points(5, Q, 0, 100)).cuda()
c.reserve_queries(Q)
if os.environ.get("PCT_AB_TIMING"):
    c.set_timing(int(os.environ["PCT_AB_TIMING"]))
idx = torch.empty(Q, dtype=torch.int32, device="cuda"); d2 = torch.empty(Q, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
res = []
for rep in range(5):
    for _ in range(3): c.nn_device(q.data_ptr(), Q, idx.data_ptr(), d2.data_ptr(), s, E.ALGO_GRID)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): c.nn_device(q.data_ptr(), Q, idx.data_ptr(), d2.data_ptr(), s, E.ALGO_GRID)
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 30 * 1e3)
km = c.kernel_ms_history(30) if int(os.environ.get("PCT_AB_TIMING", "1")) >= 1 else [float("nan")]
print("ms/step median %.4f min %.4f  kernel_ms (events) mean %.4f" % (float(np.median(res)), min(res), float(np.mean(km))))
