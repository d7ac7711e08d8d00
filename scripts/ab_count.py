"""Wall-clock ms per 1 M-query radius-count batch through the index on the C3 cloud (r = 0.5 / 1.0 / 2.0)."""
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
cnt = torch.empty(Q, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
out = []
for r in (0.5, 1.0, 2.0):
    rad = torch.full((Q,), r, dtype=torch.float32, device="cuda")
    for _ in range(30): c.radius_count_device(q.data_ptr(), rad.data_ptr(), Q, cnt.data_ptr(), s, E.ALGO_GRID)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): c.radius_count_device(q.data_ptr(), rad.data_ptr(), Q, cnt.data_ptr(), s, E.ALGO_GRID)
    torch.cuda.synchronize(); out.append("r=%.1f %.4f ms (mean count %.1f)" % (r, (time.perf_counter() - t0) / 20 * 1e3, float(cnt.float().mean())))
print("; ".join(out))
