"""Streaming NN kernel alone: probe_stream.py [N] -- Q = 1, 2, 4 at the block caps given in PROBE_BLOCKS (default 1024)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
pts = synth.uniform_points(6, N, 0.0, 200.0)
c = E.Cloud(N); c.set_input(pts); c.reserve_queries(4096)
qd = torch.from_numpy(synth.uniform_points(5, 4096, 0.0, 200.0)).cuda()
oi = torch.empty(4096, dtype=torch.int32, device="cuda"); od = torch.empty(4096, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for blocks in os.environ.get("PROBE_BLOCKS", "1024").split(","):
    os.environ["PCT_STREAM_BLOCKS"] = blocks
    for qn in (1, 2, 4):
        ms = []
        for k in range(10):
            c.nn_device(qd.data_ptr(), qn, oi.data_ptr(), od.data_ptr(), s, E.ALGO_STREAM)
            torch.cuda.synchronize()
            if k >= 3: ms.append(c.last_kernel_ms())
        m = float(np.median(ms))
        print(f"N={N} blocks<={blocks} Q={qn}: kernel {m:.4f} ms = {12*N/(m*1e-3)/1e12:.2f} TB/s = {12*N/(m*1e-3)/8e12:.3f} of peak", flush=True)
