"""C2 (1M points, 4096 queries) brute-force time vs LDS chunk size (PCT_TILE_CHUNK is read per call)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
for n in (200_000, 1_000_000, 5_000_000):
    p2 = synth.uniform_points(1, n, 0.0, 100.0)
    q2 = torch.from_numpy(synth.uniform_points(2, 4096, 0.0, 100.0)).cuda()
    c = E.Cloud(n); c.set_input(p2); c.reserve_queries(4096)
    idx = torch.empty(4096, dtype=torch.int32, device="cuda"); d2 = torch.empty(4096, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for chunk in (256, 512, 1024, 2048):
        os.environ["PCT_TILE_CHUNK"] = str(chunk)
        for _ in range(2): c.nn_device(q2.data_ptr(), 4096, idx.data_ptr(), d2.data_ptr(), s, E.ALGO_STREAM)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): c.nn_device(q2.data_ptr(), 4096, idx.data_ptr(), d2.data_ptr(), s, E.ALGO_STREAM)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"N={n:8d} chunk={chunk:5d} groups ({4*chunk} points, {12*4*chunk//1024} KiB LDS): {dt*1e3:7.3f} ms  {4096*n/dt:.3e} pairs/s", flush=True)
    c.close()
