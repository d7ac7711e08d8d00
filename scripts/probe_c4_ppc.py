"""C4-sized cloud on one card (DRAM-resident): the 1 M-query indexed batch at several cell sizes (PCT_GRID_PPC) -- kernel ms, points / runs
per query, algorithmic bytes by the 12-B rule.  usage: probe_c4_ppc.py [points] [ppc ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ppcs = sys.argv[2:] or ["6", "4", "3", "2"]
Q = 1 << 20
side = 200.0 if N > 20_000_000 else 100.0
pts = synth.uniform_points(6, N, 0.0, side)
q = synth.uniform_points(5, Q, 0.0, side)
c = E.Cloud(N); c.set_input(pts); c.reserve_queries(Q)
dq = torch.from_numpy(q).cuda(); di = torch.empty(Q, dtype=torch.int32, device="cuda"); dd = torch.empty(Q, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ref = None
for ppc in ppcs:
    os.environ["PCT_GRID_PPC"] = ppc
    t0 = time.perf_counter(); c.build_grid(); tb = time.perf_counter() - t0
    for _ in range(3): c.nn_device(dq.data_ptr(), Q, di.data_ptr(), dd.data_ptr(), s, E.ALGO_GRID)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): c.nn_device(dq.data_ptr(), Q, di.data_ptr(), dd.data_ptr(), s, E.ALGO_GRID)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
    km = float(np.mean(c.kernel_ms_history(10)))
    got = (di.cpu().numpy().copy(), dd.cpu().numpy().copy())
    same = True if ref is None else (np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]))
    ref = ref or got
    c.set_work_counters(True); c.nn_device(dq.data_ptr(), Q, di.data_ptr(), dd.data_ptr(), s, E.ALGO_GRID); torch.cuda.synchronize(); w = c.last_work_ex(); c.set_work_counters(False)
    alg = 12 * w[0] + 8 * w[1] + 24 * Q
    print(f"N={N} ppc={ppc} dims={c.grid_info()['dims']} build {tb*1e3:.1f} ms step {tg*1e3:.3f} ms kernel {km:.3f} ms points/q {w[0]/Q:.1f} runs/q {w[1]/Q:.2f} "
          f"alg {alg/1e6:.0f} MB -> {alg/km/1e6:.0f} GB/s = {alg/km/1e6/8000:.3f} of peak; same answers as first: {same}", flush=True)
