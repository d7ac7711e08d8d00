"""One sparse-cloud configuration through the cell index, for profiling: probe_pyr.py [cloud] [Q] [reps] (env PCT_PYRAMID, PCT_GRID_PPC)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "pillar10m"
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1_048_576
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
pts = {"pillar": synth.pillar_map, "clustered2m": lambda: synth.clustered_points(62, 2_000_000, 0, 100), "pillar10m": lambda: synth.pillar_map_scaled(7.4),
       "pillar100m": lambda: synth.pillar_map_scaled(23.2),          # config C4's size in the clustered variant, one card
       "uniform10m": lambda: synth.uniform_points(3, 10_000_000, 0, 100)}[name]()
lo, hi = pts.min(0), pts.max(0)
q = (lo + synth.uniform01_f32(77, 3 * Q).reshape(Q, 3) * (hi - lo)).astype(np.float32)
c = E.Cloud(len(pts)); c.set_input(pts); c.reserve_queries(Q); c.build_grid(); c.build_grid()      # steady state: the second build of a sparse cloud uses smaller cells
dq = torch.from_numpy(q).cuda()
di = torch.empty(Q, dtype=torch.int32, device="cuda"); dd = torch.empty(Q, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3): c.nn_device(dq.data_ptr(), Q, di.data_ptr(), dd.data_ptr(), s, E.ALGO_GRID)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): c.nn_device(dq.data_ptr(), Q, di.data_ptr(), dd.data_ptr(), s, E.ALGO_GRID)
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / reps
c.set_work_counters(True); c.nn_device(dq.data_ptr(), Q, di.data_ptr(), dd.data_ptr(), s, E.ALGO_GRID); torch.cuda.synchronize(); w = c.last_work_ex(); c.set_work_counters(False)
print(f"{name}: N={len(pts)} dims={c.grid_info()['dims']} pyramid={c.pyramid_info()} {Q} queries {tg*1e3:.3f} ms/step = {Q/tg:.3e} q/s kernel {c.last_kernel_ms():.3f} ms; "
      f"per query: points {w[0]/Q:.1f} runs {w[1]/Q:.1f} nodes {w[2]/Q:.1f}", flush=True)
