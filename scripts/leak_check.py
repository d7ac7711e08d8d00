"""Create / use / destroy every kind of handle repeatedly and watch the device's free memory (torch.cuda.mem_get_info)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth, voxel, kdtree as K, corridor, scenarios
E.init(0)
pts = synth.uniform_points(1, 300_000, 0, 50)
q = synth.uniform_points(2, 20_000, 0, 50)
cloud1 = scenarios.sensed_cloud(12.0)
def once():
    c = E.Cloud(len(pts)); c.set_input(pts); c.nn(q[:100], E.ALGO_STREAM); c.build_grid(); c.nn(q, E.ALGO_GRID)
    c.radius_count(q[:500], np.float32(1.0)); c.radius_crop([25, 25, 25], 5.0)
    prm = E.inflate_params((25, 25, 25), 100.0, 0.25, 1.5); c.inflate(prm, q[:64].astype(np.float64))
    plan = E.NNPlan(c, 64, E.ALGO_GRID); plan.run(q[:64]); plan.close()
    o = E.Cloud(len(pts)); c.crop_to([25, 25, 25], 8.0, o); o.close(); c.close()
    v = voxel.VoxelMap(0.1, 1000); v.add_point_cloud(pts); v.close()
    t = K.KDTree(); t.insert(pts[:3000]); t.nearest(q[:5]); t.range_ids(q[0], 2.0); t.close()
    f = corridor.SafeRegionRrtStar(80000); scenarios.timed_scenario(f, cloud1); f.close() if hasattr(f, "close") else None
once(); torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for i in range(40):
    once()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(f"free before {free0/2**20:.1f} MiB, after 40 more rounds {free1/2**20:.1f} MiB, delta {(free0-free1)/2**20:.2f} MiB")
