import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pointcloudtraj_amd import corridor, engine
from oracle import oracle as O
from corridor_scenario import run_scenario, sensed_cloud
engine.init(0)
c1 = sensed_cloud(12.0)
for name, mk in (("cpu oracle", lambda: O.PortCorridor()), ("gpu engine", lambda: corridor.SafeRegionRrtStar(80000))):
    for rep in range(2):
        f = mk()
        t0 = time.perf_counter(); ph = run_scenario(f, c1, None, expand=1500, refine=400); dt = time.perf_counter() - t0
        print(f"{name}: scenario {dt*1e3:.1f} ms; phases:", [(len(p), s['path_exists'], s['nodes'], s['inflation_queries']) for p, r, s in ph], flush=True)
