import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pointcloudtraj_amd import corridor, engine
from oracle import oracle as O
from pointcloudtraj_amd import scenarios as S
engine.init(0)
c1 = S.sensed_cloud(12.0)
p = S.PARAMS
def mk_gpu(k):
    def f():
        c = corridor.SafeRegionRrtStar(80000)
        c.setSpeculation(k)
        return c
    return f
for name, mk in (("cpu oracle", lambda: O.PortCorridor()), ("gpu engine K=1", mk_gpu(1)), ("gpu engine K=16", mk_gpu(16)), ("gpu engine K=64", mk_gpu(64)), ("gpu engine K=256", mk_gpu(256))):
    for rep in range(2):
        f = mk()
        t = [time.perf_counter()]
        f.setParam(p["safety_margin"], p["search_margin"], p["max_radius"], p["sensing_range"]); f.setInput(c1); t.append(time.perf_counter())
        f.reset(); f.setPt(S.START, S.GOAL, *S.BOUNDS, p["sensing_range"], p["max_samples"], p["sample_portion"], p["goal_portion"]); t.append(time.perf_counter())
        f.SafeRegionExpansion(1500); t.append(time.perf_counter())
        f.SafeRegionRefine(400); t.append(time.perf_counter())
        path, rad = f.getPath()
        f.setInput(S.perturbed_cloud(c1, path)); t.append(time.perf_counter())
        f.SafeRegionEvaluate(); t.append(time.perf_counter())
        f.SafeRegionRefine(200); t.append(time.perf_counter())
        d = np.diff(t) * 1e3
        extra = f.speculationStats() if hasattr(f, "speculationStats") else ""
        print(f"{name}: {extra} setInput {d[0]:.2f}  setPt {d[1]:.2f}  expansion(1500) {d[2]:.2f}  refine(400) {d[3]:.2f}  setInput2 {d[4]:.2f}  evaluate {d[5]:.2f}  refine(200) {d[6]:.2f} ms   status {f.status()}", flush=True)
