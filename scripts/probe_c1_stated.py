"""Config C1 as stated (9,383-point crop, Expansion(2000)): time and speculation statistics per speculation depth"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudtraj_amd import corridor, engine as E, scenarios, synth
E.init(0)
p = scenarios.PARAMS
cloud = synth.crop_ball(synth.pillar_map(), scenarios.START, 5.0)
for K in (1, 8, 32, 64, 256, 256):
    f = corridor.SafeRegionRrtStar(20000)
    f.setSpeculation(K)
    f.setParam(p["safety_margin"], p["search_margin"], p["max_radius"], p["sensing_range"])
    f.setInput(cloud); f.reset()
    f.setPt(scenarios.START, scenarios.GOAL, *scenarios.BOUNDS, p["sensing_range"], p["max_samples"], p["sample_portion"], p["goal_portion"])
    t0 = time.perf_counter(); f.SafeRegionExpansion(2000); t = 1e3 * (time.perf_counter() - t0)
    print(f"K={K}: {t:.2f} ms, launches {f.expansionLaunches()}, stats {f.speculationStats()}, status {f.status()}", flush=True)
