"""Workload definitions shared by tests, probes and bench.py (no oracle, no GPU code).

Config C1 corridor scenario: the clean_demo map (seed 6), start
(-10,-10,2) -> goal (9,9,2), clean_demo.launch planner constants, fixed iteration counts instead of wall-clock
limits, and a second, denser cloud for the lazy re-evaluation."""
import numpy as np

from . import synth

START, GOAL = (-10.0, -10.0, 2.0), (9.0, 9.0, 2.0)
BOUNDS = (-15.0, 15.0, -15.0, 15.0, 0.0, 4.0)
PARAMS = dict(safety_margin=0.6, search_margin=0.25, max_radius=1.5, sensing_range=30.0, max_samples=200000,
              sample_portion=0.3, goal_portion=0.1)


def sensed_cloud(radius=12.0):
    """what a 12 m sensor at the start pose has seen of the seed-6 map (shuffled, as a stream of frames would be)"""
    full = synth.pillar_map()
    crop = synth.crop_ball(full, START, radius)
    return crop[synth.shuffled_order(7, len(crop))]


def perturbed_cloud(cloud1, path):
    """a later sensor frame: the same cloud plus three new obstacle points just above the corridor's 10th, 13th and
    15th spheres -- their radii shrink (1.5 -> 1.2 / 1.3 / 1.45) but the chain stays connected and flyable"""
    extra = [path[min(k, len(path) - 1)] + np.float64([0.0, 0.0, dz]) for k, dz in ((9, 1.45), (12, 1.55), (14, 1.7))]
    return np.concatenate([cloud1, np.asarray(extra, np.float32)])


def run_scenario(finder, cloud1, cloud2=None, expand=1500, refine=400):
    """returns the (Path, Radius, status) after each planner phase; cloud2=None derives the second frame
    from the corridor found in the first phases (perturbed_cloud)"""
    p = PARAMS
    out = []
    finder.setParam(p["safety_margin"], p["search_margin"], p["max_radius"], p["sensing_range"])
    finder.setInput(cloud1)
    finder.reset()
    finder.setPt(START, GOAL, *BOUNDS, p["sensing_range"], p["max_samples"], p["sample_portion"], p["goal_portion"])
    finder.SafeRegionExpansion(expand)                 # planInitialTraj, sim_planning_demo.cpp:344-350
    out.append((*finder.getPath(), finder.status()))
    finder.SafeRegionRefine(refine)                    # planIncrementalTraj, :412
    out.append((*finder.getPath(), finder.status()))
    if cloud2 is None:
        cloud2 = perturbed_cloud(cloud1, out[-1][0])
    finder.setInput(cloud2)                            # a new sensor frame arrives (rcvPointCloudCallBack, :159-167)
    finder.SafeRegionEvaluate()                        # :413
    out.append((*finder.getPath(), finder.status()))
    finder.SafeRegionRefine(refine // 2)
    out.append((*finder.getPath(), finder.status()))
    return out


def timed_scenario(finder, cloud1, expand=1500, refine=400):
    """run_scenario with wall-clock milliseconds per planner phase (bench.py / scripts/probe_corridor.py)"""
    import time
    p = PARAMS
    t = [time.perf_counter()]
    finder.setParam(p["safety_margin"], p["search_margin"], p["max_radius"], p["sensing_range"])
    finder.setInput(cloud1); t.append(time.perf_counter())
    finder.reset()
    finder.setPt(START, GOAL, *BOUNDS, p["sensing_range"], p["max_samples"], p["sample_portion"], p["goal_portion"])
    finder.SafeRegionExpansion(expand); t.append(time.perf_counter())
    finder.SafeRegionRefine(refine); t.append(time.perf_counter())
    path, _ = finder.getPath()
    cloud2 = perturbed_cloud(cloud1, path)
    t.append(time.perf_counter())
    finder.setInput(cloud2); t.append(time.perf_counter())
    finder.SafeRegionEvaluate(); t.append(time.perf_counter())
    finder.SafeRegionRefine(refine // 2); t.append(time.perf_counter())
    d = [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])]
    out = {"set_input_ms": d[0], "expansion_ms": d[1], "refine_ms": d[2], "set_input_2_ms": d[4], "evaluate_ms": d[5], "refine_2_ms": d[6]}
    out["total_ms"] = sum(out.values())
    out["status"] = finder.status()
    out["path_len"] = len(finder.getPath()[0])
    return out
