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


def run_commit_scenario(finder, cloud1, expand=800, refine=300, commits=3):
    """planIncrementalTraj's flow (sim_planning_demo.cpp:393-460): after the first corridor the drone commits to a point on it,
    the finder moves its root there (resetRoot), refines, and re-evaluates against the next frame -- `commits` times.  The commit
    target is the centre of the corridor's third sphere (inside the root-side spheres, as the committed trajectory end is)."""
    p = PARAMS
    out = []
    finder.setParam(p["safety_margin"], p["search_margin"], p["max_radius"], p["sensing_range"])
    finder.setInput(cloud1)
    finder.reset()
    finder.setPt(START, GOAL, *BOUNDS, p["sensing_range"], p["max_samples"], p["sample_portion"], p["goal_portion"])
    finder.SafeRegionExpansion(expand)
    finder.SafeRegionRefine(refine)
    out.append((*finder.getPath(), finder.status()))
    cloud = cloud1
    for k in range(commits):
        path, _ = finder.getPath()
        if not finder.status()["path_exists"] or len(path) < 4:
            break
        target = tuple(float(v) for v in path[2])
        finder.setStartPt(target, GOAL)
        finder.resetRoot(target)
        out.append((*finder.getPath(), finder.status()))       # the path is only re-traced by the next phase; the status moves now
        finder.SafeRegionRefine(refine // 2)
        out.append((*finder.getPath(), finder.status()))
        cloud = perturbed_cloud(cloud, finder.getPath()[0])
        finder.setInput(cloud)
        finder.SafeRegionEvaluate()
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


# ---- config C5: rolling window fed one sensor frame per tick ------------------------------------------
C5_WINDOW, C5_FRAME = 5_000_000, 50_000
C5_NODES, C5_SEGMENTS, C5_ORDER = 64, 3, 6
C5_PARAMS = dict(sample_range=30.0, search_margin=0.25, max_radius=1.5)


def c5_frame(k, frame=C5_FRAME, tunnel=0.0, step=0.1):
    """sensor frame k: `frame` points uniform in a 60 m cube around a drone moving +`step` m per frame along x (seed 8),
    flattened to a 6 m slab above the ground (|z| * 0.2).  tunnel > 0 (test variant): points closer than `tunnel` to the
    flight axis (y = 0, z = 2.5) are moved sideways by 2 * tunnel, so the corridor ahead of the drone is free space and the
    inflation radii are not all negative."""
    p = synth.uniform_points(8, frame, -30.0, 30.0, offset=k * frame)
    p[:, 0] += np.float32(step * k)
    p[:, 2] = np.abs(p[:, 2]) * np.float32(0.2)
    if tunnel > 0:
        near = np.hypot(p[:, 1], p[:, 2] - np.float32(2.5)) < tunnel
        p[near, 1] += np.where(p[near, 1] >= 0, np.float32(2 * tunnel), np.float32(-2 * tunnel))
    return p


def c5_frame_clustered(k, frame=C5_FRAME, step=0.1):
    """sensor frame k of config C5's CLUSTERED variant (points on 0.1-grid pillar surfaces, as map_generator.cpp makes them): the
    same window around the same moving drone, but every point lies on a face of a square pillar -- pillars on a 3 m lattice, 0.6-1.4 m wide by
    a hash of their lattice cell, 0-6 m tall -- snapped to the 0.1 m lattice.  Frames re-sense the same surface points again and again (exact
    duplicates across frames, as the reference's rgbd mode accumulates them, camera_sensor.cpp:160-166); the corridor along the flight
    axis holds no pillar."""
    u = synth.uniform01_f32(8, 3 * frame, offset=3 * k * frame).reshape(frame, 3).astype(np.float64)
    v = synth.uniform01_f32(18, 2 * frame, offset=2 * k * frame).reshape(frame, 2).astype(np.float64)
    x0 = step * k
    px, py = u[:, 0] * 60.0 - 30.0 + x0, u[:, 1] * 60.0 - 30.0
    ci, cj = np.floor(px / 3.0), np.floor(py / 3.0)                       # the pillar's lattice cell
    cj = np.where(np.abs(cj * 3.0 + 1.5) <= 1.5, cj + np.where(py >= 0, 1.0, -1.0), cj)     # the two rows beside the flight axis move out: |y| < 3.8 m stays free
    hsh = (ci.astype(np.int64) * 73856093) ^ (cj.astype(np.int64) * 19349663)
    w = 0.6 + 0.1 * ((hsh >> 3) & 7).astype(np.float64)                   # 0.6 .. 1.3 m
    cx, cy = ci * 3.0 + 1.5, cj * 3.0 + 1.5
    face = (u[:, 2] * 4.0).astype(np.int64) & 3
    along = (v[:, 0] - 0.5) * w
    x = np.where(face == 0, cx - w / 2, np.where(face == 1, cx + w / 2, cx + along))
    y = np.where(face == 2, cy - w / 2, np.where(face == 3, cy + w / 2, cy + along))
    z = v[:, 1] * 6.0
    p = np.stack([x, y, z], 1)
    return (np.round(p * 10.0) / 10.0).astype(np.float32)


def c5_tick_queries(k):
    """what tick k asks of the cloud: the drone's pose, 64 corridor-node centres ahead of it (seed 9) and the committed
    trajectory -- 3 segments of order 6, 1 s each, control points jittered by +-0.3 m (seed 9) around a straight 12 m run;
    returns (start, nodes f64 [64,3], polycoef f64 [3,21] (control points / T as the optimizer stores them), seg_time, orders)"""
    x0 = 0.1 * k
    nodes = (synth.uniform_points(9, C5_NODES, -1.0, 1.0, offset=k * C5_NODES).astype(np.float64) * [8.0, 3.0, 1.0] + [x0 + 6.0, 0.0, 2.5])
    m = C5_ORDER + 1
    seg_time = np.ones(C5_SEGMENTS)
    orders = np.full(C5_SEGMENTS, C5_ORDER, np.int32)
    coef = np.zeros((C5_SEGMENTS, 3 * m))
    ctrl = synth.uniform_points(9, C5_SEGMENTS * m, -0.3, 0.3, offset=1_000_000 + k * C5_SEGMENTS * m).astype(np.float64)
    for sgm in range(C5_SEGMENTS):
        for d in range(3):
            for j in range(m):
                w = (sgm + j / float(C5_ORDER)) / C5_SEGMENTS
                base = [x0 + 12.0 * w, 0.0, 2.5][d]
                coef[sgm, d * m + j] = (base + (ctrl[sgm * m + j, d] if 0 < j < C5_ORDER else 0.0)) / seg_time[sgm]
    return (x0, 0.0, 2.5), nodes, coef, seg_time, orders
