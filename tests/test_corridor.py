"""Corridor finder: CPU oracle sanity on config C1, and -- on the GPU -- the engine-backed
finder against the oracle (two independent implementations, no shared code) for the same seed and iteration counts."""
import numpy as np
import pytest

from pointcloudtraj_amd.scenarios import GOAL, START, run_commit_scenario, run_scenario, sensed_cloud


def check_corridor(path, radius, cloud, safety=0.6):
    """a corridor is a chain of overlapping safe spheres from the start to the goal"""
    assert len(path) >= 2
    assert np.linalg.norm(path[0] - np.float64(START)) < 1e-9
    assert np.linalg.norm(path[-1] - np.float64(GOAL)) + 0.1 < radius[-1]            # checkEnd, corridor_finder.cpp:418-426
    assert np.all(radius >= np.float32(safety))
    for a, b, ra, rb in zip(path[:-1], path[1:], radius[:-1], radius[1:]):
        assert np.linalg.norm(a - b) + 0.1 < 0.95 * (np.float32(ra) + np.float32(rb)) + 1e-6   # checkNodeRelation == -1
    c64 = cloud.astype(np.float64)
    for p, r in zip(path, radius):                                                   # every sphere is free of obstacle points
        assert np.sqrt(((c64 - p) ** 2).sum(1).min()) >= r + 0.25 - 1e-6


def test_oracle_corridor_c1(oracle):
    from pointcloudtraj_amd.scenarios import perturbed_cloud
    cloud1 = sensed_cloud(12.0)
    for cloud2 in (None, sensed_cloud(16.0)):       # a mild change (corridor survives, radii shrink) and a drastic one
        f = oracle.PortCorridor()
        phases = run_scenario(f, cloud1, cloud2)
        (p0, r0, s0), (p1, r1, s1), (p2, r2, s2), (p3, r3, s3) = phases
        c2 = perturbed_cloud(cloud1, p1) if cloud2 is None else cloud2
        assert s0["path_exists"] and s1["path_exists"]
        check_corridor(p0, r0, cloud1)
        check_corridor(p1, r1, cloud1)
        assert s0["nodes"] > 20 and s0["inflation_queries"] > 500
        if cloud2 is None:
            assert s2["path_exists"], "the mild perturbation must leave a corridor"
            assert not np.array_equal(r1, r2) or not np.array_equal(p1, p2)
        if s2["path_exists"]:
            check_corridor(p2, r2, c2)
        if s3["path_exists"]:
            check_corridor(p3, r3, c2)
        # deterministic: a second run reproduces the corridor bit for bit
        again = run_scenario(oracle.PortCorridor(), cloud1, cloud2)
        for (pa, ra, _), (pb, rb, _) in zip(phases, again):
            assert np.array_equal(pa, pb) and np.array_equal(ra, rb)


@pytest.mark.gpu
@pytest.mark.parametrize("speculation,fused", [(1, True), (8, True), (64, True), (256, True), (8, False), (64, False), (256, False)])
def test_gpu_corridor_matches_oracle(oracle, speculation, fused):
    """speculation = samples per GPU round trip, fused = one launch per batch (nearest -> steer -> inflation -> range in one
    kernel) or three; every setting must give the one-by-one corridor"""
    from pointcloudtraj_amd import corridor, engine
    engine.init(0)
    cloud1 = sensed_cloud(12.0)
    for cloud2 in (None, sensed_cloud(16.0)):
        want = run_scenario(oracle.PortCorridor(), cloud1, cloud2, expand=600, refine=200)
        finder = corridor.SafeRegionRrtStar(80000)
        finder.setSpeculation(speculation)
        finder.setFusedExpansion(fused)
        got = run_scenario(finder, cloud1, cloud2, expand=600, refine=200)
        if speculation > 1:
            st = finder.speculationStats()
            assert st["replayed_from_batch"] > 100, st
            assert (finder.expansionLaunches() > 0) == fused
        for k, ((pw, rw, sw), (pg, rg, sg)) in enumerate(zip(want, got)):
            assert sw["path_exists"] == sg["path_exists"] and sw["nodes"] == sg["nodes"], f"phase {k}: {sw} vs {sg}"
            assert np.array_equal(pw, pg), f"phase {k}: corridor centres differ"
            assert np.array_equal(rw, rg), f"phase {k}: corridor radii differ"
            assert sw["inflation_queries"] == sg["inflation_queries"]
        assert want[0][2]["path_exists"]
        if cloud2 is not None:
            # the drastically different second frame invalidates corridor nodes: SafeRegionEvaluate hands them to treeRepair, which now
            # makes two GPU round trips per pass whatever the number of neighbours (it used to make two per neighbour)
            assert 0 < finder.repairBatches() <= 8, finder.repairBatches()


def test_oracle_commit_scenario_moves_the_root(oracle):
    """the commit scenario really exercises resetRoot: the root moves, nodes behind it are cut, and the run is deterministic"""
    cloud1 = sensed_cloud(12.0)
    phases = run_commit_scenario(oracle.PortCorridor(), cloud1)
    assert len(phases) >= 4, "at least one commit must have happened"
    assert phases[0][2]["path_exists"]
    first_path = phases[0][0]
    moved = [ph for ph in phases[2:] if ph[2]["path_exists"] and not np.array_equal(ph[0][0], first_path[0])]
    assert moved, "after a commit the corridor must start at the new root"
    again = run_commit_scenario(oracle.PortCorridor(), cloud1)
    for (pa, ra, sa), (pb, rb, sb) in zip(phases, again):
        assert np.array_equal(pa, pb) and np.array_equal(ra, rb) and sa == sb


@pytest.mark.gpu
@pytest.mark.parametrize("speculation", [1, 64, 256])
def test_gpu_commit_scenario_matches_oracle(oracle, speculation):
    """resetRoot / setStartPt / Refine / Evaluate in the planner's incremental order: corridor, radii, node and query counts equal the
    independent CPU restatement's after every phase"""
    from pointcloudtraj_amd import corridor, engine
    engine.init(0)
    cloud1 = sensed_cloud(12.0)
    want = run_commit_scenario(oracle.PortCorridor(), cloud1)
    finder = corridor.SafeRegionRrtStar(80000)
    finder.setSpeculation(speculation)
    got = run_commit_scenario(finder, cloud1)
    assert len(want) == len(got) >= 4
    for k, ((pw, rw, sw), (pg, rg, sg)) in enumerate(zip(want, got)):
        assert sw["path_exists"] == sg["path_exists"] and sw["global_navi"] == sg["global_navi"] and sw["nodes"] == sg["nodes"], f"phase {k}: {sw} vs {sg}"
        assert np.array_equal(pw, pg), f"phase {k}: corridor centres differ"
        assert np.array_equal(rw, rg), f"phase {k}: corridor radii differ"
        assert sw["inflation_queries"] == sg["inflation_queries"], f"phase {k}"
