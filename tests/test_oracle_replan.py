"""CPU tests of the checker side of config C5 (oracle.replan_tick / inflate_brute / brute_nearest_mt): the exhaustive-scan
formulation used by tests/test_gpu_ring.py must agree with the kd-tree formulation of oracle/corridor_port.c that the committed
fixtures were generated with (tests/golden/bezier_check.npz, inflate_c1.npz)."""
import numpy as np

from conftest import load_golden


def test_replan_tick_reproduces_the_bezier_fixture(oracle):
    g = load_golden("bezier_check.npz")
    pts = g["points"]
    for i in range(int(g["n_cases"])):
        r = oracle.replan_tick(pts, g["start"], float(g["sample_range"]), float(g["search_margin"]), float(g["max_radius"]), np.zeros((0, 3)),
                               g[f"case{i}_polycoef"], g["seg_time"], g["orders"], float(g[f"case{i}_t_start"]), float(g[f"case{i}_stop_time"]))
        assert r["nsamples"] == len(g[f"case{i}_pos"]) and r["first_hit_sample"] == int(g[f"case{i}_first_hit"])
        assert np.array_equal(r["sample_pos"], g[f"case{i}_pos"]) and np.array_equal(r["sample_radius"], g[f"case{i}_radius"])
        assert np.array_equal(r["sample_d2"], g[f"case{i}_d2"])
        # ties aside, the exhaustive scan and the kd-tree walk name the same point; on a tie the scan names the lowest index
        same = r["sample_idx"] == g[f"case{i}_idx"]
        d = pts[r["sample_idx"][~same]].astype(np.float64) - r["sample_pos"][~same].astype(np.float32).astype(np.float64)
        assert np.array_equal((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], r["sample_d2"][~same])
        assert np.all(r["sample_idx"][~same] < g[f"case{i}_idx"][~same])


def test_inflate_brute_equals_the_kdtree_formulation(oracle):
    from pointcloudtraj_amd import synth
    pts = synth.clustered_points(5, 4000, 0, 20)
    kd = oracle.PortKD()
    kd.insert(pts)
    q = np.concatenate([synth.uniform_points(6, 500, -3, 23).astype(np.float64) + 1e-6, pts[:50].astype(np.float64)])
    prm = oracle.corridor_params((10.0, 10.0, 10.0), 9.0, 0.25, 1.5)
    want_r, want_i, want_d, _ = oracle.inflate(kd, prm, q)
    got_r, got_i, got_d = oracle.inflate_brute(pts, (10.0, 10.0, 10.0), 9.0, 0.25, 1.5, q, threads=3)
    assert np.array_equal(got_r, want_r) and np.array_equal(got_d, want_d)
    assert (got_r == 1.25).sum() > 10                       # the early-out fired for some points
    far = np.isinf(got_d)
    assert np.all(got_i[far] == -1)
    tie_free = ~far & (got_i == want_i)
    assert tie_free.sum() > 0.3 * (~far).sum()              # the clustered cloud has exact ties; the scan takes the lowest index there
    assert np.all(got_i[~far & ~tie_free] < want_i[~far & ~tie_free])
    kd.close()


def test_control_point_list_follows_the_segment_search(oracle):
    """control points in world units = coefficient * T_i, segments from the one holding t_start on (checkSafeTrajectory's search)"""
    coef = np.arange(3 * 3 * 4, dtype=np.float64).reshape(3, 12) * 0.01 + 1.0       # 3 segments, order 3
    T = np.float64([0.5, 1.0, 2.0])
    orders = np.int32([3, 3, 3])
    pts = np.float32([[100, 100, 100]])
    for t_start, first in ((0.0, 0), (0.5, 0), (0.6, 1), (1.7, 2), (99.0, 2)):
        r = oracle.replan_tick(pts, (0, 0, 0), 1e9, 0.25, 1.5, np.zeros((0, 3)), coef, T, orders, t_start, 1.0)
        assert r["nctrl"] == 4 * (3 - first)
        want = np.concatenate([np.stack([coef[i, 0:4], coef[i, 4:8], coef[i, 8:12]], 1) * T[i] for i in range(first, 3)])
        assert np.array_equal(r["ctrl_pos"], want)
