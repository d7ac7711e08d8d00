"""Bezier evaluators beside the collision check: getStateFromBezier, the wire layout of
PolynomialTrajectoryExtra, the 1001-samples-per-segment walks of traj_postprocessing.cpp and its yaw block.

CPU part: the oracle restatement (oracle/traj_port.c) against properties that do not depend on it (the already pinned
position evaluator, finite differences, Bezier end-point interpolation, monotonicity).
GPU part: traj.hip through include/pct_traj.h against the oracle -- discrete results exactly, positions to 1e-12 relative
(device pow vs glibc pow).  PARITY UNPINNED against the reference itself (Eigen/roscpp absent)."""
import os
import re
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from pointcloudtraj_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def trajectories():
    g = np.load(os.path.join(GOLD, "bezier_check.npz"))
    out = {"fixture_6_4_8": (g["polycoef"], g["seg_time"], g["orders"])}
    # orders 1..12 in one trajectory (binomial table edge: MAX_N = 13 rows), control points from the repo's own PRNG
    orders = np.int32([1, 2, 3, 5, 7, 10, 12])
    r = (synth.splitmix64(77, 3 * 13 * len(orders)) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    coef = np.zeros((len(orders), 3 * 13))
    k = 0
    for s, n in enumerate(orders):
        m = n + 1
        for d in range(3):
            base = s * 1.5 + np.linspace(0, 1.5, m)              # a path that advances ~1.5 per segment on every axis
            coef[s, d * m:(d + 1) * m] = base + (r[k:k + m] - 0.5) * 0.6
            k += m
    times = np.float64([0.7, 1.1, 0.9, 1.6, 1.2, 2.0, 1.4])
    out["orders_1_to_12"] = (coef / times[:, None], times, orders)
    return out


def wire_of(pc, times, orders):
    cx, cy, cz = O.traj_wire_from_matrix(pc, orders)
    return SimpleNamespace(coef_x=cx, coef_y=cy, coef_z=cz, time=np.asarray(times, np.float64), order=np.asarray(orders, np.uint32))


def close(a, b, rel=1e-12):
    scale = max(1.0, float(np.max(np.abs(b)))) if np.size(b) else 1.0
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) <= rel * scale if np.size(b) else True


# ------------------------------------------------------------------------------------------------ CPU
def test_oracle_state_position_is_the_pinned_evaluator_and_derivatives_are_consistent():
    O.build()
    g = np.load(os.path.join(GOLD, "bezier_check.npz"))
    pc, orders, us = g["polycoef"], g["orders"], g["eval_u"]
    for s in range(3):
        st = O.traj_state(pc, orders, [s] * len(us), us)
        assert np.array_equal(st[:, :3], g["eval_pos"][s])            # same sums as getPosFromBezier's fixture
    for name, (pc, times, orders) in trajectories().items():
        for s in range(len(orders)):
            u = np.linspace(0.05, 0.95, 7)
            h = 1e-5
            mid, lo, hi = (O.traj_state(pc, orders, [s] * len(u), u + d) for d in (0.0, -h, h))
            assert np.allclose((hi[:, :3] - lo[:, :3]) / (2 * h), mid[:, 3:6], rtol=0, atol=1e-6 * max(1, np.abs(mid[:, 3:6]).max()))
            assert np.allclose((hi[:, 3:6] - lo[:, 3:6]) / (2 * h), mid[:, 6:9], rtol=0, atol=1e-5 * max(1, np.abs(mid[:, 6:9]).max()))


@pytest.mark.parametrize("name", list(trajectories()))
def test_oracle_wire_layout_and_walks(name):
    O.build()
    pc, times, orders = trajectories()[name]
    w = wire_of(pc, times, orders)
    assert len(w.coef_x) == int(np.sum(orders + 1))
    shift = np.concatenate([[0], np.cumsum(orders + 1)])
    for s, n in enumerate(orders):
        m = n + 1
        assert np.array_equal(w.coef_x[shift[s]:shift[s] + m], pc[s, :m]) and np.array_equal(w.coef_z[shift[s]:shift[s] + m], pc[s, 2 * m:3 * m])
    pos, step = O.traj_wire_sample(w, 1001)
    for s, n in enumerate(orders):                                   # a Bezier curve starts / ends on its end control points
        first, last = pos[s * 1001], pos[s * 1001 + 1000]
        c0 = np.array([w.coef_x[shift[s]], w.coef_y[shift[s]], w.coef_z[shift[s]]]) * times[s]
        c1 = np.array([w.coef_x[shift[s] + n], w.coef_y[shift[s] + n], w.coef_z[shift[s] + n]]) * times[s]
        assert np.array_equal(first, c0) and np.array_equal(last, c1)
    assert step[0] == 0.0 and np.all(step >= 0)
    total = float(step.sum())
    prev = (0, 0)
    for frac in (1e-9, 0.1, 0.3, 0.5, 0.8, 0.999):
        cur = O.traj_segm_index(w, frac * total)
        assert cur >= prev                                          # (segment, half) never moves backwards as the length grows
        prev = cur
    assert O.traj_segm_index(w, total * 2) == (len(orders) - 1, 1)
    assert O.traj_segm_index(w, 1e-9) == (0, 0)
    cloud, used = O.traj_nearest_traj(w, 0.1, 0.5 * total)
    acc = np.cumsum(step)
    assert used == int(np.searchsorted(acc, 0.5 * total, side="right")) + 1 or abs(used - np.searchsorted(acc, 0.5 * total)) <= 1
    assert len(cloud) <= used and len(np.unique(np.round(cloud / 0.1).astype(np.int64), axis=0)) == len(cloud)


def test_oracle_end_yaws_cases():
    O.build()
    px, py = np.float64([0, 1, 1, 1.001, 3]), np.float64([0, 0, 2, 2.001, 2])
    y = O.traj_end_yaws(px, py, [0.0, 1.0], [0.0, 1.0])
    assert y[0] == 0.0 and y[1] == np.arctan2(2.0, 0.0) and y[2] == 10 and y[3] == np.arctan2(py[4] - py[3], px[4] - px[3]) and y[4] == y[3]
    assert O.traj_end_yaws([5.0], [5.0], [0.0, 0.0], [0.0, 2.0])[0] == np.arctan2(2.0, 0.0)
    assert O.traj_end_yaws([5.0], [5.0], [0.0, 0.001], [0.0, 0.001])[0] == 10


def test_abi_declares_and_exports_traj_symbols():
    from pointcloudtraj_amd import build
    build.build_all()
    hdr = open(os.path.join(ROOT, "include", "pct_traj.h")).read()
    names = sorted(set(re.findall(r"\bint\s+(pct_\w+)\s*\(", hdr)))
    assert len(names) == 7, names          # six entry points + the pct_debug_binomials test hook
    out = subprocess.run(["nm", "-D", "--defined-only", build.ENGINE_SO], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert not [n for n in names if n not in exported]


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def T():
    import torch  # noqa: F401  (first: one HIP runtime per process)
    from pointcloudtraj_amd import engine as E, traj
    E.init(0)
    O.build()
    return traj


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(trajectories()))
def test_state_wire_and_samples_match_oracle(T, name):
    pc, times, orders = trajectories()[name]
    seg = np.repeat(np.arange(len(orders)), 9).astype(np.int32)
    u = np.tile(np.float64([0.0, 0.03, 0.25, 0.5, 0.5000001, 0.77, 0.999, 1.0, 0.1]), len(orders))
    got = T.get_state_from_bezier(pc, times, orders, seg, u)
    want = O.traj_state(pc, orders, seg, u)
    for blk in (slice(0, 3), slice(3, 6), slice(6, 9)):
        assert close(got[:, blk], want[:, blk])
    w = T.get_bezier_traj_wire(pc, times, orders)
    ow = wire_of(pc, times, orders)
    assert np.array_equal(w.coef_x, ow.coef_x) and np.array_equal(w.coef_y, ow.coef_y) and np.array_equal(w.coef_z, ow.coef_z)
    for samples in (1001, 11):
        pos, step = T.wire_sample(w, samples)
        opos, ostep = O.traj_wire_sample(ow, samples)
        assert close(pos, opos) and np.max(np.abs(step - ostep)) <= 1e-12 * max(1.0, np.abs(opos).max())
        assert np.array_equal(pos[::samples], opos[::samples])           # segment starts are exact (single non-zero term)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(trajectories()))
def test_walks_match_oracle(T, name):
    from pointcloudtraj_amd import voxel
    pc, times, orders = trajectories()[name]
    w = T.get_bezier_traj_wire(pc, times, orders)
    total = float(O.traj_wire_sample(w, 1001)[1].sum())
    for twirl in (1e-9, 0.05 * total, 0.31 * total, 0.5 * total, 0.77 * total, 1.5, total * 3):
        assert T.get_segm_index(w, twirl) == O.traj_segm_index(w, twirl)
    vm = voxel.VoxelMap(0.1, 4096)
    for res, twirl in ((0.1, 1.5), (0.1, 0.4 * total), (0.25, total * 2), (0.1, 0.0)):
        m = vm if res == 0.1 else None
        cloud, used = T.to_nearest_traj(w, res, twirl, m)
        ocloud, oused = O.traj_nearest_traj(w, res, twirl)
        assert used == oused and np.array_equal(cloud, ocloud)
    vm.close()
    rng = np.random.default_rng(5)
    px, py = rng.normal(0, 2, 9), rng.normal(0, 2, 9)
    px[4], py[4] = px[3] + 1e-3, py[3] - 1e-3                             # a step shorter than 0.01 -> marker 10
    assert np.array_equal(T.end_yaws(px, py, w.coef_x, w.coef_y), O.traj_end_yaws(px, py, w.coef_x, w.coef_y))
    assert np.array_equal(T.end_yaws(px[:1], py[:1], w.coef_x, w.coef_y), O.traj_end_yaws(px[:1], py[:1], w.coef_x, w.coef_y))


@pytest.mark.gpu
def test_traj_rejects_bad_input_loudly(T):
    from pointcloudtraj_amd import engine as E
    pc, times, orders = trajectories()["fixture_6_4_8"]
    with pytest.raises(E.EngineError):
        T.get_state_from_bezier(pc, times, orders, [3], [0.5])              # segment out of range
    with pytest.raises(E.EngineError):
        T.get_state_from_bezier(pc, times, np.int32([6, 4, 13]), [0], [0.5])   # order beyond the binomial table
    w = T.get_bezier_traj_wire(pc, times, orders)
    w.coef_x = w.coef_x[:-1]
    with pytest.raises(E.EngineError):
        T.get_segm_index(w, 1.0)                                             # fewer control points than the orders need
