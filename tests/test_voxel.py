"""Voxel de-duplication (reference Planner/src/voxel_map.cpp:5-76).

CPU part: the oracle's sequential restatement (oracle/voxel_port.c) against an independent numpy formulation.
GPU part: the HIP path (voxel.hip through include/pct_voxel.h) against the oracle, element by element: voxel order,
integer coordinates, float and double centres, per-point add_point results and voxel_value_map indices.
PARITY UNPINNED against the reference itself (voxel_map.cpp needs PCL/Eigen, absent; the reference holds no fixture)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from pointcloudtraj_amd import synth  # noqa: E402


def round_half_away(v):
    return np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5))


def numpy_voxels(pts, res):
    """first-seen voxel keys, per-point voxel index, per-point is_new -- without any sequential container"""
    k = round_half_away(pts.astype(np.float64) / res).astype(np.int64)
    uniq, first, inv = np.unique(k, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")              # voxels in the order of their first occurrence
    rank = np.empty(len(order), np.int64)
    rank[order] = np.arange(len(order))
    index = rank[inv.reshape(-1)]
    is_new = np.zeros(len(pts), bool)
    is_new[first] = True
    return uniq[order].astype(np.int32), index.astype(np.int32), is_new


def clouds():
    rng = np.random.default_rng(11)
    half = np.array([[0.25, -0.25, 0.75], [-0.75, 1.25, -1.25], [0.25, -0.25, 0.75], [0.0, -0.0, 0.5]])   # exact .5 at res 0.5
    return {
        "uniform_f32": (synth.uniform_points(31, 20000, -8, 8), 0.25),
        "pillars_f32": (synth.clustered_points(32, 30000, 0, 30), 0.1),          # already on a 0.1 lattice: many duplicates
        "normal_f64": (rng.normal(0, 3, (15000, 3)), 0.2),
        "half_cases_f64": (np.concatenate([half, half[::-1]]), 0.5),
        "negative_f32": (-synth.uniform_points(33, 5000, 0, 4), 0.05),
    }


@pytest.mark.parametrize("name", list(clouds()))
def test_oracle_matches_numpy_formulation(name):
    O.build()
    pts, res = clouds()[name]
    m = O.PortVoxelMap(res)
    n_new, is_new, index = m.add(pts)
    keys, want_index, want_new = numpy_voxels(pts, res)
    assert n_new == len(keys) == len(m)
    assert np.array_equal(m.keys(), keys)
    assert np.array_equal(index, want_index) and np.array_equal(is_new.astype(bool), want_new)
    assert np.array_equal(m.cloud_f64(), keys * res)
    assert np.array_equal(m.cloud_f32(), (keys * res).astype(np.float32))
    # adding the same cloud again changes nothing; adding in two halves equals adding at once
    assert m.add(pts)[0] == 0 and len(m) == len(keys)
    m2 = O.PortVoxelMap(res)
    a = m2.add(pts[: len(pts) // 3])[0]
    b = m2.add(pts[len(pts) // 3:])[0]
    assert a + b == len(keys) and np.array_equal(m2.keys(), keys)


def test_abi_declares_and_exports_voxel_symbols():
    import ctypes
    import re
    from pointcloudtraj_amd import build
    build.build_all()
    hdr = open(os.path.join(ROOT, "include", "pct_voxel.h")).read()
    names = sorted(set(re.findall(r"\b(pct_voxel_map_\w+)\s*\(", hdr)))
    assert len(names) == 11, names
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", build.ENGINE_SO], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert not [n for n in names if n not in exported]
    assert not [s for s in exported if s.startswith("_ZN12pct_internal")], "internal helpers must stay hidden"


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def V():
    import torch  # noqa: F401  (first: one HIP runtime per process)
    from pointcloudtraj_amd import engine as E, voxel
    E.init(0)
    O.build()
    return voxel


def _compare(vm, om, pts_batches):
    for pts in pts_batches:
        n_new, is_new, index = vm.add_points(pts)
        want_new, want_is_new, want_index = om.add(pts)
        assert n_new == want_new
        assert np.array_equal(is_new, want_is_new.astype(bool))
        assert np.array_equal(index, want_index)
    assert len(vm) == len(om)
    assert np.array_equal(vm.keys(), om.keys())
    assert np.array_equal(vm.get_voxel_cloud(np.float32), om.cloud_f32())
    assert np.array_equal(vm.get_voxel_cloud(np.float64), om.cloud_f64())


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(clouds()))
def test_voxel_map_matches_oracle(V, name):
    pts, res = clouds()[name]
    vm, om = V.VoxelMap(res), O.PortVoxelMap(res)
    _compare(vm, om, [pts])
    vm.close()


@pytest.mark.gpu
def test_voxel_map_incremental_growth_strides_and_clear(V):
    res = 0.1
    vm, om = V.VoxelMap(res, capacity_hint=16), O.PortVoxelMap(res)      # forces store growth and table rehashes
    frames = [synth.uniform_points(40 + k, 30000, k * 0.5, 6 + k * 0.5) for k in range(6)]    # overlapping sensor frames
    rec16 = np.zeros((len(frames[0]), 4), np.float32)                      # pcl::PointXYZ records (16-byte stride)
    rec16[:, :3] = frames[0]
    rec16[:, 3] = 1.0
    _compare(vm, om, [rec16] + frames[1:] + [frames[2].astype(np.float64), np.zeros((0, 3), np.float32)])
    assert vm.add_point_cloud(frames[3]) == 0                              # idempotent
    vm.clear()
    assert len(vm) == 0
    om2 = O.PortVoxelMap(res)
    _compare(vm, om2, [frames[4]])
    vm.close()


@pytest.mark.gpu
def test_voxel_map_rejects_out_of_range_loudly(V):
    from pointcloudtraj_amd import engine as E
    vm = V.VoxelMap(0.1)
    pts = np.array([[0.0, 0.0, 0.0], [2.0e5, 0.0, 0.0], [np.nan, 1.0, 1.0], [0.3, 0.3, 0.3]], np.float32)
    with pytest.raises(E.EngineError):
        vm.add_point_cloud(pts)
    assert len(vm) == 2            # the two valid points were added, the others skipped (documented)
    with pytest.raises(E.EngineError):
        V.VoxelMap(0.0)
    vm.close()


@pytest.mark.gpu
def test_voxel_cloud_feeds_the_obstacle_cloud_on_the_device(V):
    """rgbd-style accumulation (camera_sensor.cpp:160-166 keeps every frame's points) -> de-dup -> obstacle cloud -> NN"""
    from pointcloudtraj_amd import engine as E
    res = 0.1
    world = synth.pillar_map(6)[:60000]
    vm = V.VoxelMap(res)
    for k in range(4):
        vm.add_point_cloud(world[k * 10000: k * 10000 + 30000])           # overlapping frames: 2/3 duplicates
    want = V.to_voxel_cloud(world[:60000], res)
    got = vm.get_voxel_cloud()
    assert np.array_equal(got, want) and len(got) == 60000          # 120 000 points went in: half were repeats
    c = E.Cloud(len(vm))
    vm.to_cloud(c)
    c.build_grid()
    q = synth.uniform_points(50, 4096, -20, 20)
    idx, d2 = c.nn(q)
    wi, wd = O.brute_nearest(got, q)
    assert np.array_equal(d2, wd) and np.array_equal(idx, wi)
    c.close()
    vm.close()


@pytest.mark.gpu
def test_voxel_map_full_size_properties(V):
    """10 M points (config C3's cloud) at res 0.25: size equals the number of distinct keys, order is first-seen,
    a second pass adds nothing; checked with numpy instead of the sequential oracle (which would take ~10 s)."""
    pts = synth.uniform_points(3, 10_000_000, 0.0, 100.0)
    res = 0.25
    vm = V.VoxelMap(res, len(pts))
    n_new, is_new, index = vm.add_points(pts)
    keys, want_index, want_new = numpy_voxels(pts, res)
    assert n_new == len(keys)
    assert np.array_equal(vm.keys(), keys)
    assert np.array_equal(index, want_index) and np.array_equal(is_new, want_new)
    assert vm.add_point_cloud(pts[::7]) == 0
    vm.close()
