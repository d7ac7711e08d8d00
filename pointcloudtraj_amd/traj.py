"""Bezier-trajectory evaluators beside the collision check: ctypes over include/pct_traj.h (libpct_engine.so).

Mirrors (names and argument meaning) the reference's free functions:
    get_state_from_bezier(poly_coeff, orders, t_now, seg_now)   sim_planning_demo.cpp:688-713 (batched over samples)
    get_bezier_traj_wire(poly_coeff, orders)                    sim_planning_demo.cpp:543-562 (PolyCoeff -> coef_x/y/z)
    get_segm_index(wire, twirl_len)                             traj_postprocessing.cpp:29-57
    to_nearest_traj(wire, res, twirl_len)                       traj_postprocessing.cpp:59-90 (returns the voxel cloud)
    end_yaws(path_x, path_y, coef_x, coef_y)                    traj_postprocessing.cpp:152-179
No CPU fallback: the evaluation runs in traj.hip's kernels."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import engine as E
from . import voxel as V


class _BezierTraj(C.Structure):
    _fields_ = [("polycoef", C.c_void_p), ("row_stride", C.c_int64), ("seg_time", C.c_void_p), ("orders", C.c_void_p), ("nseg", C.c_int32)]


class _Wire(C.Structure):
    _fields_ = [("coef_x", C.c_void_p), ("coef_y", C.c_void_p), ("coef_z", C.c_void_p), ("ncoef", C.c_int64), ("time", C.c_void_p),
                ("order", C.c_void_p), ("num_segment", C.c_int32)]


@dataclass
class WireTraj:
    """quadrotor_msgs/PolynomialTrajectoryExtra's trajectory fields (msg:19-31)"""
    coef_x: np.ndarray
    coef_y: np.ndarray
    coef_z: np.ndarray
    time: np.ndarray
    order: np.ndarray

    def __post_init__(self):
        self.coef_x = np.ascontiguousarray(self.coef_x, np.float64)
        self.coef_y = np.ascontiguousarray(self.coef_y, np.float64)
        self.coef_z = np.ascontiguousarray(self.coef_z, np.float64)
        self.time = np.ascontiguousarray(self.time, np.float64)
        self.order = np.ascontiguousarray(self.order, np.uint32)

    @property
    def num_segment(self) -> int:
        return len(self.time)

    def _c(self) -> _Wire:
        return _Wire(self.coef_x.ctypes.data, self.coef_y.ctypes.data, self.coef_z.ctypes.data, len(self.coef_x), self.time.ctypes.data,
                     self.order.ctypes.data, self.num_segment)


def _lib():
    L = E.lib()
    if not getattr(L, "_traj_bound", False):
        vp, i64 = C.c_void_p, C.c_int64
        L.pct_bezier_state_batch.argtypes = [C.POINTER(_BezierTraj), vp, vp, i64, vp]
        L.pct_traj_wire_from_matrix.argtypes = [C.POINTER(_BezierTraj), vp, vp, vp, i64, C.POINTER(i64)]
        L.pct_traj_wire_sample.argtypes = [C.POINTER(_Wire), C.c_int32, vp, vp]
        L.pct_traj_segm_index.argtypes = [C.POINTER(_Wire), C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.pct_traj_nearest_voxels.argtypes = [C.POINTER(_Wire), C.c_double, vp, C.POINTER(i64)]
        L.pct_traj_end_yaws.argtypes = [vp, vp, i64, vp, vp, vp]
        L._traj_bound = True
    return L


def _matrix(poly_coeff, seg_time, orders):
    pc = np.ascontiguousarray(poly_coeff, np.float64)
    st = np.ascontiguousarray(seg_time, np.float64)
    od = np.ascontiguousarray(orders, np.int32)
    t = _BezierTraj(pc.ctypes.data, pc.shape[1], st.ctypes.data, od.ctypes.data, len(od))
    return t, (pc, st, od)


def get_state_from_bezier(poly_coeff, seg_time, orders, seg_now, t_now) -> np.ndarray:
    """[n, 9]: position / velocity / acceleration sums of getStateFromBezier for every (seg_now[i], t_now[i])."""
    t, keep = _matrix(poly_coeff, seg_time, orders)
    seg = np.ascontiguousarray(np.atleast_1d(seg_now), np.int32)
    u = np.ascontiguousarray(np.atleast_1d(t_now), np.float64)
    out = np.zeros((len(seg), 9))
    E._chk(_lib().pct_bezier_state_batch(C.byref(t), seg.ctypes.data, u.ctypes.data, len(seg), out.ctypes.data))
    return out


def get_bezier_traj_wire(poly_coeff, seg_time, orders) -> WireTraj:
    t, keep = _matrix(poly_coeff, seg_time, orders)
    total = int(np.sum(np.asarray(orders, np.int64) + 1))
    cx, cy, cz = np.zeros(total), np.zeros(total), np.zeros(total)
    n = C.c_int64()
    E._chk(_lib().pct_traj_wire_from_matrix(C.byref(t), cx.ctypes.data, cy.ctypes.data, cz.ctypes.data, total, C.byref(n)))
    return WireTraj(cx, cy, cz, np.asarray(seg_time, np.float64), np.asarray(orders, np.uint32))


def wire_sample(w: WireTraj, samples: int = 1001):
    """(pos [S*samples, 3], step_len [S*samples]) of the per-segment sampling loops"""
    total = w.num_segment * samples
    pos, step = np.zeros((total, 3)), np.zeros(total)
    cw = w._c()
    E._chk(_lib().pct_traj_wire_sample(C.byref(cw), samples, pos.ctypes.data, step.ctypes.data))
    return pos, step


def get_segm_index(w: WireTraj, twirl_len: float) -> tuple[int, int]:
    segm, part = C.c_int32(), C.c_int32()
    cw = w._c()
    E._chk(_lib().pct_traj_segm_index(C.byref(cw), float(twirl_len), C.byref(segm), C.byref(part)))
    return segm.value, part.value


def to_nearest_traj(w: WireTraj, res: float, twirl_len: float, vmap: "V.VoxelMap | None" = None):
    """(voxel cloud float32 [k, 3], samples used).  Pass a VoxelMap of resolution `res` to reuse its buffers."""
    own = vmap is None
    vmap = vmap or V.VoxelMap(res, 4096)
    try:
        used = C.c_int64()
        cw = w._c()
        E._chk(_lib().pct_traj_nearest_voxels(C.byref(cw), float(twirl_len), vmap._h, C.byref(used)))
        return vmap.get_voxel_cloud(np.float32), used.value
    finally:
        if own:
            vmap.close()


def end_yaws(path_x, path_y, coef_x, coef_y) -> np.ndarray:
    px, py = np.ascontiguousarray(path_x, np.float64), np.ascontiguousarray(path_y, np.float64)
    cx, cy = np.ascontiguousarray(coef_x, np.float64), np.ascontiguousarray(coef_y, np.float64)
    out = np.zeros(len(px))
    E._chk(_lib().pct_traj_end_yaws(px.ctypes.data, py.ctypes.data, len(px), cx.ctypes.data, cy.ctypes.data, out.ctypes.data))
    return out
