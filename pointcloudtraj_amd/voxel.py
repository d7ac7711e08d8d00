"""Voxel de-duplication in front of the obstacle cloud: ctypes over include/pct_voxel.h (libpct_engine.so).

Mirrors the reference's containers (Planner/include/pointcloudTraj/voxel_map.h:10-45):
    VoxelMap(res).add_point_cloud(points)   voxel_map<Cont>::add_point_cloud      (voxel_map.cpp:24-33)
    VoxelMap.add_points(points)             per-point results of add_point        (:35-44) / voxel_value_map::add_point (:62-72)
    VoxelMap.get_voxel_cloud(dtype)         get_voxel_cloud                       (:46-49, 74-76)
    to_voxel_cloud(points, res)             voxel_map<Cont>::to_voxel_cloud       (:51-57)
Everything runs in the HIP kernels of voxel.hip; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import engine as E


def _lib():
    L = E.lib()
    if not getattr(L, "_voxel_bound", False):
        vp, i64 = C.c_void_p, C.c_int64
        L.pct_voxel_map_create.argtypes = [C.c_double, i64, C.POINTER(vp)]
        L.pct_voxel_map_destroy.argtypes = [vp]
        L.pct_voxel_map_clear.argtypes = [vp]
        L.pct_voxel_map_size.argtypes = [vp, C.POINTER(i64)]
        L.pct_voxel_map_add.argtypes = [vp, vp, i64, i64, C.c_int, C.POINTER(i64), vp, vp]
        L.pct_voxel_map_add_dev.argtypes = [vp, vp, i64, i64, C.c_int, C.POINTER(i64), vp, vp]
        L.pct_voxel_map_get_f32.argtypes = [vp, i64, i64, vp, i64]
        L.pct_voxel_map_get_f64.argtypes = [vp, i64, i64, vp]
        L.pct_voxel_map_get_keys.argtypes = [vp, i64, i64, vp]
        L.pct_voxel_map_soa_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
        L.pct_voxel_map_last_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L._voxel_bound = True
    return L


class VoxelMap:
    def __init__(self, res: float, capacity_hint: int = 0):
        h = C.c_void_p()
        E._chk(_lib().pct_voxel_map_create(float(res), int(capacity_hint), C.byref(h)))
        self._h = h
        self.res = float(res)

    def close(self):
        if getattr(self, "_h", None):
            _lib().pct_voxel_map_destroy(self._h)
            self._h = None

    __del__ = close

    def __len__(self) -> int:
        n = C.c_int64()
        E._chk(_lib().pct_voxel_map_size(self._h, C.byref(n)))
        return n.value

    def clear(self):
        E._chk(_lib().pct_voxel_map_clear(self._h))

    @staticmethod
    def _records(points):
        a = np.asarray(points)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        a = np.ascontiguousarray(a)
        if a.ndim != 2 or a.shape[1] < 3:
            raise ValueError("points must be [n, >=3]")
        return a, a.strides[0] if len(a) else a.shape[1] * a.itemsize, int(a.dtype == np.float64)

    def add_point_cloud(self, points) -> int:
        """Adds every point; returns how many voxels were new."""
        a, stride, f64 = self._records(points)
        n_new = C.c_int64()
        E._chk(_lib().pct_voxel_map_add(self._h, a.ctypes.data, len(a), stride, f64, C.byref(n_new), None, None))
        return n_new.value

    def add_points(self, points):
        """Returns (n_new, is_new[n] bool, voxel_index[n] int32): the sequential add_point results for each point."""
        a, stride, f64 = self._records(points)
        n_new = C.c_int64()
        is_new = np.zeros(len(a), np.uint8)
        index = np.zeros(len(a), np.int32)
        E._chk(_lib().pct_voxel_map_add(self._h, a.ctypes.data, len(a), stride, f64, C.byref(n_new), is_new.ctypes.data,
                                        index.ctypes.data))
        return n_new.value, is_new.astype(bool), index

    def add_device(self, ptr: int, n: int, stride_bytes: int, is_f64: bool = False) -> int:
        n_new = C.c_int64()
        E._chk(_lib().pct_voxel_map_add_dev(self._h, ptr, int(n), int(stride_bytes), int(is_f64), C.byref(n_new), None, None))
        return n_new.value

    def get_voxel_cloud(self, dtype=np.float32, first: int = 0, count: int | None = None) -> np.ndarray:
        count = len(self) - first if count is None else count
        if np.dtype(dtype) == np.float32:
            out = np.zeros((count, 3), np.float32)
            E._chk(_lib().pct_voxel_map_get_f32(self._h, first, count, out.ctypes.data, 3))
        else:
            out = np.zeros((count, 3), np.float64)
            E._chk(_lib().pct_voxel_map_get_f64(self._h, first, count, out.ctypes.data))
        return out

    def keys(self) -> np.ndarray:
        out = np.zeros((len(self), 3), np.int32)
        E._chk(_lib().pct_voxel_map_get_keys(self._h, 0, len(self), out.ctypes.data))
        return out

    def to_cloud(self, cloud: "E.Cloud"):
        """Make the de-duplicated voxel cloud the obstacle cloud, device to device (pct_cloud_upload_soa_dev)."""
        x, y, z, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        E._chk(_lib().pct_voxel_map_soa_dev(self._h, C.byref(x), C.byref(y), C.byref(z), C.byref(n)))
        cloud.set_input_device(x.value, y.value, z.value, n.value)

    def last_ms(self) -> float:
        ms = C.c_float()
        E._chk(_lib().pct_voxel_map_last_ms(self._h, C.byref(ms)))
        return ms.value


def to_voxel_cloud(points, res: float, dtype=None) -> np.ndarray:
    """voxel_map<Cont>::to_voxel_cloud (voxel_map.cpp:51-57): same container type out as in."""
    a = np.asarray(points)
    m = VoxelMap(res, len(a))
    try:
        m.add_point_cloud(a)
        return m.get_voxel_cloud(dtype or (np.float64 if a.dtype == np.float64 else np.float32))
    finally:
        m.close()
