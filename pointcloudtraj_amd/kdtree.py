"""ctypes binding of the drop-in libkdtree.so (include/kdtree/kdtree.h): the reference's kd_*
API served by the GPU engine.  Used by the parity tests the way corridor_finder.cpp uses the
C API (kd_insertf / kd_nearestf / kd_nearest_rangef / kd_res_*)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from . import engine as _engine

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_build.KDTREE_SO):
            raise FileNotFoundError(f"{_build.KDTREE_SO} is missing: run __graft_entry__.build()")
        _engine.lib()      # loads the HIP runtime + libpct_engine.so first (same runtime as torch)
        L = C.CDLL(_build.KDTREE_SO)
        vp = C.c_void_p
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.kd_create.restype = vp
        L.kd_create.argtypes = [C.c_int]
        L.kd_free.argtypes = [vp]
        L.kd_clear.argtypes = [vp]
        L.kd_data_destructor.argtypes = [vp, vp]
        L.kd_insert.argtypes = [vp, dp, vp]
        L.kd_insertf.argtypes = [vp, fp, vp]
        L.kd_insert3.argtypes = [vp, C.c_double, C.c_double, C.c_double, vp]
        L.kd_insert3f.argtypes = [vp, C.c_float, C.c_float, C.c_float, vp]
        for n in ("kd_nearest", "kd_nearestf", "kd_nearest3", "kd_nearest3f", "kd_nearest_range", "kd_nearest_rangef",
                  "kd_nearest_range3", "kd_nearest_range3f"):
            getattr(L, n).restype = vp
        L.kd_nearest.argtypes = [vp, dp]
        L.kd_nearestf.argtypes = [vp, fp]
        L.kd_nearest3.argtypes = [vp, C.c_double, C.c_double, C.c_double]
        L.kd_nearest3f.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        L.kd_nearest_range.argtypes = [vp, dp, C.c_double]
        L.kd_nearest_rangef.argtypes = [vp, fp, C.c_float]
        L.kd_nearest_range3.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double]
        L.kd_nearest_range3f.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float]
        for n in ("kd_res_free", "kd_res_size", "kd_res_rewind", "kd_res_end", "kd_res_next"):
            getattr(L, n).argtypes = [vp]
        for n in ("kd_res_item", "kd_res_itemf", "kd_res_item3", "kd_res_item3f", "kd_res_item_data"):
            getattr(L, n).restype = vp
        L.kd_res_item.argtypes = [vp, dp]
        L.kd_res_itemf.argtypes = [vp, fp]
        L.kd_res_item3.argtypes = [vp, dp, dp, dp]
        L.kd_res_item3f.argtypes = [vp, fp, fp, fp]
        L.kd_res_item_data.argtypes = [vp]
        L.kdx_set_host_threshold.argtypes = [C.c_int64]
        L.kdx_host_threshold.restype = C.c_int64
        _lib = L
    return _lib


def set_host_threshold(nodes: int):
    """node sets of up to `nodes` nodes answer single queries on the host (0 = always the device; negative = the default)"""
    lib().kdx_set_host_threshold(int(nodes))


def host_threshold() -> int:
    return int(lib().kdx_host_threshold())


class KDTree:
    """Thin convenience wrapper; payload of point i is (void*)(i+1)."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.kd_create(3)
        if not self.h:
            raise RuntimeError("kd_create failed (no HIP device?)")
        self.n = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.kd_free(self.h)
            self.h = None

    __del__ = close

    def insert(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        for p in xyz:
            if self.L.kd_insertf(self.h, p.ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(self.n + 1)):
                raise MemoryError
            self.n += 1

    def nearest(self, q):
        """payload-derived ids and fp64 positions for each float query"""
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        ids = np.empty(len(q), np.int32)
        pos = np.empty((len(q), 3), np.float64)
        for i, qq in enumerate(q):
            r = self.L.kd_nearestf(self.h, qq.ctypes.data_as(C.POINTER(C.c_float)))
            if not r:
                raise RuntimeError("kd_nearestf returned NULL")
            d = self.L.kd_res_item(r, pos[i].ctypes.data_as(C.POINTER(C.c_double)))
            ids[i] = int(d or 0) - 1
            self.L.kd_res_free(r)
        return ids, pos

    def range_ids(self, q, r):
        q = np.ascontiguousarray(q, np.float32).reshape(3)
        rs = self.L.kd_nearest_rangef(self.h, q.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(r))
        if not rs:
            raise RuntimeError("kd_nearest_rangef returned NULL")
        out = []
        while not self.L.kd_res_end(rs):
            out.append(int(self.L.kd_res_item_data(rs) or 0) - 1)
            self.L.kd_res_next(rs)
        assert len(out) == self.L.kd_res_size(rs)
        self.L.kd_res_free(rs)
        return np.asarray(out, np.int32)


class KDTreeN:
    """kd_* on a tree of any dimension with double positions (kd_create(k), kd_insert, kd_nearest, kd_nearest_range);
    payload of row i is (void*)(i+1)."""

    def __init__(self, dim: int):
        self.L = lib()
        self.dim = int(dim)
        self.h = self.L.kd_create(self.dim)
        if not self.h:
            raise RuntimeError(f"kd_create({dim}) failed")
        self.n = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.kd_free(self.h)
            self.h = None

    __del__ = close

    def insert(self, rows):
        rows = np.ascontiguousarray(rows, np.float64).reshape(-1, self.dim)
        dp = C.POINTER(C.c_double)
        for p in rows:
            if self.L.kd_insert(self.h, p.ctypes.data_as(dp), C.c_void_p(self.n + 1)):
                raise MemoryError
            self.n += 1

    def nearest(self, q):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, self.dim)
        ids = np.empty(len(q), np.int32)
        pos = np.empty((len(q), self.dim), np.float64)
        dp = C.POINTER(C.c_double)
        for i, qq in enumerate(q):
            r = self.L.kd_nearest(self.h, qq.ctypes.data_as(dp))
            if not r:
                raise RuntimeError("kd_nearest returned NULL")
            d = self.L.kd_res_item(r, pos[i].ctypes.data_as(dp))
            ids[i] = int(d or 0) - 1
            self.L.kd_res_free(r)
        return ids, pos

    def range_ids(self, q, r):
        q = np.ascontiguousarray(q, np.float64).reshape(self.dim)
        rs = self.L.kd_nearest_range(self.h, q.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(r))
        if not rs:
            raise RuntimeError("kd_nearest_range returned NULL")
        out = []
        while not self.L.kd_res_end(rs):
            out.append(int(self.L.kd_res_item_data(rs) or 0) - 1)
            self.L.kd_res_next(rs)
        assert len(out) == self.L.kd_res_size(rs)
        self.L.kd_res_free(rs)
        return np.asarray(out, np.int32)
