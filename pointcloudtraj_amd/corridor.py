"""ctypes binding of libpct_corridor.so (include/pct_corridor.h): the safe-region RRT* corridor finder on
the engine, with the reference's method names (corridor_finder.h:81-149)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from . import engine as _engine
from . import kdtree as _kdtree

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_build.CORRIDOR_SO):
            raise FileNotFoundError(f"{_build.CORRIDOR_SO} is missing: run __graft_entry__.build()")
        _engine.lib()
        _kdtree.lib()
        L = C.CDLL(_build.CORRIDOR_SO)
        vp, d3 = C.c_void_p, C.POINTER(C.c_double)
        L.pct_corridor_last_error.restype = C.c_char_p
        L.pct_corridor_create.argtypes = [C.c_int64, C.c_int, C.POINTER(vp)]
        L.pct_corridor_destroy.argtypes = [vp]
        L.pct_corridor_set_param.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double]
        L.pct_corridor_reset.argtypes = [vp]
        L.pct_corridor_set_speculation.argtypes = [vp, C.c_int]
        L.pct_corridor_speculation_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.pct_corridor_set_fused_expansion.argtypes = [vp, C.c_int]
        L.pct_corridor_expansion_launches.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.pct_corridor_repair_batches.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.pct_corridor_set_input.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int]
        L.pct_corridor_set_pt.argtypes = [vp, d3, d3] + [C.c_double] * 7 + [C.c_int, C.c_double, C.c_double]
        L.pct_corridor_set_start_pt.argtypes = [vp, d3, d3]
        L.pct_corridor_reset_root.argtypes = [vp, d3]
        L.pct_corridor_expansion.argtypes = [vp, C.c_int64]
        L.pct_corridor_refine.argtypes = [vp, C.c_int64]
        L.pct_corridor_evaluate.argtypes = [vp]
        L.pct_corridor_expansion_timed.argtypes = [vp, C.c_double, C.POINTER(C.c_int64)]
        L.pct_corridor_refine_timed.argtypes = [vp, C.c_double, C.POINTER(C.c_int64)]
        L.pct_corridor_evaluate_timed.argtypes = [vp, C.c_double]
        L.pct_corridor_check_traj_pt_col.argtypes = [vp, d3, C.POINTER(C.c_int)]
        L.pct_corridor_get_path.argtypes = [vp, vp, vp, C.c_int64, C.POINTER(C.c_int64)]
        L.pct_corridor_status.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class SafeRegionRrtStar:
    def __init__(self, cloud_capacity: int = 1 << 20, device: int = 0):
        self.L = lib()
        self.h = C.c_void_p()
        self._chk(self.L.pct_corridor_create(int(cloud_capacity), device, C.byref(self.h)))

    def _chk(self, rc):
        if rc:
            raise RuntimeError("pct_corridor: " + self.L.pct_corridor_last_error().decode(errors="replace"))

    def close(self):
        if getattr(self, "h", None) and self.h.value and _lib is not None:
            _lib.pct_corridor_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def setParam(self, safety_margin, search_margin, max_radius, sample_range):
        self._chk(self.L.pct_corridor_set_param(self.h, safety_margin, search_margin, max_radius, sample_range))

    def reset(self):
        self._chk(self.L.pct_corridor_reset(self.h))

    def setSpeculation(self, k: int):
        self._chk(self.L.pct_corridor_set_speculation(self.h, int(k)))

    def setFusedExpansion(self, on: bool):
        self._chk(self.L.pct_corridor_set_fused_expansion(self.h, int(bool(on))))

    def expansionLaunches(self) -> int:
        n = C.c_uint64()
        self._chk(self.L.pct_corridor_expansion_launches(self.h, C.byref(n)))
        return n.value

    def repairBatches(self) -> int:
        n = C.c_uint64()
        self._chk(self.L.pct_corridor_repair_batches(self.h, C.byref(n)))
        return n.value

    def speculationStats(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.L.pct_corridor_speculation_stats(self.h, C.byref(a), C.byref(b)))
        return dict(replayed_from_batch=a.value, fell_back=b.value)

    def setInput(self, points, build_index=True):
        a = np.ascontiguousarray(points, np.float32)
        self._chk(self.L.pct_corridor_set_input(self.h, a.ctypes.data_as(C.c_void_p), len(a), a.shape[1] * 4, int(build_index)))

    def setPt(self, start, end, xl, xh, yl, yh, zl, zh, local_range, max_iter, sample_portion, goal_portion):
        self._chk(self.L.pct_corridor_set_pt(self.h, _d3(start), _d3(end), xl, xh, yl, yh, zl, zh, local_range, int(max_iter),
                                             sample_portion, goal_portion))

    def setStartPt(self, start, end):
        self._chk(self.L.pct_corridor_set_start_pt(self.h, _d3(start), _d3(end)))

    def resetRoot(self, target):
        self._chk(self.L.pct_corridor_reset_root(self.h, _d3(target)))

    # The reference's three entry points take seconds of wall clock (corridor_finder.h:97-99): pass a float.  An int is an
    # iteration count (the deterministic form, C++: ExpansionIterations / RefineIterations / EvaluateOnce).  The timed forms return
    # the number of samples they consumed.
    def ExpansionIterations(self, iterations: int):
        self._chk(self.L.pct_corridor_expansion(self.h, int(iterations)))

    def RefineIterations(self, iterations: int):
        self._chk(self.L.pct_corridor_refine(self.h, int(iterations)))

    def EvaluateOnce(self):
        self._chk(self.L.pct_corridor_evaluate(self.h))

    def SafeRegionExpansion(self, limit):
        if isinstance(limit, (int, np.integer)):
            return self.ExpansionIterations(limit)
        n = C.c_int64()
        self._chk(self.L.pct_corridor_expansion_timed(self.h, float(limit), C.byref(n)))
        return n.value

    def SafeRegionRefine(self, limit):
        if isinstance(limit, (int, np.integer)):
            return self.RefineIterations(limit)
        n = C.c_int64()
        self._chk(self.L.pct_corridor_refine_timed(self.h, float(limit), C.byref(n)))
        return n.value

    def SafeRegionEvaluate(self, time_limit=None):
        if time_limit is None:
            return self.EvaluateOnce()
        self._chk(self.L.pct_corridor_evaluate_timed(self.h, float(time_limit)))

    def checkTrajPtCol(self, p) -> bool:
        c = C.c_int()
        self._chk(self.L.pct_corridor_check_traj_pt_col(self.h, _d3(p), C.byref(c)))
        return bool(c.value)

    def getPath(self):
        n = C.c_int64()
        path = np.zeros((4096, 3)); rad = np.zeros(4096)
        self._chk(self.L.pct_corridor_get_path(self.h, path.ctypes.data_as(C.c_void_p), rad.ctypes.data_as(C.c_void_p), 4096, C.byref(n)))
        return path[:n.value].copy(), rad[:n.value].copy()

    def status(self):
        pe, gn, nn, ni = C.c_int(), C.c_int(), C.c_int64(), C.c_uint64()
        self._chk(self.L.pct_corridor_status(self.h, C.byref(pe), C.byref(gn), C.byref(nn), C.byref(ni)))
        return dict(path_exists=bool(pe.value), global_navi=bool(gn.value), nodes=nn.value, inflation_queries=ni.value)
