"""ctypes loader for the CHECKER libraries under oracle/.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Nothing under pointcloudtraj_amd/ does.

Two backends with one interface:
  PortKD  -- oracle/liboracle.so, our CPU restatement (okd_* symbols)
  RefKD   -- oracle/_ref/libkdtree_ref.so, the reference's own kdtree.c compiled unmodified,
             driven through oracle/_ref/librefshim.so (our batch loops over its public API)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libkdtree_ref.so")
_SHIM = os.path.join(_HERE, "_ref", "librefshim.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    """Compile the checker (and, when /root/reference is present, oracle/_ref)."""
    if force or not os.path.exists(_PORT) or os.path.exists("/root/reference/Utils/kdtree/src/kdtree.c"):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True, stdout=subprocess.DEVNULL)


def have_ref() -> bool:
    return os.path.exists(_REF) and os.path.exists(_SHIM)


_port = None
_ref = None
_shim = None


def port_lib():
    global _port
    if _port is None:
        if not os.path.exists(_PORT):
            build()
        L = C.CDLL(_PORT)
        L.okd_create.restype = C.c_void_p
        L.okd_create.argtypes = [C.c_int]
        L.okd_free.argtypes = [C.c_void_p]
        L.okd_clear.argtypes = [C.c_void_p]
        L.okd_insertf_batch.argtypes = [C.c_void_p, _f32p, C.c_int64]
        L.okd_nearestf_batch.argtypes = [C.c_void_p, _f32p, C.c_int64, _i32p, _f64p]
        L.okd_range_countf_batch.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int64, _i32p]
        L.okd_brute_nearestf.argtypes = [_f32p, C.c_int64, _f32p, C.c_int64, _i32p, _f64p]
        L.okd_brute_countf.argtypes = [_f32p, C.c_int64, _f32p, _f32p, C.c_int64, _i32p]
        for n in ("okd_nearest_rangef",):
            getattr(L, n).restype = C.c_void_p
        L.okd_nearest_rangef.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_float]
        L.okd_nearestf.restype = C.c_void_p
        L.okd_nearestf.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.okd_res_free.argtypes = [C.c_void_p]
        L.okd_res_size.argtypes = [C.c_void_p]
        L.okd_res_end.argtypes = [C.c_void_p]
        L.okd_res_next.argtypes = [C.c_void_p]
        L.okd_res_rewind.argtypes = [C.c_void_p]
        L.okd_res_item_id.argtypes = [C.c_void_p]
        L.okd_res_item_id.restype = C.c_int32
        L.okd_res_item_data.argtypes = [C.c_void_p]
        L.okd_res_item_data.restype = C.c_void_p
        L.okd_res_item.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.okd_res_item.restype = C.c_void_p
        L.okd_res_item3.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 3
        L.okd_res_item3.restype = C.c_void_p
        L.okd_insert.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
        L.okd_data_destructor.argtypes = [C.c_void_p, C.c_void_p]
        # corridor port
        L.ocor_inflate_batch.argtypes = [C.c_void_p, C.c_void_p, _f64p, C.c_int64, _f64p, _i32p, _f64p, _u8p]
        L.ocor_bezier_pos.argtypes = [_f64p, C.c_int, C.c_double, _f64p]
        L.ocor_bezier_samples.restype = C.c_int64
        L.ocor_bezier_samples.argtypes = [_f64p, C.c_int64, _f64p, _i32p, C.c_int32, C.c_double, C.c_double, C.c_double,
                                          _i32p, _f64p, _f64p, C.c_int64]
        L.ocor_check_safe_trajectory.restype = C.c_int64
        L.ocor_check_safe_trajectory.argtypes = [C.c_void_p, C.c_void_p, _f64p, C.c_int64, _f64p, _i32p, C.c_int32,
                                                 C.c_double, C.c_double, C.c_double, C.c_int64,
                                                 C.POINTER(C.c_int64), _f64p, _f64p, _f64p, _i32p]
        _port = L
    return _port


def ref_libs():
    global _ref, _shim
    if _ref is None:
        if not have_ref():
            raise FileNotFoundError("oracle/_ref is not built (needs /root/reference; run `make -C oracle`)")
        # RTLD_DEEPBIND: the shim's kd_* calls must bind to the reference library it was linked against even when the
        # product's libkdtree.so (same symbol names) is loaded in the same process
        deep = getattr(os, "RTLD_DEEPBIND", 0)
        R = C.CDLL(_REF, mode=C.RTLD_LOCAL | deep)
        R.kd_create.restype = C.c_void_p
        R.kd_create.argtypes = [C.c_int]
        R.kd_free.argtypes = [C.c_void_p]
        R.kd_clear.argtypes = [C.c_void_p]
        R.kd_nearestf.restype = C.c_void_p
        R.kd_nearestf.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        R.kd_nearest_rangef.restype = C.c_void_p
        R.kd_nearest_rangef.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_float]
        R.kd_res_free.argtypes = [C.c_void_p]
        R.kd_res_size.argtypes = [C.c_void_p]
        R.kd_res_item3.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 3
        R.kd_res_item3.restype = C.c_void_p
        R.kd_res_item_data.argtypes = [C.c_void_p]
        R.kd_res_item_data.restype = C.c_void_p
        R.kd_insert.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
        R.kd_data_destructor.argtypes = [C.c_void_p, C.c_void_p]
        S = C.CDLL(_SHIM, mode=C.RTLD_LOCAL | deep)
        S.refshim_insertf_batch.restype = C.c_int64
        S.refshim_insertf_batch.argtypes = [C.c_void_p, _f32p, C.c_int64, C.c_int64]
        S.refshim_nearestf_batch.argtypes = [C.c_void_p, _f32p, C.c_int64, _i32p, _f64p]
        S.refshim_nearestf_timed.restype = C.c_double
        S.refshim_nearestf_timed.argtypes = [C.c_void_p, _f32p, C.c_int64, _i32p]
        S.refshim_rangef.restype = C.c_int64
        S.refshim_rangef.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_float, _i32p, C.c_int64]
        S.refshim_range_countf_batch.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int64, _i32p]
        _ref, _shim = R, S
    return _ref, _shim


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class PortKD:
    """Our CPU restatement behind the same batch interface as RefKD."""

    kind = "port"

    def __init__(self):
        self.L = port_lib()
        self.h = self.L.okd_create(3)
        self.n = 0

    def close(self):
        if self.h:
            self.L.okd_free(self.h)
            self.h = None

    __del__ = close

    def insert(self, xyz):
        xyz = _f32c(xyz).reshape(-1, 3)
        assert self.n == 0, "batch insert assigns ids from 0"
        if self.L.okd_insertf_batch(self.h, xyz, len(xyz)):
            raise MemoryError
        self.n += len(xyz)

    def nearest(self, q):
        q = _f32c(q).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float64)
        if self.L.okd_nearestf_batch(self.h, q, len(q), idx, d2):
            raise RuntimeError("empty tree")
        return idx, d2

    def range_ids(self, q, r):
        """ids in the reference's iteration order for one query."""
        q = _f32c(q).reshape(3)
        res = self.L.okd_nearest_rangef(self.h, q.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(r))
        out = []
        while not self.L.okd_res_end(res):
            out.append(self.L.okd_res_item_id(res))
            self.L.okd_res_next(res)
        assert len(out) == self.L.okd_res_size(res)
        self.L.okd_res_free(res)
        return np.asarray(out, np.int32)

    def range_count(self, q, r):
        q = _f32c(q).reshape(-1, 3)
        r = _f32c(np.broadcast_to(r, (len(q),)))
        cnt = np.empty(len(q), np.int32)
        if self.L.okd_range_countf_batch(self.h, q, r, len(q), cnt):
            raise MemoryError
        return cnt


class _AnyDimKD:
    """kd_create(k) / kd_insert / kd_nearest / kd_nearest_range with double positions through the plain C API of either library
    (`pre` = "kd_" on the reference, "okd_" on the port); payload of row i is (void*)(i+1)."""

    def __init__(self, lib, pre, dim):
        self.L, self.pre, self.dim, self.n = lib, pre, int(dim), 0
        dp, vp = C.POINTER(C.c_double), C.c_void_p
        f = lambda n: getattr(lib, pre + n)                      # noqa: E731
        f("create").restype = vp; f("create").argtypes = [C.c_int]
        f("free").argtypes = [vp]
        f("insert").argtypes = [vp, dp, vp]
        f("nearest").restype = vp; f("nearest").argtypes = [vp, dp]
        f("nearest_range").restype = vp; f("nearest_range").argtypes = [vp, dp, C.c_double]
        for n in ("res_free", "res_size", "res_end", "res_next"):
            f(n).argtypes = [vp]
        f("res_item_data").restype = vp; f("res_item_data").argtypes = [vp]
        self.f = f
        self.h = f("create")(self.dim)

    def close(self):
        if self.h:
            self.f("free")(self.h)
            self.h = None

    __del__ = close

    def insert(self, rows):
        rows = np.ascontiguousarray(rows, np.float64).reshape(-1, self.dim)
        for p in rows:
            if self.f("insert")(self.h, p.ctypes.data_as(C.POINTER(C.c_double)), C.c_void_p(self.n + 1)):
                raise MemoryError
            self.n += 1

    def nearest(self, q):
        q = np.ascontiguousarray(q, np.float64).reshape(-1, self.dim)
        ids = np.empty(len(q), np.int32)
        for i, qq in enumerate(q):
            r = self.f("nearest")(self.h, qq.ctypes.data_as(C.POINTER(C.c_double)))
            ids[i] = int(self.f("res_item_data")(r) or 0) - 1
            self.f("res_free")(r)
        return ids

    def range_ids(self, q, r):
        q = np.ascontiguousarray(q, np.float64).reshape(self.dim)
        rs = self.f("nearest_range")(self.h, q.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(r))
        out = []
        while not self.f("res_end")(rs):
            out.append(int(self.f("res_item_data")(rs) or 0) - 1)
            self.f("res_next")(rs)
        assert len(out) == self.f("res_size")(rs)
        self.f("res_free")(rs)
        return np.asarray(out, np.int32)


def RefKDN(dim):
    """the reference's compiled kdtree.c, any dimension, double positions"""
    return _AnyDimKD(ref_libs()[0], "kd_", dim)


def PortKDN(dim):
    """the CPU restatement (oracle/kdtree_port.c), any dimension, double positions"""
    return _AnyDimKD(port_lib(), "okd_", dim)


class RefKD:
    """The reference's compiled kdtree.c."""

    kind = "reference"

    def __init__(self):
        self.R, self.S = ref_libs()
        self.h = self.R.kd_create(3)
        self.n = 0

    def close(self):
        if self.h:
            self.R.kd_free(self.h)
            self.h = None

    __del__ = close

    def insert(self, xyz):
        xyz = _f32c(xyz).reshape(-1, 3)
        got = self.S.refshim_insertf_batch(self.h, xyz, len(xyz), self.n)
        if got != len(xyz):
            raise MemoryError
        self.n += len(xyz)

    def nearest(self, q):
        q = _f32c(q).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float64)
        if self.S.refshim_nearestf_batch(self.h, q, len(q), idx, d2):
            raise RuntimeError("empty tree")
        return idx, d2

    def nearest_timed(self, q):
        q = _f32c(q).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        return self.S.refshim_nearestf_timed(self.h, q, len(q), idx), idx

    def nearest_timed_mt(self, q, threads: int):
        """T threads on the shared read-only tree through kd_nearest (the *f variants are not re-entrant): (seconds, ids)"""
        q = _f32c(q).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        self.S.refshim_nearest_timed_mt.restype = C.c_double
        self.S.refshim_nearest_timed_mt.argtypes = [C.c_void_p, _f32p, C.c_int64, _i32p, C.c_int]
        return self.S.refshim_nearest_timed_mt(self.h, q, len(q), idx, int(threads)), idx

    def range_ids(self, q, r):
        q = _f32c(q).reshape(3)
        cap = max(self.n, 1)
        out = np.empty(cap, np.int32)
        n = self.S.refshim_rangef(self.h, q.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(r), out, cap)
        if n < 0:
            raise RuntimeError(f"refshim_rangef -> {n}")
        return out[:n].copy()

    def range_count(self, q, r):
        q = _f32c(q).reshape(-1, 3)
        r = _f32c(np.broadcast_to(r, (len(q),)))
        cnt = np.empty(len(q), np.int32)
        if self.S.refshim_range_countf_batch(self.h, q, r, len(q), cnt):
            raise MemoryError
        return cnt


def brute_nearest(xyz, q):
    """Exhaustive fp64 scan, lowest index wins ties (the engine's tie rule)."""
    L = port_lib()
    xyz = _f32c(xyz).reshape(-1, 3)
    q = _f32c(q).reshape(-1, 3)
    idx = np.empty(len(q), np.int32)
    d2 = np.empty(len(q), np.float64)
    L.okd_brute_nearestf(xyz, len(xyz), q, len(q), idx, d2)
    return idx, d2


def brute_count(xyz, q, r):
    L = port_lib()
    xyz = _f32c(xyz).reshape(-1, 3)
    q = _f32c(q).reshape(-1, 3)
    r = _f32c(np.broadcast_to(r, (len(q),)))
    cnt = np.empty(len(q), np.int32)
    L.okd_brute_countf(xyz, len(xyz), q, r, len(q), cnt)
    return cnt


def brute_nearest_mt(xyz, q, threads: int = 0):
    """brute_nearest with the queries split over host threads (ctypes releases the GIL): same results"""
    import concurrent.futures as cf
    q = _f32c(q).reshape(-1, 3)
    xyz = _f32c(xyz).reshape(-1, 3)
    threads = threads or min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 32)
    threads = max(1, min(threads, len(q)))
    if threads == 1:
        return brute_nearest(xyz, q)
    parts = np.array_split(np.arange(len(q)), threads)
    with cf.ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(lambda ix: brute_nearest(xyz, q[ix]), parts))
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])


def inflate_brute(xyz, start, sample_range, search_margin, max_radius, pts, threads: int = 0):
    """radiusSearch (corridor_finder.cpp:113-133) over an exhaustive nearest neighbour: (radius, idx, d2) for (n,3) fp64 points;
    early-out rows give max_radius - search_margin, idx -1, d2 inf"""
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    dx, dy, dz = pts[:, 0] - start[0], pts[:, 1] - start[1], pts[:, 2] - start[2]
    far = np.sqrt(dx * dx + dy * dy + dz * dz) > sample_range + max_radius
    idx, d2 = brute_nearest_mt(xyz, pts.astype(np.float32), threads)
    rad = np.minimum(np.sqrt(d2) - search_margin, max_radius)
    skip = far | (len(np.asarray(xyz).reshape(-1, 3)) == 0)
    rad = np.where(skip, max_radius - search_margin, rad)
    return rad, np.where(skip, -1, idx), np.where(skip, np.inf, d2)


def replan_tick(xyz, start, sample_range, search_margin, max_radius, nodes, polycoef, seg_time, orders, t_start, stop_time, dt=0.02,
                cap=4096, threads: int = 0):
    """CPU statement of one replan tick's query side: corridor-node inflation (corridor_finder.cpp:829-835), the sampled Bezier
    check (sim_planning_demo.cpp:729-781) and the control-point check against the cloud `xyz`, exhaustively."""
    L = port_lib()
    polycoef = np.ascontiguousarray(polycoef, np.float64)
    seg_time = np.ascontiguousarray(seg_time, np.float64)
    orders = np.ascontiguousarray(orders, np.int32)
    seg = np.zeros(cap, np.int32)
    tt = np.zeros(cap, np.float64)
    pos = np.zeros((cap, 3), np.float64)
    ns = L.ocor_bezier_samples(polycoef, polycoef.shape[1], seg_time, orders, len(seg_time), float(t_start), float(stop_time), float(dt),
                               seg, tt, pos.reshape(-1), cap)
    n = min(ns, cap)
    # control points in world units, segments from the one holding t_start on (the segment search of checkSafeTrajectory)
    t_s, first = float(t_start), 0
    while first < len(seg_time):
        if t_s > seg_time[first] and first + 1 < len(seg_time):
            t_s -= seg_time[first]
            first += 1
        else:
            break
    ctrl = []
    for i in range(first, len(seg_time)):
        m = int(orders[i]) + 1
        for j in range(m):
            ctrl.append([polycoef[i, j] * seg_time[i], polycoef[i, m + j] * seg_time[i], polycoef[i, 2 * m + j] * seg_time[i]])
    ctrl = np.asarray(ctrl, np.float64).reshape(-1, 3)
    nodes = np.ascontiguousarray(nodes, np.float64).reshape(-1, 3)
    allp = np.concatenate([nodes, pos[:n], ctrl])
    rad, idx, d2 = inflate_brute(xyz, start, sample_range, search_margin, max_radius, allp, threads)
    a, b = len(nodes), len(nodes) + n

    def first_neg(r):
        w = np.nonzero(r < 0.0)[0]
        return int(w[0]) if len(w) else -1
    return dict(node_radius=rad[:a], node_idx=idx[:a], node_d2=d2[:a], nsamples=int(ns), sample_pos=pos[:n], sample_radius=rad[a:b],
                sample_idx=idx[a:b], sample_d2=d2[a:b], first_hit_sample=first_neg(rad[a:b]), nctrl=len(ctrl), ctrl_pos=ctrl,
                ctrl_radius=rad[b:], ctrl_idx=idx[b:], ctrl_d2=d2[b:], first_hit_ctrl=first_neg(rad[b:]))


class _CorParams(C.Structure):
    _fields_ = [("start", C.c_double * 3), ("sample_range", C.c_double), ("search_margin", C.c_double),
                ("max_radius", C.c_double), ("cloud_empty", C.c_int)]


def corridor_params(start, sample_range, search_margin, max_radius, cloud_empty=False):
    p = _CorParams()
    p.start[:] = [float(v) for v in start]
    p.sample_range, p.search_margin, p.max_radius = float(sample_range), float(search_margin), float(max_radius)
    p.cloud_empty = int(bool(cloud_empty))
    return p


def inflate(kd: PortKD, params, pts):
    """ocor_inflate_batch: radii (f64), NN idx, d2, collide flags for (n,3) f64 points."""
    L = port_lib()
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    n = len(pts)
    rad = np.empty(n, np.float64)
    idx = np.empty(n, np.int32)
    d2 = np.empty(n, np.float64)
    col = np.empty(n, np.uint8)
    L.ocor_inflate_batch(C.byref(params), kd.h, pts, n, rad, idx, d2, col)
    return rad, idx, d2, col


def bezier_pos(coef_row, order, u):
    L = port_lib()
    out = np.empty(3, np.float64)
    L.ocor_bezier_pos(np.ascontiguousarray(coef_row, np.float64), int(order), float(u), out)
    return out


def check_safe_trajectory(kd: PortKD, params, polycoef, seg_time, orders, t_start, stop_time, dt=0.02, cap=4096):
    """Returns dict(first_hit, n, pos, radius, d2, idx) following checkSafeTrajectory."""
    L = port_lib()
    polycoef = np.ascontiguousarray(polycoef, np.float64)
    seg_time = np.ascontiguousarray(seg_time, np.float64)
    orders = np.ascontiguousarray(orders, np.int32)
    nseg = len(seg_time)
    pos = np.zeros((cap, 3), np.float64)
    rad = np.zeros(cap, np.float64)
    d2 = np.zeros(cap, np.float64)
    idx = np.zeros(cap, np.int32)
    ns = C.c_int64(0)
    first = L.ocor_check_safe_trajectory(C.byref(params), kd.h, polycoef, polycoef.shape[1], seg_time, orders, nseg,
                                         float(t_start), float(stop_time), float(dt), cap, C.byref(ns),
                                         pos.reshape(-1), rad, d2, idx)
    n = ns.value
    return dict(first_hit=int(first), n=int(n), pos=pos[:n], radius=rad[:n], d2=d2[:n], idx=idx[:n])


# ---- safe-region RRT* corridor finder (oracle/rrt_port.c) ---------------------------------------------------------
class PortCorridor:
    """CPU oracle of the corridor finder; same method names as pointcloudtraj_amd.corridor.SafeRegionRrtStar."""

    def __init__(self):
        L = port_lib()
        vp, d3 = C.c_void_p, C.POINTER(C.c_double)
        L.orrt_create.restype = vp
        L.orrt_destroy.argtypes = [vp]
        L.orrt_set_param.argtypes = [vp] + [C.c_double] * 4
        L.orrt_reset.argtypes = [vp]
        L.orrt_set_input.argtypes = [vp, _f32p, C.c_int64]
        L.orrt_set_start_pt.argtypes = [vp, d3, d3]
        L.orrt_set_pt.argtypes = [vp, d3, d3] + [C.c_double] * 7 + [C.c_int, C.c_double, C.c_double]
        L.orrt_expansion.argtypes = [vp, C.c_int64]
        L.orrt_refine.argtypes = [vp, C.c_int64]
        L.orrt_evaluate.argtypes = [vp]
        L.orrt_evaluate_exhausted.argtypes = [vp]
        L.orrt_reset_root.argtypes = [vp, d3]
        L.orrt_check_traj_pt_col.argtypes = [vp, d3]
        L.orrt_get_path.restype = C.c_int64
        L.orrt_get_path.argtypes = [vp, _f64p, _f64p, C.c_int64]
        L.orrt_status.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
        self.L = L
        self.h = L.orrt_create()

    @staticmethod
    def _d3(v):
        return (C.c_double * 3)(*[float(x) for x in v])

    def close(self):
        if self.h:
            self.L.orrt_destroy(self.h)
            self.h = None

    __del__ = close

    def setParam(self, safety_margin, search_margin, max_radius, sample_range):
        self.L.orrt_set_param(self.h, safety_margin, search_margin, max_radius, sample_range)

    def reset(self):
        self.L.orrt_reset(self.h)

    def setInput(self, points, build_index=True):
        a = np.ascontiguousarray(np.asarray(points, np.float32)[:, :3])
        self.L.orrt_set_input(self.h, a, len(a))

    def setPt(self, start, end, xl, xh, yl, yh, zl, zh, local_range, max_iter, sample_portion, goal_portion):
        self.L.orrt_set_pt(self.h, self._d3(start), self._d3(end), xl, xh, yl, yh, zl, zh, local_range, int(max_iter),
                           sample_portion, goal_portion)

    def setStartPt(self, start, end):
        self.L.orrt_set_start_pt(self.h, self._d3(start), self._d3(end))

    def resetRoot(self, target):
        self.L.orrt_reset_root(self.h, self._d3(target))

    def SafeRegionExpansion(self, iterations):
        self.L.orrt_expansion(self.h, int(iterations))

    def SafeRegionRefine(self, iterations):
        self.L.orrt_refine(self.h, int(iterations))

    def SafeRegionEvaluate(self, time_limit=None):
        """time_limit None = no clock (iteration-count form); a negative limit = the clock has already run out at every check"""
        if time_limit is None:
            self.L.orrt_evaluate(self.h)
        elif time_limit < 0:
            self.L.orrt_evaluate_exhausted(self.h)
        else:
            raise ValueError("the CPU restatement states the time box only at its deterministic ends (None or < 0)")

    def checkTrajPtCol(self, p) -> bool:
        return bool(self.L.orrt_check_traj_pt_col(self.h, self._d3(p)))

    def getPath(self):
        path = np.zeros(3 * 4096)
        rad = np.zeros(4096)
        n = self.L.orrt_get_path(self.h, path, rad, 4096)
        return path.reshape(-1, 3)[:n].copy(), rad[:n].copy()

    def status(self):
        pe, gn, nn, ni = C.c_int(), C.c_int(), C.c_int64(), C.c_uint64()
        self.L.orrt_status(self.h, C.byref(pe), C.byref(gn), C.byref(nn), C.byref(ni))
        return dict(path_exists=bool(pe.value), global_navi=bool(gn.value), nodes=nn.value, inflation_queries=ni.value)


class PortVoxelMap:
    """oracle/voxel_port.c: sequential restatement of voxel_map / voxel_value_map (voxel_map.cpp:5-76)."""

    def __init__(self, res: float):
        L = port_lib()
        L.ovox_create.restype = C.c_void_p
        L.ovox_create.argtypes = [C.c_double]
        L.ovox_destroy.argtypes = [C.c_void_p]
        L.ovox_size.restype = C.c_int64
        L.ovox_size.argtypes = [C.c_void_p]
        L.ovox_add.restype = C.c_int64
        L.ovox_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.ovox_get_keys.argtypes = [C.c_void_p, _i32p]
        L.ovox_get_f64.argtypes = [C.c_void_p, _f64p]
        L.ovox_get_f32.argtypes = [C.c_void_p, _f32p]
        self.L = L
        self.h = L.ovox_create(float(res))

    def close(self):
        if self.h:
            self.L.ovox_destroy(self.h)
            self.h = None

    __del__ = close

    def __len__(self):
        return int(self.L.ovox_size(self.h))

    def add(self, pts: np.ndarray):
        """pts: [n, >=3] float32 or float64 (row stride = the array's).  Returns (n_new, is_new[n], voxel_index[n])."""
        pts = np.ascontiguousarray(pts)
        assert pts.dtype in (np.float32, np.float64) and pts.ndim == 2 and pts.shape[1] >= 3
        n = len(pts)
        is_new = np.zeros(n, np.uint8)
        index = np.zeros(n, np.int32)
        n_new = self.L.ovox_add(self.h, pts.ctypes.data, n, pts.strides[0], int(pts.dtype == np.float64),
                                is_new.ctypes.data, index.ctypes.data)
        return int(n_new), is_new, index

    def keys(self):
        out = np.zeros((len(self), 3), np.int32)
        self.L.ovox_get_keys(self.h, out)
        return out

    def cloud_f64(self):
        out = np.zeros((len(self), 3), np.float64)
        self.L.ovox_get_f64(self.h, out)
        return out

    def cloud_f32(self):
        out = np.zeros((len(self), 3), np.float32)
        self.L.ovox_get_f32(self.h, out)
        return out


# ---- oracle/traj_port.c: Bezier evaluators beside the collision check ------------------------------------------------
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def _traj_lib():
    L = port_lib()
    if not getattr(L, "_traj_bound", False):
        L.otraj_state.argtypes = [_f64p, C.c_int, C.c_double, _f64p]
        L.otraj_wire_from_matrix.restype = C.c_int64
        L.otraj_wire_from_matrix.argtypes = [_f64p, C.c_int64, _i32p, C.c_int32, _f64p, _f64p, _f64p]
        L.otraj_wire_sample.argtypes = [_f64p, _f64p, _f64p, _f64p, _u32p, C.c_int32, C.c_int32, _f64p, _f64p]
        L.otraj_segm_index.argtypes = [_f64p, _f64p, _f64p, _f64p, _u32p, C.c_int32, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.otraj_nearest_points.restype = C.c_int64
        L.otraj_nearest_points.argtypes = [_f64p, _f64p, _f64p, _f64p, _u32p, C.c_int32, C.c_double, _f64p, C.c_int64]
        L.otraj_end_yaws.argtypes = [_f64p, _f64p, C.c_int64, _f64p, _f64p, _f64p]
        L._traj_bound = True
    return L


def traj_state(poly_coeff, orders, seg, u):
    pc = np.ascontiguousarray(poly_coeff, np.float64)
    out = np.zeros((len(seg), 9))
    L = _traj_lib()
    for i, (s, t) in enumerate(zip(seg, u)):
        L.otraj_state(np.ascontiguousarray(pc[s]), int(orders[s]), float(t), out[i])
    return out


def traj_wire_from_matrix(poly_coeff, orders):
    pc = np.ascontiguousarray(poly_coeff, np.float64)
    od = np.ascontiguousarray(orders, np.int32)
    total = int(np.sum(od.astype(np.int64) + 1))
    cx, cy, cz = np.zeros(total), np.zeros(total), np.zeros(total)
    n = _traj_lib().otraj_wire_from_matrix(pc, pc.shape[1], od, len(od), cx, cy, cz)
    assert n == total
    return cx, cy, cz


def _wire_args(w):
    return (np.ascontiguousarray(w.coef_x, np.float64), np.ascontiguousarray(w.coef_y, np.float64), np.ascontiguousarray(w.coef_z, np.float64),
            np.ascontiguousarray(w.time, np.float64), np.ascontiguousarray(w.order, np.uint32), len(w.time))


def traj_wire_sample(w, samples=1001):
    a = _wire_args(w)
    pos, step = np.zeros((a[5] * samples, 3)), np.zeros(a[5] * samples)
    _traj_lib().otraj_wire_sample(*a, samples, pos, step)
    return pos, step


def traj_segm_index(w, twirl_len):
    segm, part = C.c_int32(), C.c_int32()
    _traj_lib().otraj_segm_index(*_wire_args(w), float(twirl_len), C.byref(segm), C.byref(part))
    return segm.value, part.value


def traj_nearest_traj(w, res, twirl_len):
    """(voxel cloud float32, samples used): to_nearest_traj through the sequential voxel container restatement"""
    a = _wire_args(w)
    buf = np.zeros((a[5] * 1001 + 1, 3))
    n = _traj_lib().otraj_nearest_points(*a, float(twirl_len), buf, len(buf))
    m = PortVoxelMap(res)
    m.add(buf[:n])
    return m.cloud_f32(), int(n)


def traj_end_yaws(path_x, path_y, coef_x, coef_y):
    px, py = np.ascontiguousarray(path_x, np.float64), np.ascontiguousarray(path_y, np.float64)
    out = np.zeros(len(px))
    _traj_lib().otraj_end_yaws(px, py, len(px), np.ascontiguousarray(coef_x, np.float64), np.ascontiguousarray(coef_y, np.float64), out)
    return out
