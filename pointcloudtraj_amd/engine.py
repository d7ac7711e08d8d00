"""ctypes binding of libpct_engine.so (include/pct_engine.h) -- the host-side mirror used by the
tests, bench.py and the torch.distributed sharding layer.  Plumbing only: every number comes
from the HIP kernels; a missing library or a missing GPU raises, there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os

import numpy as np

from . import build as _build

NO_INDEX = 0xFFFFFFFF
ALGO_AUTO, ALGO_STREAM, ALGO_GRID, ALGO_STREAM_EXACT = 0, 1, 2, 3

_lib = None


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pct_engine status {code}: {msg}")
        self.code = code


class InflateParams(C.Structure):
    _fields_ = [("start", C.c_double * 3), ("sample_range", C.c_double), ("search_margin", C.c_double),
                ("max_radius", C.c_double)]


class ReplanOut(C.Structure):
    _fields_ = [("node_radius", C.c_void_p), ("node_idx", C.c_void_p), ("node_d2", C.c_void_p),
                ("sample_pos", C.c_void_p), ("sample_radius", C.c_void_p), ("sample_d2", C.c_void_p), ("sample_idx", C.c_void_p),
                ("ctrl_pos", C.c_void_p), ("ctrl_radius", C.c_void_p), ("ctrl_d2", C.c_void_p), ("ctrl_idx", C.c_void_p),
                ("nsamples", C.c_int64), ("first_hit_sample", C.c_int64), ("nctrl", C.c_int64), ("first_hit_ctrl", C.c_int64)]


class BezierTraj(C.Structure):
    _fields_ = [("polycoef", C.POINTER(C.c_double)), ("row_stride", C.c_int64), ("seg_time", C.POINTER(C.c_double)),
                ("orders", C.POINTER(C.c_int32)), ("nseg", C.c_int32)]


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (soname
    libamdhip64.so.7, the same as /opt/rocm's); two copies in one process each open the KFD device
    and the second one sees no GPU.  Loading torch's copy FIRST makes libpct_engine.so's
    DT_NEEDED libamdhip64.so.7 resolve to it by soname, and a later `import torch` finds the same
    file already mapped.  Without torch installed the system runtime is used as linked."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib():
    """Load (never build implicitly on a GPU box: the .so travels with the snapshot)."""
    global _lib
    if _lib is None:
        _preload_hip_runtime()
        if not os.path.exists(_build.ENGINE_SO):
            raise FileNotFoundError(f"{_build.ENGINE_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(the HIP extension is mandatory; there is no CPU fallback)")
        L = C.CDLL(_build.ENGINE_SO)
        vp, i64, i32, u32p, f32p, f64p = C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p
        L.pct_last_error.restype = C.c_char_p
        L.pct_init.argtypes = [i32]
        L.pct_cloud_create.argtypes = [i64, C.POINTER(vp)]
        L.pct_cloud_destroy.argtypes = [vp]
        L.pct_cloud_size.restype = i64
        L.pct_cloud_size.argtypes = [vp]
        L.pct_cloud_capacity.restype = i64
        L.pct_cloud_capacity.argtypes = [vp]
        L.pct_cloud_set_index_base.argtypes = [vp, i64]
        L.pct_cloud_upload_aos.argtypes = [vp, vp, i64, i64]
        L.pct_cloud_upload_soa_dev.argtypes = [vp, vp, vp, vp, i64]
        L.pct_cloud_upload_fields.argtypes = [vp, vp, i64, i64, i64, i64, i64]
        L.pct_cloud_append_aos.argtypes = [vp, vp, i64, i64]
        L.pct_cloud_build_grid.argtypes = [vp, C.c_float]
        L.pct_cloud_drop_grid.argtypes = [vp]
        L.pct_cloud_has_grid.argtypes = [vp]
        L.pct_cloud_grid_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(i64)]
        L.pct_nodeset_create.argtypes = [C.c_int, i64, C.POINTER(vp)]
        L.pct_nodeset_destroy.argtypes = [vp]
        L.pct_nodeset_clear.argtypes = [vp]
        L.pct_nodeset_size.argtypes = [vp]
        L.pct_nodeset_size.restype = i64
        L.pct_nodeset_dim.argtypes = [vp]
        L.pct_nodeset_append.argtypes = [vp, C.POINTER(C.c_double), i64]
        L.pct_nodeset_nearest.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        L.pct_nodeset_radius_indices_r2.argtypes = [vp, C.POINTER(C.c_double), C.c_double, vp, i64, C.POINTER(i64)]
        L.pct_cloud_ring_bucket_records.argtypes = [vp]
        L.pct_nn_batch.argtypes = [vp, f32p, i64, u32p, f64p]
        L.pct_nn_batch_algo.argtypes = [vp, i32, f32p, i64, u32p, f64p]
        L.pct_radius_count_batch.argtypes = [vp, f32p, f32p, i64, u32p]
        L.pct_radius_count_batch_algo.argtypes = [vp, i32, f32p, f32p, i64, u32p]
        L.pct_radius_indices.argtypes = [vp, f32p, C.c_float, u32p, i64, C.POINTER(i64)]
        L.pct_radius_crop.argtypes = [vp, vp, C.c_double, C.c_int, i64, vp, vp, vp, C.POINTER(i64)]
        L.pct_cloud_crop_to.argtypes = [vp, vp, C.c_double, vp]
        L.pct_inflate_batch.argtypes = [vp, C.POINTER(InflateParams), f64p, i64, f64p, u32p, f64p]
        L.pct_bezier_check.argtypes = [vp, C.POINTER(BezierTraj), C.POINTER(InflateParams), C.c_double, C.c_double, C.c_double,
                                       C.POINTER(i64), C.POINTER(i64), i64, f64p, f64p, f64p, u32p]
        L.pct_nn_batch_dev.argtypes = [vp, i32, vp, i64, vp, vp, vp]
        L.pct_inflate_batch_dev.argtypes = [vp, C.POINTER(InflateParams), vp, i64, vp, vp, vp, vp]
        L.pct_bezier_check_dev.argtypes = [vp, C.POINTER(BezierTraj), C.POINTER(InflateParams), C.c_double, C.c_double, C.c_double, i64,
                                           vp, vp, vp, vp, vp, vp, vp]
        L.pct_radius_count_batch_dev.argtypes = [vp, i32, vp, vp, i64, vp, vp]
        L.pct_cloud_reserve_queries.argtypes = [vp, i64]
        L.pct_plan_create_nn.argtypes = [vp, i32, i64, C.POINTER(vp)]
        L.pct_plan_run.argtypes = [vp, f32p, u32p, f64p]
        L.pct_plan_destroy.argtypes = [vp]
        L.pct_cloud_ring_index.argtypes = [vp, C.c_float, vp]
        L.pct_cloud_ring_drop.argtypes = [vp]
        L.pct_cloud_has_ring_index.argtypes = [vp]
        L.pct_cloud_ring_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(i64)]
        L.pct_ctrl_points_check.argtypes = [vp, C.POINTER(BezierTraj), C.POINTER(InflateParams), C.c_double, C.POINTER(i64), C.POINTER(i64),
                                            i64, f64p, f64p, f64p, u32p]
        L.pct_plan_create_replan.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.pct_plan_last_run_us.argtypes = [vp, C.POINTER(C.c_double)]
        L.pct_plan_replan_run.argtypes = [vp, C.POINTER(InflateParams), f64p, i64, C.POINTER(BezierTraj), C.c_double, C.c_double, C.c_double,
                                          C.c_int, C.POINTER(ReplanOut)]
        L.pct_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.pct_last_batch_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.pct_kernel_ms_history.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
        L.pct_cloud_frame_buffer.argtypes = [vp, i64, C.POINTER(vp)]
        L.pct_cloud_append_frame.argtypes = [vp, i64, i64]
        L.pct_debug_read_grid.argtypes = [vp, vp, vp]
        L.pct_set_timing.argtypes = [vp, C.c_int]
        L.pct_set_timing_stride.argtypes = [vp, C.c_int]
        L.pct_kernel_ms_samples.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.pct_merge_mask_dev.argtypes = [vp, vp, vp, vp, i64, vp]
        L.pct_last_work.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.pct_set_work_counters.argtypes = [vp, i32]
        L.pct_last_work_ex.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.pct_debug_verify_grid.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.pct_cloud_pyramid_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(i64), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _chk(code):
    if code != 0:
        raise EngineError(code, lib().pct_last_error().decode(errors="replace"))


def init(device: int = 0):
    _chk(lib().pct_init(device))


def device_count() -> int:
    return lib().pct_device_count()


def sync():
    _chk(lib().pct_sync())


def set_filter_mode(mode: int):
    """-1 automatic, 0 direct (p-q)^2 filter, 1 expanded |p|^2 - 2 p.q filter (pct_debug_set_filter_mode); results are identical"""
    _chk(lib().pct_debug_set_filter_mode(int(mode)))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def merge_mask_device(d2_local_ptr: int, d2_best_ptr: int, idx_local_ptr: int, cand_ptr: int, Q: int, stream: int = 0):
    """cand[i] = idx_local[i] where d2_local[i] is the reduced minimum (and finite), INT32_MAX elsewhere (pct_merge_mask_dev)"""
    _chk(lib().pct_merge_mask_dev(d2_local_ptr, d2_best_ptr, idx_local_ptr, cand_ptr, int(Q), stream))


def inflate_params(start, sample_range, search_margin, max_radius) -> InflateParams:
    p = InflateParams()
    p.start[:] = [float(v) for v in start]
    p.sample_range, p.search_margin, p.max_radius = float(sample_range), float(search_margin), float(max_radius)
    return p


class Cloud:
    """An obstacle cloud resident in HBM (pct_cloud).  Mirrors the planner seam:
    set_input ~ safeRegionRrtStar::setInput (corridor_finder.cpp:93-99), inflate ~ radiusSearch
    (:113-133), bezier_check ~ checkSafeTrajectory (sim_planning_demo.cpp:729-781)."""

    def __init__(self, capacity: int):
        self._h = C.c_void_p()
        _chk(lib().pct_cloud_create(int(capacity), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.pct_cloud_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    def __len__(self):
        return lib().pct_cloud_size(self._h)

    @property
    def capacity(self):
        return lib().pct_cloud_capacity(self._h)

    def set_index_base(self, base: int):
        _chk(lib().pct_cloud_set_index_base(self._h, int(base)))

    @staticmethod
    def _aos(points):
        a = np.asarray(points)
        if a.dtype != np.float32 or not a.flags.c_contiguous:
            a = np.ascontiguousarray(a, np.float32)
        if a.ndim != 2 or a.shape[1] not in (3, 4):
            raise ValueError("points must be (n,3) packed xyz or (n,4) pcl::PointXYZ-style records")
        return a, a.shape[1] * 4

    def set_input(self, points):
        a, stride = self._aos(points)
        _chk(lib().pct_cloud_upload_aos(self._h, _ptr(a), len(a), stride))

    def set_input_pointcloud2(self, data: bytes, n: int, point_step: int, off_x: int, off_y: int, off_z: int):
        """sensor_msgs/PointCloud2 payload: `data` = msg.data, offsets = msg.fields[*].offset"""
        buf = np.frombuffer(data, np.uint8)
        _chk(lib().pct_cloud_upload_fields(self._h, _ptr(buf), int(n), int(point_step), int(off_x), int(off_y), int(off_z)))

    def set_input_device(self, x_ptr: int, y_ptr: int, z_ptr: int, n: int):
        _chk(lib().pct_cloud_upload_soa_dev(self._h, x_ptr, y_ptr, z_ptr, int(n)))

    def append(self, points):
        a, stride = self._aos(points)
        _chk(lib().pct_cloud_append_aos(self._h, _ptr(a), len(a), stride))

    def frame_buffer(self, n: int, floats_per_point: int = 3):
        """host-mapped staging buffer for zero-copy appends, as a float32 [n, floats_per_point] array the producer fills in place"""
        p = C.c_void_p()
        _chk(lib().pct_cloud_frame_buffer(self._h, int(n) * floats_per_point * 4, C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(int(n), floats_per_point))

    def append_frame(self, n: int, floats_per_point: int = 3):
        """append the first n records of the frame buffer (pct_cloud_append_frame)"""
        _chk(lib().pct_cloud_append_frame(self._h, int(n), floats_per_point * 4))

    def build_grid(self, cell_size: float = 0.0):
        _chk(lib().pct_cloud_build_grid(self._h, float(cell_size)))

    def drop_grid(self):
        _chk(lib().pct_cloud_drop_grid(self._h))

    @property
    def has_grid(self) -> bool:
        return bool(lib().pct_cloud_has_grid(self._h))

    def grid_info(self):
        dims = (C.c_int32 * 3)()
        h = C.c_float()
        org = (C.c_float * 3)()
        nc = C.c_int64()
        _chk(lib().pct_cloud_grid_info(self._h, dims, C.byref(h), org, C.byref(nc)))
        return dict(dims=tuple(dims), cell_size=h.value, origin=tuple(org), ncells=nc.value)

    def ring_index(self, cell_size: float = 0.0, extent=None):
        """rolling-map index (pct_cloud_ring_index): appends update it in place instead of dropping an index"""
        ext = None if extent is None else np.ascontiguousarray(extent, np.float32).reshape(3)
        _chk(lib().pct_cloud_ring_index(self._h, float(cell_size), None if ext is None else ext.ctypes.data))

    def ring_drop(self):
        _chk(lib().pct_cloud_ring_drop(self._h))

    @property
    def has_ring_index(self) -> bool:
        return bool(lib().pct_cloud_has_ring_index(self._h))

    def ring_info(self):
        dims = (C.c_int32 * 3)()
        h = C.c_double()
        ov = C.c_int64()
        _chk(lib().pct_cloud_ring_info(self._h, dims, C.byref(h), C.byref(ov)))
        return dict(dims=tuple(dims), cell_size=h.value, overflow_entries=ov.value, bucket_records=int(lib().pct_cloud_ring_bucket_records(self._h)))

    def reserve_queries(self, Q: int):
        _chk(lib().pct_cloud_reserve_queries(self._h, int(Q)))

    def nn(self, queries, algo: int = ALGO_AUTO):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, 3)
        idx = np.empty(len(q), np.uint32)
        d2 = np.empty(len(q), np.float64)
        _chk(lib().pct_nn_batch_algo(self._h, algo, _ptr(q), len(q), _ptr(idx), _ptr(d2)))
        return idx, d2

    def radius_count(self, queries, radii, algo: int = ALGO_AUTO):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, 3)
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(radii, np.float32), (len(q),)))
        cnt = np.empty(len(q), np.uint32)
        _chk(lib().pct_radius_count_batch_algo(self._h, algo, _ptr(q), _ptr(r), len(q), _ptr(cnt)))
        return cnt

    def radius_indices(self, center, radius, cap=None):
        q = np.ascontiguousarray(center, np.float32).reshape(3)
        cap = int(cap if cap is not None else max(len(self), 1))
        out = np.empty(cap, np.uint32)
        n = C.c_int64()
        _chk(lib().pct_radius_indices(self._h, _ptr(q), float(radius), _ptr(out), cap, C.byref(n)))
        return out[:min(n.value, cap)].copy(), n.value

    def radius_crop(self, center, radius, sort_by_distance=False):
        """lidar crop (camera_sensor.cpp:133-145): (indices u32, d2 fp64, cropped cloud float32 [k,3]) of the points within radius"""
        q = np.ascontiguousarray(center, np.float64).reshape(3)
        cap = max(len(self), 1)
        idx, d2, xyz = np.empty(cap, np.uint32), np.empty(cap, np.float64), np.empty((cap, 3), np.float32)
        n = C.c_int64()
        _chk(lib().pct_radius_crop(self._h, q.ctypes.data, float(radius), int(bool(sort_by_distance)), cap, idx.ctypes.data,
                                   d2.ctypes.data, xyz.ctypes.data, C.byref(n)))
        k = min(n.value, cap)
        return idx[:k].copy(), d2[:k].copy(), xyz[:k].copy()

    def crop_to(self, center, radius, dst: "Cloud"):
        """dst := points within radius of center, in this cloud's order, device to device"""
        q = np.ascontiguousarray(center, np.float64).reshape(3)
        _chk(lib().pct_cloud_crop_to(self._h, q.ctypes.data, float(radius), dst._h))

    def inflate(self, params: InflateParams, points):
        p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        rad = np.empty(len(p), np.float64)
        idx = np.empty(len(p), np.uint32)
        d2 = np.empty(len(p), np.float64)
        _chk(lib().pct_inflate_batch(self._h, C.byref(params), _ptr(p), len(p), _ptr(rad), _ptr(idx), _ptr(d2)))
        return rad, idx, d2

    def bezier_check(self, params: InflateParams, polycoef, seg_time, orders, t_start, stop_time, dt=0.02, cap=4096):
        coef = np.ascontiguousarray(polycoef, np.float64)
        st = np.ascontiguousarray(seg_time, np.float64)
        od = np.ascontiguousarray(orders, np.int32)
        traj = BezierTraj(coef.ctypes.data_as(C.POINTER(C.c_double)), coef.shape[1], st.ctypes.data_as(C.POINTER(C.c_double)),
                          od.ctypes.data_as(C.POINTER(C.c_int32)), len(st))
        pos = np.zeros((cap, 3), np.float64)
        rad = np.zeros(cap, np.float64)
        d2 = np.zeros(cap, np.float64)
        idx = np.zeros(cap, np.uint32)
        fh, ns = C.c_int64(), C.c_int64()
        _chk(lib().pct_bezier_check(self._h, C.byref(traj), C.byref(params), float(t_start), float(stop_time), float(dt),
                                    C.byref(fh), C.byref(ns), cap, _ptr(pos), _ptr(rad), _ptr(d2), _ptr(idx)))
        n = min(ns.value, cap)
        return dict(first_hit=fh.value, n=ns.value, pos=pos[:n], radius=rad[:n], d2=d2[:n], idx=idx[:n])

    def ctrl_points_check(self, params: InflateParams, polycoef, seg_time, orders, t_start=0.0, cap=1024):
        """pct_ctrl_points_check: the collision threshold test on the raw control points (world units)"""
        traj, keep = _traj(polycoef, seg_time, orders)
        pos = np.zeros((cap, 3), np.float64)
        rad = np.zeros(cap, np.float64)
        d2 = np.zeros(cap, np.float64)
        idx = np.zeros(cap, np.uint32)
        fh, nc = C.c_int64(), C.c_int64()
        _chk(lib().pct_ctrl_points_check(self._h, C.byref(traj), C.byref(params), float(t_start), C.byref(fh), C.byref(nc), cap,
                                         _ptr(pos), _ptr(rad), _ptr(d2), _ptr(idx)))
        n = min(nc.value, cap)
        return dict(first_hit=fh.value, n=nc.value, pos=pos[:n], radius=rad[:n], d2=d2[:n], idx=idx[:n])

    # device-buffer variants: raw pointers (e.g. torch tensor .data_ptr()) and a hipStream_t handle
    def nn_device(self, q_ptr: int, Q: int, idx_ptr: int, d2_ptr: int, stream: int = 0, algo: int = ALGO_AUTO):
        _chk(lib().pct_nn_batch_dev(self._h, algo, q_ptr, int(Q), idx_ptr, d2_ptr, stream))

    def inflate_device(self, params: InflateParams, pts_ptr: int, Q: int, radius_ptr: int, idx_ptr: int = 0, d2_ptr: int = 0, stream: int = 0):
        _chk(lib().pct_inflate_batch_dev(self._h, C.byref(params), pts_ptr, int(Q), radius_ptr, idx_ptr or None, d2_ptr or None, stream))

    def bezier_check_device(self, params: InflateParams, polycoef, seg_time, orders, t_start, stop_time, dt, cap, pos_ptr, radius_ptr, d2_ptr,
                            idx_ptr, first_hit_ptr, nsamples_ptr, stream: int = 0):
        traj, keep = _traj(polycoef, seg_time, orders)
        _chk(lib().pct_bezier_check_dev(self._h, C.byref(traj), C.byref(params), float(t_start), float(stop_time), float(dt), int(cap),
                                        pos_ptr or None, radius_ptr, d2_ptr or None, idx_ptr or None, first_hit_ptr, nsamples_ptr, stream))
        return keep                                     # host arrays the stream copies from: keep them alive until it has passed

    def radius_count_device(self, q_ptr: int, r_ptr: int, Q: int, cnt_ptr: int, stream: int = 0, algo: int = ALGO_AUTO):
        _chk(lib().pct_radius_count_batch_dev(self._h, algo, q_ptr, r_ptr, int(Q), cnt_ptr, stream))

    def set_timing(self, level: int):
        """0 = no events, 1 = dominant kernel only (default), 2 = + whole batch (needed by last_batch_ms)"""
        _chk(lib().pct_set_timing(self._h, int(level)))

    def debug_read_grid(self):
        """(cell_start uint32 [ncells + 1], records float32 [n, 4]) as built -- test hook"""
        cs = np.empty(self.grid_info()["ncells"] + 1, np.uint32)
        rec = np.empty((len(self), 4), np.float32)
        _chk(lib().pct_debug_read_grid(self._h, cs.ctypes.data, rec.ctypes.data))
        return cs, rec

    def set_timing_stride(self, stride: int):
        """the index path times only every stride-th launch from now on (pct_set_timing_stride)"""
        _chk(lib().pct_set_timing_stride(self._h, int(stride)))

    def kernel_ms_samples(self) -> int:
        n = C.c_uint64()
        _chk(lib().pct_kernel_ms_samples(self._h, C.byref(n)))
        return int(n.value)

    def kernel_ms_history(self, n: int = 64):
        """dominant-kernel durations (ms) of the last <= n batches, oldest first (HIP events on the launch stream)"""
        buf = (C.c_float * max(n, 1))()
        got = C.c_int()
        _chk(lib().pct_kernel_ms_history(self._h, buf, int(n), C.byref(got)))
        return [buf[i] for i in range(got.value)]

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _chk(lib().pct_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_batch_ms(self) -> float:
        ms = C.c_float()
        _chk(lib().pct_last_batch_ms(self._h, C.byref(ms)))
        return ms.value

    def set_work_counters(self, on: bool):
        _chk(lib().pct_set_work_counters(self._h, int(bool(on))))

    def last_work(self):
        a, b = C.c_uint64(), C.c_uint64()
        _chk(lib().pct_last_work(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_work_ex(self):
        """(points scanned, cell runs scanned, pyramid node visits) of the last instrumented batch"""
        out = (C.c_uint64 * 3)()
        _chk(lib().pct_last_work_ex(self._h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def verify_grid(self):
        """device-side structural check of the built index (pct_debug_verify_grid): dict of fault counts; sound = all zero and
        cell_start running from 0 to n"""
        out = (C.c_uint64 * 6)()
        _chk(lib().pct_debug_verify_grid(self._h, out))
        return dict(bad_ids=int(out[0]), duplicates=int(out[1]), misplaced=int(out[2]), decreasing=int(out[3]), first=int(out[4]), last=int(out[5]))

    def pyramid_info(self):
        lv, nodes, ef = C.c_int32(), C.c_int64(), C.c_double()
        _chk(lib().pct_cloud_pyramid_info(self._h, C.byref(lv), C.byref(nodes), C.byref(ef)))
        return dict(levels=lv.value, nodes=nodes.value, empty_fraction=ef.value)


def _traj(polycoef, seg_time, orders):
    coef = np.ascontiguousarray(polycoef, np.float64)
    st = np.ascontiguousarray(seg_time, np.float64)
    od = np.ascontiguousarray(orders, np.int32)
    t = BezierTraj(coef.ctypes.data_as(C.POINTER(C.c_double)), coef.shape[1], st.ctypes.data_as(C.POINTER(C.c_double)),
                   od.ctypes.data_as(C.POINTER(C.c_int32)), len(st))
    return t, (coef, st, od)


class ReplanPlan:
    """hipGraph-captured query side of one replan tick (pct_plan_create_replan): corridor-node inflation + sampled Bezier
    check + control-point check in one launch."""

    def __init__(self, cloud: Cloud, max_nodes: int, max_samples: int, max_segments: int):
        self._h = C.c_void_p()
        self._cloud = cloud
        self.max_nodes, self.max_samples, self.max_segments = int(max_nodes), int(max_samples), int(max_segments)
        _chk(lib().pct_plan_create_replan(cloud.handle, self.max_nodes, self.max_samples, self.max_segments, C.byref(self._h)))
        nc = 13 * self.max_segments
        self._b = dict(node_radius=np.zeros(self.max_nodes), node_idx=np.zeros(self.max_nodes, np.uint32), node_d2=np.zeros(self.max_nodes),
                       sample_pos=np.zeros((self.max_samples, 3)), sample_radius=np.zeros(self.max_samples), sample_d2=np.zeros(self.max_samples),
                       sample_idx=np.zeros(self.max_samples, np.uint32), ctrl_pos=np.zeros((nc, 3)), ctrl_radius=np.zeros(nc), ctrl_d2=np.zeros(nc),
                       ctrl_idx=np.zeros(nc, np.uint32))
        self._o = ReplanOut()
        for k, v in self._b.items():
            setattr(self._o, k, v.ctypes.data)

    def run(self, params: InflateParams, nodes, polycoef=None, seg_time=None, orders=None, t_start=0.0, stop_time=2.0, dt=0.02, want_nn=True,
            copy=True):
        nd = np.ascontiguousarray(nodes, np.float64).reshape(-1, 3)
        traj, keep = (None, None) if polycoef is None else _traj(polycoef, seg_time, orders)
        _chk(lib().pct_plan_replan_run(self._h, C.byref(params), _ptr(nd), len(nd), None if traj is None else C.byref(traj), float(t_start),
                                       float(stop_time), float(dt), int(bool(want_nn)), C.byref(self._o)))
        o, b = self._o, self._b
        ns, nc = min(o.nsamples, self.max_samples), o.nctrl
        cp = (lambda a: a.copy()) if copy else (lambda a: a)
        return dict(node_radius=cp(b["node_radius"][:len(nd)]), node_idx=cp(b["node_idx"][:len(nd)]), node_d2=cp(b["node_d2"][:len(nd)]),
                    nsamples=o.nsamples, first_hit_sample=o.first_hit_sample, sample_pos=cp(b["sample_pos"][:ns]),
                    sample_radius=cp(b["sample_radius"][:ns]), sample_d2=cp(b["sample_d2"][:ns]), sample_idx=cp(b["sample_idx"][:ns]),
                    nctrl=nc, first_hit_ctrl=o.first_hit_ctrl, ctrl_pos=cp(b["ctrl_pos"][:nc]), ctrl_radius=cp(b["ctrl_radius"][:nc]),
                    ctrl_d2=cp(b["ctrl_d2"][:nc]), ctrl_idx=cp(b["ctrl_idx"][:nc]))

    def last_run_us(self):
        """host wall time of the last run inside the library: (fill, graph launch, wait, read-out) in microseconds"""
        us = (C.c_double * 4)()
        _chk(lib().pct_plan_last_run_us(self._h, us))
        return tuple(us)

    def close(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.pct_plan_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


class NNPlan:
    """hipGraph-captured fixed-shape NN batch (config C5)."""

    def __init__(self, cloud: Cloud, Q: int, algo: int = ALGO_AUTO):
        self._h = C.c_void_p()
        self.Q = int(Q)
        self._cloud = cloud
        _chk(lib().pct_plan_create_nn(cloud.handle, algo, self.Q, C.byref(self._h)))

    def run(self, queries):
        q = np.ascontiguousarray(queries, np.float32).reshape(self.Q, 3)
        idx = np.empty(self.Q, np.uint32)
        d2 = np.empty(self.Q, np.float64)
        _chk(lib().pct_plan_run(self._h, _ptr(q), _ptr(idx), _ptr(d2)))
        return idx, d2

    def close(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:      # _lib is None during interpreter shutdown
            _lib.pct_plan_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close
