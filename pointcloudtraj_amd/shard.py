"""ctypes binding of libpct_shard.so (include/pct_shard.h): the multi-GPU exchange step in the C ABI -- index-range shards
(all_reduce(min) pair) and the routed form (slab ownership, owned answers exchanged as records).  One process per GPU over RCCL;
the rendezvous token travels by whatever channel the caller has (bench.py broadcasts it through torch.distributed)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from . import engine as _engine

ID_BYTES = 128
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_build.SHARD_SO):
            raise FileNotFoundError(f"{_build.SHARD_SO} is missing: run __graft_entry__.build()")
        _engine.lib()
        L = C.CDLL(_build.SHARD_SO)
        vp, i64 = C.c_void_p, C.c_int64
        L.pct_shard_unique_id.argtypes = [vp]
        L.pct_shard_init.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.pct_shard_destroy.argtypes = [vp]
        L.pct_shard_route_build.argtypes = [vp, vp, i64, i64, i64, C.c_double, C.POINTER(vp)]
        L.pct_shard_route_nn_dev.argtypes = [vp, vp, i64, vp, vp, vp]
        L.pct_shard_route_nn_partitioned_dev.argtypes = [vp, vp, i64, vp, vp, vp]
        L.pct_shard_route_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.pct_shard_route_destroy.argtypes = [vp]
        _lib = L
    return _lib


def _chk(rc, what):
    if rc:
        raise RuntimeError(f"{what} failed ({rc}): {_engine.lib().pct_last_error().decode(errors='replace')}")


def unique_id() -> bytes:
    buf = (C.c_ubyte * ID_BYTES)()
    _chk(lib().pct_shard_unique_id(buf), "pct_shard_unique_id")
    return bytes(buf)


class Shard:
    """this rank's communicator (ncclCommInitRank: collective over all ranks)"""

    def __init__(self, token: bytes, rank: int, world: int, device: int):
        self.h = C.c_void_p()
        buf = (C.c_ubyte * ID_BYTES).from_buffer_copy(token)
        _chk(lib().pct_shard_init(buf, rank, world, device, C.byref(self.h)), "pct_shard_init")
        self.rank, self.world = rank, world

    def route(self, local_points, index_begin: int, halo_spacings: float = 4.0) -> "Route":
        return Route(self, local_points, index_begin, halo_spacings)

    def close(self):
        if getattr(self, "h", None) and self.h.value and _lib is not None:
            _lib.pct_shard_destroy(self.h)
            self.h = C.c_void_p()


class Route:
    """slab ownership (pct_shard_route_build: collective) and the routed batch"""

    def __init__(self, shard: Shard, local_points, index_begin: int, halo_spacings: float):
        a = np.ascontiguousarray(local_points, np.float32).reshape(-1, 3)
        self.h = C.c_void_p()
        _chk(lib().pct_shard_route_build(shard.h, a.ctypes.data_as(C.c_void_p), len(a), 12, int(index_begin), float(halo_spacings), C.byref(self.h)),
             "pct_shard_route_build")

    def nn_device(self, q_ptr: int, Q: int, idx_ptr: int, d2_ptr: int, stream: int = 0):
        _chk(lib().pct_shard_route_nn_dev(self.h, q_ptr, int(Q), idx_ptr, d2_ptr, stream), "pct_shard_route_nn_dev")

    def nn_partitioned_device(self, q_ptr: int, Q: int, idx_ptr: int, d2_ptr: int, stream: int = 0):
        """this rank's OWN Q queries -> its own Q answers (the queries travel to the owners of their slabs and the answers back)"""
        _chk(lib().pct_shard_route_nn_partitioned_dev(self.h, q_ptr, int(Q), idx_ptr, d2_ptr, stream), "pct_shard_route_nn_partitioned_dev")

    def stats(self):
        sp, ow, un, ba = C.c_int64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(lib().pct_shard_route_stats(self.h, C.byref(sp), C.byref(ow), C.byref(un), C.byref(ba)), "pct_shard_route_stats")
        return dict(slab_points=sp.value, owned=ow.value, uncertified=un.value, batches=ba.value)

    def close(self):
        if getattr(self, "h", None) and self.h.value and _lib is not None:
            _lib.pct_shard_route_destroy(self.h)
            self.h = C.c_void_p()
