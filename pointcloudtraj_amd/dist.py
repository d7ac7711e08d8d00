"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" for the CPU tests).  SURVEY.md section 8(e): the cloud is split into contiguous
index ranges, every rank answers the whole (replicated) query batch against its shard, and one
exchange step merges the per-shard winners:

    d2*  = all_reduce(min) over ranks of the per-shard squared distances      (fp64, exact)
    idx* = all_reduce(min) of  (idx_r  if d2_r == d2*  else  INT64_MAX)        -> lowest global index
    count = all_reduce(sum) of per-shard radius counts

Global index = shard base + local index, and shards are contiguous ascending ranges, so "lowest
index among fp64-equal minima" is preserved across the merge.  Messages are Q*8 bytes: the
collective is latency-bound, so it is issued once per batch on the stream the kernels ran on.

torch is plumbing here (device memory, streams, process groups); all distance arithmetic happens
in the HIP kernels of libpct_engine.so.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

_I64_MAX = torch.iinfo(torch.int64).max


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """[begin, end) of rank's contiguous slice: rank r owns [r*n/world, (r+1)*n/world)."""
    return (rank * n_total) // world, ((rank + 1) * n_total) // world


def _staging(t: torch.Tensor, group=None) -> torch.Tensor:
    """gloo rehearsals of the multi-rank path on a GPU box stage through host memory (gloo's device
    support is not guaranteed on ROCm); with RCCL the tensors stay in HBM."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return t.cpu()
    return t


def merge_nearest(d2: torch.Tensor, idx: torch.Tensor, group=None) -> tuple[torch.Tensor, torch.Tensor]:
    """Exchange step for 1-NN.  d2: fp64 [Q] per-shard minima (+inf for an empty shard);
    idx: int64 [Q] GLOBAL indices (anything where d2 is +inf).  Returns the global (d2, idx)."""
    dev = d2.device
    d2, idx = _staging(d2, group), _staging(idx, group)
    best, cand = _merge_nearest(d2, idx, group)
    return best.to(dev), cand.to(dev)


def _merge_nearest(d2, idx, group):
    best = d2.clone()
    dist.all_reduce(best, op=dist.ReduceOp.MIN, group=group)
    # second message: 4 bytes per query while global indices fit int32 (NO_INDEX maps to INT32_MAX), else 8
    narrow = idx.dtype == torch.int32
    none = torch.iinfo(idx.dtype).max
    cand = torch.where((d2 == best) & torch.isfinite(d2), idx, torch.full_like(idx, none))
    dist.all_reduce(cand, op=dist.ReduceOp.MIN, group=group)
    if narrow:
        cand = torch.where(cand == none, torch.full_like(cand, -1), cand).to(torch.int64) & 0xFFFFFFFF
    return best, cand


def merge_counts(count: torch.Tensor, group=None) -> torch.Tensor:
    dev = count.device
    total = _staging(count, group).clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return total.to(dev)


def init_process_group_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment (torch.distributed.run)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = os.environ.get("PCT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class ShardedCloud:
    """This rank's contiguous shard of a global cloud, resident in this rank's HBM."""

    def __init__(self, n_total: int, rank: int, world: int, device_index: int, group=None):
        from . import engine as E   # HIP runtime is loaded here; torch is already imported (see engine._preload_hip_runtime)
        self.E = E
        self.rank, self.world, self.group = rank, world, group
        self.n_total = int(n_total)
        self.begin, self.end = shard_range(self.n_total, rank, world)
        torch.cuda.set_device(device_index)
        E.init(device_index)
        self.device = torch.device("cuda", device_index)
        self.cloud = E.Cloud(max(self.end - self.begin, 1))
        self.cloud.set_index_base(self.begin)

    def set_input_local(self, local_points: np.ndarray):
        """local_points = rows [begin, end) of the global cloud."""
        assert len(local_points) == self.end - self.begin
        self.cloud.set_input(local_points)

    def build_grid(self, cell_size: float = 0.0):
        if len(self.cloud):
            self.cloud.build_grid(cell_size)

    def reserve(self, Q: int, depth: int = 2):
        """Result buffers for batches of up to Q queries; `depth` sets of them so that the exchange step of one
        batch can overlap the kernels of the next (nn_submit).  A result handed out by nn_submit lives in its set and is
        overwritten by the depth-th submit after it."""
        self.cloud.reserve_queries(Q)
        self._slots = [(torch.empty(Q, dtype=torch.int32, device=self.device),
                        torch.empty(Q, dtype=torch.float64, device=self.device)) for _ in range(max(depth, 1))]
        self._idx32, self._d2 = self._slots[0]
        self._cnt = torch.empty(Q, dtype=torch.int32, device=self.device)
        # pipelined exchange (nn_submit over RCCL): reduced distances and candidate / merged indices per slot
        self._merged = [(torch.empty(Q, dtype=torch.float64, device=self.device), torch.empty(Q, dtype=torch.int32, device=self.device))
                        for _ in self._slots] if self.world > 1 else []
        self._slot_free = [None] * len(self._slots)   # event: the exchange that last read this slot has finished
        self._next_slot = 0
        self._comm = torch.cuda.Stream(device=self.device) if self.world > 1 else None

    def _nn_into(self, q: torch.Tensor, algo: int, idx32: torch.Tensor, d2: torch.Tensor):
        Q = q.shape[0]
        s = torch.cuda.current_stream().cuda_stream
        self.cloud.nn_device(q.data_ptr(), Q, idx32.data_ptr(), d2.data_ptr(), s, algo)
        if self.world > 1 and self.n_total < 2 ** 31 - 1:
            # indices < 2^31 are exchanged as int32 (the u32 bit pattern is non-negative; NO_INDEX -> INT32_MAX).  A non-empty
            # shard answers every query, so the fix-up kernels are only queued for an empty one (they would sit on the
            # compute stream, in front of the next batch)
            i32 = idx32[:Q]
            if self.end > self.begin:
                return d2[:Q], i32
            return d2[:Q], torch.where(i32 < 0, torch.full_like(i32, torch.iinfo(torch.int32).max), i32)
        # u32 -> int64 (NO_INDEX stays recognisable through d2 == +inf)
        return d2[:Q], idx32[:Q].to(torch.int64) & 0xFFFFFFFF

    def nn_local(self, q: torch.Tensor, algo: int = 0):
        """Per-shard kernel on torch's current stream.  q: float32 [Q,3] on this device."""
        return self._nn_into(q, algo, self._idx32, self._d2)

    def nn_submit(self, q: torch.Tensor, algo: int = 0):
        """Pipelined form of nn(): the shard kernels run on the current stream into the next result slot, the
        exchange step runs behind them on a side stream, and the call returns at once with
        (d2, idx, done_event).  The next batch's kernels therefore overlap this batch's all_reduce pair
        (8+4 bytes per query over xGMI, which at Q = 1M costs about as much as the kernels).  The consumer waits
        on done_event (torch.cuda.current_stream().wait_event(ev) or ev.synchronize()) before reading; with a single rank
        done_event is None and the results are simply ordered on the current stream.

        Index contract, the same on every branch (one rank, RCCL, gloo rehearsal): int32 global indices (int64 once the cloud
        holds 2^31 - 1 points or more), -1 = no point anywhere (every shard empty).  The returned tensors live in this
        batch's result slot: they stay valid until the `depth`-th later submit, whose kernels first wait for this batch's
        exchange (slot_free event) -- a consumer that needs them longer copies them after waiting on done_event."""
        if self.world == 1:
            # nothing to merge: hand back the engine's own buffers (idx = the u32 indices as an int32 view, no
            # conversion kernel in the step; PCT_NO_INDEX reads as -1)
            Q = q.shape[0]
            self.cloud.nn_device(q.data_ptr(), Q, self._idx32.data_ptr(), self._d2.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream, algo)
            return self._d2[:Q], self._idx32[:Q], None         # ordered on the current stream: no event needed (one costs ~4 us)
        slot = self._next_slot
        self._next_slot = (slot + 1) % len(self._slots)
        cur = torch.cuda.current_stream()
        if self._slot_free[slot] is not None:
            cur.wait_event(self._slot_free[slot])       # the exchange `depth` batches ago still owns these buffers
        idx32, d2buf = self._slots[slot]
        d2, idx = self._nn_into(q, algo, idx32, d2buf)
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self._comm):
            self._comm.wait_event(ready)
            idx.record_stream(self._comm)
            staged = idx.is_cuda and dist.get_backend(self.group) == "gloo" and os.environ.get("PCT_DIST_DEVICE_COLLECTIVES") != "1"
            if staged:
                best, cand = merge_nearest(d2, idx, self.group)         # rehearsal: staged through the host
                cand = self._normalise(cand, idx.dtype == torch.int32)
            elif idx.dtype == torch.int32:
                # RCCL, indices < 2^31: copy + all_reduce(min) on d2, ONE fused mask kernel (pct_merge_mask_dev), all_reduce(min)
                # on the masked indices -- into this slot's preallocated buffers (valid until the slot is reused, i.e. until the
                # second submit after this one).  idx comes back as int32, INT32_MAX = no point anywhere.
                Q = q.shape[0]
                best, cand = self._merged[slot][0][:Q], self._merged[slot][1][:Q]
                best.copy_(d2)
                dist.all_reduce(best, op=dist.ReduceOp.MIN, group=self.group)
                self.E.merge_mask_device(d2.data_ptr(), best.data_ptr(), idx32.data_ptr(), cand.data_ptr(), Q, self._comm.cuda_stream)
                dist.all_reduce(cand, op=dist.ReduceOp.MIN, group=self.group)
                cand.masked_fill_(cand == torch.iinfo(torch.int32).max, -1)      # on the exchange stream, off the kernels' path
            else:
                best, cand = _merge_nearest(d2, idx, self.group)
                cand = self._normalise(cand, False)
            done = torch.cuda.Event()
            done.record(self._comm)
        self._slot_free[slot] = done
        best.record_stream(cur)
        cand.record_stream(cur)
        return best, cand, done

    @staticmethod
    def _normalise(cand: torch.Tensor, narrow: bool) -> torch.Tensor:
        """merged indices of merge_nearest / _merge_nearest (int64, 0xFFFFFFFF or INT64_MAX = none) -> nn_submit's contract"""
        none = (cand == 0xFFFFFFFF) | (cand == _I64_MAX)
        cand = torch.where(none, torch.full_like(cand, -1), cand)
        return cand.to(torch.int32) if narrow else cand

    def nn(self, q: torch.Tensor, algo: int = 0):
        d2, idx = self.nn_local(q, algo)
        if self.world == 1:
            return d2, idx
        return merge_nearest(d2, idx, self.group)

    def radius_count(self, q: torch.Tensor, r: torch.Tensor, algo: int = 0):
        Q = q.shape[0]
        s = torch.cuda.current_stream().cuda_stream
        self.cloud.radius_count_device(q.data_ptr(), r.data_ptr(), Q, self._cnt.data_ptr(), s, algo)
        cnt = self._cnt[:Q].to(torch.int64)
        return cnt if self.world == 1 else merge_counts(cnt, self.group)

    def close(self):
        self.cloud.close()


# ---------------------------------------------------------------------------------------------------------------------------
# Spatial ownership of queries.  Index-range shards make every rank answer every query, which scales the
# brute-force kernels (per-rank work = Q x N/W pairs) but not the cell-pruned one (a query costs ~49 scanned points whatever the
# shard's size).  Here the cloud is re-distributed once into W slabs along its longest axis (equal point counts, a halo of a few
# point spacings on both sides), the replicated query batch is split by slab, each rank answers only the queries whose slab it
# owns, and the SAME exchange step (all_reduce(min) on d2, all_reduce(min) on the matching global indices) delivers the merged
# answer -- non-owners simply offer +inf.  An answer is certified when the point found is closer than the edge of the owner's halo;
# the rare uncertified query is flagged through the distance reduction itself (the owner offers -1) and answered by everybody in
# a second, small round.  Results are identical to the single-cloud answer (lowest global index on exact ties: points are kept
# in ascending global-index order inside a slab, and the index reduction takes the minimum across slabs).
# ---------------------------------------------------------------------------------------------------------------------------
class _EngineSearcher:
    """local search on this rank's slab through libpct_engine.so (cell-pruned kernel)"""

    def __init__(self, device_index: int):
        from . import engine as E
        self.E = E
        torch.cuda.set_device(device_index)
        E.init(device_index)
        self.device = torch.device("cuda", device_index)
        self.cloud = None

    def load(self, pts: np.ndarray):
        if self.cloud is not None:
            self.cloud.close()
        self.cloud = self.E.Cloud(max(len(pts), 1))
        self.cloud.set_input(pts)
        if len(pts):
            self.cloud.build_grid()
        self._cap = 0

    def search(self, q: torch.Tensor):
        """q: float32 [m,3] on self.device -> (local idx int64 [m], d2 float64 [m]) on the same device"""
        m = q.shape[0]
        if m > self._cap:
            self.cloud.reserve_queries(m)
            self._idx = torch.empty(m, dtype=torch.int32, device=self.device)
            self._d2 = torch.empty(m, dtype=torch.float64, device=self.device)
            self._cap = m
        if m:
            self.cloud.nn_device(q.data_ptr(), m, self._idx.data_ptr(), self._d2.data_ptr(), torch.cuda.current_stream().cuda_stream, 0)
        return self._idx[:m].to(torch.int64) & 0xFFFFFFFF, self._d2[:m]

    def close(self):
        if self.cloud is not None:
            self.cloud.close()


class SpatialShardedCloud:
    """Slab-owned shards with routed queries.  `searcher`: object with load(points float32 [n,3]), search(q) -> (local idx, d2) and
    a `device` attribute (default: the engine on `device_index`; the CPU tests plug the oracle in)."""

    def __init__(self, rank: int, world: int, device_index: int = 0, group=None, searcher=None, halo_spacings: float = 4.0):
        self.rank, self.world, self.group = rank, world, group
        self.searcher = searcher if searcher is not None else _EngineSearcher(device_index)
        self.device = self.searcher.device
        self.halo_spacings = float(halo_spacings)
        self.stats = {"owned": 0, "uncertified": 0, "batches": 0}

    def _ar(self, t: torch.Tensor, op):
        s = _staging(t, self.group)
        dist.all_reduce(s, op=op, group=self.group)
        return s.to(t.device) if s is not t else t

    def build(self, local_points: np.ndarray, index_begin: int):
        """local_points: this rank's rows [index_begin, index_begin + n) of the global cloud (float32 [n,3]).  Collective."""
        W = self.world
        pts = np.ascontiguousarray(local_points, np.float32).reshape(-1, 3)
        n = len(pts)
        # collectives run on device tensors over RCCL, on host tensors over gloo
        cdev = self.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        # global bounding box and point count
        lo = torch.from_numpy(pts.min(0).astype(np.float64) if n else np.full(3, np.inf)).to(cdev)
        hi = torch.from_numpy(pts.max(0).astype(np.float64) if n else np.full(3, -np.inf)).to(cdev)
        cnt = torch.tensor([float(n)], dtype=torch.float64, device=cdev)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=self.group)
        self.n_total = int(cnt.item())
        if self.n_total == 0:
            self.axis, self.cuts, self.halo = 0, np.zeros(W + 1), 0.0
            self.searcher.load(np.zeros((0, 3), np.float32))
            self.gidx = torch.zeros(0, dtype=torch.int64, device=self.device)
            return
        lo, hi = lo.cpu().numpy(), hi.cpu().numpy()
        ext = np.maximum(hi - lo, 1e-30)
        self.axis = int(np.argmax(ext))
        a = self.axis
        # equal-count cuts along the axis from a global histogram
        nb = 4096
        if n:
            h = torch.from_numpy(np.histogram(pts[:, a].astype(np.float64), bins=nb, range=(lo[a], hi[a] + ext[a] * 1e-9))[0].astype(np.float64))
        else:
            h = torch.zeros(nb, dtype=torch.float64)
        h = h.to(cdev)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        cum = np.cumsum(h.cpu().numpy())
        edges = np.linspace(lo[a], hi[a] + ext[a] * 1e-9, nb + 1)
        cuts = [-np.inf]
        for k in range(1, W):
            cuts.append(float(edges[min(int(np.searchsorted(cum, self.n_total * k / W)) + 1, nb)]))
        cuts.append(np.inf)
        self.cuts = np.asarray(cuts)
        spacing = float(np.cbrt(np.prod(np.maximum(ext, ext.max() * 1e-3)) / self.n_total))
        self.halo = self.halo_spacings * spacing
        # every point goes to its owner slab and to every slab whose halo it falls into
        x = pts[:, a].astype(np.float64)
        gid = index_begin + np.arange(n, dtype=np.int64)
        send_pts, send_gid = [], []
        for k in range(W):
            m = (x >= self.cuts[k] - self.halo) & (x < self.cuts[k + 1] + self.halo)
            send_pts.append(pts[m])
            send_gid.append(gid[m])
        counts = torch.tensor([len(g) for g in send_gid], dtype=torch.int64, device=cdev)
        recv_counts = torch.zeros(W, dtype=torch.int64, device=cdev)
        dist.all_to_all_single(recv_counts, counts, group=self.group)
        sp = torch.from_numpy(np.concatenate(send_pts).astype(np.float32).reshape(-1, 3)).to(cdev)
        sg = torch.from_numpy(np.concatenate(send_gid)).to(cdev)
        ins, outs = counts.cpu().tolist(), recv_counts.cpu().tolist()
        rp = torch.empty((sum(outs), 3), dtype=torch.float32, device=cdev)
        rg = torch.empty(sum(outs), dtype=torch.int64, device=cdev)
        dist.all_to_all_single(rp, sp, output_split_sizes=outs, input_split_sizes=ins, group=self.group)
        dist.all_to_all_single(rg, sg, output_split_sizes=outs, input_split_sizes=ins, group=self.group)
        order = torch.argsort(rg, stable=True)                   # ascending global index: "lowest local index" = "lowest global index"
        self.searcher.load(rp[order].cpu().numpy())
        self.gidx = rg[order].to(self.device)
        self.slab_points = int(rg.numel())

    def _owner_of(self, q: torch.Tensor) -> torch.Tensor:
        cuts = torch.as_tensor(self.cuts[1:-1], dtype=torch.float64, device=q.device)
        return torch.bucketize(q[:, self.axis].to(torch.float64), cuts, right=True)

    def nn(self, q: torch.Tensor):
        """q: float32 [Q,3] on self.device, the same on every rank.  Returns (d2 float64 [Q], idx int64 [Q], -1 = empty cloud) on every rank."""
        Q = q.shape[0]
        dev = q.device
        inf = float("inf")
        d2 = torch.full((Q,), inf, dtype=torch.float64, device=dev)
        gi = torch.full((Q,), _I64_MAX, dtype=torch.int64, device=dev)
        if self.n_total == 0:
            return d2, torch.full((Q,), -1, dtype=torch.int64, device=dev)
        mine = torch.nonzero(self._owner_of(q) == self.rank).squeeze(1)
        if mine.numel() and self.gidx.numel():
            li, ld = self.searcher.search(q[mine].contiguous())
            valid = torch.isfinite(ld)
            g = torch.where(valid, self.gidx[torch.where(valid, li, torch.zeros_like(li))], torch.full_like(li, _I64_MAX))
            # certified: nothing outside this slab's halo can be nearer than what was found
            x = q[mine, self.axis].to(torch.float64)
            lo_edge, hi_edge = self.cuts[self.rank] - self.halo, self.cuts[self.rank + 1] + self.halo
            margin = torch.minimum(x - lo_edge if np.isfinite(lo_edge) else torch.full_like(x, inf),
                                   hi_edge - x if np.isfinite(hi_edge) else torch.full_like(x, inf))
            cert = valid & (margin > 0) & (ld < margin * margin)       # strictly: a point just outside the halo at exactly that distance could tie with a lower index
            d2[mine] = torch.where(cert, ld, torch.full_like(ld, -1.0))     # -1 = "ask everybody": it wins the min-reduction
            gi[mine] = torch.where(cert, g, torch.full_like(g, _I64_MAX))
            self.stats["owned"] += int(mine.numel())
            self.stats["uncertified"] += int((~cert).sum().item())
        elif mine.numel():
            d2[mine] = -1.0                                      # an owner without points: everybody answers
        self.stats["batches"] += 1
        best = self._ar(d2.clone(), dist.ReduceOp.MIN)
        cand = torch.where((d2 == best) & torch.isfinite(d2) & (d2 >= 0), gi, torch.full_like(gi, _I64_MAX))
        cand = self._ar(cand, dist.ReduceOp.MIN)
        flagged = torch.nonzero(best < 0).squeeze(1)             # the same set on every rank (derived from reduced data)
        if flagged.numel():
            fd = torch.full((flagged.numel(),), inf, dtype=torch.float64, device=dev)
            fg = torch.full((flagged.numel(),), _I64_MAX, dtype=torch.int64, device=dev)
            if self.gidx.numel():
                li, ld = self.searcher.search(q[flagged].contiguous())
                valid = torch.isfinite(ld)
                fd = torch.where(valid, ld, fd)
                fg = torch.where(valid, self.gidx[torch.where(valid, li, torch.zeros_like(li))], fg)
            fbest = self._ar(fd.clone(), dist.ReduceOp.MIN)
            fcand = self._ar(torch.where((fd == fbest) & torch.isfinite(fd), fg, torch.full_like(fg, _I64_MAX)), dist.ReduceOp.MIN)
            best[flagged] = fbest
            cand[flagged] = fcand
        return best, torch.where(cand == _I64_MAX, torch.full_like(cand, -1), cand)

    def close(self):
        if hasattr(self.searcher, "close"):
            self.searcher.close()
