"""CPU tests (gloo, world_size 2 and 3) of the multi-GPU exchange step in pointcloudtraj_amd/dist.py:
contiguous index-range shards, per-shard nearest results, all_reduce(min) on fp64 d2 followed by
all_reduce(min) on the indices that attain it, all_reduce(sum) for radius counts.

There is no GPU here, so the per-shard kernel is played by the oracle (tests may use it); what is
under test is the merge: the sharded answer must equal the single-cloud answer bit for bit,
including the lowest-global-index rule on exact ties and empty shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, case, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from pointcloudtraj_amd import synth
    from pointcloudtraj_amd.dist import merge_counts, merge_nearest, shard_range
    if case == "uniform":
        pts = synth.uniform_points(7, n_total, 0, 50)
    else:   # grid-aligned cloud with every point duplicated at the far end of the index range: exact ties across shards
        base = synth.clustered_points(8, n_total // 2, 0, 20)
        pts = np.concatenate([base, base])
    q = np.concatenate([synth.uniform_points(9, 300, -5, 55), pts[:40]])
    b, e = shard_range(len(pts), rank, world)
    local = pts[b:e]
    if len(local):
        li, ld = O.brute_nearest(local, q)
        gi = torch.from_numpy(li.astype(np.int64) + b)
        gd = torch.from_numpy(ld.copy())
        cnt = torch.from_numpy(O.brute_count(local, q, 3.0).astype(np.int64))
    else:   # empty shard: what the engine reports (d2 = +inf, index = NO_INDEX)
        gi = torch.full((len(q),), 0xFFFFFFFF, dtype=torch.int64)
        gd = torch.full((len(q),), float("inf"), dtype=torch.float64)
        cnt = torch.zeros(len(q), dtype=torch.int64)
    d2, idx = merge_nearest(gd, gi)
    # the int32 exchange form used when all global indices fit 31 bits
    gi32 = torch.where(torch.isfinite(gd), gi, torch.full_like(gi, 2 ** 31 - 1)).to(torch.int32)
    d2b, idxb = merge_nearest(gd, gi32)
    assert torch.equal(d2b, d2) and torch.equal(idxb, idx)
    total = merge_counts(cnt)
    if rank == 0:
        wi, wd = O.brute_nearest(pts, q)
        wc = O.brute_count(pts, q, 3.0)
        ok = bool(np.array_equal(d2.numpy(), wd) and np.array_equal(idx.numpy(), wi.astype(np.int64))
                  and np.array_equal(total.numpy(), wc.astype(np.int64)))
        with open(os.path.join(out_dir, f"ok_{case}_{world}"), "w") as f:
            f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,case", [(2, 5000, "uniform"), (2, 4000, "ties"), (3, 1001, "uniform"), (3, 2, "uniform")])
def test_sharded_merge_equals_single_cloud(tmp_path, world, n_total, case):
    from oracle import oracle as O
    O.build()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, case, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / f"ok_{case}_{world}").read() == "1"


def test_shard_ranges_partition_the_cloud():
    sys.path.insert(0, ROOT)
    from pointcloudtraj_amd.dist import shard_range
    for n in (0, 1, 7, 8, 100, 100_000_001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


class _OracleSearcher:
    """stands in for the per-slab HIP kernel on CPU: exhaustive fp64 scan, lowest local index on ties"""

    def __init__(self):
        self.device = torch.device("cpu")

    def load(self, pts):
        self.pts = np.ascontiguousarray(pts, np.float32)

    def search(self, q):
        from oracle import oracle as O
        m = q.shape[0]
        if m == 0 or len(self.pts) == 0:
            return torch.zeros(m, dtype=torch.int64), torch.full((m,), float("inf"), dtype=torch.float64)
        i, d = O.brute_nearest(self.pts, q.numpy())
        return torch.from_numpy(i.astype(np.int64)), torch.from_numpy(d.copy())


def _spatial_worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from pointcloudtraj_amd import synth
    from pointcloudtraj_amd.dist import SpatialShardedCloud, shard_range
    if case == "uniform":
        pts = synth.uniform_points(17, 30_000, 0, 50)
        halo = 4.0
    elif case == "thin_halo":      # a halo far thinner than the point spacing: most boundary queries must take the second round
        pts = synth.uniform_points(18, 4_000, 0, 50)
        halo = 0.05
    elif case == "ties":           # grid-aligned points, every one duplicated at the far end of the index range
        base = synth.clustered_points(8, 3_000, 0, 20)
        pts = np.concatenate([base, base])
        halo = 2.0
    else:                          # two points in all: most slabs are empty
        pts = synth.uniform_points(19, 2, 0, 50)
        halo = 4.0
    q = np.concatenate([synth.uniform_points(9, 600, -5, 55), pts[:40], synth.uniform_points(10, 20, -400, 400)])
    b, e = shard_range(len(pts), rank, world)
    sc = SpatialShardedCloud(rank, world, searcher=_OracleSearcher(), halo_spacings=halo)
    sc.build(pts[b:e], b)
    d2, idx = sc.nn(torch.from_numpy(q))
    d2b, idxb = sc.nn(torch.from_numpy(q[::-1].copy()))          # a second batch through the same shards
    own = torch.tensor([float(sc.stats["owned"]), float(sc.stats["uncertified"])], dtype=torch.float64)
    dist.all_reduce(own)
    if rank == 0:
        wi, wd = O.brute_nearest(pts, q)
        ok = bool(np.array_equal(d2.numpy(), wd) and np.array_equal(idx.numpy(), wi.astype(np.int64)) and
                  np.array_equal(d2b.numpy(), wd[::-1]) and np.array_equal(idxb.numpy(), wi[::-1].astype(np.int64)))
        ok = ok and int(own[0].item()) == 2 * len(q)             # every query had exactly one owner
        if case == "uniform":
            ok = ok and own[1].item() < 0.1 * own[0].item()      # a 4-spacing halo certifies nearly everything in the first round
        if case == "thin_halo":
            ok = ok and own[1].item() > 0                         # ... and a thin one exercises the second round
        with open(os.path.join(out_dir, f"ok_spatial_{case}_{world}"), "w") as f:
            f.write("1" if ok else f"0 owned {own.tolist()}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "uniform"), (3, "uniform"), (3, "thin_halo"), (2, "ties"), (3, "tiny")])
def test_spatially_routed_queries_equal_single_cloud(tmp_path, world, case):
    """SpatialShardedCloud (slab ownership + halo + certified answers + second round): merged answers must equal the single-cloud
    answer bit for bit, each query answered by exactly one owner in the first round"""
    from oracle import oracle as O
    O.build()
    port = _free_port()
    mp.spawn(_spatial_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / f"ok_spatial_{case}_{world}").read() == "1"
