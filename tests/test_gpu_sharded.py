"""GPU test of pointcloudtraj_amd.dist.ShardedCloud end to end: two ranks (gloo, both on the box's one card),
each holding a contiguous index-range shard in HBM and running the HIP kernels; the merged answers of nn() and of
the pipelined nn_submit() must equal the single-cloud oracle bit for bit (indices and fp64 d2), ties included.
With RCCL on a multi-GPU node the same code path keeps the tensors in HBM (dist._staging)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from test_dist_gloo import ROOT, _free_port

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir, device_collectives):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", PCT_DIST_BACKEND="gloo", PCT_DIST_DEVICE_COLLECTIVES="1" if device_collectives else "0")
    import torch
    import torch.distributed as dist
    from pointcloudtraj_amd import dist as D, engine as E, synth
    D.init_process_group_from_env()
    base = synth.clustered_points(8, 60_000, 0, 30)
    pts = np.concatenate([synth.uniform_points(7, 80_000, 0, 30), base, base])     # exact ties across the two shards
    sc = D.ShardedCloud(len(pts), rank, world, 0)
    sc.set_input_local(pts[sc.begin:sc.end])
    sc.build_grid()
    batches = [np.concatenate([synth.uniform_points(20 + k, 20_000, -3, 33), base[k::97][:500]]) for k in range(4)]
    sc.reserve(max(len(b) for b in batches), depth=4)     # the three pipelined results below are read after the last submit
    outs = []
    for k, qh in enumerate(batches):
        q = torch.from_numpy(qh).to(sc.device)
        if k == 0:
            d2, idx = sc.nn(q, E.ALGO_GRID)
            outs.append((d2, idx, None))
        else:
            outs.append(sc.nn_submit(q, E.ALGO_GRID if k % 2 else E.ALGO_STREAM))   # queued back to back, read afterwards
    cq = torch.from_numpy(batches[0]).to(sc.device)
    cnt = sc.radius_count(cq, torch.full((len(cq),), 1.5, dtype=torch.float32, device=sc.device))
    if rank == 0:
        from oracle import oracle as O
        bad = []
        for k, ((d2, idx, ev), qh) in enumerate(zip(outs, batches)):
            if ev is not None:
                ev.synchronize()
            wi, wd = O.brute_nearest(pts, qh)
            nd = int((d2.cpu().numpy() != wd).sum())
            ni = int((idx.cpu().numpy() != wi.astype(np.int64)).sum())
            if nd or ni:
                bad.append(f"batch {k}: {nd} d2 and {ni} index mismatches")
        nc = int((cnt.cpu().numpy() != O.brute_count(pts, batches[0], 1.5).astype(np.int64)).sum())
        if nc:
            bad.append(f"{nc} count mismatches")
        with open(os.path.join(out_dir, "ok"), "w") as f:
            f.write("; ".join(bad) if bad else "1")
    torch.cuda.synchronize()
    dist.barrier()
    sc.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("device_collectives", [False, True])
def test_two_rank_sharded_cloud_on_one_card(tmp_path, device_collectives):
    """device_collectives=True runs nn_submit's RCCL code path (collectives on device tensors, the fused mask kernel, per-slot
    result buffers) with gloo doing the transport; False is the host-staged rehearsal path."""
    from oracle import oracle as O
    O.build()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), device_collectives), nprocs=2, join=True)
    assert open(tmp_path / "ok").read() == "1"


def test_fused_merge_mask_equals_the_elementwise_recipe():
    """pct_merge_mask_dev (the one kernel between the two all_reduce(min) calls over RCCL) against dist._merge_nearest's
    elementwise formula, including +inf distances (empty shard), NO_INDEX and exact ties."""
    import torch
    from pointcloudtraj_amd import engine as E
    E.init(0)
    g = torch.Generator(device="cpu").manual_seed(5)
    Q = 100_003
    mine = torch.rand(Q, generator=g, dtype=torch.float64)
    other = torch.rand(Q, generator=g, dtype=torch.float64)
    other[::7] = mine[::7]                                  # exact ties
    mine[::11] = float("inf")                               # this shard saw nothing
    other[::33] = float("inf")
    idx = torch.randint(0, 2 ** 31 - 2, (Q,), generator=g, dtype=torch.int64)
    idx[::11] = 0xFFFFFFFF
    best = torch.minimum(mine, other)
    want = torch.where((mine == best) & torch.isfinite(mine), idx, torch.full_like(idx, 2 ** 31 - 1)).to(torch.int32)
    d_m, d_b = mine.cuda(), best.cuda()
    d_i32 = torch.from_numpy((idx.numpy() & 0xFFFFFFFF).astype("uint32").view("int32")).cuda()    # u32 bit patterns
    out = torch.empty(Q, dtype=torch.int32, device="cuda")
    E.merge_mask_device(d_m.data_ptr(), d_b.data_ptr(), d_i32.data_ptr(), out.data_ptr(), Q, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), want)


def test_bench_two_ranks_on_one_card():
    """`python bench.py --gpus 2` started plainly (no launcher, no RANK in the environment) must start its two ranks itself and
    print ONE line with n_gpus = 2 and the C4 workload (strong scaling, merged answers/s); on this one-card box the ranks share
    the card and exchange over gloo (rehearsal).  Reduced sizes: the point of the test is the launch path and the line's shape."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--points-total", "4000000",
                        "--queries", "65536", "--preheat-steps", "10"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_points"] == 4000000
    assert d["config"]["points_per_gpu"] == 2000000 and d["config"]["backend"] == "gloo"
    assert abs(d["config"]["query_shard_evaluations_per_s"] - 2 * d["value"]) < 1e-6 * d["value"]
    assert d["one_gpu_whole_cloud"]["same_answers_as_sharded"] is True
    assert d["c4_q4096"]["brute_force_ms_per_batch"] > 0
    sp = d["spatial_routing"]                              # slab-owned queries through the HIP kernels, same answers, each rank ~half the batch
    assert "error" not in sp, sp
    assert sp["same_answers_as_index_range_shards"] is True
    assert 0.3 * 65536 < sp["owned_queries_rank0_per_batch"] < 0.7 * 65536 and sp["uncertified_rank0_per_batch"] < 0.02 * 65536


def test_cpp_shard_client_over_rccl():
    """examples/shard_client.cpp through include/pct_shard.h: ncclCommInitRank, the per-shard kernels, ncclAllReduce(min) x 2 and the
    two merge kernels, all from a plain C++ process.  This box has one card and RCCL refuses two ranks on one device, so the
    communicator has ONE rank here (the collectives still run); examples/run_shard_client.sh 8 is the 8-GPU form."""
    import subprocess
    r = subprocess.run(["bash", os.path.join(ROOT, "examples", "run_shard_client.sh"), "1", "3000000", "2048"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


@pytest.mark.parametrize("world,halo", [(2, 4.0), (3, 4.0), (8, 4.0), (3, 0.0)])
def test_routed_form_in_the_c_abi_equals_the_single_cloud(world, halo):
    """include/pct_shard.h routed form (slab ownership: pct_shard_route_build_world / pct_shard_route_nn_world) with 2, 3 and 8 ranks in
    ONE process on the box's one card (RCCL refuses two ranks on one device; the phases are the same, the bytes are moved by copies):
    every rank's answers must equal the single cloud's -- indices and fp64 distances, duplicated rows across slabs (lowest global
    index) and queries on top of points included.  halo 0 forces the second round (everybody answers what the owner cannot
    certify) for a good share of the batch."""
    import subprocess
    from pointcloudtraj_amd import build
    r = subprocess.run([build.SHARD_CLIENT, "local", str(world), "1500000", "40000", str(halo)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert " 0 mismatches" in r.stdout, r.stdout
    last = r.stdout.strip().splitlines()[-1]
    second = int(last.split(" in the second round")[0].split()[-1])
    assert (second > 100) if halo == 0.0 else (second < 400), last


def test_routed_c_abi_inside_a_torch_distributed_process():
    """bench.py --gpus N drives the routed form of libpct_shard.so from a torch.distributed rank: the library's own RCCL communicator
    next to torch's, the rendezvous token broadcast through torch.distributed, device buffers owned by torch.  One rank here (one
    card): the communicator, the slab build (collectives on a one-rank communicator) and the routed batch must work in that process
    and equal the plain cloud's answers."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["PCT_ROOT"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
t = torch.ones(4, device="cuda"); dist.all_reduce(t)
from pointcloudtraj_amd import engine as E, shard as SH, synth
E.init(0)
pts = synth.uniform_points(11, 600_000, 0, 80); q = synth.uniform_points(12, 50_000, -2, 82)
tok = torch.zeros(SH.ID_BYTES, dtype=torch.uint8, device="cuda")
tok.copy_(torch.frombuffer(bytearray(SH.unique_id()), dtype=torch.uint8)); dist.broadcast(tok, src=0)
sh = SH.Shard(bytes(tok.cpu().numpy().tobytes()), 0, 1, 0)
route = sh.route(pts, 0, 4.0)
dq = torch.from_numpy(q).cuda(); di = torch.empty(len(q), dtype=torch.int32, device="cuda"); dd = torch.empty(len(q), dtype=torch.float64, device="cuda")
for _ in range(2):
    route.nn_device(dq.data_ptr(), len(q), di.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
c = E.Cloud(len(pts)); c.set_input(pts); c.build_grid()
wi, wd = c.nn(q, E.ALGO_GRID)
ok = np.array_equal(di.cpu().numpy().view(np.uint32), wi) and np.array_equal(dd.cpu().numpy(), wd)
di.zero_(); dd.zero_()
route.nn_partitioned_device(dq.data_ptr(), len(q), di.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream)      # the weak-scaling form
torch.cuda.synchronize()
ok = ok and np.array_equal(di.cpu().numpy().view(np.uint32), wi) and np.array_equal(dd.cpu().numpy(), wd)
print("stats", route.stats(), "equal", ok)
route.close(); sh.close(); c.close(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''
    env = dict(os.environ, PCT_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "equal True" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
