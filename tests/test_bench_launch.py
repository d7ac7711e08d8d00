"""CPU test of bench.py's self-launch path (`python bench.py --gpus N` with no RANK / WORLD_SIZE in the environment): the parent
must start N ranks with torch.distributed's environment contract on 127.0.0.1, hand rank 0's JSON line through and fail when a
rank fails.  The ranks are played by tests/helpers/fake_rank.py (gloo, no GPU, no engine) through the PCT_BENCH_CHILD hook;
the real ranks run in tests/test_gpu_sharded.py::test_bench_two_ranks_on_one_card."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, fail_rank=None, die_early=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PCT_BENCH_CHILD"] = os.path.join(ROOT, "tests", "helpers", "fake_rank.py")
    if fail_rank is not None:
        env["FAKE_FAIL_RANK"] = str(fail_rank)
    if die_early is not None:
        env["FAKE_DIE_EARLY_RANK"] = str(die_early)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", *extra],
                          env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks_gloo():
    r = _run([])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly ONE line on stdout"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] == 3.0 and d["steps"] == 2 and d["warmup"] == 1


def test_self_launch_propagates_a_failing_rank():
    r = _run([], fail_rank=1)
    assert r.returncode != 0
    assert "exit codes" in r.stderr


def test_self_launch_does_not_hang_when_a_rank_dies_before_the_rendezvous():
    """rank 1 exits at once; rank 0 would wait for it in the rendezvous (default timeout: half an hour) -- the launcher must end it"""
    import time
    t0 = time.time()
    r = _run([], die_early=1)
    assert r.returncode != 0 and "exit codes" in r.stderr
    assert time.time() - t0 < 120
