"""Reproduce a failure of tests/test_gpu_ring.py::test_ring_index_randomised_appends and dump the state of the missed point."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from pointcloudtraj_amd import engine as E
from oracle import oracle as O
from test_gpu_ring import Mirror
E.init(0); O.build()
L = E.lib(); L.pct_debug_ring_slot.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_uint32)]
import time
rng = np.random.default_rng(4242)
done = 0
LIMIT = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
while done < LIMIT:
    t_cfg = time.perf_counter()
    cap = int(rng.choice([1000, 5000, 40_000, 150_000])); ext = float(rng.choice([1.0, 30.0, 400.0]))
    c, m = E.Cloud(cap), Mirror(cap)
    if rng.random() < 0.5: c.ring_index(); cfg = "auto"
    else:
        cs = ext / float(rng.choice([20, 60, 200])); ex = (ext, ext, ext * float(rng.choice([1.0, 0.1]))); c.ring_index(cs, ex); cfg = f"cell {cs} ext {ex}"
    drift = np.float32(rng.uniform(-0.3, 0.3, 3) * ext); centre = np.zeros(3, np.float32)
    hist = []
    for _ in range(int(rng.integers(4, 16))):
        n = int(min(cap, max(1, rng.choice([1, 17, cap // 50, cap // 7, cap // 2, cap]))))
        kind = rng.choice(["uniform", "dups", "clusters"]); u = rng.random((n, 3))
        if kind == "uniform": p = u * ext
        elif kind == "dups":
            k = max(1, n // 80); p = (rng.random((k, 3)) * ext)[rng.integers(0, k, n)]
        else:
            cc = rng.random((5, 3)) * ext; p = cc[rng.integers(0, 5, n)] + rng.normal(0, ext * 0.004, (n, 3))
        centre = centre + drift
        f = (p * [1, 1, 0.2] + centre).astype(np.float32)
        before = (m.nxt, m.count)
        ta = time.perf_counter(); c.append(f); tb = time.perf_counter(); m.append(f); hist.append((kind, n, before))
        q = np.concatenate([(rng.random((60, 3)) * ext * 1.3 - 0.15 * ext) * [1, 1, 0.2] + centre, f[rng.integers(0, n, 20)],
                            (rng.random((8, 3)) - 0.5) * ext * 30 + centre]).astype(np.float32)
        bi, bd = O.brute_nearest_mt(m.live(), q)
        tc = time.perf_counter(); i1, d1 = c.nn(q); td = time.perf_counter()
        if tb - ta > 0.5 or td - tc > 0.5: print(f"   slow step: append {tb - ta:.2f} s ({kind} n {n}), nn {td - tc:.2f} s, info {c.ring_info()}", flush=True)
        done += 1
        bad = np.nonzero(d1 != bd)[0]
        if len(bad):
            print("FAIL at step", done, "cap", cap, "ext", ext, cfg, "info", c.ring_info(), "history", hist)
            for k in bad[:4]:
                out = (C.c_uint32 * 6)()
                L.pct_debug_ring_slot(c.handle, int(bi[k]), out)
                print(" query", k, q[k], "true idx", bi[k], "d2", bd[k], "got idx", i1[k], d1[k], "true point", m.xyz[bi[k]],
                      "where %08x bucket %d head %d tail %d id %d ovf_len %d" % tuple(out))
            sys.exit(1)
    print(f"steps {done} cap {cap} ext {ext} {cfg}: {time.perf_counter() - t_cfg:.2f} s, info {c.ring_info()}", flush=True)
    c.close()
print("no failure in", done, "steps")
