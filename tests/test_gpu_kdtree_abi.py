"""GPU parity tests for the drop-in libkdtree.so (the reference's 26 kd_* functions served by the
engine), against the golden vectors produced by the reference's own kdtree.c."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["device", "host_below_4096"])
def K(request):
    """every test of this file runs twice: with every single query answered by the HIP kernels (threshold 0), and with the default
    dispatch -- node sets of up to 4096 nodes scanned on the host (kdtree_ext.h kdx_set_host_threshold; BASELINE config C1), larger
    ones on the device.  Same fixtures, same required answers."""
    from pointcloudtraj_amd import engine, kdtree
    engine.init(0)
    kdtree.set_host_threshold(0 if request.param == "device" else 4096)
    yield kdtree
    kdtree.set_host_threshold(-1)


@pytest.mark.parametrize("name", ["kd_nn_n1.npz", "kd_nn_n2.npz", "kd_nn_n17.npz", "kd_nn_n1000.npz",
                                  "kd_nn_duplicates.npz", "kd_nn_clustered.npz", "kd_nn_c1_crop5m.npz"])
def test_kd_nearestf_golden(K, name):
    """every row equals the compiled reference's answer -- ON EXACT TIES TOO (kd_nn_duplicates: 128 tied queries, 67 of them with
    a winner other than the lowest index; kd_nn_clustered: 127 tied, 92 such): the drop-in replays the reference's walk order
    among the tied nodes (kdtree.c:345-402, 432-436)"""
    g = load_golden(name)
    pts = g["points"]
    t = K.KDTree()
    t.insert(pts)
    nq = min(len(g["queries"]), 400)
    ids, pos = t.nearest(g["queries"][:nq])
    assert np.array_equal(ids, g["ref_idx"][:nq])
    assert np.array_equal(pos, pts[ids].astype(np.float64))     # kd_res_item hands back the stored fp64 position
    t.close()


@pytest.mark.parametrize("name", ["kd_range_n1000.npz", "kd_range_lattice.npz", "kd_range_c1_crop5m.npz"])
def test_kd_nearest_rangef_iteration_order(K, name):
    """Same hits in the same ITERATION ORDER as the reference (reverse pre-order of its walk),
    including the lattice case where the reference drops hits with fabs(dx) == range."""
    g = load_golden(name)
    t = K.KDTree()
    t.insert(g["points"])
    offs = g["offsets"]
    for i, (q, r) in enumerate(zip(g["queries"], g["radii"])):
        want = g["ids"][offs[i]:offs[i + 1]]
        got = t.range_ids(q, float(r))
        assert np.array_equal(got, want), f"query {i}"
    t.close()


def test_interleaved_insert_and_query(K, oracle):
    """The RRT* usage pattern: insert one node, query, insert, range-query ... against the oracle port."""
    from pointcloudtraj_amd import synth
    pts = synth.uniform_points(111, 300, 0, 10)
    L = oracle.port_lib()
    ot = L.okd_create(3)
    t = K.KDTree()
    for i, p in enumerate(pts):
        t.insert(p[None])
        L.okd_insertf_batch  # noqa: B018 (documenting the oracle entry point used below)
        w = np.ascontiguousarray(p, np.float64)
        assert L.okd_insert(ot, w.ctypes.data_as(C.POINTER(C.c_double)), C.c_void_p(i + 1)) == 0
        if i % 7 == 0:
            q = synth.uniform_points(500 + i, 1, 0, 10)[0]
            ids, _ = t.nearest(q[None])
            r = L.okd_nearestf(ot, q.ctypes.data_as(C.POINTER(C.c_float)))
            want = L.okd_res_item_id(r)
            L.okd_res_free(r)
            assert ids[0] == want
            got = t.range_ids(q, 2.5)
            rr = L.okd_nearest_rangef(ot, q.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(2.5))
            exp = []
            while not L.okd_res_end(rr):
                exp.append(L.okd_res_item_id(rr))
                L.okd_res_next(rr)
            L.okd_res_free(rr)
            assert list(got) == exp
    L.okd_free(ot)
    t.close()


def test_kd_api_edges(K):
    g = load_golden("kd_api_edges.npz")
    L = K.lib()
    t = L.kd_create(3)
    q = (C.c_float * 3)(1, 2, 3)
    assert int(L.kd_nearestf(t, q) is None) == int(g["nn_empty_is_null"]) == 1
    rs = L.kd_nearest_rangef(t, q, C.c_float(5.0))
    assert int(rs is not None) == int(g["range_empty_valid"]) == 1
    assert L.kd_res_size(rs) == int(g["range_empty_size"]) == 0 and L.kd_res_end(rs)
    L.kd_res_free(rs)
    order = []
    CB = C.CFUNCTYPE(None, C.c_void_p)
    cb = CB(lambda p: order.append(int(p or 0)))
    L.kd_data_destructor(t, C.cast(cb, C.c_void_p))
    for i, p in enumerate(np.ascontiguousarray(g["destructor_points"], np.float64)):
        assert L.kd_insert(t, p.ctypes.data_as(C.POINTER(C.c_double)), C.c_void_p(i + 1)) == 0
    q2 = (C.c_float * 3)(*[float(v) for v in g["item3_query"]])
    rs = L.kd_nearestf(t, q2)
    xin = g["item3_in"]
    x, y, z = C.c_double(xin[0]), C.c_double(xin[1]), C.c_double(xin[2])
    ret = L.kd_res_item3(rs, C.byref(x), C.byref(y), C.byref(z))
    assert int(ret is None) == int(g["item3_ret_null"]) == 1
    assert np.array_equal(np.float64([x.value, y.value, z.value]), g["item3_out"])
    assert int(L.kd_res_item_data(rs) or 0) == int(g["item3_nn_payload"])
    assert L.kd_res_size(rs) == 1 and not L.kd_res_end(rs) and L.kd_res_next(rs) == 0 and L.kd_res_end(rs)
    L.kd_res_rewind(rs)
    assert not L.kd_res_end(rs)
    L.kd_res_free(rs)
    # the 3/3f/double entry points agree with the f one
    r1 = L.kd_nearest3f(t, 2.2, 6.1, 1.3)
    r2 = L.kd_nearest3(t, float(np.float32(2.2)), float(np.float32(6.1)), float(np.float32(1.3)))
    assert int(L.kd_res_item_data(r1) or 0) == int(L.kd_res_item_data(r2) or 0) == int(g["item3_nn_payload"])
    L.kd_res_free(r1); L.kd_res_free(r2)
    rr = L.kd_nearest_range3f(t, 5.0, 5.0, 5.0, 0.5)
    assert L.kd_res_size(rr) == 2          # the two coincident points at (5,5,5)
    L.kd_res_free(rr)
    L.kd_clear(t)
    assert order == list(g["destructor_order"])
    assert int(L.kd_nearestf(t, q) is None) == int(g["after_clear_nn_null"]) == 1
    L.kd_free(t)
    assert L.kd_create(0) is None and L.kd_create(1025) is None      # documented limits: 1 <= k <= 1024
    t5 = L.kd_create(5)
    assert t5 is not None and L.kd_insert3(t5, 1.0, 2.0, 3.0, None) == -1 and L.kd_nearest3(t5, 1.0, 2.0, 3.0) is None   # x,y,z forms: <= 3 dimensions
    L.kd_free(t5)


@pytest.mark.parametrize("name", ["k2", "k3_f64", "k7", "k17", "k1_dups", "k5_lattice", "k3_mixed", "k4_big"])
def test_kd_general_k_golden(K, name):
    """kd_create(k) for k = 1 .. 17 and doubles fp32 cannot hold (kdtree.c:112-131, 167-209): the drop-in keeps such trees as fp64
    columns in HBM (csrc/nodeset.hip) and returns the compiled reference's nearest node -- its walk's winner on exact ties
    (k5_lattice, k1_dups) -- and its range ITERATION order.  k3_mixed starts as an fp32 tree and is handed an unrepresentable double
    halfway; k4_big (20 000 rows) is answered by the kernels under either dispatch."""
    from test_oracle_golden import general_k_case
    dim, rows, q, rad, nn, ids, offs = general_k_case(load_golden("kd_general_k.npz"), name)
    t = K.KDTreeN(dim)
    t.insert(rows)
    got, pos = t.nearest(q)
    assert np.array_equal(got, nn)
    assert np.array_equal(pos, rows[nn])                              # kd_res_item: the stored doubles, all k of them
    for i in range(len(q)):
        assert np.array_equal(t.range_ids(q[i], float(rad[i])), ids[offs[i]:offs[i + 1]]), f"query {i}"
    t.close()


def test_nodeset_abi_direct():
    """include/pct_engine.h "node sets": lowest node number + tie count at the minimum, ascending hit lists, growth, clear"""
    from pointcloudtraj_amd import engine as E
    E.init(0)
    L = E.lib()
    dp = C.POINTER(C.c_double)
    h = C.c_void_p()
    assert L.pct_nodeset_create(4, 2, C.byref(h)) == 0
    idx, ties, d2, n = C.c_uint32(), C.c_uint32(), C.c_double(), C.c_int64()
    q = np.float64([1, 1, 1, 1])
    assert L.pct_nodeset_nearest(h, q.ctypes.data_as(dp), C.byref(idx), C.byref(d2), C.byref(ties)) == 0
    assert idx.value == 0xFFFFFFFF and d2.value == np.inf and ties.value == 0
    rng = np.random.default_rng(5)
    rows = np.round(rng.random((5000, 4)) * 3.0)                      # integer lattice: many exact ties
    for a in range(0, 5000, 1237):                                    # several appends across two capacity growths
        blk = np.ascontiguousarray(rows[a:a + 1237])
        assert L.pct_nodeset_append(h, blk.ctypes.data_as(dp), len(blk)) == 0
    assert L.pct_nodeset_size(h) == 5000 and L.pct_nodeset_dim(h) == 4
    for qq in (np.float64([1, 1, 1, 1]), np.float64([0.5, 2.5, 1.0, 3.0]), np.float64([9, 9, 9, 9])):
        d = ((rows - qq) ** 2)
        s = ((d[:, 0] + d[:, 1]) + d[:, 2]) + d[:, 3]
        assert L.pct_nodeset_nearest(h, qq.ctypes.data_as(dp), C.byref(idx), C.byref(d2), C.byref(ties)) == 0
        assert d2.value == s.min() and idx.value == int(np.flatnonzero(s == s.min())[0]) and ties.value == int((s == s.min()).sum())
        out = np.empty(5000, np.uint32)
        r2 = 2.25
        assert L.pct_nodeset_radius_indices_r2(h, qq.ctypes.data_as(dp), C.c_double(r2), out.ctypes.data_as(C.c_void_p), 5000, C.byref(n)) == 0
        assert np.array_equal(out[:n.value], np.flatnonzero(s <= r2).astype(np.uint32))
        assert L.pct_nodeset_radius_indices_r2(h, qq.ctypes.data_as(dp), C.c_double(r2), out.ctypes.data_as(C.c_void_p), 3, C.byref(n)) == 0
        assert n.value == int((s <= r2).sum())                        # the count is reported whatever the capacity
    assert L.pct_nodeset_clear(h) == 0 and L.pct_nodeset_size(h) == 0
    assert L.pct_nodeset_destroy(h) == 0


def test_cpp_client_of_both_libraries():
    """examples/seam_demo.cpp: a plain g++ client that uses pct::ObstacleMap (include/pct_obstacle_map.hpp)
    and the kd_* C API the way the planner does, checking every answer against host loops."""
    import os
    import subprocess
    from pointcloudtraj_amd import build
    assert os.path.exists(build.DEMO_BIN), "run __graft_entry__.build() first"
    r = subprocess.run([build.DEMO_BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


def test_planner_node_call_sites_compile_and_run_against_the_replacement_class():
    """examples/node_call_sites.cpp: sim_planning_demo.cpp:344-356, 381-416 with `_rrtPathPlaner` declared as
    pct::SafeRegionRrtStar -- SafeRegionExpansion(_path_find_limit), SafeRegionRefine(_time_limit_1),
    SafeRegionEvaluate(_time_limit_2) are the reference's own statements (wall-clock seconds, corridor_finder.h:97-99)"""
    import os
    import subprocess
    from pointcloudtraj_amd import build
    assert os.path.exists(build.NODE_BIN), "run __graft_entry__.build() first"
    r = subprocess.run([build.NODE_BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "initial corridor" in r.stdout and "all checks passed" in r.stdout, r.stdout


@pytest.mark.parametrize("seed,n,lattice", [(1, 700, False), (2, 1500, True), (3, 70_000, False)])
def test_differential_sequences_against_the_port(K, oracle, seed, n, lattice):
    """Random op sequences (insert bursts, kd_nearestf, kd_nearest_rangef with assorted radii, kd_clear + refill) on the
    drop-in and on the oracle port side by side: same range ITERATION order, same nearest distance, same nearest id
    wherever the minimum is unique.  lattice=True puts the points on a 0.5 grid (ties, hits exactly at the radius, and
    split-plane cases with fabs(dx) == range); n = 70 000 crosses from the host-mapped node set to the device-resident one."""
    from pointcloudtraj_amd import synth
    rng = np.random.default_rng(seed)
    pts = synth.uniform_points(900 + seed, n, 0, 20)
    if lattice:
        pts = (np.round(pts * 2) / 2).astype(np.float32)
    L = oracle.port_lib()
    ot = L.okd_create(3)
    t = K.KDTree()
    state = {"base": 0, "n": 0}

    def insert(k):
        chunk = np.ascontiguousarray(pts[state["base"] + state["n"]: state["base"] + state["n"] + k])
        t.insert(chunk)
        assert L.okd_insertf_batch(ot, chunk, len(chunk)) == 0          # ids below are insertion indices, not payloads
        state["n"] += len(chunk)

    def check():
        q = (rng.uniform(-1, 21, 3)).astype(np.float32)
        if lattice and rng.random() < 0.5:
            q = (np.round(q * 2) / 2).astype(np.float32)
        qp = q.ctypes.data_as(C.POINTER(C.c_float))
        ids, pos = t.nearest(q[None])
        r = L.okd_nearestf(ot, qp)
        want = L.okd_res_item_id(r)
        L.okd_res_free(r)
        cur = pts[state["base"]: state["base"] + state["n"]].astype(np.float64)
        d = cur - q.astype(np.float64)
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        assert d2[ids[0]] == d2[want] == d2.min()
        assert ids[0] == want                                        # ties included: the reference's walk order decides
        rad = float(rng.choice([0.5, 1.0, 1.5, 2.5, 0.0, 4.0]))
        got = t.range_ids(q, rad)
        rr = L.okd_nearest_rangef(ot, qp, C.c_float(rad))
        exp = []
        while not L.okd_res_end(rr):
            exp.append(L.okd_res_item_id(rr))
            L.okd_res_next(rr)
        L.okd_res_free(rr)
        assert list(got) == exp

    big = n > 65536
    insert(3)
    check()
    while state["n"] < (n if big else n // 2):
        insert(int(rng.integers(1, 40)) if not big else 23_000)
        for _ in range(2 if big else 1):
            check()
    if not big:                                                          # kd_clear, then a different set of points
        t.L.kd_clear(t.h)
        t.n = 0
        L.okd_clear(ot)
        state["base"] += state["n"]
        state["n"] = 0
        while state["base"] + state["n"] < n - 40:
            insert(int(rng.integers(1, 40)))
            check()
    L.okd_free(ot)
    t.close()
