"""CPU tests: the built shared libraries load and export every symbol the public headers declare.
No compute call is made (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    from pointcloudtraj_amd import build
    build.build_all()
    return build


def test_engine_exports_every_declared_symbol(built):
    names = declared("pct_engine.h", "pct_")
    assert len(names) >= 30
    L = C.CDLL(built.ENGINE_SO)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_engine_fails_loudly_without_gpu(built):
    from pointcloudtraj_amd import engine as E
    if E.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(E.EngineError) as ei:
        E.init(0)
    assert ei.value.code == 1 and "no host fallback" in str(ei.value)
    with pytest.raises(E.EngineError):
        E.Cloud(16)


def test_kdtree_exports_every_reference_function(built):
    names = declared("kdtree/kdtree.h", "kd_")
    assert len(names) == 26
    import ctypes as C2
    from pointcloudtraj_amd import engine as E
    E._preload_hip_runtime()
    L = C2.CDLL(built.KDTREE_SO)          # libpct_engine.so comes in as its DT_NEEDED dependency (RUNPATH $ORIGIN), local scope
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    want = ["kd_create", "kd_free", "kd_clear", "kd_data_destructor", "kd_insert", "kd_insertf", "kd_insert3", "kd_insert3f",
            "kd_nearest", "kd_nearestf", "kd_nearest3", "kd_nearest3f", "kd_nearest_range", "kd_nearest_rangef",
            "kd_nearest_range3", "kd_nearest_range3f", "kd_res_free", "kd_res_size", "kd_res_rewind", "kd_res_end",
            "kd_res_next", "kd_res_item", "kd_res_itemf", "kd_res_item3", "kd_res_item3f", "kd_res_item_data"]
    assert sorted(want) == names


def test_kdtree_create_fails_loudly_without_gpu(built, capfd):
    from pointcloudtraj_amd import engine as E, kdtree as K
    if E.device_count() > 0:
        pytest.skip("a GPU is present")
    assert K.lib().kd_create(3) is None
    assert "no host fallback" in capfd.readouterr().err


def test_corridor_exports_every_declared_symbol(built):
    import ctypes as C2
    from pointcloudtraj_amd import engine as E
    names = declared("pct_corridor.h", "pct_corridor_")
    assert len(names) >= 14
    E._preload_hip_runtime()
    L = C2.CDLL(built.CORRIDOR_SO)        # pulls libkdtree.so and libpct_engine.so through DT_NEEDED, local scope
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_shard_library_exports_every_declared_symbol_and_links_rccl(built):
    """libpct_shard.so (include/pct_shard.h): the multi-GPU exchange step for C / C++ callers; it must resolve RCCL's
    ncclAllReduce itself (DT_NEEDED librccl) and the example client must have linked against it"""
    import ctypes as C2
    import subprocess
    from pointcloudtraj_amd import engine as E
    names = declared("pct_shard.h", "pct_shard_")
    assert len(names) >= 11
    E._preload_hip_runtime()
    L = C2.CDLL(built.SHARD_SO)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    needed = subprocess.run(["readelf", "-d", built.SHARD_SO], capture_output=True, text=True).stdout
    assert "librccl.so" in needed and "libpct_engine.so" in needed
    und = subprocess.run(["nm", "-D", "--undefined-only", built.SHARD_SO], capture_output=True, text=True).stdout
    assert "ncclAllReduce" in und and "ncclCommInitRank" in und and "pct_merge_mask_dev" in und
    assert os.path.exists(built.SHARD_CLIENT)


def test_public_headers_are_plain_c99_and_cxx17():
    """The drop-in boundary is a C ABI: every include/*.h must compile as C99 on its own (no C++, no torch/HIP types), and the
    two C++ mirrors as C++17 without any GPU toolchain header."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    for h in ("pct_engine.h", "pct_shard.h", "pct_voxel.h", "pct_traj.h", "pct_corridor.h", "kdtree/kdtree.h", "kdtree/kdtree_ext.h"):
        r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + inc, "-x", "c", "-fsyntax-only", "-"],
                           input=f'#include "{h}"\n', text=True, capture_output=True)
        assert r.returncode == 0, h + ":\n" + r.stderr
    for h in ("pct_obstacle_map.hpp", "pct_corridor_finder.hpp"):
        r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + inc, "-x", "c++", "-fsyntax-only", "-"],
                           input=f'#include "{h}"\n', text=True, capture_output=True)
        assert r.returncode == 0, h + ":\n" + r.stderr
