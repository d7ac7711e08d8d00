"""In-tree build of the HIP libraries for gfx950 (hipcc cross-compiles without a GPU).

Products (git-ignored, but they travel to the GPU box with the gpurun snapshot):
    pointcloudtraj_amd/lib/libpct_engine.so   kernels + batched C ABI (include/pct_engine.h)
    pointcloudtraj_amd/lib/libkdtree.so       drop-in kd_* ABI (include/kdtree/kdtree.h) over the engine
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib")
ENGINE_SO = os.environ.get("PCT_ENGINE_SO") or os.path.join(LIB, "libpct_engine.so")     # PCT_ENGINE_SO: an A/B build of the engine (scripts/ab_build.sh)
KDTREE_SO = os.path.join(LIB, "libkdtree.so")
DEMO_BIN = os.path.join(LIB, "seam_demo")
NODE_BIN = os.path.join(LIB, "node_call_sites")
CORRIDOR_SO = os.path.join(LIB, "libpct_corridor.so")
SHARD_SO = os.path.join(LIB, "libpct_shard.so")
SHARD_CLIENT = os.path.join(LIB, "shard_client")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
COMMON = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
          "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")]


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_all(force: bool = False, verbose: bool = False) -> None:
    os.makedirs(LIB, exist_ok=True)
    hdrs = [os.path.join(ROOT, "include", "pct_engine.h"), os.path.join(ROOT, "include", "kdtree", "kdtree.h"),
            os.path.join(ROOT, "include", "kdtree", "kdtree_ext.h")]
    hdrs += [os.path.join(ROOT, "include", "pct_voxel.h"), os.path.join(ROOT, "include", "pct_traj.h")]
    eng_units = [os.path.join(CSRC, "engine.hip"), os.path.join(CSRC, "voxel.hip"), os.path.join(CSRC, "traj.hip"), os.path.join(CSRC, "nodeset.hip")]
    eng_src = eng_units + [os.path.join(CSRC, f) for f in ("kernels.hpp", "gridbuild.hpp", "pyramid.hpp", "bernstein.hpp", "brute2.hpp", "ring.hpp", "ring_host.inc", "engine_internal.hpp")]
    if force or _stale(ENGINE_SO, eng_src + hdrs):
        cmd = [HIPCC, *COMMON, "-o", ENGINE_SO, *eng_units]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    kd_src = [os.path.join(CSRC, "kdtree_gpu.cpp")]
    if os.path.exists(kd_src[0]) and (force or _stale(KDTREE_SO, kd_src + hdrs + [ENGINE_SO])):
        # host-only C++ (no device code): it reaches the GPU through libpct_engine.so's C ABI
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
               "-o", KDTREE_SO, kd_src[0], "-L" + LIB, "-lpct_engine", "-Wl,-rpath,$ORIGIN", "-Wl,-soname,libkdtree.so"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)


    cor_src = os.path.join(CSRC, "corridor.cpp")
    cor_hdrs = [os.path.join(ROOT, "include", h) for h in ("pct_corridor.h", "pct_corridor_finder.hpp", "pct_obstacle_map.hpp")]
    if os.path.exists(cor_src) and os.path.exists(KDTREE_SO) and (force or _stale(CORRIDOR_SO, [cor_src, KDTREE_SO] + cor_hdrs + hdrs)):
        # host-only C++: the corridor finder's bookkeeping; every distance it needs comes from the two libraries above
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
               "-o", CORRIDOR_SO, cor_src, "-L" + LIB, "-lkdtree", "-lpct_engine", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + LIB,
               "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-soname,libpct_corridor.so"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    demo_src = os.path.join(ROOT, "examples", "seam_demo.cpp")
    if os.path.exists(demo_src) and os.path.exists(KDTREE_SO) and (force or _stale(DEMO_BIN, [demo_src, KDTREE_SO] + hdrs + [os.path.join(ROOT, "include", "pct_obstacle_map.hpp")])):
        # a plain g++ client of the two libraries: the link line INTEGRATION.md gives the planner
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-o", DEMO_BIN, demo_src,
               "-L" + LIB, "-lkdtree", "-lpct_engine", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + LIB,
               "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)


    # multi-GPU exchange step: host-only C++ over libpct_engine.so's C ABI + RCCL (ncclAllReduce over xGMI)
    shard_src = os.path.join(CSRC, "shard.cpp")
    shard_hdr = os.path.join(ROOT, "include", "pct_shard.h")
    if os.path.exists(shard_src) and (force or _stale(SHARD_SO, [shard_src, shard_hdr, ENGINE_SO] + hdrs)):
        cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", SHARD_SO, shard_src,
               "-L" + LIB, "-lpct_engine", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-soname,libpct_shard.so"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    client_src = os.path.join(ROOT, "examples", "shard_client.cpp")
    if os.path.exists(client_src) and os.path.exists(SHARD_SO) and (force or _stale(SHARD_CLIENT, [client_src, SHARD_SO, shard_hdr] + hdrs)):
        # the link line a planner node adds: -lpct_shard -lpct_engine (RCCL and the HIP runtime come in through libpct_shard.so)
        # (the routed form takes device buffers: the client allocates them with the HIP runtime API, hence -I/opt/rocm/include -lamdhip64)
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
               "-o", SHARD_CLIENT, client_src,
               "-L" + LIB, "-lpct_shard", "-lpct_engine", "-pthread", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + LIB,
               "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    node_src = os.path.join(ROOT, "examples", "node_call_sites.cpp")
    if os.path.exists(node_src) and os.path.exists(KDTREE_SO) and (force or _stale(NODE_BIN, [node_src, KDTREE_SO] + hdrs + cor_hdrs)):
        # the planner node's own statements (sim_planning_demo.cpp:344-416) compiled against the replacement class
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-o", NODE_BIN, node_src,
               "-L" + LIB, "-lkdtree", "-lpct_engine", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + LIB,
               "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    lat_src = os.path.join(ROOT, "examples", "latency_bench.cpp")
    lat_bin = os.path.join(LIB, "latency_bench")
    if os.path.exists(lat_src) and os.path.exists(KDTREE_SO) and (force or _stale(lat_bin, [lat_src, KDTREE_SO] + hdrs + [os.path.join(ROOT, "include", "pct_obstacle_map.hpp")])):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-o", lat_bin, lat_src,
               "-L" + LIB, "-lkdtree", "-lpct_engine", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + LIB,
               "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)


if __name__ == "__main__":
    build_all(force=True, verbose=True)
