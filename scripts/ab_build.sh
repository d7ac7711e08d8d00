# A/B builds of the engine: bash scripts/ab_build.sh TAG [-DMACRO ...]  ->  pointcloudtraj_amd/lib/variants/libpct_engine_TAG.so
# (select at run time with PCT_ENGINE_SO=<that file>; libkdtree / libpct_corridor keep resolving libpct_engine.so by soname, so use
# the variants for engine-level probes and bench.py --replan-probe 0 only)
TAG=$1; shift
mkdir -p pointcloudtraj_amd/lib/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function -Iinclude "$@" \
  -o pointcloudtraj_amd/lib/variants/libpct_engine_$TAG.so pointcloudtraj_amd/csrc/engine.hip pointcloudtraj_amd/csrc/voxel.hip pointcloudtraj_amd/csrc/traj.hip pointcloudtraj_amd/csrc/nodeset.hip
