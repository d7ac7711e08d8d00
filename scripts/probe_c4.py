"""Config C4 probe alone (bench.c4_probe): the 100 M-point cloud on one card -- streaming kernel at Q = 1, 2, 4 and the 1 M-query
batch through the index.  Run under rocprofv3 by scripts/prof_r2.sh (profiles/r02_c4_*)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointcloudtraj_amd import engine as E, synth
E.init(0)
print(json.dumps(bench.c4_probe(E, synth, torch, torch.device("cuda", 0), 1 << 20)))
