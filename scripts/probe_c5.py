"""Config C5 probe alone (bench.replan_probe): python scripts/probe_c5.py [ticks]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process)
import bench
from pointcloudtraj_amd import engine as E, synth
E.init(0)
print(json.dumps(bench.replan_probe(E, synth, int(sys.argv[1]) if len(sys.argv) > 1 else 200), indent=1))
