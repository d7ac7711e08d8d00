"""The C1 corridor scenario at ONE speculation depth (argv[1], default 256), a few repetitions: for profiling the expansion phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudtraj_amd import corridor, engine, scenarios
engine.init(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cloud1 = scenarios.sensed_cloud(12.0)
for rep in range(6):
    f = corridor.SafeRegionRrtStar(80000)
    f.setSpeculation(K)
    t = scenarios.timed_scenario(f, cloud1)
    print({k: round(v, 3) for k, v in t.items() if k.endswith("_ms")}, f.speculationStats(), f.expansionLaunches(), flush=True)
