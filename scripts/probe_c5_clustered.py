"""Config C5 (rolling 5 M-point window, one captured replan graph per tick) in its clustered variant and as stated: ms per tick, the state of the index"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
r = bench.replan_probe(E, synth, ticks=100, clustered=True)
print({k: r[k] for k in ("ms_per_tick_p50","ms_per_tick_p99","ingest_ms_p50","replan_graph_ms_p50","ticks_with_a_colliding_sample","unindexed_brute_force_ms_per_tick_p50")}); print(r["ring_index"], r["replan_inside_library_us_p50"], r["worst_tick"])
r = bench.replan_probe(E, synth, ticks=100, clustered=False)
print({k: r[k] for k in ("ms_per_tick_p50","ms_per_tick_p99","ingest_ms_p50","replan_graph_ms_p50")}); print(r["ring_index"])
