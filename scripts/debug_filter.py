import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth
E.init(0)
N = 10_000_000
pts = synth.uniform_points(3, N, 0, 100)
c = E.Cloud(N); c.set_input(pts)
q = synth.uniform_points(5, 1024, 0, 100)
for _ in range(3):
    idx, d2 = c.nn(q, E.ALGO_STREAM)
