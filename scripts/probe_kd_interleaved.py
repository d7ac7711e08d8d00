import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudtraj_amd import engine as E, synth, kdtree as K
E.init(0)
L = K.lib()
t = L.kd_create(3)
pts = synth.uniform_points(5, 3000, 0, 10)
qq = (C.c_float * 3)(5, 5, 5)
def tm(fn, n):
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    return 1e6 * (time.perf_counter() - t0) / n
for i in range(100):
    L.kd_insertf(t, pts[i].ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(i + 1))
def q_only(i):
    r = L.kd_nearestf(t, qq); L.kd_res_free(r)
def ins_q(i):
    L.kd_insertf(t, pts[100 + i].ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(101 + i))
    r = L.kd_nearestf(t, qq); L.kd_res_free(r)
def ins_q_r(i):
    L.kd_insertf(t, pts[1500 + i].ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(1501 + i))
    r = L.kd_nearestf(t, qq); L.kd_res_free(r)
    r = L.kd_nearest_rangef(t, qq, C.c_float(1.0)); L.kd_res_free(r)
print("query only      %.1f us" % tm(q_only, 500))
print("insert + query  %.1f us" % tm(ins_q, 1000))
print("insert+q+range  %.1f us" % tm(ins_q_r, 1000))
def clear_rebuild(i):
    L.kd_clear(t)
    for k in range(60):
        L.kd_insertf(t, pts[k].ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(k + 1))
    r = L.kd_nearestf(t, qq); L.kd_res_free(r)
print("clear+60 inserts+query %.1f us" % tm(clear_rebuild, 200))
c = E.Cloud(50000); c.set_input(synth.uniform_points(3, 45000, -10, 10)); c.build_grid()
prm = E.inflate_params((0, 0, 0), 30, 0.25, 1.5)
p1 = np.float64([[1.0, 2.0, 3.0]])
print("inflate Q=1 on 45k cloud  %.1f us" % tm(lambda i: c.inflate(prm, p1), 500))
t2 = L.kd_create(3)
for i in range(2500):
    L.kd_insertf(t2, pts[i].ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(i + 1))
def r_only(i):
    r = L.kd_nearest_rangef(t2, qq, C.c_float(1.0)); L.kd_res_free(r)
def q2(i):
    r = L.kd_nearestf(t2, qq); L.kd_res_free(r)
print("static 2500: query %.1f us  range(1.0) %.1f us" % (tm(q2, 300), tm(r_only, 300)))
def r_big(i):
    r = L.kd_nearest_rangef(t2, qq, C.c_float(3.0)); n = L.kd_res_size(r); L.kd_res_free(r); return n
print("static 2500: range(3.0) %.1f us, hits %d" % (tm(r_big, 300), r_big(0)))
