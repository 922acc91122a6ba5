import sys, time, ctypes, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracerfortran_amd as rt
from raytracerfortran_amd.synthetic import c2_workload
rt.init(0); lib = rt.load_library()
v, d, off, dep = c2_workload(4096)
dp = ctypes.POINTER(ctypes.c_double)
for n in (20, 256, 4096):
    o, z, t = off[:n].copy(), dep[:n].copy(), np.zeros(n)
    cn, cl, kd = ctypes.c_int(n), ctypes.c_int(8), ctypes.c_int(-1)
    a = (v.ctypes.data_as(dp), d.ctypes.data_as(dp), ctypes.byref(cl), o.ctypes.data_as(dp), z.ctypes.data_as(dp), ctypes.byref(cn), t.ctypes.data_as(dp), ctypes.byref(kd))
    for _ in range(50): lib.dff_(*a)
    ts = []
    for _ in range(500):
        t0 = time.perf_counter(); lib.dff_(*a); ts.append(time.perf_counter() - t0)
    print("NSrc=%5d  median %.1f us  min %.1f us" % (n, np.median(ts) * 1e6, np.min(ts) * 1e6))
