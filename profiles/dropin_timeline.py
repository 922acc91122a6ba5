#!/usr/bin/env python3
"""profiles/dropin_timeline.py -- five dff_ calls of 1e6 sources on pageable arrays (run under
rocprofv3 --kernel-trace --memory-copy-trace; profiles/dropin_timeline.sh prints the last call's timeline)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracerfortran_amd as rt
from raytracerfortran_amd.synthetic import c2_workload
rt.init(0); lib = rt.load_library()
n = 1_000_000
v, d, off, dep = c2_workload(n)
t = np.zeros(n)
dp = ctypes.POINTER(ctypes.c_double)
cn, cl, kd = ctypes.c_int(n), ctypes.c_int(8), ctypes.c_int(-1)
a = (v.ctypes.data_as(dp), d.ctypes.data_as(dp), ctypes.byref(cl), off.ctypes.data_as(dp), dep.ctypes.data_as(dp), ctypes.byref(cn), t.ctypes.data_as(dp), ctypes.byref(kd))
for _ in range(6):
    t0 = time.perf_counter(); lib.dff_(*a); print("call %.3f ms" % ((time.perf_counter() - t0) * 1e3))
    time.sleep(0.01)
