"""Drop-in dff_ on FRESH host arrays every call (what R's .Fortran hands over: newly allocated copies), against the same
arrays reused (a Fortran caller's static arrays): does pinning the caller's buffers per call still pay?"""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracerfortran_amd as rt
from raytracerfortran_amd.synthetic import c2_workload
rt.init(0); lib = rt.load_library()
n = 1_000_000
v, d, off0, dep0 = c2_workload(n)
dp = ctypes.POINTER(ctypes.c_double)
cn, cl, kd = ctypes.c_int(n), ctypes.c_int(8), ctypes.c_int(-1)
def call(off, dep, t):
    t0 = time.perf_counter()
    lib.dff_(v.ctypes.data_as(dp), d.ctypes.data_as(dp), ctypes.byref(cl), off.ctypes.data_as(dp), dep.ctypes.data_as(dp),
             ctypes.byref(cn), t.ctypes.data_as(dp), ctypes.byref(kd))
    return time.perf_counter() - t0
t = np.zeros(n)
for _ in range(4):
    call(off0, dep0, t)
same = [call(off0, dep0, t) for _ in range(15)]
fresh = []
keep = []
for i in range(15):
    off, dep, tt = off0.copy(), dep0.copy(), np.zeros(n)   # new allocations, touched (like .Fortran's copies)
    keep.append((off, dep, tt))                               # and not recycled by the allocator during the test
    fresh.append(call(off, dep, tt))
print("register=%s  same arrays: median %.3f ms   fresh arrays: median %.3f ms (first %.3f)" % (
    os.environ.get("RTMI_DROPIN_REGISTER", "1"), np.median(same) * 1e3, np.median(fresh) * 1e3, fresh[0] * 1e3))
