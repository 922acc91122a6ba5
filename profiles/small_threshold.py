#!/usr/bin/env python3
"""profiles/small_threshold.py -- where the mapped-buffer path of dff_ (no copy commands: the kernel reads the caller's
data from, and writes the times to, one pinned host buffer) stops paying against the copy pipeline: per-call time of dff_
at several NSrc for several values of the threshold (one child process per value)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json, time, ctypes, numpy as np
sys.path.insert(0, %r)
import raytracerfortran_amd as rt
from raytracerfortran_amd.synthetic import c2_workload
rt.init(0); lib = rt.load_library()
dp = ctypes.POINTER(ctypes.c_double)
out = {}
for n in (4096, 10000, 20000, 50000, 100000, 200000):
    v, d, off, dep = c2_workload(n)
    t = np.zeros(n)
    cn, cl, kd = ctypes.c_int(n), ctypes.c_int(8), ctypes.c_int(-1)
    a = (v.ctypes.data_as(dp), d.ctypes.data_as(dp), ctypes.byref(cl), off.ctypes.data_as(dp), dep.ctypes.data_as(dp),
         ctypes.byref(cn), t.ctypes.data_as(dp), ctypes.byref(kd))
    for _ in range(5):
        lib.dff_(*a)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); lib.dff_(*a); ts.append(time.perf_counter() - t0)
    out[n] = round(float(np.median(ts)) * 1e6, 1)
print(json.dumps(out))
""" % ROOT
for thr in (4096, 16384, 65536, 131072, 262144):
    p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=dict(os.environ, RTMI_SMALL_CALL_NSRC=str(thr)))
    try:
        print("threshold %7d  us per call by NSrc: %s" % (thr, p.stdout.strip().splitlines()[-1]), flush=True)
    except Exception:
        print("threshold", thr, "failed:", p.stderr[-300:])
