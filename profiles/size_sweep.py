#!/usr/bin/env python3
"""profiles/size_sweep.py -- the C2 recipe at other source counts: device-resident launch (HIP events, median of 9 groups
of 10 launches) and the drop-in dff_ on host arrays (PCIe inclusive, median of 9 calls on the same arrays)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd.synthetic import c2_workload
lib = rt.load_library(); rt.init(0)
dp = ctypes.POINTER(ctypes.c_double)
print("%10s %14s %14s %16s %14s" % ("sources", "kernel ms", "tt/s device", "dff_ ms (PCIe)", "tt/s dff_"))
for n in (20, 1000, 10_000, 100_000, 320_000, 1_000_000, 3_000_000, 10_000_000):
    v, d, off, dep = c2_workload(n)
    dv, dd, do, dz = (torch.tensor(x, device="cuda") for x in (v, d, off, dep))
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    ts = []
    for r in range(9):
        rt.trace_device(dv, dd, do, dz, timeP=out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            rt.trace_device(dv, dd, do, dz, timeP=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    km = float(np.median(ts))
    t = np.zeros(n)
    nl, ns, kd = ctypes.c_int(8), ctypes.c_int(n), ctypes.c_int(-1)
    args = (v.ctypes.data_as(dp), d.ctypes.data_as(dp), ctypes.byref(nl), off.ctypes.data_as(dp), dep.ctypes.data_as(dp),
            ctypes.byref(ns), t.ctypes.data_as(dp), ctypes.byref(kd))
    hs = []
    for r in range(12):
        t0 = time.perf_counter(); lib.dff_(*args); hs.append((time.perf_counter() - t0) * 1e3)
    hm = float(np.median(hs[3:]))
    assert np.array_equal(t, out.cpu().numpy(), equal_nan=True)
    print("%10d %14.4f %14.3e %16.4f %14.3e" % (n, km, n / km * 1e3, hm, n / hm * 1e3), flush=True)
