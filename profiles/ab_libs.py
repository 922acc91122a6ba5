#!/usr/bin/env python3
"""profiles/ab_libs.py -- interleaved A/B of two BUILDS of librtmi.so in one process on the C2 workload:
profiles/ab_prev/librtmi.so (built from the previous commit) against the in-tree library."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd import raymod
import bench
cur = rt.load_library(); rt.init(0)
prev = ctypes.CDLL(os.path.join(ROOT, "profiles", "ab_prev", "librtmi.so"))  # plain dlopen: older builds may lack newer symbols
prev.rtmi_init.restype = ctypes.c_int; prev.rtmi_trace_device.restype = ctypes.c_int
prev.rtmi_trace_device.argtypes = cur.rtmi_trace_device.argtypes
prev.rtmi_init(0)
v, d, off, dep = bench.c2_workload(1_000_000)
dev = "cuda"
dv, dd, do, dz = (torch.tensor(x, device=dev) for x in (v, d, off, dep))
outs = {"prev": torch.empty(off.size, dtype=torch.float64, device=dev), "cur": torch.empty(off.size, dtype=torch.float64, device=dev)}
libs = {"prev": prev, "cur": cur}
vp = ctypes.c_void_p
def launch(name):
    lib = libs[name]
    rc = lib.rtmi_trace_device(vp(dv.data_ptr()), vp(dd.data_ptr()), 8, vp(do.data_ptr()), vp(dz.data_ptr()), off.size,
                               vp(outs[name].data_ptr()), None, vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
res = {"prev": [], "cur": []}
for r in range(9):
    for name in ("prev", "cur"):
        launch(name); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            launch(name)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10)
assert torch.equal(outs["prev"], outs["cur"]), "builds disagree"
for name in ("prev", "cur"):
    a = np.array(res[name]); print("%-5s median %.4f ms  min %.4f ms" % (name, np.median(a), a.min()))
