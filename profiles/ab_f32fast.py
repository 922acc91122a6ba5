#!/usr/bin/env python3
"""profiles/ab_f32fast.py -- interleaved A/B of two BUILDS of librtmi.so on config 5's per-GPU share (1.25e6 sources x 20
interfaces, rtmi_trace_device_f32_fast) in one process: profiles/ab_prev/raytracerfortran_amd/librtmi.so against the in-tree one."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracerfortran_amd as rt
cur = rt.load_library(); rt.init(0)
prev = ctypes.CDLL(os.path.join(ROOT, "profiles", "ab_prev", "raytracerfortran_amd", "librtmi.so"))
prev.rtmi_init.restype = ctypes.c_int
prev.rtmi_trace_device_f32_fast.restype = ctypes.c_int
prev.rtmi_trace_device_f32_fast.argtypes = cur.rtmi_trace_device_f32_fast.argtypes
prev.rtmi_init(0)
vp = ctypes.c_void_p
for n in (1_250_000, 4_000_000):
    rng = np.random.default_rng(2025)
    d = np.cumsum(rng.uniform(200, 600, 20)); v = np.sort(rng.uniform(1500, 6500, 21))
    off, dep = rng.uniform(10, 8000, n), rng.uniform(50, d[-1] + 500, n)
    tv, td, to, tz = (torch.tensor(np.asarray(a, np.float32), device="cuda") for a in (v, d, off, dep))
    outs = {k: torch.empty(n, dtype=torch.float32, device="cuda") for k in ("prev", "cur")}
    libs = {"prev": prev, "cur": cur}
    def launch(name):
        rc = libs[name].rtmi_trace_device_f32_fast(vp(tv.data_ptr()), vp(td.data_ptr()), 20, vp(to.data_ptr()), vp(tz.data_ptr()), n,
                                                   vp(outs[name].data_ptr()), None, vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    res = {"prev": [], "cur": []}
    for r in range(11):
        for name in ("prev", "cur"):
            launch(name); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                launch(name)
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 20)
    assert torch.equal(outs["prev"], outs["cur"]), "builds disagree"
    for name in ("prev", "cur"):
        a = np.array(res[name]); print("c5 fast %8d  %-5s median %.4f ms (%.3e tt/s)  min %.4f ms" % (n, name, np.median(a), n / np.median(a) * 1e3, a.min()))
