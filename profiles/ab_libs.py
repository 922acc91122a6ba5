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
    a = np.array(res[name]); print("C2 1e6        %-5s median %.4f ms  min %.4f ms" % (name, np.median(a), a.min()))

# the batch shape of a parallel-tempering slot: 32 models (k <= 10) x 1e4 shared sources, fused logL, no travel times
from raytracerfortran_amd.synthetic import prior_models, sampler_sources
for nm, n, kmax in ((32, 10_000, 10), (32, 100_000, 30)):
    V, D, NL = prior_models(nm, 2, kmax, seed=5, ld=kmax + 1)
    so, sz = sampler_sources(n, 6)
    rng = np.random.default_rng(7)
    tV, tD, tN, to, tz = (torch.tensor(a, device=dev) for a in (V, D, NL, so, sz))
    obs, sd = torch.tensor(rng.uniform(0.5, 2.0, n), device=dev), torch.full((nm,), 0.02, dtype=torch.float64, device=dev)
    for lib in libs.values():
        lib.rtmi_trace_batch_device.restype = ctypes.c_int
        lib.rtmi_trace_batch_device.argtypes = cur.rtmi_trace_batch_device.argtypes
        lib.rtmi_trace_batch_work_bytes.restype = ctypes.c_int64
        lib.rtmi_trace_batch_work_bytes.argtypes = cur.rtmi_trace_batch_work_bytes.argtypes
    work = torch.empty(max(int(cur.rtmi_trace_batch_work_bytes(nm, n)), int(prev.rtmi_trace_batch_work_bytes(nm, n))) // 8 + 8,
                       dtype=torch.float64, device=dev)
    ll = {k: torch.empty(nm, dtype=torch.float64, device=dev) for k in libs}
    def blaunch(name):
        rc = libs[name].rtmi_trace_batch_device(vp(tV.data_ptr()), vp(tD.data_ptr()), vp(tN.data_ptr()), kmax + 1, nm,
                                                vp(to.data_ptr()), vp(tz.data_ptr()), n, None, vp(obs.data_ptr()),
                                                vp(sd.data_ptr()), vp(ll[name].data_ptr()), vp(work.data_ptr()),
                                                vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    bres = {"prev": [], "cur": []}
    for r in range(9):
        for name in ("prev", "cur"):
            blaunch(name); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                blaunch(name)
            e1.record(); torch.cuda.synchronize()
            bres[name].append(e0.elapsed_time(e1) / 10)
    assert torch.equal(ll["prev"], ll["cur"]), "builds disagree on logL"
    for name in ("prev", "cur"):
        a = np.array(bres[name]); print("batch %3dx%-6d k<=%-2d %-5s median %.4f ms  min %.4f ms" % (nm, n, kmax, name, np.median(a), a.min()))
