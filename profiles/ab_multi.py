#!/usr/bin/env python3
"""profiles/ab_multi.py name=lib.so [name=lib.so ...] -- interleaved A/B of several BUILDS of librtmi.so in one process on
three shapes: C2 (1e6 sources x 8 interfaces), the tempering slot (32 models x 1e4, k <= 10) and config 4's share
(32 x 1e5, k <= 30).  Outputs are checked equal across the builds.  `cur` = the in-tree library."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracerfortran_amd as rt
import bench
from raytracerfortran_amd.synthetic import prior_models, sampler_sources
cur = rt.load_library(); rt.init(0)
libs = {}
for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    if name == "cur":
        libs[name] = cur
        continue
    lib = ctypes.CDLL(os.path.abspath(path))
    for fn in ("rtmi_init", "rtmi_trace_device", "rtmi_trace_batch_device"):
        getattr(lib, fn).restype = ctypes.c_int
    lib.rtmi_trace_device.argtypes = cur.rtmi_trace_device.argtypes
    lib.rtmi_trace_batch_device.argtypes = cur.rtmi_trace_batch_device.argtypes
    lib.rtmi_trace_batch_work_bytes.restype = ctypes.c_int64
    lib.rtmi_trace_batch_work_bytes.argtypes = cur.rtmi_trace_batch_work_bytes.argtypes
    lib.rtmi_init(0)
    libs[name] = lib
dev = "cuda"; vp = ctypes.c_void_p
v, d, off, dep = bench.c2_workload(1_000_000)
dv, dd, do, dz = (torch.tensor(x, device=dev) for x in (v, d, off, dep))
out = {k: torch.empty(off.size, dtype=torch.float64, device=dev) for k in libs}
def c2(name):
    assert libs[name].rtmi_trace_device(vp(dv.data_ptr()), vp(dd.data_ptr()), 8, vp(do.data_ptr()), vp(dz.data_ptr()), off.size,
                                        vp(out[name].data_ptr()), None, vp(torch.cuda.current_stream().cuda_stream)) == 0
    return out[name]
shapes = {"C2 1e6": c2}
for nm, n, kmax in ((32, 10_000, 10), (32, 100_000, 30)):
    V, D, NL = prior_models(nm, 2, kmax, seed=5, ld=kmax + 1)
    so, sz = sampler_sources(n, 6)
    rng = np.random.default_rng(7)
    ten = [torch.tensor(a, device=dev) for a in (V, D, NL, so, sz)]
    obs, sd = torch.tensor(rng.uniform(0.5, 2.0, n), device=dev), torch.full((nm,), 0.02, dtype=torch.float64, device=dev)
    work = torch.empty(max(int(l.rtmi_trace_batch_work_bytes(nm, n)) for l in libs.values()) // 8 + 8, dtype=torch.float64, device=dev)
    ll = {k: torch.empty(nm, dtype=torch.float64, device=dev) for k in libs}
    def b(name, ten=ten, obs=obs, sd=sd, work=work, ll=ll, nm=nm, n=n, kmax=kmax):
        tV, tD, tN, to, tz = ten
        assert libs[name].rtmi_trace_batch_device(vp(tV.data_ptr()), vp(tD.data_ptr()), vp(tN.data_ptr()), kmax + 1, nm,
                                                  vp(to.data_ptr()), vp(tz.data_ptr()), n, None, vp(obs.data_ptr()), vp(sd.data_ptr()),
                                                  vp(ll[name].data_ptr()), vp(work.data_ptr()), vp(torch.cuda.current_stream().cuda_stream)) == 0
        return ll[name]
    shapes["batch %dx%d k<=%d" % (nm, n, kmax)] = b
res = {(s, m): [] for s in shapes for m in libs}
for r in range(9):
    for m in libs:
        for s, fn in shapes.items():
            fn(m); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn(m)
            e1.record(); torch.cuda.synchronize()
            res[(s, m)].append(e0.elapsed_time(e1) / 10)
first = next(iter(libs))
for s, fn in shapes.items():
    ref = fn(first).clone()
    for m in libs:
        assert torch.equal(ref, fn(m)), ("builds disagree", s, m)
    print("%-24s" % s, "  ".join("%s %.4f" % (m, np.median(res[(s, m)])) for m in libs))
