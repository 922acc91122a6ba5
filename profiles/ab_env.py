#!/usr/bin/env python3
"""profiles/ab_env.py VAR v0 v1 ... -- interleaved sweep of an experiment's environment knob (read by the launcher at
every launch of an experimental build; VAR = opt:<name> sweeps the scheduling knob rtmi_set_option(<name>) instead) on three shapes: C2 1e6, the tempering slot (32 x 1e4, k <= 10) and config 4's
share (32 x 1e5, k <= 30).  Results are checked equal across the values."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd.synthetic import c2_workload, prior_models, sampler_sources
cur = rt.load_library(); rt.init(0)
dev = "cuda"; vp = ctypes.c_void_p
var, vals = sys.argv[1], sys.argv[2:]
v, d, off, dep = c2_workload(1_000_000)
dv, dd, do, dz = (torch.tensor(x, device=dev) for x in (v, d, off, dep))
out = torch.empty(off.size, dtype=torch.float64, device=dev)
def c2():
    assert cur.rtmi_trace_device(vp(dv.data_ptr()), vp(dd.data_ptr()), 8, vp(do.data_ptr()), vp(dz.data_ptr()), off.size,
                                 vp(out.data_ptr()), None, vp(torch.cuda.current_stream().cuda_stream)) == 0
    return out
shapes = {"C2": c2}
for nm, n, kmax in ((32, 10_000, 10), (32, 100_000, 30)):
    V, D, NL = prior_models(nm, 2, kmax, seed=5, ld=kmax + 1)
    so, sz = sampler_sources(n, 6)
    rng = np.random.default_rng(7)
    ten = [torch.tensor(a, device=dev) for a in (V, D, NL, so, sz)]
    obs, sd = torch.tensor(rng.uniform(0.5, 2.0, n), device=dev), torch.full((nm,), 0.02, dtype=torch.float64, device=dev)
    work = torch.empty(int(cur.rtmi_trace_batch_work_bytes(nm, n)) // 8 + 8, dtype=torch.float64, device=dev)
    ll = torch.empty(nm, dtype=torch.float64, device=dev)
    def b(ten=ten, obs=obs, sd=sd, work=work, ll=ll, nm=nm, n=n, kmax=kmax):
        tV, tD, tN, to, tz = ten
        assert cur.rtmi_trace_batch_device(vp(tV.data_ptr()), vp(tD.data_ptr()), vp(tN.data_ptr()), kmax + 1, nm,
                                           vp(to.data_ptr()), vp(tz.data_ptr()), n, None, vp(obs.data_ptr()), vp(sd.data_ptr()),
                                           vp(ll.data_ptr()), vp(work.data_ptr()), vp(torch.cuda.current_stream().cuda_stream)) == 0
        return ll
    shapes["b%dx%d" % (nm, n)] = b
res = {(s, m): [] for s in shapes for m in vals}
ref = {}
for r in range(7):
    for m in vals:
        if var.startswith("opt:"):
            rt.set_option(var[4:], int(m))
        else:
            os.environ[var] = m
        for s, fn in shapes.items():
            o = fn(); torch.cuda.synchronize()
            if s not in ref:
                ref[s] = o.clone()
            assert torch.equal(ref[s], o), (s, m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[(s, m)].append(e0.elapsed_time(e1) / 10)
for s in shapes:
    print("%-12s" % s, "  ".join("%s=%s %.4f" % (var, m, np.median(res[(s, m)])) for m in vals))
