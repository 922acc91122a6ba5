#!/usr/bin/env python3
"""profiles/pt_state_dump.py <librtmi.so> <out.npz> -- run the device-resident sampler from a fixed seed with the given
build and dump every chain's state, the statistics and the sample ring: two builds whose sweep kernels follow the same
rules and random streams must produce identical files (compare with numpy)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from raytracerfortran_amd import raymod
raymod.LIB_PATH = os.path.abspath(sys.argv[1])
import ctypes
_probe = ctypes.CDLL(raymod.LIB_PATH)   # an older build may lack newer entry points: they are not what is being compared
for _name in [n for n in raymod.SIGNATURES if not hasattr(_probe, n)]:
    del raymod.SIGNATURES[_name]
import raytracerfortran_amd as rt
from raytracerfortran_amd import tempering as T
rt.load_library(); rt.init(0)
out = {}
for tag, kmax, nch, nsrc, sweeps in (("k10", 10, 32, 10000, 200), ("k30", 30, 8, 3000, 60)):
    rng = np.random.default_rng(2024)
    x, y = rng.uniform(10, 6500, nsrc), rng.uniform(10, 4500, nsrc)
    off, dep = np.sqrt(x * x + y * y), rng.uniform(1050, 4200, nsrc)
    v = [3100.0, 2400.0, 3730.0, 4200.0, 5800.0, 4420.0, 6200.0]
    d = [1000.0, 1500.0, 2000.0, 2500.0, 3260.0, 3960.0]
    f64 = torch.float64
    tt = rt.trace_device(torch.tensor(v, device="cuda", dtype=f64), torch.tensor(d, device="cuda", dtype=f64),
                         torch.tensor(off, device="cuda"), torch.tensor(dep, device="cuda")).cpu().numpy()
    ev = T.GpuEvaluator(off, dep, tt + rng.normal(0, 0.016, nsrc))
    pr = T.Prior(kmax=kmax) if kmax != 10 else T.Prior()
    s = T.NativeSampler(pr, ev, ndata=nsrc, nchains_total=nch, nptchains1=1, dTlog=1.15, seed=11, rank=0, world=1,
                        dist=None, dist_device="cpu")
    s.init_from_model(np.column_stack([[0.0] + d, v]), 0.016)
    s.native.sample_buffer(4096)
    s.native.run(sweeps, 1, s.comm, 4)
    def masked(dct):  # node slots at and beyond k hold leftovers: not part of the state
        m = np.arange(dct["z"].shape[1])[None, :] < dct["k"][:, None]
        return {**dct, "z": np.where(m, dct["z"], 0.0), "v": np.where(m, dct["v"], 0.0)}
    for name, a in masked(s.native.get_all()).items():
        out["%s_state_%s" % (tag, name)] = np.asarray(a)
    stats = s.native.stats()
    out[tag + "_stats"] = np.array([float(stats[k]) for k in sorted(stats)])
    for name, a in masked(s.native.samples()).items():
        out["%s_samples_%s" % (tag, name)] = np.asarray(a)
    lev, sw = s.native.get_ladder()[:2]
    out[tag + "_level"] = np.asarray(lev)
np.savez(sys.argv[2], **out)
print("dumped", len(out), "arrays")
