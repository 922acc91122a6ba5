import json, os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd import tempering as T
rt.load_library(); rt.init(0)
nsweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
nsrc = 10000
rng = np.random.default_rng(2024)
x, y = rng.uniform(10, 6500, nsrc), rng.uniform(10, 4500, nsrc)
off, dep = np.sqrt(x * x + y * y), rng.uniform(1050, 4200, nsrc)
v = [3100.0, 2400.0, 3730.0, 4200.0, 5800.0, 4420.0, 6200.0]
d = [1000.0, 1500.0, 2000.0, 2500.0, 3260.0, 3960.0]
f64 = torch.float64
tt = rt.trace_device(torch.tensor(v, device="cuda", dtype=f64), torch.tensor(d, device="cuda", dtype=f64),
                     torch.tensor(off, device="cuda"), torch.tensor(dep, device="cuda")).cpu().numpy()
obs = tt + rng.normal(0, 0.016, nsrc)
ev = T.GpuEvaluator(off, dep, obs)
s = T.NativeSampler(T.Prior(), ev, ndata=nsrc, nchains_total=32, nptchains1=1, dTlog=1.15, seed=11)
s.init_from_model(np.array([[0.0, 3500.0], [2500.0, 5500.0]]), 0.05)
samples = s.run(nsweeps, burnin=nsweeps // 2, thin=5)
st = s.native.stats()
a = s.native.get_all()
h = hashlib.sha256()
for k in ("k", "sd", "logL", "ss", "beta"):
    h.update(np.ascontiguousarray(a[k]).tobytes())
live = np.arange(a["z"].shape[1])[None, :] < a["k"][:, None]
h.update(np.where(live, a["z"], 0.0).tobytes()); h.update(np.where(live, a["v"], 0.0).tobytes())
lev, acc, prop = s.native.get_ladder()
print(os.environ.get("TAG", ""), "state", h.hexdigest()[:16], "levels", hashlib.sha256(lev.tobytes()).hexdigest()[:8], "swaps", acc, prop,
      {k: st[k] for k in ("acc_bd", "prop_bd", "acc_mh", "prop_mh", "acc_sd", "prop_sd")}, "samples", len(samples), "launches", st["launches"])
