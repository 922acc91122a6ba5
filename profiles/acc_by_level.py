#!/usr/bin/env python3
"""profiles/acc_by_level.py [burn] [nmeas] -- config 3's problem: acceptance of the perturbation / birth-death proposals BY TEMPERATURE
LEVEL (the chains are run without exchanges for `nmeas` sweeps after `burn` sweeps with them), the chains' k, and what a window of
d rows retires per pass at that acceptance: the question is whether one speculation depth fits the whole ladder."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd import tempering as T
rt.load_library(); rt.init(0)
burn = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
nmeas = int(sys.argv[2]) if len(sys.argv) > 2 else 400
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
s.run(burn)
s.sync()
lev, _, _ = s.native.get_ladder()
a0 = s.native.chain_stats()
sp0 = s.native.speculation()
s.native.run(nmeas, 0, None, 0)   # no exchanges: a chain keeps its level
s.sync()
a1 = s.native.chain_stats() - a0
al = s.native.get_all()
rows = []
for c in np.argsort(lev):
    acc_mh = a1[c, 2] / max(1, a1[c, 3]); acc_bd = a1[c, 0] / max(1, a1[c, 1])
    r = 1.0 - acc_mh
    exp = {dd: (1 - r ** dd) / (1 - r) if r < 1 else dd for dd in (1, 2, 3, 4, 6, 8)}
    rows.append({"level": int(lev[c]), "chain": int(c), "beta": float(al["beta"][c]), "k": int(al["k"][c]), "acc_mh": round(float(acc_mh), 3),
                 "acc_bd": round(float(acc_bd), 3), "decided": int(a1[c, 6]), "discarded": int(a1[c, 7]),
                 "slots_per_pass": {k_: round(v_, 2) for k_, v_ in exp.items()}})
for r in rows:
    print(json.dumps(r))
