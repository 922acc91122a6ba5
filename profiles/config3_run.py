#!/usr/bin/env python3
"""profiles/config3_run.py [nsweeps] -- BASELINE config 3 at full size on one MI355X: 32 tempered chains x 1e4 sweeps x
1e4 sources (synthetic data of the reference's 7-layer test model + N(0, 0.016 s) noise), native sweep engine.
Prints wall time, travel times traced, acceptance rates and a posterior summary of the T = 1 chain."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd import tempering as T
rt.load_library(); rt.init(0)
nsweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
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
# start far from the truth: a 2-layer model
s.init_from_model(np.array([[0.0, 3500.0], [2500.0, 5500.0]]), 0.05)
t0 = time.perf_counter()
samples = s.run(nsweeps, burnin=nsweeps // 2, thin=5)
el = time.perf_counter() - t0
st = s.native.stats()
ks = np.array([k for k, *_ in samples]); sds = np.array([x[2] for x in samples]); lls = np.array([x[3] for x in samples])
# posterior mean velocity at a few depths
zq = np.array([500.0, 1250.0, 1750.0, 2250.0, 2900.0, 3600.0, 4100.0])
vq = np.zeros((len(samples), zq.size))
for i, (k, nodes, sd, ll) in enumerate(samples):
    idx = np.searchsorted(nodes[:, 0], zq, side="right") - 1
    vq[i] = nodes[idx, 1]
out = {"sweeps": nsweeps, "chains": 32, "nsrc": nsrc, "wall_s": el, "ms_per_sweep": el / nsweeps * 1e3,
       "models_evaluated": st["models"], "launches": st["launches"], "travel_times_per_s": st["models"] * nsrc / el,
       "acc_bd": st["acc_bd"] / max(1, st["prop_bd"]), "acc_mh": st["acc_mh"] / max(1, st["prop_mh"]),
       "acc_sd": st["acc_sd"] / max(1, st["prop_sd"]), "acc_swap": s.stats["acc_swap"] / max(1, s.stats["prop_swap"]),
       "T1_samples": len(samples), "k_hist": {int(k): int((ks == k).sum()) for k in np.unique(ks)},
       "sd_mean": float(sds.mean()), "sd_true": 0.016, "logL_median": float(np.median(lls)),
       "logL_true_model": float(T.gaussian_loglike(float(np.sum((obs - tt) ** 2)), nsrc, 0.016)),
       "depths_m": zq.tolist(), "v_true": [3100.0, 2400.0, 3730.0, 4200.0, 5800.0, 4420.0, 6200.0],
       "v_post_mean": vq.mean(0).round(1).tolist(), "v_post_std": vq.std(0).round(1).tolist()}
print(json.dumps(out, indent=1))
