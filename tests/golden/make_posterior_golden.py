#!/usr/bin/env python3
"""tests/golden/make_posterior_golden.py -- exact posteriors of two small problems, by quadrature on the ORACLE's likelihood
(oracle/dff_oracle.c: the reference's arithmetic), for the tests that assert what the device sampler samples
(tests/test_gpu_posterior.py).  Writes tests/golden/posterior_small.json.  Run in the build container; ~1 min of CPU.

Problem (both): 8 sources (depths 1000 .. 4000 m), observations = forward(two layers: 3300 m/s over 5000 m/s, interface at
2500 m) + N(0, 0.1 s) -- a contrast the data resolve only partly, so that neither k dominates and the odds move with beta (0.60 at
beta = 1, 0.51 at beta = 0.5) --, sigma FIXED at
0.1 s (sdmin = sdmax), prior box v in [2000, 7000] m/s, hmx = 6000 m, hmin = 1e-3 m, Poisson lambda = 2.
 * fixed dimension k = 2 (three parameters: v1, z2, v2): marginal CDFs of the target  L(theta)^beta x uniform prior  at
   beta = 1 and 0.5, on the midpoints of a 200^3 grid.
 * trans-dimensional k in {1, 2}: P(k = 2) under the chain's stationary law.  The sampler's birth draws the new node from
   q(z) = U(0, hmx - hmin), q(v) = U(vmin, vmax) and accepts on  pk(k')/pk(k) x (L'/L)^beta  alone (prjmh_temper_rf.f90:697-706,
   BIRTH_FULL :997-1095), its death removes one of the k - 1 deeper nodes at random (DEATH_FULL :917-995): detailed balance then
   holds for  pi(k, theta) = pk(k) x prod q(node) x L^beta  over LABELLED nodes inside CHECKBOUNDS' box, so
       P(k = 2) / P(k = 1) = pk(2)/pk(1) x Z2 / Z1,   Z1 = int L1^beta dv1 / dv,   Z2 = int L2^beta dv1 dz dv2 / (dv^2 (hmx - hmin))
   with z over [hmin, hmx - hmin) (what a birth can reach; the sliver (hmx - hmin, hmx] that only a perturbation reaches has
   relative measure 2e-7 here).  A Hastings ratio that forgot the prior ratio, the proposal density or the temperature moves this
   number by tens of per cent.
"""
import json, math, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O

rng = np.random.default_rng(20261005)
N = 8
x, y = rng.uniform(500, 4500, N), rng.uniform(500, 3000, N)
off, dep = np.sqrt(x * x + y * y), rng.uniform(1000.0, 4000.0, N)
SD = 0.1
TRUE_K2 = (3300.0, 2500.0, 5000.0)   # v1, z2, v2 of the generating model
obs = O.oracle_dff(np.array([TRUE_K2[0], TRUE_K2[2]]), np.array([TRUE_K2[1]]), off, dep) + rng.normal(0.0, SD, N)
prior = {"kmin": 1, "kmax": 2, "lam": 2.0, "hmx": 6000.0, "hmin": 1e-3, "vmin": 2000.0, "vmax": 7000.0, "sdmin": SD, "sdmax": SD,
         "pertsdsc": 30.0, "pertsdsdsc": 10.0}
G = 200
vmid = prior["vmin"] + (np.arange(G) + 0.5) * (prior["vmax"] - prior["vmin"]) / G
zlo, zhi = prior["hmin"], prior["hmx"] - prior["hmin"]
zmid = zlo + (np.arange(G) + 0.5) * (zhi - zlo) / G

def ss_models(V, D):
    """SUM(DresRT**2) of every model row (NLayers = 1), threads over slices (the C code holds no shared state)."""
    nm = V.shape[0]
    out = np.empty(nm)
    NL = np.ones(nm, np.int32)
    bounds = np.linspace(0, nm, 17).astype(int)
    def work(i):
        a, b = bounds[i], bounds[i + 1]
        out[a:b] = O.oracle_ss_batch(V[a:b], D[a:b], NL[a:b], off, dep, obs)
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(work, range(16)))
    return out

# k = 1: LOGLHOOD_RT passes (v, v) over one interface at 9999.9 m (loglhood.f90:137-144)
V1 = np.column_stack([vmid, vmid]); D1 = np.full((G, 2), 9999.9)
ss1 = ss_models(V1, D1)
# k = 2: (v1, v2) over the interface z
ss2 = np.empty((G, G, G))  # [z][v1][v2]
v1g, v2g = np.meshgrid(vmid, vmid, indexing="ij")
for iz, z in enumerate(zmid):
    V2 = np.column_stack([v1g.ravel(), v2g.ravel()]); D2 = np.full((G * G, 2), z)
    ss2[iz] = ss_models(V2, D2).reshape(G, G)
    if iz % 20 == 0:
        print("z slice", iz, file=sys.stderr)

out = {"generator": "tests/golden/make_posterior_golden.py", "sources": {"off": off.tolist(), "dep": dep.tolist(), "obs": obs.tolist()},
       "sd": SD, "true_k2": list(TRUE_K2), "prior": prior, "grid": G, "v_mid": vmid.tolist(), "z_mid": zmid.tolist(), "betas": [1.0, 0.5], "fixed_k2": {}, "transd": {}}
for beta in (1.0, 0.5):
    w1 = np.exp(-beta * (ss1 - ss1.min()) / (2 * SD * SD))
    w2 = np.exp(-beta * (ss2 - ss1.min()) / (2 * SD * SD))
    # marginals of the k = 2 target (uniform prior inside the box; hmin = 1e-3 m: CHECKBOUNDS' spacing rule cuts nothing else)
    tot = w2.sum()
    cdf = lambda p: (np.cumsum(p) / tot).tolist()
    out["fixed_k2"][str(beta)] = {"cdf_z": cdf(w2.sum((1, 2))), "cdf_v1": cdf(w2.sum((0, 2))), "cdf_v2": cdf(w2.sum((0, 1))),
                                  "mean": [float((w2.sum((1, 2)) * zmid).sum() / tot), float((w2.sum((0, 2)) * vmid).sum() / tot),
                                           float((w2.sum((0, 1)) * vmid).sum() / tot)]}
    Z1 = w1.mean()                       # int L^beta dv1 / dv (midpoint rule)
    Z2 = w2.mean()                       # int L^beta dz dv1 dv2 / ((hmx - hmin - hmin) dv^2) ...
    Z2 *= (zhi - zlo) / (prior["hmx"] - prior["hmin"])   # ... q(z) = 1 / (hmx - hmin) on [0, hmx - hmin), of which [hmin, .) is inside the bounds
    pk = lambda k: math.exp(-prior["lam"] + k * math.log(prior["lam"]) - math.lgamma(k + 1.0))
    odds = pk(2) / pk(1) * Z2 / Z1
    out["transd"][str(beta)] = {"Z2_over_Z1": float(Z2 / Z1), "odds_k2_k1": float(odds), "p_k2": float(odds / (1.0 + odds))}
    print("beta", beta, "Z2/Z1", Z2 / Z1, "P(k=2)", odds / (1 + odds), "means", out["fixed_k2"][str(beta)]["mean"], file=sys.stderr)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "posterior_small.json"), "w"))
