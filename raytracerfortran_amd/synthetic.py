"""Seeded synthetic workloads of BASELINE.json's configurations -- inputs only.

Shared by bench.py, the parity tests and tests/golden/make_golden.py so that a number, a test and a
committed fixture always talk about the same arrays.  numpy `default_rng` streams are stable across
numpy versions; the fixtures additionally carry a SHA-256 of the inputs they were generated from.
"""
from __future__ import annotations

import hashlib

import numpy as np

TRUE_V = [3100.0, 2400.0, 3730.0, 4200.0, 5800.0, 4420.0, 6200.0]   # raytracerR-export-data-to-MCMC.Rmd:61
TRUE_D = [1000.0, 1500.0, 2000.0, 2500.0, 3260.0, 3960.0]           # :62


def c2_workload(n: int, seed: int = 12345, shard: int = 0):
    """Config C2: 8 interfaces, 9 velocities with one LVZ; shard > 0 draws other sources for the same model."""
    rng = np.random.default_rng(seed)
    d = np.cumsum(rng.uniform(300, 800, 8))
    v = np.sort(rng.uniform(1500, 6500, 9))
    v[[2, 3]] = v[[3, 2]]
    if shard:
        rng = np.random.default_rng([seed, shard])
    off = rng.uniform(10, 8000, n)
    dep = rng.uniform(50, d[-1] + 500, n)
    return v, d, off, dep


def sampler_sources(n: int, seed: int):
    """Source geometry of the sampler's example (raytracerR-export-data-to-MCMC.Rmd:44-46):
    offset = sqrt(x^2 + y^2), x ~ U(10, 6500), y ~ U(10, 4500), depth ~ U(1050, 4200)."""
    rng = np.random.default_rng(seed)
    x, y = rng.uniform(10, 6500, n), rng.uniform(10, 4500, n)
    return np.sqrt(x * x + y * y), rng.uniform(1050, 4200, n)


def prior_models(nmodels: int, kmin: int, kmax: int, seed: int, ld: int | None = None, force_k=()):
    """Models drawn from the sampler's prior (read_input.f90:205-214): k ~ U{kmin..kmax} nodes, interface depths
    sorted U(100.1, 10000.1), Vp ~ U(1500, 10000) (unsorted: low-velocity zones are the rule).  Rows the way
    LOGLHOOD_RT passes them to TraceRays (loglhood.f90:130-135): V[m, :k], D[m, :k-1], NL[m] = k - 1.
    `force_k`: the first models get these k (e.g. (30,) pins one model at the upper bound)."""
    rng = np.random.default_rng(seed)
    ld = ld or (kmax + 1)
    V, D = np.zeros((nmodels, ld)), np.zeros((nmodels, ld))
    NL = np.zeros(nmodels, np.int32)
    for m in range(nmodels):
        k = int(force_k[m]) if m < len(force_k) else int(rng.integers(kmin, kmax + 1))
        D[m, :k - 1] = np.sort(rng.uniform(100.1, 10000.1, k - 1))
        V[m, :k] = rng.uniform(1500, 10000, k)
        NL[m] = k - 1
    return V, D, NL


def c4_workload(nmodels: int, nsrc: int = 100_000, seed: int = 404, kmin: int = 2, kmax: int = 30):
    """Config C4 (per-GPU share 32 models, whole job 256): shared sources, models from the prior with
    k in [2, 30] (model 0 pinned at k = 30, model 1 at k = 2), sigma ~ U(0.001, 0.07), observations =
    times of the generating 7-layer model + N(0, 0.016 s) -- `obs_noise` is added by the caller to the
    forward times of (TRUE_V, TRUE_D), which the caller computes with whatever it is testing against."""
    off, dep = sampler_sources(nsrc, seed)
    V, D, NL = prior_models(nmodels, kmin, kmax, seed + 1, ld=kmax + 2, force_k=(kmax, kmin))
    rng = np.random.default_rng(seed + 2)
    sd = rng.uniform(0.001, 0.07, nmodels)
    obs_noise = rng.normal(0.0, 0.016, nsrc)
    return {"off": off, "dep": dep, "V": V, "D": D, "NL": NL, "sd": sd, "obs_noise": obs_noise}


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
