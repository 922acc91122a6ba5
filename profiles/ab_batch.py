#!/usr/bin/env python3
"""profiles/ab_batch.py -- scheduling knobs on the batch shape of BASELINE config 3 (32 models x 1e4 sources, fused logL)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
rt.load_library(); rt.init(0)
def make(nm, n, kmax):
    rng = np.random.default_rng(5); ld = 32
    x, y = rng.uniform(10, 6500, n), rng.uniform(10, 4500, n)
    so, sz = np.sqrt(x * x + y * y), rng.uniform(1050, 4200, n)
    V, D, NL = np.zeros((nm, ld)), np.zeros((nm, ld)), np.zeros(nm, np.int32)
    for m in range(nm):
        k = int(rng.integers(2, kmax + 1))
        D[m, :k - 1] = np.sort(rng.uniform(100.1, 10000.1, k - 1)); V[m, :k] = rng.uniform(1500, 10000, k); NL[m] = k - 1
    dev = "cuda"
    return [torch.tensor(a, device=dev) for a in (V, D, NL, so, sz)] + [torch.tensor(rng.uniform(0.5, 2.0, n), device=dev),
            torch.full((nm,), 0.02, dtype=torch.float64, device=dev)]
for nm, n, kmax in ((32, 10000, 10), (32, 100000, 30), (256, 10000, 30)):
    tV, tD, tN, to, tz, obs, sd = make(nm, n, kmax)
    settings = [dict(waves_per_simd=w, refill_min=m, tail_group=t) for w in (3, 4) for m in (16, 32) for t in (0, 128, 512)]
    res = {i: [] for i in range(len(settings))}
    for r in range(5):
        for i, st in enumerate(settings):
            for k, val in st.items():
                rt.set_option(k, val)
            rt.trace_batch_device(tV, tD, tN, to, tz, obs=obs, sd=sd, want_timeP=False); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                rt.trace_batch_device(tV, tD, tN, to, tz, obs=obs, sd=sd, want_timeP=False)
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) / 5)
    print(f"--- {nm} models x {n} sources, k <= {kmax}")
    for i, st in enumerate(settings):
        x = np.median(res[i]); print("%-62s %.4f ms  %.3e tt/s" % (st, x, nm * n / x * 1e3))
