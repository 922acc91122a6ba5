#!/usr/bin/env python3
"""profiles/ab_large.py -- scheduling knobs at config 5's size (1e7 sources x 20 interfaces, several LDS rounds per CU)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
rt.load_library(); rt.init(0)
rng = np.random.default_rng(2025)
d = np.cumsum(rng.uniform(200, 600, 20)); v = np.sort(rng.uniform(1500, 6500, 21))
n = 10_000_000
off = rng.uniform(10, 8000, n); dep = rng.uniform(50, d[-1] + 500, n)
dev = "cuda"
a = [torch.tensor(x, device=dev) for x in (v, d, off, dep)]
out = torch.empty(n, dtype=torch.float64, device=dev)
settings = [dict(waves_per_simd=4, refill_min=16, tail_group=t, retire_max=r, promote_age=p) for p in (0, 8) for t in (0, 512) for r in (0, 8, 16)]
res = {i: [] for i in range(len(settings))}
for r in range(4):
    for i, st in enumerate(settings):
        for k, val in st.items():
            rt.set_option(k, val)
        rt.trace_device(*a, timeP=out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            rt.trace_device(*a, timeP=out)
        e1.record(); torch.cuda.synchronize()
        res[i].append(e0.elapsed_time(e1) / 3)
for i, st in enumerate(settings):
    x = np.array(res[i]); print("%-60s median %.3f ms  -> %.3e tt/s" % (st, np.median(x), n / np.median(x) * 1e3))
