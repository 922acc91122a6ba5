#!/usr/bin/env python3
"""profiles/fp32_vs_fp64.py -- BASELINE config 5 shape (20 interfaces) in both precisions: kernel time per launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
rt.load_library(); rt.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(5)
d = np.cumsum(rng.uniform(200, 600, 20)); v = np.sort(rng.uniform(1500, 6500, 21))
off, dep = rng.uniform(10, 8000, n), rng.uniform(50, d[-1] + 500, n)
for dt in (torch.float64, torch.float32):
    a = [torch.tensor(x, device="cuda", dtype=dt) for x in (v, d, off, dep)]
    out = torch.empty(n, dtype=dt, device="cuda")
    for _ in range(3):
        rt.trace_device(*a, timeP=out)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            rt.trace_device(*a, timeP=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    t = float(np.median(ts))
    print("%s  n=%d x 20 interfaces: %.3f ms per launch = %.3e tt/s" % (dt, n, t, n / t * 1e3))
