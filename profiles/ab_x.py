#!/usr/bin/env python3
"""profiles/ab_x.py N [N ...] -- refill_min x promote_age x retire_max at several source counts (interleaved A/B)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
import bench
rt.load_library(); rt.init(0)
dev = "cuda"
for N in [int(x) for x in sys.argv[1:]] or [1_000_000]:
    v, d, off, dep = bench.c2_workload(N)
    dv, dd, do, dz = (torch.tensor(x, device=dev) for x in (v, d, off, dep))
    out = torch.empty(off.size, dtype=torch.float64, device=dev)
    settings = [("m%d a%d r%d" % (m, a, r), dict(refill_min=m, promote_age=a, retire_max=r))
                for m in (24, 32, 40) for a in (2, 3, 4) for r in (8, 12, 16)]
    res = {n: [] for n, _ in settings}
    for r in range(7):
        for n, val in settings:
            for k, x in val.items():
                rt.set_option(k, x)
            rt.trace_device(dv, dd, do, dz, timeP=out); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                rt.trace_device(dv, dd, do, dz, timeP=out)
            e1.record(); torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / 10)
    print("N =", N)
    for n, _ in settings:
        a = np.array(res[n]); print("  %-12s median %.4f ms  min %.4f ms" % (n, np.median(a), a.min()))
    for k, x in dict(refill_min=32, retire_max=12, promote_age=3, tail_group=128, waves_per_simd=4).items():
        rt.set_option(k, x)
