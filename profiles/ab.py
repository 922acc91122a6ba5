#!/usr/bin/env python3
"""profiles/ab.py -- interleaved A/B of scheduling-knob settings of rtmi_trace_device in ONE process (same device,
same clocks): for each setting the median / min kernel time over ROUNDS rounds of STEPS launches on the C2 workload."""
import itertools, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
import bench
rt.load_library(); rt.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
v, d, off, dep = bench.c2_workload(N)
dev = "cuda"
dv, dd, do, dz = (torch.tensor(x, device=dev) for x in (v, d, off, dep))
out = torch.empty(off.size, dtype=torch.float64, device=dev)
settings = [dict(path=2, waves_per_simd=4, refill_min=m, tail_group=t, retire_max=r, promote_age=p) for m in (8, 16) for t in (0, 64, 128, 256, 512) for r in (8, 16) for p in (8, 10, 12)]
ROUNDS, STEPS = 7, 10
res = {i: [] for i in range(len(settings))}
for r in range(ROUNDS):
    for i, st in enumerate(settings):
        for k, val in st.items():
            rt.set_option(k, val)
        rt.trace_device(dv, dd, do, dz, timeP=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(STEPS):
            rt.trace_device(dv, dd, do, dz, timeP=out)
        e1.record(); torch.cuda.synchronize()
        res[i].append(e0.elapsed_time(e1) / STEPS)
for i, st in enumerate(settings):
    a = np.array(res[i])
    print("%-70s median %.4f ms  min %.4f ms" % (st, np.median(a), a.min()))
