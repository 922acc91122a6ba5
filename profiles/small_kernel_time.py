#!/usr/bin/env python3
"""profiles/small_kernel_time.py -- the forward kernel alone at the reference's own call sizes (device-resident arrays, events
around back-to-back launches): what of the drop-in's 33-39 us at NSrc = 20 is the kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd.synthetic import c2_workload
rt.load_library(); rt.init(0)
v, d, off, dep = c2_workload(131072)
sizes = [int(x) for x in sys.argv[1:]] or [1, 20, 64, 256, 4096, 10000]
for n in sizes:
    a = [torch.tensor(x, device="cuda") for x in (v, d, off[:n], dep[:n])]
    for _ in range(20):
        rt.trace_device(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 300
    e0.record()
    for _ in range(reps):
        rt.trace_device(*a)
    e1.record(); torch.cuda.synchronize()
    print(f"NSrc = {n:6d}: {e0.elapsed_time(e1) / reps * 1e3:7.1f} us per launch (back to back, events)   kernel {rt.last_kernel_name()[:60]}")
