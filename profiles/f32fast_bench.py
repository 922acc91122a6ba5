#!/usr/bin/env python3
"""profiles/f32fast_bench.py -- BASELINE config 5's shape on one GPU (4e6 sources x 20 interfaces; the whole job is 1e7 over
8 GPUs): the three single-GPU paths side by side -- fp64 (the reference's arithmetic), fp32 with the reference's iteration
(rtmi_trace_device_f32), fp32 fast (rtmi_trace_device_f32_fast) -- kernel time by events, and the deviation of both fp32
paths from the fp64 result on the same (float-rounded) inputs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
rt.load_library(os.environ.get('RTMI_AB_LIB') or None); rt.init(0)   # RTMI_AB_LIB: another build (profiles/ab_prev/...)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(2025)
d = np.cumsum(rng.uniform(200, 600, 20)); v = np.sort(rng.uniform(1500, 6500, 21))
off, dep = rng.uniform(10, 8000, n), rng.uniform(50, d[-1] + 500, n)
dev = "cuda"
f32 = [torch.tensor(np.asarray(a, np.float32), device=dev) for a in (v, d, off, dep)]
f64 = [t.double() for t in f32]
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out
ms64, t64 = timeit(lambda: rt.trace_device(*f64))
ms32, t32 = timeit(lambda: rt.trace_device(*f32))
msf, tf = timeit(lambda: rt.trace_device(*f32, fast=True))
t64 = t64.cpu().numpy()
print(f"config 5 shape, {n} sources x 20 interfaces, one MI355X")
for name, ms, t, b in (("fp64 (reference arithmetic)", ms64, None, 24), ("fp32, reference iteration", ms32, t32, 12), ("fp32 fast", msf, tf, 12)):
    line = f"{name:30s} {ms:8.4f} ms  {n / ms * 1e3:.3e} tt/s  {b * n / ms * 1e-6:7.1f} GB/s algorithmic"
    if t is not None:
        t = t.cpu().numpy().astype(np.float64)
        both = (t64 != -999.0) & (t != -999.0)
        e = np.abs(t[both] - t64[both])
        line += (f"   |dt| vs fp64: median {np.median(e):.2e}  99.9 % {np.percentile(e, 99.9):.2e}  max {e.max():.2e} s;"
                 f" sentinels {int((t == -999.0).sum())} (fp64: {int((t64 == -999.0).sum())})")
    print(line)
