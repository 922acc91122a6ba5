#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (not the bench metric): PCIe-inclusive rate of the drop-in `dff_`,
its per-call latency at the reference's NSrc=20, and the many-models batch (BASELINE configs 3/4 shapes)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
import bench

rt.load_library(); rt.init(0)
out = {}
v, d, off, dep = bench.c2_workload(1_000_000)
for _ in range(3):
    rt.dff(v, d, 8, off, dep, off.size)
t0 = time.perf_counter(); reps = 10
for _ in range(reps):
    rt.dff(v, d, 8, off, dep, off.size)
dt = (time.perf_counter() - t0) / reps
out["dff_host_1e6"] = {"ms_per_call": dt * 1e3, "tt_per_s": off.size / dt,
                       "note": "pageable host buffers -> HBM -> kernel -> host, incl. the ctypes mirror's 3 array copies"}
# raw C symbol without the Python copies
import ctypes
lib = rt.load_library()
dp = ctypes.POINTER(ctypes.c_double)
def raw(n):
    o, z, t = off[:n].copy(), dep[:n].copy(), np.zeros(n)
    nl, ns, kd = ctypes.c_int(8), ctypes.c_int(n), ctypes.c_int(-1)
    args = (v.ctypes.data_as(dp), d.ctypes.data_as(dp), ctypes.byref(nl), o.ctypes.data_as(dp), z.ctypes.data_as(dp),
            ctypes.byref(ns), t.ctypes.data_as(dp), ctypes.byref(kd))
    for _ in range(5):
        lib.dff_(*args)
    reps = 200 if n < 10000 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.dff_(*args)
    return (time.perf_counter() - t0) / reps
for n in (20, 1000, 100000, 1000000):
    dt = raw(n)
    out[f"dff_raw_symbol_n{n}"] = {"us_per_call": dt * 1e6, "tt_per_s": n / dt}

# batch shapes
def batch(nm, n, kmax, steps=20):
    rng = np.random.default_rng(5)
    ld = 32
    x, y = rng.uniform(10, 6500, n), rng.uniform(10, 4500, n)
    so, sz = np.sqrt(x * x + y * y), rng.uniform(1050, 4200, n)
    V, D, NL = np.zeros((nm, ld)), np.zeros((nm, ld)), np.zeros(nm, np.int32)
    for m in range(nm):
        k = int(rng.integers(2, kmax + 1))
        D[m, :k - 1] = np.sort(rng.uniform(100.1, 10000.1, k - 1)); V[m, :k] = rng.uniform(1500, 10000, k); NL[m] = k - 1
    dev = "cuda"
    tV, tD, tN, to, tz = (torch.tensor(a, device=dev) for a in (V, D, NL, so, sz))
    obs = torch.tensor(rng.uniform(0.5, 2.0, n), device=dev); sd = torch.full((nm,), 0.02, dtype=torch.float64, device=dev)
    T = torch.empty((nm, n), dtype=torch.float64, device=dev)
    res = {}
    for label, kw in (("timeP+logL", dict(timeP=T, obs=obs, sd=sd)), ("logL only", dict(obs=obs, sd=sd, want_timeP=False))):
        for _ in range(3):
            rt.trace_batch_device(tV, tD, tN, to, tz, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            rt.trace_batch_device(tV, tD, tN, to, tz, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        res[label] = {"ms_per_launch": dt * 1e3, "tt_per_s": nm * n / dt}
    return res
out["batch_c3_32x1e4_k2..10"] = batch(32, 10000, 10)
out["batch_c4_shard_32x1e5_k2..30"] = batch(32, 100000, 30, steps=5)
out["batch_256x1e4_k2..30"] = batch(256, 10000, 30, steps=5)
print(json.dumps(out, indent=1))
