#!/usr/bin/env python3
"""tests/soak_gpu.py [iterations] [seed] -- extended randomized parity soak of the HIP path against the checker
(not collected by pytest: run by hand on a GPU box).  Random models of every shape (see
test_fused_kernel_random_model_soak), random source sets, random scheduling knobs, both kernels, fp64 and fp32."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
from oracle import oracle as O

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rt.load_library(); rt.init(0)
rng = np.random.default_rng(seed)
total = 0
for it in range(iters):
    k = int(rng.integers(2, 80))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        v = np.sort(rng.uniform(1500, 8000, k))
    elif kind == 1:
        v = rng.uniform(1500, 10000, k)
    elif kind == 2:
        v = rng.uniform(2000, 6000, k); v[rng.integers(0, k, 3)] = 6000.0
    elif kind == 3:
        v = np.concatenate([rng.uniform(300, 900, k - 1), [7000.0]])
    elif kind == 4:
        v = np.exp(rng.uniform(np.log(50.0), np.log(50000.0), k))
    else:
        v = np.sort(rng.uniform(1500, 8000, k))[::-1].copy()
    thick = rng.uniform(1.0, 900.0, k - 1) if rng.random() < 0.5 else np.exp(rng.uniform(np.log(0.5), np.log(3000.0), k - 1))
    d = np.cumsum(thick)
    n = int(rng.choice([rng.integers(1, 9000), rng.integers(9000, 400000)]))
    off = np.where(rng.random(n) < 0.5, rng.uniform(0.0, 3.0 * d[-1] + 100.0, n), np.exp(rng.uniform(-3, 11, n)))
    dep = np.where(rng.random(n) < 0.7, rng.uniform(0.01, d[-1] * 1.2 + 10.0, n), rng.choice(d, n) + rng.choice([0.0, 1e-9, -1e-9, 0.5], n))
    dep = np.abs(dep) + 1e-6
    knobs = dict(path=int(rng.choice([0, 1, 2])), waves_per_simd=int(rng.integers(1, 9)), refill_min=int(rng.integers(1, 65)),
                 tail_group=int(rng.choice([0, 64, 256, 1024])), retire_max=int(rng.integers(0, 33)),
                 promote_age=int(rng.choice([0, 1, 4, 10, 20])))
    f32 = rng.random() < 0.25
    try:
        for kk, vv in knobs.items():
            rt.set_option(kk, vv)
        dt = torch.float32 if f32 else torch.float64
        a = [torch.tensor(x, device="cuda", dtype=dt) for x in (v, d, off, dep)]
        p = torch.empty(n, dtype=dt, device="cuda")
        t = rt.trace_device(*a, p_final=p).cpu().numpy()
    finally:
        for kk, vv in dict(path=0, waves_per_simd=4, refill_min=32, tail_group=128, retire_max=12, promote_age=3).items():
            rt.set_option(kk, vv)
    if f32:
        ref, pref, nl = O.oracle_dff_f32(v, d, off, dep, want_p=True)
    else:
        ref, pref, nl = O.oracle_dff(v, d, off, dep, want_p=True)
    bad = int((~((t == ref) | (np.isnan(t) & np.isnan(ref)))).sum())
    good = nl > 1
    pg = p.cpu().numpy()[good]
    badp = int((~((pg == pref[good]) | (np.isnan(pg) & np.isnan(pref[good])))).sum())
    total += n
    if bad or badp:
        print("MISMATCH it=%d kind=%d k=%d n=%d f32=%s knobs=%s: times %d, p %d" % (it, kind, k, n, f32, knobs, bad, badp)); sys.exit(1)
    if it % 20 == 0:
        print("it %d ok (%d sources so far, status %d)" % (it, total, rt.raymod.load_library().rtmi_device_status()), flush=True)
print("soak ok: %d cases, %d sources, device status %d" % (iters, total, rt.raymod.load_library().rtmi_device_status()))
