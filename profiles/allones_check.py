#!/usr/bin/env python3
"""profiles/allones_check.py -- the guarded fp64 layer term on the one family of operands where a refined reciprocal
is not correctly rounded (Markstein's exception): w whose square root has an all-ones mantissa, random numerators."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, raytracerfortran_amd as rt
from raytracerfortran_amd import raymod
rt.load_library(); rt.init(0)
ws = []
for k in range(0, 60):
    g = np.nextafter(np.float64(2.0) ** (-k), 0.0)          # all-ones mantissa, just below 2^-k
    w0 = g * g
    for w in (np.nextafter(w0, 0), w0, np.nextafter(w0, 1), np.nextafter(np.nextafter(w0, 1), 1)):
        if np.sqrt(w) == g and 1e-60 < w <= 1.0:
            ws.append(w)
ws = np.unique(np.array(ws))
print("w values whose sqrt has an all-ones mantissa:", ws.size)
g = torch.Generator(device="cuda").manual_seed(3)
n = 1 << 22
tot = np.zeros(3, dtype=np.int64)
for w in ws:
    wt = torch.full((n,), float(w), dtype=torch.float64, device="cuda")
    num = torch.exp(40.0 * (torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5))
    den = torch.exp(40.0 * (torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5))
    tot += np.array(raymod.selftest_fastmath(wt, num, den))
print("mismatches (sqrt, divide, layer term):", tot, "over", ws.size * n, "operand triples")
# special numerators: exact powers of two, small integers, and multiples of the square root itself
tot2 = np.zeros(3, dtype=np.int64)
spec = np.concatenate([2.0 ** np.arange(-40, 41), np.arange(1, 200, dtype=np.float64), 1.0 / np.arange(1, 200)])
for w in ws:
    gq = np.sqrt(w)
    nums = np.concatenate([spec, gq * spec[:81], gq * gq * spec[:81], (gq ** 3) * spec[:81]])
    wt = torch.full((nums.size,), float(w), dtype=torch.float64, device="cuda")
    nt = torch.tensor(nums, device="cuda")
    tot2 += np.array(raymod.selftest_fastmath(wt, nt, nt.clone()))
print("special numerators: mismatches (sqrt, divide, layer term):", tot2)
