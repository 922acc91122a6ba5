"""Which divisors near an all-ones mantissa break the shared-reciprocal layer term (guard ignored)?  s = 1 - m 2^-53
(and other exponents), m = 1 .. 2^15, numerators built to sit on rounding midpoints (tests/hardcases.py)."""
import sys, numpy as np, math, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, raytracerfortran_amd as rt
import hardcases as H
from raytracerfortran_amd import raymod
rt.init(0)
dev = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), device="cuda")
W, N, Dn, M = [], [], [], []
full = (1 << 53) - 1
for m in list(range(0, 4096)) + list(range(4096, 1 << 17, 37)):
    D = full - m
    for e in (0, 1, 40):
        w, s = H.w_with_sqrt_mantissa(D, e)
        if w is None:
            continue
        s3 = (s * s) * s
        D3, _ = H.mant_exp(s3)
        hs = H.hard_numerators(D, limit=8) or [(1 << 52, 54, 0)]
        h3 = H.hard_numerators(D3, limit=8) or [(1 << 52, 54, 0)]
        for i in range(max(len(hs), len(h3))):
            W.append(w); N.append(float(hs[i % len(hs)][0])); Dn.append(float(h3[i % len(h3)][0])); M.append((m, full - D3))
W, N, Dn = np.array(W), np.array(N), np.array(Dn)
print("cases", W.size)
bad_m, bad_m3 = set(), set()
step = 50000
for a in range(0, W.size, step):
    sl = slice(a, a + step)
    # tf alone (den = 1 is a trivial quotient), then tp alone
    r1 = raymod.selftest_fastmath_ex(dev(W[sl]), dev(N[sl]), dev(np.zeros(W[sl].size)))
    off1 = list(raymod.selftest_fastmath_ex.last_offenders)
    r2 = raymod.selftest_fastmath_ex(dev(W[sl]), dev(np.zeros(W[sl].size)), dev(Dn[sl]))
    off2 = list(raymod.selftest_fastmath_ex.last_offenders)
    print(a, "tf raw mismatches", r1[3], "accepted-and-wrong", r1[2], "| tp raw", r2[3], "accepted-and-wrong", r2[2])
    for i, acc in off1:
        bad_m.add(M[a + i][0])
    for i, acc in off2:
        bad_m3.add(M[a + i])
print("m (s = all-ones - m) with a wrong a/s (first 58 per chunk):", sorted(bad_m))
print("(m, m3) with a wrong hv/s3:", sorted(bad_m3))
# details of the first offenders of chunk 0 (tf only)
sl = slice(0, step)
r1 = raymod.selftest_fastmath_ex(dev(W[sl]), dev(N[sl]), dev(np.zeros(W[sl].size)))
for i, acc in raymod.selftest_fastmath_ex.last_offenders:
    w, n = float(W[i]), float(N[i])
    tw, tn = dev([w]), dev([n])
    plain = float((tn / torch.sqrt(tw)).cpu()[0])
    print("offender", i, "accepted" if acc else "rejected", "M=", M[i][0] + 1, "w=", w.hex(), "a=", n.hex(), "gpu plain a/s=", plain.hex(),
          "cpu a/s=", (n / math.sqrt(w)).hex())
