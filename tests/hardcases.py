"""Constructed hard cases for the guarded sqrt / divide / shared-reciprocal layer term of the kernels
(raytracerfortran_amd/csrc/rtmi_device.hpp: sqrt_guarded, div_guarded, term_guarded_n) -- operands that sit ON the
failure boundaries of a reciprocal-based divide instead of being drawn at random.

Division.  For a divisor with p-bit mantissa D (p = 53 or 24) the quotients hardest to round are those closest to a
rounding midpoint X/2 (X odd): n * 2^(p+1) = X * D + r with a tiny odd residue r puts n/D at distance |r| / (2 D) from the
midpoint -- about 2^-(2p+1) relative, the closest a quotient of two p-bit numbers can come.  Given D and r, X follows
from a modular inverse (X = -r D^-1 mod 2^(p+1)) and n from the exact division; both roundings of such a quotient (r > 0:
just above the midpoint, r < 0: just below) are generated.  A reciprocal that is one ulp off, or a residual correction
that loses a bit, returns the wrong neighbour exactly here.

Divisor patterns.  The divisors the kernels see are s = sqrt(w) and s3 = (s*s)*s.  s is steered through w = RN(D^2):
all-ones mantissas (Markstein's exception), all-ones minus a few ulps, 2^52 + small, single ones, long runs of ones
and zeros, alternating bits, all-ones low words (the kernel's guard pattern) and random mantissas, over several
exponents of the guarded range; s3 is whatever the rounded cube of such an s is, and gets its own hard numerators.

Square root.  w on both sides of squares of midpoints, (M + 1/2)^2 rounded down and up and their neighbours, and around
perfect squares M^2.

Pure Python integers (exact), no GPU.  tests/test_hardcases_cpu.py checks the construction itself with rationals.
"""
from __future__ import annotations

import math
import random
from fractions import Fraction

P64 = 53


def mant_exp(x: float, p: int = P64):
    """x = m * 2^e with 2^(p-1) <= m < 2^p (x > 0, normal)."""
    f, e = math.frexp(x)
    m = int(f * (1 << p))
    return m, e - p


def hard_numerators(D: int, p: int = P64, residues=(1, -1, 3, -3, 5, -5, 7, -7, 9, -9), limit: int = 4):
    """p-bit numerators n (as integers) whose quotient by the p-bit divisor mantissa D lies within ~2^-(2p+1) relative of a
    rounding midpoint.  Returns [(n, shift, r)]: n * 2^shift / D = X/2 + r/(2D), X odd, q in [2^(p-1), 2^p)."""
    lo, hi = 1 << (p - 1), 1 << p
    assert lo <= D < hi
    t = (D & -D).bit_length() - 1          # trailing zeros
    Dp = D >> t
    out = []
    for shift in (p + 1, p):               # q = n * 2^(shift-1) / D:  n < D  |  D <= n < 2D
        bits = shift - t
        if bits < 2:
            continue                       # D is (almost) a power of two: every quotient is exact
        mod = 1 << bits
        inv = pow(Dp, -1, mod)
        for rp in residues:
            x0 = (-rp * inv) % mod         # odd, because rp and inv are odd
            # X = x0 + j * mod, X in [2^p, 2^(p+1)): quotient q = X/2 in [2^(p-1), 2^p)
            j0 = max(0, ((1 << p) - x0 + mod - 1) // mod)
            for j in range(j0, j0 + 3):
                X = x0 + j * mod
                if X >= (1 << (p + 1)):
                    break
                num = X * D + (rp << t)
                if num % (1 << shift):
                    continue
                n = num >> shift
                if lo <= n < hi:
                    out.append((n, shift, rp << t))
                    break
            if len(out) >= limit:
                return out
    return out


def mantissa_patterns(p: int = P64, nrandom: int = 300, seed: int = 1):
    """p-bit mantissas with the structures that break reciprocal refinements."""
    top, full = 1 << (p - 1), (1 << p) - 1
    pats = {full, full - 1, full - 2, full - 4, top + 1, top + 2, top + 3, top | (full >> 1 & 0x5555555555555555),
            top | (0xAAAAAAAAAAAAAAAA & (full >> 1))}
    for k in range(1, p - 1):
        pats.add(full + 1 - (1 << k))          # ones down to bit k, zeros below
        pats.add(top + (1 << k))               # a single one
        pats.add(top + (1 << k) - 1)           # zeros, then a run of ones
        pats.add(full - (1 << k))              # all ones with a single zero
    rng = random.Random(seed)
    low = (1 << 32) - 1 if p == P64 else (1 << 12) - 1
    for _ in range(nrandom):
        pats.add(top | rng.getrandbits(p - 1))
        pats.add((top | rng.getrandbits(p - 1)) | low)                      # all-ones low word: the kernel's guard pattern
        pats.add((top | rng.getrandbits(p - 1)) & ~low | (low - 1))        # ... and its neighbour, which the guard lets through
        hi_run = rng.randrange(8, p - 8)
        pats.add(full & ~((1 << hi_run) - 1) | rng.getrandbits(hi_run))    # long run of leading ones, random tail
        pats.add(top | rng.getrandbits(hi_run))                             # long run of zeros after the top bit
    return sorted(x for x in pats if top <= x <= full)


def w_with_sqrt_mantissa(D: int, e: int):
    """A double w in (0, 1] with sqrt(w) == D * 2^(-53 - e) exactly rounded (w = RN(D^2 * 4^(-53 - e)))."""
    w = (D * D) / (1 << (2 * (P64 + e)))     # int / int: correctly rounded in CPython
    s = math.sqrt(w)
    return (w, s) if mant_exp(s)[0] == D else (None, None)


def fp64_term_cases(exponents=(0, 1, 9, 40, 88), nrandom: int = 300, seed: int = 7):
    """(w, num, den) triples for rtmi_selftest_fastmath: num is hard for the divisor s = sqrt(w), den for s3 = (s*s)*s."""
    rng = random.Random(seed)
    W, N, Dn = [], [], []
    for D in mantissa_patterns(P64, nrandom, seed):
        for e in exponents:
            w, s = w_with_sqrt_mantissa(D, e)
            if w is None or not (1e-60 < w <= 1.0):
                continue
            s3 = (s * s) * s
            D3, _ = mant_exp(s3)
            hs = hard_numerators(D) or [(1 << 52, 54, 0)]
            h3 = hard_numerators(D3) or [(1 << 52, 54, 0)]
            for i in range(max(len(hs), len(h3))):
                n1 = hs[i % len(hs)][0]
                n3 = h3[i % len(h3)][0]
                sgn = -1.0 if rng.random() < 0.3 else 1.0
                W.append(w)
                N.append(sgn * math.ldexp(float(n1), rng.randrange(-90, -10)))   # |num| in (1e-12, 1e13): inside the guard
                Dn.append(math.ldexp(float(n3), rng.randrange(-90, -10)))
    return W, N, Dn


def fp64_divide_cases(nrandom: int = 300, seed: int = 11):
    """(num, den) pairs for div_guarded: den > 0 in (1e-90, 1e30), num 0 or of moderate magnitude, quotient at a midpoint."""
    rng = random.Random(seed)
    N, Dn = [], []
    for D in mantissa_patterns(P64, nrandom, seed):
        for n, _, _ in hard_numerators(D):
            ed = rng.randrange(-290, 40)                 # den = D * 2^(ed - 52) in (1e-88, 1e12)
            en = rng.randrange(-190, 190)
            N.append((-1.0 if rng.random() < 0.5 else 1.0) * math.ldexp(float(n), en - 52))
            Dn.append(math.ldexp(float(D), ed - 52))
    return N, Dn


def fp64_sqrt_cases(nrandom: int = 300, seed: int = 13):
    """w around squares of midpoints and around perfect squares, over the guarded range (1e-60, 1e60)."""
    rng = random.Random(seed)
    out = set()
    for M in mantissa_patterns(P64, nrandom, seed):
        for e in (0, 1, 30, 99, -60, -98):
            for num in ((2 * M + 1) ** 2, (2 * M - 1) ** 2, 4 * M * M):     # (M +- 1/2)^2 and M^2, times 4
                sh = 2 * (P64 + e) + 2
                w = num / (1 << sh) if sh >= 0 else float(num * (1 << -sh))
                for _ in range(2):
                    for x in (w, math.nextafter(w, 0.0), math.nextafter(w, math.inf)):
                        if 1e-60 < x < 1e60:
                            out.add(x)
                    w = math.nextafter(w, math.inf if rng.random() < 0.5 else 0.0)
    return sorted(out)


def midpoint_distance(n: int, shift: int, D: int):
    """Exact relative distance of q = n * 2^(shift-1) / D from the nearest rounding midpoint (as a Fraction)."""
    q = Fraction(n << (shift - 1), D)
    fl = q.numerator // q.denominator
    mid = Fraction(2 * fl + 1, 2)
    return abs(q - mid) / q
