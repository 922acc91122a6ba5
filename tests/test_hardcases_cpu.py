"""The hard-case generator of tests/hardcases.py, checked with exact rational arithmetic on the CPU: the operands it
hands to the GPU self-tests really sit where it says (quotients within 2^-100 relative of a rounding midpoint, divisors
with the promised mantissas), so a green GPU test means something."""
import math
from fractions import Fraction

import numpy as np

import hardcases as H


def test_quotients_sit_on_rounding_midpoints_fp64():
    pats = H.mantissa_patterns(53, nrandom=60, seed=2)
    assert (1 << 53) - 1 in pats and (1 << 52) + 1 in pats and len(pats) > 400
    seen = 0
    for D in pats[::7]:
        for n, shift, r in H.hard_numerators(D):
            assert (1 << 52) <= n < (1 << 53)
            d = H.midpoint_distance(n, shift, D)
            q = Fraction(n << (shift - 1), D)
            assert d == Fraction(abs(r), 2 * D) / q              # exactly |r| / (2 D) away from the midpoint
            assert d > 0
            if D & 1:
                assert d < Fraction(1, 1 << 100), (D, n, float(d))   # odd divisors: as close as a quotient can come
            # and the correctly rounded quotient is the neighbour on the side of r
            fl = q.numerator // q.denominator
            assert (q > Fraction(2 * fl + 1, 2)) == (r > 0)
            seen += 1
    assert seen > 150


def test_quotients_sit_on_rounding_midpoints_fp32():
    for D in H.mantissa_patterns(24, nrandom=40, seed=3)[::5]:
        for n, shift, r in H.hard_numerators(D, p=24):
            assert (1 << 23) <= n < (1 << 24)
            d = H.midpoint_distance(n, shift, D)
            assert d > 0 and (d < Fraction(1, 1 << 44) or not D & 1)


def test_w_steers_the_square_root_mantissa():
    full = (1 << 53) - 1
    for D in (full, full - 1, (1 << 52) + 1, 0x1AAAAAAAAAAAAA, 0x1FFFFFFF00000001 >> 8 | 1 << 52):
        for e in (0, 9, 88):
            w, s = H.w_with_sqrt_mantissa(D, e)
            assert w is not None and H.mant_exp(s)[0] == D and 1e-60 < w <= 1.0
    W, N, Dn = H.fp64_term_cases(nrandom=20)
    W, N, Dn = np.array(W), np.array(N), np.array(Dn)
    assert W.size > 3000 and W.min() > 1e-60 and W.max() <= 1.0
    assert np.all((np.abs(N) > 1e-30) & (np.abs(N) < 1e30) & (Dn > 1e-30) & (Dn < 1e30))
    # the family the kernel's guard exists for is in there: square roots with an all-ones mantissa
    s = np.sqrt(W)
    assert np.any((s.view(np.uint64) & ((1 << 52) - 1)) == (1 << 52) - 1)


def test_sqrt_cases_straddle_midpoint_squares():
    ws = H.fp64_sqrt_cases(nrandom=10)
    assert len(ws) > 5000
    near = 0
    for w in ws[::97]:
        s = math.sqrt(w)
        m, e = H.mant_exp(s)
        # distance of the exact root from the nearer midpoint neighbour of s, in ulps of s
        exact2 = Fraction(w)
        lo, hi = Fraction(2 * m - 1, 2) * Fraction(2) ** e, Fraction(2 * m + 1, 2) * Fraction(2) ** e
        assert lo * lo <= exact2 <= hi * hi            # s is the correctly rounded root (sanity of math.sqrt)
        d = min(abs(exact2 - lo * lo), abs(exact2 - hi * hi)) / (2 * Fraction(s) * Fraction(2) ** e)
        near += d < Fraction(1, 4)
    assert near > 10
