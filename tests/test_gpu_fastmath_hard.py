"""Constructed hard cases for the kernels' guarded sqrt / divide / shared-reciprocal layer term ("the
bit-exact fast path rests on sampling").  Operands come from tests/hardcases.py (checked on the CPU with exact
rationals): quotients within ~2^-107 relative of a rounding midpoint for divisors whose mantissas are all ones, all
ones minus a few ulps, single ones, long runs, all-ones low words ... -- for the divisors the kernels really form,
s = sqrt(w) and s3 = (s*s)*s -- and w around squares of midpoints.  The bar is zero mismatches against the plain
IEEE operators evaluated in the same launch, and the test must FAIL if the all-ones-divisor guard is taken out
(`test_all_ones_guard_has_teeth`)."""
import numpy as np
import pytest

import hardcases as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def raymod():
    import raytracerfortran_amd as r
    r.load_library()
    r.init(0)
    return r.raymod


def dev(a):
    import torch
    return torch.tensor(np.asarray(a, dtype=np.float64), device="cuda")


def values(raymod, w, num, den):
    import torch
    out = raymod.selftest_values(dev(w), dev(num), dev(den))
    return {k: v.cpu().numpy() for k, v in out.items()}


def same_bits(a, b):
    return a.view(np.uint64) == b.view(np.uint64)


def test_layer_term_on_midpoint_quotients(raymod):
    """a / s and hv / s3 with numerators built for the divisors s = sqrt(w) and s3 = (s*s)*s themselves, against the
    IEEE results computed on the HOST (numpy): the GPU's own `/` is not a valid checker (see
    test_compilers_divide_is_only_faithful)."""
    W, N, D = (np.array(x) for x in H.fp64_term_cases(nrandom=400))
    assert W.size > 40000
    for num, den in ((N, D), (D * -0.25, np.abs(N) * 8.0)):       # and the other pairing of the same divisors
        v = values(raymod, W, num, den)
        s = np.sqrt(W)
        assert np.all(same_bits(v["sqrt"], s))
        acc = ~np.isnan(v["tf"])
        assert acc.sum() > 0.7 * W.size and (~acc).sum() > 0         # the guard saw its patterns, and only them
        assert np.all(same_bits(v["tf"][acc], (num / s)[acc])), "a / sqrt(w): fast form differs from IEEE"
        assert np.all(same_bits(v["tp"][acc], (den / ((s * s) * s))[acc])), "hv / sqrt(w)^3: fast form differs from IEEE"
        # what the guard rejects is what the kernels route to the plain operators: those must be IEEE as well
        assert np.all(same_bits(values(raymod, W, num, s)["div"], num / s))
        assert np.all(same_bits(values(raymod, W, den, (s * s) * s)["div"], den / ((s * s) * s)))


def test_divides_on_midpoint_quotients(raymod):
    """div_ieee (every division outside the fast forms) and div_guarded (initial p0, time pass) on quotients at
    rounding midpoints, all divisor patterns and exponents, against numpy."""
    N, D = (np.array(x) for x in H.fp64_divide_cases(nrandom=400))
    assert N.size > 4000
    v = values(raymod, np.full(N.size, 0.75), N, D)
    assert np.all(same_bits(v["div"], N / D))
    acc = ~np.isnan(v["div_guarded"])
    assert acc.sum() > 0.8 * N.size and np.all(same_bits(v["div_guarded"][acc], (N / D)[acc]))
    # negative divisors and numerators (f / f' of the solver has a negative divisor)
    v = values(raymod, np.full(N.size, 0.75), -N, -D)
    assert np.all(same_bits(v["div"], N / D))
    v = values(raymod, np.full(N.size, 0.75), N, -D)
    assert np.all(same_bits(v["div"], N / -D))


def test_compilers_divide_is_only_faithful(raymod):
    """Why div_ieee exists: on these operands hipcc's own fp64 `/` (as torch's GPU division uses it too) returns the
    WRONG neighbour for some quotients, while never being more than one ulp off.  Informational on the compiler side
    (a future ROCm may fix it: no assertion that it fails), binding on ours (div_ieee must be exact on the same set)."""
    import torch
    N, D = (np.array(x) for x in H.fp64_divide_cases(nrandom=400))
    q_gpu = (dev(N) / dev(D)).cpu().numpy()
    q_cpu = N / D
    wrong = int((~same_bits(q_gpu, q_cpu)).sum())
    ulps = np.abs(q_gpu.view(np.int64) - q_cpu.view(np.int64)).max()
    print(f"compiler's fp64 divide: {wrong} of {N.size} constructed quotients are not correctly rounded (max {ulps} ulp)")
    assert ulps <= 1
    assert np.all(same_bits(values(raymod, np.full(N.size, 0.75), N, D)["div"], q_cpu))


def test_sqrt_guarded_around_midpoint_squares(raymod):
    ws = np.array(H.fp64_sqrt_cases(nrandom=400))
    assert ws.size > 50000
    one = np.ones(ws.size)
    v = values(raymod, ws, one, one)
    assert np.all(same_bits(v["sqrt"], np.sqrt(ws)))
    inside = ws[(ws > 1e-60) & (ws <= 1.0)]
    v = values(raymod, inside, np.ones(inside.size), np.ones(inside.size))
    s = np.sqrt(inside)
    acc = ~np.isnan(v["tf"])
    assert np.all(same_bits(v["tf"][acc], (1.0 / s)[acc])) and np.all(same_bits(v["tp"][acc], (1.0 / ((s * s) * s))[acc]))


def test_near_all_ones_divisors_every_distance(raymod):
    """The family that broke the one-step reciprocal of round 1: s = (2^53 - M) 2^e for EVERY M up to 4096 (and a sample
    up to 2^17) -- what the first evaluation of each bisection produces (x1 = 1e-10: M ~ 5e-5 v^2) -- with eight
    midpoint numerators each, for a / s and hv / s3."""
    full = (1 << 53) - 1
    W, N, Dn = [], [], []
    for m in list(range(0, 4096)) + list(range(4096, 1 << 17, 37)):
        for e in (0, 1, 40):
            w, s = H.w_with_sqrt_mantissa(full - m, e)
            if w is None:
                continue
            D3, _ = H.mant_exp((s * s) * s)
            hs = H.hard_numerators(full - m, limit=8) or [(1 << 52, 54, 0)]
            h3 = H.hard_numerators(D3, limit=8) or [(1 << 52, 54, 0)]
            for i in range(max(len(hs), len(h3))):
                W.append(w), N.append(float(hs[i % len(hs)][0])), Dn.append(float(h3[i % len(h3)][0]))
    W, N, Dn = np.array(W), np.array(N), np.array(Dn)
    assert W.size > 100000
    v = values(raymod, W, N, Dn)
    s = np.sqrt(W)
    acc = ~np.isnan(v["tf"])
    assert (~acc).sum() < 200                                    # only M = 1 (s) and M3 <= 7 (s3) are sent to the plain path
    assert np.all(same_bits(v["tf"][acc], (N / s)[acc])) and np.all(same_bits(v["tp"][acc], (Dn / ((s * s) * s))[acc]))
    assert np.all(same_bits(values(raymod, W, N, s)["div"], N / s))


def test_all_ones_guard_has_teeth(raymod):
    """Markstein's exception is real and the guard is what catches it: over square roots with an all-ones mantissa and
    short numerators the UNGUARDED term disagrees with IEEE (count > 0), the guarded one never does, and the guard
    rejects exactly those operands.  If the guard were compiled out of term_guarded_n, `guarded` below would equal
    `unguarded` and this test would fail."""
    ws = []
    for k in range(0, 60):
        g = np.nextafter(np.float64(2.0) ** (-k), 0.0)
        w0 = g * g
        for w in (np.nextafter(w0, 0), w0, np.nextafter(w0, 1)):
            if np.sqrt(w) == g and 1e-60 < w <= 1.0:
                ws.append(w)
    ws = np.unique(ws)
    spec = np.concatenate([2.0 ** np.arange(-40, 41), np.arange(1, 200, dtype=np.float64), 1.0 / np.arange(1, 200)])
    W = np.repeat(ws, spec.size)
    Nn = np.tile(spec, ws.size)
    s_, d_, guarded, unguarded, rejected = raymod.selftest_fastmath_ex(dev(W), dev(Nn), dev(Nn[::-1].copy()))
    assert guarded == 0 and s_ == 0
    assert unguarded > 0, "the all-ones family no longer trips the unguarded term: the test lost its teeth"
    assert rejected == W.size


def test_fp32_layer_term_every_w_with_midpoint_numerators(raymod):
    """Single precision, EVERY float w of the guarded range (1e-12, 1]: numerators constructed on the GPU (integer
    arithmetic, modular inverse by Newton's iteration) so that num / sqrtf(w) and den / sqrtf(w)^3 lie within ~2^-49
    relative of a rounding midpoint -- not just num = 1."""
    import torch

    def hard_num(divisor):
        """For a float32 tensor of divisors: float32 numerators whose quotient sits next to a midpoint (1.0 where the
        construction finds none).  n * 2^shift = X * D + r, X odd in [2^24, 2^25), r in {+-1, +-3, +-5, +-7} << t
        (tests/hardcases.py: hard_numerators, vectorised for p = 24)."""
        bits = divisor.view(torch.int32).to(torch.int64)
        D = (bits & 0x7FFFFF) | 0x800000
        t = torch.zeros_like(D)
        Dp = D.clone()
        for _ in range(23):                       # strip trailing zeros
            ev = (Dp & 1) == 0
            Dp = torch.where(ev, Dp >> 1, Dp)
            t = t + ev.to(torch.int64)
        inv = Dp.clone()                          # Newton: inv <- inv (2 - Dp inv), doubling the valid bits from 3
        for _ in range(4):
            inv = (inv * (2 - Dp * inv)) & 0x3FFFFFF
        out = torch.ones_like(divisor)
        found = torch.zeros_like(D, dtype=torch.bool)
        for shift in (25, 24):                    # n < D: q = n 2^24 / D  |  D <= n < 2 D: q = n 2^23 / D
            nb = shift - t                        # X is determined modulo 2^nb
            mask = (torch.ones_like(D) << torch.clamp(nb, min=0)) - 1
            span = mask + 1
            for rp in (1, -1, 3, -3, 5, -5, 7, -7):
                x0 = (-rp * inv) & mask
                j = torch.clamp(((1 << 24) - x0 + span - 1) // span, min=0)
                X = x0 + j * span
                num = X * D + (rp << t)
                n = num >> shift
                ok = (~found) & (nb >= 2) & (X < (1 << 25)) & ((num & ((1 << shift) - 1)) == 0) & (n >= (1 << 23)) & \
                     (n < (1 << 24))
                out = torch.where(ok, n.to(torch.float32), out)
                found |= ok
        return out, found

    lo = int(np.float32(1.0000001e-12).view(np.uint32))
    hi = int(np.float32(1.0).view(np.uint32))
    tot = [0, 0]
    covered = total = 0
    step = 1 << 25
    for start in range(lo, hi + 1, step):
        b = torch.arange(start, min(start + step, hi + 1), dtype=torch.int64, device="cuda").to(torch.int32)
        w = b.view(torch.float32)
        s = torch.sqrt(w)
        s3 = (s * s) * s
        num, f1 = hard_num(s)
        den, f3 = hard_num(s3)
        scale = torch.exp2(torch.remainder(b, 41).to(torch.float32) - 20.0)   # exact powers of two, 2^-20 .. 2^20
        nn, dd = (num * scale).contiguous(), (den / scale).contiguous()
        a_, b_ = raymod.selftest_fastmath_f32(w.contiguous(), nn, dd)
        tot[0] += a_
        tot[1] += b_
        # a thirteenth of the operands also against the IEEE host (numpy float32): sqrtf, the repaired divide and the term
        sub = slice(start % 13, None, 13)
        ws_, ns_, ds_ = w[sub].contiguous(), nn[sub].contiguous(), dd[sub].contiguous()
        v = raymod.selftest_values(ws_, ns_, s[sub].contiguous())
        hw, hn, hd = ws_.cpu().numpy(), ns_.cpu().numpy(), ds_.cpu().numpy()
        hs = np.sqrt(hw)
        assert np.array_equal(v["div"].cpu().numpy().view(np.uint32), (hn / hs).view(np.uint32)), "float divide not IEEE"
        v = raymod.selftest_values(ws_, ns_, ds_)
        tf, tp = v["tf"].cpu().numpy(), v["tp"].cpu().numpy()
        acc = ~np.isnan(tf)
        assert np.array_equal(tf[acc].view(np.uint32), (hn / hs)[acc].view(np.uint32))
        assert np.array_equal(tp[acc].view(np.uint32), (hd / ((hs * hs) * hs))[acc].view(np.uint32))
        checked_on_host = locals().get("checked_on_host", 0) + int(acc.sum())
        covered += int(f1.sum()) + int(f3.sum())
        total += 2 * w.numel()
    assert tot == [0, 0], tot
    assert covered > 0.9 * total, (covered, total)   # the construction succeeds for nearly every divisor
    assert checked_on_host > 2e7
