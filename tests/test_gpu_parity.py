"""Parity of the HIP path with the oracle / the reference fixtures -- needs an MI355X.

Everything here calls through the C ABI of librtmi.so (via the ctypes mirror
raytracerfortran_amd.raymod).  The bar (BASELINE.json): identical -999 sentinel
positions and |dt| < 1e-9 s in fp64; because the kernel repeats the reference's
operation order with correctly rounded sqrt and divide, the results are in fact
expected to be BIT-IDENTICAL, and the tests that say so assert err == 0.
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import compare_times, golden_case_names

pytestmark = pytest.mark.gpu

TOL = 1e-9  # seconds; BASELINE.json north_star


@pytest.fixture(scope="module")
def rt():
    import raytracerfortran_amd as r
    r.load_library()
    r.init(0)
    return r


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def c2_workload(n, seed=12345):
    """8 interfaces, 9 velocities with one LVZ."""
    rng = np.random.default_rng(seed)
    d = np.cumsum(rng.uniform(300, 800, 8))
    v = np.sort(rng.uniform(1500, 6500, 9))
    v[[2, 3]] = v[[3, 2]]
    off = rng.uniform(10, 8000, n)
    dep = rng.uniform(50, d[-1] + 500, n)
    return v, d, off, dep


# ------------------------------------------------------------------------------------------
# the reference's own example and the author's known answers, through `dff_`
# ------------------------------------------------------------------------------------------
def test_config1_test_1(rt, kat):
    m = kat["model"]
    r = rt.dff(vels=m["vels"], depths=m["depths"], NLayers=len(m["depths"]),
               src_offset=kat["test_1_src_offset"], src_depth=kat["test_1_src_depth"],
               NSrc=len(kat["test_1_src_depth"]), timeP=np.zeros(20), keep_delta=-1)
    assert compare_times(r["timeP"], kat["config1_ref_timeP"], TOL) == 0.0


def test_kat_a_notebook_print(rt, kat):
    m, a = kat["model"], kat["kat_a"]
    t = rt.dff(m["vels"], m["depths"], 6, a["src_offset"], a["src_depth"], 5)["timeP"]
    assert [f"{x:.7f}" for x in t] == [f"{x:.7f}" for x in a["printed_timeP"]]


def test_kat_b_rays_dat(rt, kat, tmp_path):
    """keep_delta = 10 writes ./rays.dat; its token stream must equal the author's committed file."""
    m = kat["model"]
    old = os.getcwd()
    os.chdir(tmp_path)
    try:
        rt.dff(m["vels"], m["depths"], 6, kat["test_1_src_offset"], kat["test_1_src_depth"], 20, keep_delta=10)
        lines = open("rays.dat").read().splitlines()
    finally:
        os.chdir(old)
    assert len(lines) == 40  # one record per line (read.rays scans line-wise, MiscFunctions.R:13-26)
    ours = np.array([float(x) for ln in lines for x in ln.split()])
    ref = np.array(kat["kat_b"]["rays_dat_tokens"])
    assert ours.size == ref.size
    assert (np.abs(ours - ref) / np.abs(ref)).max() < 5e-13


def test_raytracer_r_example(rt, kat):
    e = kat["rayTracerR_example"]
    t = rt.dff(e["vels"], e["depths"], 2, e["src_offset"], e["src_depth"], len(e["src_depth"]), keep_delta=-1)
    assert compare_times(t["timeP"], e["ref_timeP"], TOL) == 0.0


def test_tracerays_and_7arg_flavours(rt, kat):
    m = kat["model"]
    off, dep = kat["test_1_src_offset"], kat["test_1_src_depth"]
    t = rt.TraceRays(m["vels"], m["depths"], 6, off, dep, 20, np.zeros(20), -1)
    assert compare_times(t, kat["config1_ref_timeP"], TOL) == 0.0
    t7 = rt.dff7(m["vels"], m["depths"], 6, off, dep, 20)["timeP"]
    assert compare_times(t7, kat["config1_ref_timeP"], TOL) == 0.0
    # librtmi_dff7.so: symbol `dff_` with seven arguments (subroutineR.f90:405)
    lib7 = ctypes.CDLL(rt.raymod.LIB7_PATH)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    lib7.dff_.argtypes = [dp, dp, ip, dp, dp, ip, dp]
    lib7.dff_.restype = None
    v, d = np.array(m["vels"]), np.array(m["depths"])
    o, z, out = np.array(off), np.array(dep), np.zeros(20)
    nl, ns = ctypes.c_int(6), ctypes.c_int(20)
    lib7.dff_(v.ctypes.data_as(dp), d.ctypes.data_as(dp), ctypes.byref(nl), o.ctypes.data_as(dp),
              z.ctypes.data_as(dp), ctypes.byref(ns), out.ctypes.data_as(dp))
    assert compare_times(out, kat["config1_ref_timeP"], TOL) == 0.0


# ------------------------------------------------------------------------------------------
# committed fixtures generated from the compiled reference
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_case_names())
def test_reference_fixture(rt, name, golden_batches):
    c = golden_batches[name]
    t = rt.dff(c["vels"], c["depths"], c["depths"].size, c["off"], c["dep"], c["off"].size)["timeP"]
    err = compare_times(t, c["timeP"], TOL)
    assert err == 0.0, f"{name}: not bit-identical ({err:.3e} s)"


# ------------------------------------------------------------------------------------------
# seeded random inputs against the oracle
# ------------------------------------------------------------------------------------------
def test_random_models_vs_oracle(rt, O):
    rng = np.random.default_rng(99)
    for it in range(25):
        k = int(rng.integers(2, 31))
        d = np.sort(rng.uniform(100.1, 10000.1, k - 1))
        v = rng.uniform(1500, 10000, k)
        n = int(rng.integers(1, 5000))
        off = rng.uniform(0, 12000, n)
        dep = rng.uniform(1, 11000, n)
        t = rt.dff(v, d, k - 1, off, dep, n)["timeP"]
        assert compare_times(t, O.oracle_dff(v, d, off, dep), TOL) == 0.0


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 255, 256, 257, 1000])
def test_ragged_sizes(rt, O, n):
    v, d, off, dep = c2_workload(max(n, 1), seed=5)
    off, dep = off[:n], dep[:n]
    t = rt.dff(v, d, 8, off, dep, n)["timeP"]
    assert t.shape == (n,)
    if n:
        assert compare_times(t, O.oracle_dff(v, d, off, dep), TOL) == 0.0


def test_bad_inputs_do_not_hang(rt, O):
    """NaN / Inf source data: the reference loops forever (p0 never becomes 'good'); the HIP path and the
    oracle report -999 for those sources and leave the others untouched."""
    v, d, off, dep = c2_workload(512, seed=6)
    off[3], dep[7], off[11], dep[13] = np.nan, np.nan, np.inf, np.inf
    t = rt.dff(v, d, 8, off, dep, 512)["timeP"]
    ref = O.oracle_dff(v, d, off, dep)
    ok = np.ones(512, bool)
    ok[[3, 7, 11, 13]] = False
    assert compare_times(t[ok], ref[ok], TOL) == 0.0
    assert np.array_equal(t[~ok], ref[~ok], equal_nan=True)  # -999 for NaN data; Inf offset follows the arithmetic


def test_small_launches_a_group_of_lanes_per_source(rt, O):
    """Launches whose sources x group width fit the chip's resident lanes once (the reference's own example has 20 sources, its
    sampler calls TraceRays with 1e4) take the kernel with a GROUP of lanes per source (csrc/rtmi_kernels.hip::trace_small_kernel: one layer term of costFunc / costFunc_Prime per lane, added up
    in layer order, subroutineR-quiet.f90:184-221): travel times AND final ray parameters must be the oracle's bits for every
    group width (8, 16, 32, 64 lanes: models of 1 ... 63 interfaces), for sources in the top layer, exactly on interfaces, below
    the last interface, with NaN / Inf data, with interfaces out of order -- and equal to the one-lane-per-source kernel that
    rtmi_set_option("path", 1) selects, strictly."""
    import torch
    rng = np.random.default_rng(20260)
    dev = "cuda"
    for nl in (1, 2, 3, 7, 8, 9, 15, 16, 20, 31, 32, 47, 63):
        d = np.sort(rng.uniform(80.0, 9000.0, nl))
        if nl >= 4 and nl % 2 == 0:
            d[2] = d[1]                                   # a zero-thickness layer
        if nl in (9, 47):
            d[[1, nl - 2]] = d[[nl - 2, 1]]               # interfaces out of order: whichLayer's own scan
        v = rng.uniform(1500.0, 8000.0, nl + 1)
        for n in (1, 20, 333, 1024, 4099):
            off = rng.uniform(0.0, 12000.0, n)
            dep = rng.uniform(1.0, d.max() + 800.0, n)
            m = min(max(n - 8, 1), nl)
            dep[:m] = d[-m:]                              # sources exactly on interfaces (the last one included)
            if n >= 20:
                dep[n - 1] = 0.5 * d.min()                # top layer: straight ray
                off[n - 2] = 0.0                          # vertical ray
                off[n - 3] = -off[n - 4]                  # negative offset: the mirror ray
                off[n - 5], dep[n - 6] = np.nan, np.nan
                off[n - 7] = np.inf
            t_ref, p_ref, _ = O.oracle_dff(v, d, off, dep, want_p=True)
            tv, td = torch.tensor(v, device=dev), torch.tensor(d, device=dev)
            to, tz = torch.tensor(off, device=dev), torch.tensor(dep, device=dev)
            p = torch.empty(n, dtype=torch.float64, device=dev)
            t = rt.trace_device(tv, td, to, tz, p_final=p).cpu().numpy()
            assert np.array_equal(t, t_ref, equal_nan=True), (nl, n, int((t != t_ref).sum()))
            assert np.array_equal(p.cpu().numpy(), p_ref, equal_nan=True), (nl, n)
            t2 = rt.trace_device(tv, td, to, tz).cpu().numpy()                         # (the variant without p_final)
            assert np.array_equal(t2, t_ref, equal_nan=True), (nl, n)
            if n in (20, 1024):                                                       # the same sources, one lane each
                try:
                    rt.set_option("path", 1)
                    lane = rt.trace_device(tv, td, to, tz).cpu().numpy()
                    assert rt.last_kernel_name().startswith("void rtmi::trace_kernel<double")
                finally:
                    rt.set_option("path", 0)
                assert np.array_equal(lane, t, equal_nan=True), nl
                rt.trace_device(tv, td, to, tz)
                assert "trace_small_kernel" in rt.last_kernel_name(), rt.last_kernel_name()
            th = rt.dff(v, d, nl, off, dep, n)["timeP"]                               # ... and through the drop-in symbol
            assert np.array_equal(th, t_ref, equal_nan=True), (nl, n)
    assert rt.load_library().rtmi_device_status() == 0


def test_bad_arguments_raise(rt):
    import torch
    with pytest.raises(rt.RtmiError):
        rt.dff([3000.0], [], 0, [1.0], [1.0], 1)
    z = torch.zeros(4, dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError):
        rt.trace_device(z[:3], z[:3], z, z)  # vels must have NLayers+1 entries
    with pytest.raises(ValueError):
        rt.trace_device(z[:3].cpu(), z[:2], z, z)  # host tensor


# ------------------------------------------------------------------------------------------
# device-resident entry points
# ------------------------------------------------------------------------------------------
def test_trace_device_full_c2(rt, O):
    """BASELINE config 2 at full size: 1e6 sources, 8 interfaces, single forward, fp64."""
    import torch
    n = 1_000_000
    v, d, off, dep = c2_workload(n)
    dv, dd = torch.tensor(v, device="cuda"), torch.tensor(d, device="cuda")
    do, dz = torch.tensor(off, device="cuda"), torch.tensor(dep, device="cuda")
    p = torch.empty(n, dtype=torch.float64, device="cuda")
    t = rt.trace_device(dv, dd, do, dz, p_final=p)
    torch.cuda.synchronize()
    t = t.cpu().numpy()
    ref, pref, nl = O.oracle_dff(v, d, off, dep, want_p=True)
    assert compare_times(t, ref, TOL) == 0.0
    assert (ref == -999.0).sum() == 12  # the C2 stream has 12 non-converged sources
    # final ray parameters: NaN for straight rays, identical elsewhere
    p = p.cpu().numpy()
    assert np.array_equal(np.isnan(p), nl == 1)
    assert np.array_equal(p[nl > 1], pref[nl > 1])
    # size-independent properties: order independence and determinism
    perm = np.random.default_rng(0).permutation(n)
    t2 = rt.trace_device(dv, dd, torch.tensor(off[perm], device="cuda"), torch.tensor(dep[perm], device="cuda"))
    assert np.array_equal(t2.cpu().numpy(), t[perm])
    t3 = rt.trace_device(dv, dd, torch.tensor(np.sort(off), device="cuda"),
                         torch.tensor(dep[np.argsort(off)], device="cuda"))  # sorted input: other chunking
    assert np.array_equal(t3.cpu().numpy(), t[np.argsort(off)])


@pytest.mark.parametrize("n", [1, 64, 8192, 40001, 300000, 1200000])
@pytest.mark.parametrize("wps,refill,tail,retire,promo", [(4, 16, 256, 8, 10), (1, 1, 0, 8, 3), (8, 64, 4096, 64, 0),
                                                           (5, 7, 100, 1, 1), (3, 16, 0, 0, 0), (4, 8, 0, 21, 2),
                                                           (2, 4, 64, 16, 5), (4, 16, 512, 0, 4)])
def test_fused_kernel_equals_plain_kernel(rt, O, n, wps, refill, tail, retire, promo):
    """Large source sets take the fused kernel (wave-local sort by layer, lane refill); small ones the plain
    one-lane-per-source kernel.  Same arithmetic per source: bit-identical results under every scheduling
    knob, final ray parameters included, with NaN data, straight rays, interface hits and -999 sources."""
    import torch
    v, d, off, dep = c2_workload(n, seed=77 + n)
    if n > 400:
        off[5], dep[9] = np.nan, np.inf
        dep[100:200] = 10.0          # a run of straight top-layer rays
        dep[300] = d[-1]             # exactly on the last interface
        dep[301] = d[3]              # exactly on an inner interface
    dv, dd = torch.tensor(v, device="cuda"), torch.tensor(d, device="cuda")
    do, dz = torch.tensor(off, device="cuda"), torch.tensor(dep, device="cuda")
    p1 = torch.empty(n, dtype=torch.float64, device="cuda")
    p2 = torch.empty(n, dtype=torch.float64, device="cuda")
    try:
        rt.set_option("path", 1)
        t1 = rt.trace_device(dv, dd, do, dz, p_final=p1).cpu().numpy()
        rt.set_option("path", 2)
        rt.set_option("waves_per_simd", wps)
        rt.set_option("refill_min", refill)
        rt.set_option("tail_group", tail)
        rt.set_option("retire_max", retire)
        rt.set_option("promote_age", promo)
        t2 = rt.trace_device(dv, dd, do, dz, p_final=p2).cpu().numpy()
        t3 = rt.trace_device(dv, dd, do, dz).cpu().numpy()  # without p_final
    finally:
        rt.set_option("path", 0)
        rt.set_option("waves_per_simd", 4)
        rt.set_option("refill_min", 32)
        rt.set_option("tail_group", 128)
        rt.set_option("retire_max", 12)
        rt.set_option("promote_age", 3)
    assert np.array_equal(t1, t2, equal_nan=True) and np.array_equal(t2, t3, equal_nan=True)
    ref, pref, nl = O.oracle_dff(v, d, off, dep, want_p=True)
    assert np.array_equal(t2, ref, equal_nan=True)
    p1, p2 = p1.cpu().numpy(), p2.cpu().numpy()
    good = (nl > 1) & np.isfinite(off) & np.isfinite(dep)
    assert np.array_equal(p2[good], pref[good]) and np.all(np.isnan(p2[nl == 1]))
    assert np.array_equal(p1[good], p2[good])


def test_fused_kernel_degenerate_rounds(rt, O):
    """Rounds with nothing to solve (every source a straight top-layer ray), only hopeless data (NaN), every source
    finished by its first evaluation, and a one-interface model -- through the fused kernel."""
    import torch
    v, d, off, dep = c2_workload(20000, seed=4)
    cases = {"all straight": (off, np.full_like(dep, 0.5 * d[0])),
             "all NaN": (np.full_like(off, np.nan), dep),
             "zero offset": (np.zeros_like(off), dep)}
    try:
        rt.set_option("path", 2)
        for name, (o, z) in cases.items():
            t = rt.trace_device(torch.tensor(v, device="cuda"), torch.tensor(d, device="cuda"),
                                torch.tensor(o, device="cuda"), torch.tensor(z, device="cuda")).cpu().numpy()
            assert np.array_equal(t, O.oracle_dff(v, d, o, z), equal_nan=True), name
        v1, d1 = np.array([3000.0, 5000.0]), np.array([1000.0])
        t = rt.trace_device(torch.tensor(v1, device="cuda"), torch.tensor(d1, device="cuda"),
                            torch.tensor(off, device="cuda"), torch.tensor(dep, device="cuda")).cpu().numpy()
        assert np.array_equal(t, O.oracle_dff(v1, d1, off, dep))
    finally:
        rt.set_option("path", 0)


def test_fast_math_equals_ieee(rt):
    """The guarded sqrt / divide (compiler expansions without their exponent-range wrappers) return the same
    bits as the plain operators over the guarded operand ranges: random operands, operands shaped like the
    solver's (w = 1 - p^2 v^2 near 0 and near 1), and the guard's boundary decades."""
    import torch
    from raytracerfortran_amd import raymod
    g = torch.Generator(device="cuda").manual_seed(5)
    n = 1 << 24
    total_s = total_d = total_t = 0
    for trial in range(8):
        u = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
        e = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
        if trial == 0:      # solver-like: w in (0, 1], numerators ~ H*V*p, denominators s and s^3
            w = 1.0 - u * u * (1.0 - 1e-12)
            den = torch.sqrt(w) ** (1 + 2 * (e > 0.5))
            num = (u + 0.1) * 1e4 * torch.exp(-20 * e)
        elif trial == 1:    # near-critical: w tiny
            w = torch.exp(-100.0 * u) * (1.0 - 1e-16) + 1e-59
            den = torch.exp(-190.0 * e) + 1e-89
            num = torch.exp(138.0 * (u - 0.5)) * torch.sign(e - 0.3)
        elif trial == 2:    # whole guarded range, log-uniform
            w = torch.exp(-138.0 * u)
            den = torch.exp(-207.0 * e)
            num = torch.exp(276.0 * (u - 0.5))
        elif trial == 3:    # exact zeros and ones among ordinary values
            w = torch.where(u < 0.1, torch.ones_like(u), u)
            den = torch.where(e < 0.1, torch.ones_like(e), e + 1e-3)
            num = torch.where(u > 0.9, torch.zeros_like(u), u - 0.5)
        elif trial == 4:    # the solver's own operands: w = 1 - p^2 v^2, numerators H V p and H V
            pv = u * (1.0 - 1e-9)
            w = 1.0 - pv * pv
            hv = (50.0 + 3000.0 * e) * (1500.0 + 8500.0 * torch.rand(n, dtype=torch.float64, device="cuda", generator=g))
            den = hv
            num = hv * (pv / 6000.0)
        elif trial == 5:    # near-critical layers: w down to 1e-12
            w = torch.exp(-27.0 * u)
            den = 1e5 + 1e7 * e
            num = den * 1e-4 * torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
        elif trial == 7:    # initial_p0: sqrt(R^2 + z^2) and quotients of lengths and times, both sides of 1
            w = torch.exp(276.0 * (u - 0.5))                  # 1e-60 .. 1e60
            den = torch.exp(138.0 * (e - 0.5))                # 1e-30 .. 1e30
            num = torch.where(u > 0.95, torch.zeros_like(u), torch.exp(138.0 * (u - 0.5)) * torch.sign(e - 0.2))
        else:               # the time pass: H / (V cos) with the whole guarded range of V cos, H = 0 included
            w = torch.exp(-138.0 * u)
            den = torch.exp(207.0 * (e - 2.0 / 3.0))          # 1e-60 .. 1e30
            num = torch.where(u > 0.95, torch.zeros_like(u), torch.exp(138.0 * (u - 0.5)) * torch.sign(e - 0.2))
        bs, bd, bt = raymod.selftest_fastmath(w.contiguous(), num.contiguous(), den.contiguous())
        total_s += bs
        total_d += bd
        if 3 <= trial <= 5:  # the layer term is only ever fed solver-shaped operands (guards: w in (1e-60, 1])
            total_t += bt
    assert (total_s, total_d, total_t) == (0, 0, 0)


def test_fast_math_all_ones_divisors(rt):
    """Markstein's exception: for a divisor with an all-ones mantissa a refined reciprocal is not correctly rounded and
    quotients of short numerators (powers of two, small integers, multiples of the divisor) come out one ulp low.
    The shared-reciprocal layer term flags such divisors (the kernels then take the plain operators): over the whole
    family -- every w whose square root has an all-ones mantissa, short and random numerators -- nothing that the
    fast form accepts differs from sqrt and the plain divides."""
    import torch
    from raytracerfortran_amd import raymod
    ws = []
    for k in range(0, 60):
        g = np.nextafter(np.float64(2.0) ** (-k), 0.0)
        w0 = g * g
        for w in (np.nextafter(w0, 0), w0, np.nextafter(w0, 1), np.nextafter(np.nextafter(w0, 1), 1)):
            if np.sqrt(w) == g and 1e-60 < w <= 1.0:
                ws.append(w)
    ws = np.unique(np.array(ws))
    assert ws.size >= 100
    spec = np.concatenate([2.0 ** np.arange(-40, 41), np.arange(1, 200, dtype=np.float64), 1.0 / np.arange(1, 200)])
    gen = torch.Generator(device="cuda").manual_seed(3)
    tot = np.zeros(3, dtype=np.int64)
    for w in ws:
        gq = np.sqrt(w)
        nums = np.concatenate([spec, gq * spec[:81], gq * gq * spec[:81], (gq ** 3) * spec[:81]])
        rnd = torch.exp(40.0 * (torch.rand(1 << 16, dtype=torch.float64, device="cuda", generator=gen) - 0.5))
        nt = torch.cat([torch.tensor(nums, device="cuda"), rnd])
        wt = torch.full((nt.numel(),), float(w), dtype=torch.float64, device="cuda")
        tot += np.array(raymod.selftest_fastmath(wt, nt, nt.flip(0).contiguous()))
    assert tot[2] == 0 and tot[0] == 0, tot


def test_fast_math_equals_ieee_f32(rt):
    """The single-precision guarded layer term returns the bits of sqrtf and the plain divides: for EVERY float w of
    the guarded range (1e-12, 1] with unit numerators (i.e. 1/sqrt(w) and 1/sqrt(w)^3 exhaustively), and for random
    numerators over their guarded range."""
    import torch
    from raytracerfortran_amd import raymod
    lo = int(np.float32(1.0000001e-12).view(np.uint32))
    hi = int(np.float32(1.0).view(np.uint32))
    tot = [0, 0]
    step = 1 << 26
    for start in range(lo, hi + 1, step):
        bits = torch.arange(start, min(start + step, hi + 1), dtype=torch.int64, device="cuda").to(torch.int32)
        w = bits.view(torch.float32)
        one = torch.ones_like(w)
        a, b = raymod.selftest_fastmath_f32(w.contiguous(), one, one)
        tot[0] += a
        tot[1] += b
    assert tot == [0, 0], tot
    g = torch.Generator(device="cuda").manual_seed(11)
    n = 1 << 25
    for trial in range(4):
        u = torch.rand(n, dtype=torch.float32, device="cuda", generator=g)
        e = torch.rand(n, dtype=torch.float32, device="cuda", generator=g)
        if trial == 0:    # solver-like: w = 1 - (pv)^2, numerators H V p and H V
            pv = u * (1.0 - 1e-6)
            w = 1.0 - pv * pv
            den = (50.0 + 3000.0 * e) * (1500.0 + 8500.0 * torch.rand(n, dtype=torch.float32, device="cuda", generator=g))
            num = den * (pv / 6000.0)
        elif trial == 1:  # whole guarded range, log-uniform
            w = torch.exp(-27.0 * u)
            den = torch.exp(82.0 * (e - 0.5))
            num = torch.exp(82.0 * (u - 0.5)) * torch.sign(e - 0.3)
        elif trial == 2:  # zeros and ones among ordinary values
            w = torch.where(u < 0.1, torch.ones_like(u), u * 0.999 + 1e-3)
            den = torch.where(e < 0.1, torch.zeros_like(e), e + 1e-3)
            num = torch.where(u > 0.9, torch.zeros_like(u), u - 0.5)
        else:             # near-critical layers
            w = torch.exp(-20.0 * u) + 2e-12
            den = 1e5 + 1e7 * e
            num = den * 1e-4 * torch.rand(n, dtype=torch.float32, device="cuda", generator=g)
        w = torch.clamp(w, 1.1e-12, 1.0)
        a, b = raymod.selftest_fastmath_f32(w.contiguous(), num.contiguous(), den.contiguous())
        assert (a, b) == (0, 0), (trial, a, b)


def test_fused_kernel_many_layers_and_host_path(rt, O):
    """dff_ takes the fused kernel for NSrc >= 8192; with 300 and 1024 interfaces the sort bins are
    proportional and the LDS share of a wave shrinks (several rounds per wave)."""
    rng = np.random.default_rng(17)
    for nlay, n in ((300, 50000), (1024, 20000)):
        d = np.cumsum(rng.uniform(5, 40, nlay))
        v = rng.uniform(1500, 8000, nlay + 1)
        off, dep = rng.uniform(10, 9000, n), rng.uniform(1, d[-1] + 100, n)
        t = rt.dff(v, d, nlay, off, dep, n)["timeP"]
        assert compare_times(t, O.oracle_dff(v, d, off, dep), TOL) == 0.0


def test_fused_kernel_several_rounds(rt, O):
    """More sources than the resident waves hold in LDS at once: every wave runs several load/solve/time rounds."""
    import torch
    n = 3_000_001
    v, d, off, dep = c2_workload(n, seed=3)
    try:
        rt.set_option("waves_per_simd", 8)  # smallest LDS share per wave -> most rounds
        t = rt.trace_device(torch.tensor(v, device="cuda"), torch.tensor(d, device="cuda"),
                            torch.tensor(off, device="cuda"), torch.tensor(dep, device="cuda")).cpu().numpy()
    finally:
        rt.set_option("waves_per_simd", 4)
    assert compare_times(t, O.oracle_dff(v, d, off, dep), TOL) == 0.0


def test_trace_batch_device_and_loglike(rt, O):
    """Many models over shared sources (what LOGLHOOD_RT does per proposal), with the fused logL."""
    import torch
    rng = np.random.default_rng(31)
    nm, n, ld = 37, 3001, 32
    x, y = rng.uniform(10, 6500, n), rng.uniform(10, 4500, n)
    off, dep = np.sqrt(x * x + y * y), rng.uniform(1050, 4200, n)
    V = np.zeros((nm, ld))
    D = np.zeros((nm, ld))
    NL = np.zeros(nm, np.int32)
    sd = rng.uniform(0.001, 0.07, nm)
    for m in range(nm):
        k = int(rng.integers(2, 31))
        if m == 0:  # k = 1 as LOGLHOOD_RT calls it (loglhood.f90:137-144)
            V[m, :2], D[m, 0], NL[m] = 4321.0, 9999.9, 1
            continue
        D[m, :k - 1] = np.sort(rng.uniform(100.1, 10000.1, k - 1))
        V[m, :k] = rng.uniform(1500, 10000, k)
        NL[m] = k - 1
    true_t = O.oracle_dff(V[5, :NL[5] + 1], D[5, :NL[5]], off, dep)
    obs = true_t + rng.normal(0, 0.016, n)
    dev = "cuda"
    tV, tD, tN = torch.tensor(V, device=dev), torch.tensor(D, device=dev), torch.tensor(NL, device=dev)
    to, tz = torch.tensor(off, device=dev), torch.tensor(dep, device=dev)
    tobs, tsd = torch.tensor(obs, device=dev), torch.tensor(sd, device=dev)
    T, L = rt.trace_batch_device(tV, tD, tN, to, tz, obs=tobs, sd=tsd)
    _, L2 = rt.trace_batch_device(tV, tD, tN, to, tz, obs=tobs, sd=tsd, want_timeP=False)
    torch.cuda.synchronize()
    T, L, L2 = T.cpu().numpy(), L.cpu().numpy(), L2.cpu().numpy()
    assert np.array_equal(L, L2)
    for m in range(nm):
        ref = O.oracle_dff(V[m, :NL[m] + 1], D[m, :NL[m]], off, dep)
        assert compare_times(T[m], ref, TOL) == 0.0
        ll = O.oracle_loglhood_rt(obs, ref, sd[m], stable=True)
        assert abs(L[m] - ll) <= 1e-11 * abs(ll), (m, L[m], ll)


def test_batch_small_leading_dimension_rows(rt, O):
    """Batch of models with 1..11 interfaces in a leading dimension of 12: the launch is small enough for the layer
    tables with one row per stack height, and every model carves its own (ragged) rows."""
    import torch
    rng = np.random.default_rng(77)
    nm, n, ld = 24, 20011, 12
    off, dep = rng.uniform(1, 9000, n), rng.uniform(1, 9000, n)
    V, D, NL = np.zeros((nm, ld)), np.zeros((nm, ld)), np.zeros(nm, np.int32)
    for m in range(nm):
        k = 2 + m % 11  # 2..12 velocities = 1..11 interfaces
        D[m, :k - 1] = np.sort(rng.uniform(50.0, 8000.0, k - 1))
        V[m, :k] = rng.uniform(1500, 9000, k)
        NL[m] = k - 1
    obs = rng.uniform(0.1, 3.0, n)
    sd = np.full(nm, 0.05)
    dev = "cuda"
    args = [torch.tensor(a, device=dev) for a in (V, D, NL, off, dep)]
    try:
        rt.set_option("path", 2)
        T, L = rt.trace_batch_device(*args, obs=torch.tensor(obs, device=dev), sd=torch.tensor(sd, device=dev))
    finally:
        rt.set_option("path", 0)
    torch.cuda.synchronize()
    T, L = T.cpu().numpy(), L.cpu().numpy()
    for m in range(nm):
        ref = O.oracle_dff(V[m, :NL[m] + 1], D[m, :NL[m]], off, dep)
        assert compare_times(T[m], ref, TOL) == 0.0, m
        ll = O.oracle_loglhood_rt(obs, ref, sd[m], stable=True)
        assert abs(L[m] - ll) <= 1e-11 * abs(ll), (m, L[m], ll)


@pytest.mark.parametrize("path", [1, 2])
def test_batch_paths_agree_c3_shape(rt, O, path):
    """BASELINE config 3 shape: 32 models (k = 2..10) over 1e4 shared sources, travel times and fused logL,
    through the plain kernel (path 1) and the fused kernel (path 2)."""
    import torch
    rng = np.random.default_rng(3)
    nm, n, ld = 32, 10000, 16
    x, y = rng.uniform(10, 6500, n), rng.uniform(10, 4500, n)
    off, dep = np.sqrt(x * x + y * y), rng.uniform(1050, 4200, n)
    V, D, NL = np.zeros((nm, ld)), np.zeros((nm, ld)), np.zeros(nm, np.int32)
    for m in range(nm):
        k = int(rng.integers(2, 11))
        D[m, :k - 1] = np.sort(rng.uniform(100.1, 10000.1, k - 1))
        V[m, :k] = rng.uniform(1500, 10000, k)
        NL[m] = k - 1
    sd = rng.uniform(0.001, 0.07, nm)
    obs = O.oracle_dff(V[0, :NL[0] + 1], D[0, :NL[0]], off, dep) + rng.normal(0, 0.016, n)
    dev = "cuda"
    args = [torch.tensor(a, device=dev) for a in (V, D, NL, off, dep)]
    try:
        rt.set_option("path", path)
        T, L = rt.trace_batch_device(*args, obs=torch.tensor(obs, device=dev), sd=torch.tensor(sd, device=dev))
        _, L2 = rt.trace_batch_device(*args, obs=torch.tensor(obs, device=dev), sd=torch.tensor(sd, device=dev),
                                      want_timeP=False)
    finally:
        rt.set_option("path", 0)
    T, L, L2 = T.cpu().numpy(), L.cpu().numpy(), L2.cpu().numpy()
    assert np.array_equal(L, L2)
    for m in range(nm):
        ref = O.oracle_dff(V[m, :NL[m] + 1], D[m, :NL[m]], off, dep)
        assert compare_times(T[m], ref, TOL) == 0.0
        ll = O.oracle_loglhood_rt(obs, ref, sd[m], stable=True)
        assert abs(L[m] - ll) <= 1e-11 * abs(ll)


def test_loglike_test_1_map(rt, kat):
    """logL of the test_1 MAP model through the fused path = the reference's LOGLHOOD_RT value."""
    import torch
    m = kat["model"]
    dev = "cuda"
    V = torch.tensor([m["vels"] + [0.0]], device=dev, dtype=torch.float64)
    D = torch.tensor([m["depths"] + [0.0, 0.0]], device=dev, dtype=torch.float64)
    N = torch.tensor([6], device=dev, dtype=torch.int32)
    g = kat["test_1_map_logL_survey"]
    f64 = torch.float64
    T, L = rt.trace_batch_device(V, D, N, torch.tensor(kat["test_1_src_offset"], device=dev, dtype=f64),
                                 torch.tensor(kat["test_1_src_depth"], device=dev, dtype=f64),
                                 obs=torch.tensor(kat["test_1_obs_RT"], device=dev, dtype=f64),
                                 sd=torch.tensor([g["sd"]], device=dev, dtype=torch.float64))
    assert abs(float(L[0]) - g["value"]) < 1e-10 * abs(g["value"])
    assert compare_times(T[0].cpu().numpy(), kat["config1_ref_timeP"], TOL) == 0.0


def test_loglhood_batch_host_entry(rt, O, kat):
    """rtmi_set_sources + rtmi_loglhood_batch: the likelihood as the sampler calls it (READDATA once, LOGLHOOD per
    proposal; loglhood.f90:35-207), host arrays in and out."""
    m = kat["model"]
    off, dep, obs = kat["test_1_src_offset"], kat["test_1_src_depth"], kat["test_1_obs_RT"]
    rt.set_sources(off, dep, obs)
    V = np.zeros((3, 16)); D = np.zeros((3, 16)); NL = np.array([6, 1, 2], np.int32)
    V[0, :7], D[0, :6] = m["vels"], m["depths"]
    V[1, :2], D[1, 0] = 4321.0, 9999.9                       # k = 1 (loglhood.f90:137-144)
    V[2, :3], D[2, :2] = [3000.0, 4500.0, 6000.0], [1200.0, 3000.0]
    sd = np.array([0.02, 0.05, 0.03])
    logl, ss = rt.loglhood_batch(V, D, NL, sd, want_ss=True)
    assert abs(logl[0] - kat["test_1_map_logL_survey"]["value"]) < 1e-10 * 55.0
    for i in range(3):
        t = O.oracle_dff(V[i, :NL[i] + 1], D[i, :NL[i]], off, dep)
        assert abs(logl[i] - O.oracle_loglhood_rt(obs, t, sd[i], stable=True)) <= 1e-11 * abs(logl[i])
        assert abs(ss[i] - np.sum((np.array(obs) - t) ** 2)) <= 1e-12 * ss[i]
    assert np.array_equal(rt.loglhood_batch(V, D, NL, sd), logl)   # deterministic
    # a large resident set takes the fused kernel
    v, d, off2, dep2 = c2_workload(50000, seed=21)
    obs2 = O.oracle_dff(v, d, off2, dep2) + 0.01
    obs2[obs2 < -900] = 1.0
    rt.set_sources(off2, dep2, obs2)
    V = np.zeros((2, 16)); D = np.zeros((2, 16)); NL = np.array([8, 8], np.int32)
    V[:, :9], D[:, :8] = v, d
    V[1, 4] += 50.0
    l2 = rt.loglhood_batch(V, D, NL, np.array([0.02, 0.02]))
    for i in range(2):
        t = O.oracle_dff(V[i, :9], D[i, :8], off2, dep2)
        assert abs(l2[i] - O.oracle_loglhood_rt(obs2, t, 0.02, stable=True)) <= 1e-11 * abs(l2[i])
    with pytest.raises(rt.RtmiError):
        rt.loglhood_batch(V, D, np.array([8, 99], np.int32), np.array([0.02, 0.02]))


def c5_workload(n, seed=2025):
    """20 interfaces, 21 sorted velocities."""
    rng = np.random.default_rng(seed)
    d = np.cumsum(rng.uniform(200, 600, 20))
    v = np.sort(rng.uniform(1500, 6500, 21))
    return v, d, rng.uniform(10, 8000, n), rng.uniform(50, d[-1] + 500, n)


@pytest.mark.parametrize("path", [1, 2])
def test_fp32_path_config5(rt, O, path):
    """BASELINE config 5 (fp32 kernel path, 20 interfaces): the single-precision kernels equal the oracle built in
    single precision bit for bit (same operation order), and the deviation from the fp64 result is reported."""
    import torch
    n = 1_000_000 if path == 2 else 50_000
    v, d, off, dep = c5_workload(n)
    f32 = torch.float32
    a = [torch.tensor(x.astype(np.float32), device="cuda") for x in (v, d, off, dep)]
    p = torch.empty(n, dtype=f32, device="cuda")
    try:
        rt.set_option("path", path)
        t32 = rt.trace_device(*a, p_final=p).cpu().numpy()
    finally:
        rt.set_option("path", 0)
    assert t32.dtype == np.float32
    ref32, pref, nl = O.oracle_dff_f32(v, d, off, dep, want_p=True)
    assert np.array_equal(t32, ref32), f"{(t32 != ref32).sum()} of {n} differ from the fp32 oracle"
    pp = p.cpu().numpy()
    assert np.array_equal(pp[nl > 1], pref[nl > 1], equal_nan=True)
    # delta-t vs fp64 on the same (float-rounded) inputs
    t64 = O.oracle_dff(v.astype(np.float32), d.astype(np.float32), off.astype(np.float32), dep.astype(np.float32))
    both = (t32 != -999.0) & (t64 != -999.0)
    dt = np.abs(t32[both].astype(np.float64) - t64[both])
    print(f"fp32 vs fp64, {n} sources x 20 interfaces: max |dt| = {dt.max():.3e} s, 99.9 pct = "
          f"{np.quantile(dt, 0.999):.3e} s, median = {np.median(dt):.3e} s; -999: fp32 {int((t32 == -999).sum())}, "
          f"fp64 {int((t64 == -999).sum())}")
    assert np.median(dt) < 1e-5 and np.quantile(dt, 0.999) < 5e-2


def test_device_entry_points_are_graph_capturable(rt, O):
    """rtmi_trace_device / rtmi_trace_batch_device allocate nothing and never synchronise: a HIP graph captured
    around them replays to the same bits (launch-bound inner loops of a sampler can be replayed as graphs)."""
    import torch
    n = 200_000
    v, d, off, dep = c2_workload(n, seed=9)
    dv, dd, do, dz = (torch.tensor(x, device="cuda") for x in (v, d, off, dep))
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    rt.trace_device(dv, dd, do, dz, timeP=out)  # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rt.trace_device(dv, dd, do, dz, timeP=out)
    out.zero_()
    do.copy_(torch.tensor(off[::-1].copy(), device="cuda"))  # new inputs in the captured buffers
    dz.copy_(torch.tensor(dep[::-1].copy(), device="cuda"))
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), O.oracle_dff(v, d, off[::-1], dep[::-1]))


def test_many_layers(rt, O):
    """Up to the 1024-interface limit of the kernels (the reference's sampler copy stops at 99)."""
    rng = np.random.default_rng(8)
    for nl in (99, 100, 400, 1024):
        d = np.cumsum(rng.uniform(5, 30, nl))
        v = rng.uniform(1500, 8000, nl + 1)
        off, dep = rng.uniform(10, 9000, 700), rng.uniform(1, d[-1] + 100, 700)
        t = rt.dff(v, d, nl, off, dep, 700)["timeP"]
        assert compare_times(t, O.oracle_dff(v, d, off, dep), TOL) == 0.0
    with pytest.raises(rt.RtmiError):
        rt.dff(np.ones(1026) * 3000, np.arange(1, 1026.0), 1025, [1.0], [1.0], 1)


def test_plain_operators_everywhere(rt, O):
    """Lengths scaled by 2^110 (2^40 in single precision) put every H, H*V, offset and depth outside the guards of the
    fast sqrt/divide forms, so every evaluation, start value and travel time of the run goes through the plain
    operators -- the path a wave otherwise takes only for a stray operand.  Same bits as the oracle on the same
    scaled inputs, plain and fused kernel, fp64 and fp32."""
    import torch
    rng = np.random.default_rng(2468)
    cases = [c2_workload(120_000, seed=5)]
    for k in (3, 12, 40):
        d = np.cumsum(rng.uniform(5.0, 700.0, k - 1))
        cases.append((rng.uniform(400, 9000, k), d, rng.uniform(0, 2.5 * d[-1], 60_000), rng.uniform(0.5, 1.1 * d[-1], 60_000)))
    for v, d, off, dep in cases:
        for dt, scale in ((torch.float64, 2.0 ** 110), (torch.float32, 2.0 ** 40)):
            ds, offs, deps = d * scale, off * scale, dep * scale
            want = O.oracle_dff(v, ds, offs, deps) if dt == torch.float64 else O.oracle_dff_f32(v, ds, offs, deps)
            assert np.isfinite(want[want != -999.0]).all()
            for path in (1, 2):
                a = [torch.tensor(x, device="cuda", dtype=dt) for x in (v, ds, offs, deps)]
                try:
                    rt.set_option("path", path)
                    t = rt.trace_device(*a).cpu().numpy()
                finally:
                    rt.set_option("path", 0)
                assert np.array_equal(t, want, equal_nan=True), (path, dt, int((t != want).sum()))


def test_unsorted_and_tied_interfaces(rt, O):
    """whichLayer is the reference's scan ("first interface below z"); the kernels search by bisection only when the
    interfaces are in non-decreasing order.  Tied interfaces (zero-thickness layers) keep the bisection, interfaces
    out of order take the scan: both return the oracle's bits, through the plain and the fused kernel."""
    import torch
    rng = np.random.default_rng(4242)
    for case in range(6):
        k = int(rng.integers(4, 14))
        d = np.sort(rng.uniform(100.0, 6000.0, k - 1))
        if case % 2 == 0:
            d[rng.integers(1, k - 1)] = d[0]            # out of order (a shallow interface listed late)
            if case == 4:
                d = d[::-1].copy()                      # fully reversed
        else:
            j = int(rng.integers(1, k - 1))
            d[j] = d[j - 1]                              # a tie: zero-thickness layer
        v = rng.uniform(1500, 8000, k)
        n = 30011
        off, dep = rng.uniform(0.0, 9000.0, n), rng.uniform(0.5, 7000.0, n)
        dep[:50] = rng.choice(d, 50)                     # sources exactly on interfaces
        ref = O.oracle_dff(v, d, off, dep)
        for path in (1, 2):
            try:
                rt.set_option("path", path)
                t = rt.trace_device(torch.tensor(v, device="cuda"), torch.tensor(d, device="cuda"),
                                    torch.tensor(off, device="cuda"), torch.tensor(dep, device="cuda")).cpu().numpy()
            finally:
                rt.set_option("path", 0)
            assert np.array_equal(t, ref, equal_nan=True), (case, path, int((t != ref).sum()))


def test_fused_kernel_random_model_soak(rt, O):
    """Random models of every shape the sampler's prior can draw and a few it cannot (monotone, low-velocity zones,
    tied maxima, thin and thick layers, 1..60 interfaces, velocities over three decades), source sets from 9 000 to
    150 000, shallow and deep, near and far: the fused kernel -- whose solve order comes from a per-model estimate
    of the evaluation count -- returns the oracle's bits for every source, final ray parameters included."""
    import torch
    rng = np.random.default_rng(987654321)
    nsrc_total = 0
    for it in range(48):
        k = int(rng.integers(2, 62))
        kind = it % 6
        if kind == 0:      # monotone increasing
            v = np.sort(rng.uniform(1500, 8000, k))
        elif kind == 1:    # prior draw
            v = rng.uniform(1500, 10000, k)
        elif kind == 2:    # tied maxima and a low-velocity zone
            v = rng.uniform(2000, 6000, k)
            v[rng.integers(0, k, 3)] = 6000.0
        elif kind == 3:    # slow sediments over a fast basement
            v = np.concatenate([rng.uniform(300, 900, k - 1), [7000.0]])
        elif kind == 4:    # velocities over three decades
            v = np.exp(rng.uniform(np.log(50.0), np.log(50000.0), k))
        else:              # decreasing with depth
            v = np.sort(rng.uniform(1500, 8000, k))[::-1].copy()
        thick = rng.uniform(1.0, 900.0, k - 1) if it % 2 else np.exp(rng.uniform(np.log(0.5), np.log(3000.0), k - 1))
        d = np.cumsum(thick)
        n = int(rng.integers(9000, 150000))
        off = np.where(rng.random(n) < 0.5, rng.uniform(0.0, 3.0 * d[-1] + 100.0, n), np.exp(rng.uniform(-3, 11, n)))
        dep = np.where(rng.random(n) < 0.7, rng.uniform(0.01, d[-1] * 1.2 + 10.0, n), rng.choice(d, n) + rng.choice([0.0, 1e-9, -1e-9, 0.5], n))
        dep = np.abs(dep) + 1e-6
        p = torch.empty(n, dtype=torch.float64, device="cuda")
        try:
            rt.set_option("path", 2)
            t = rt.trace_device(torch.tensor(v, device="cuda"), torch.tensor(d, device="cuda"),
                                torch.tensor(off, device="cuda"), torch.tensor(dep, device="cuda"), p_final=p).cpu().numpy()
        finally:
            rt.set_option("path", 0)
        ref, pref, nl = O.oracle_dff(v, d, off, dep, want_p=True)
        assert np.array_equal(t, ref, equal_nan=True), (it, kind, k, n, int((t != ref).sum()))
        good = nl > 1
        assert np.array_equal(p.cpu().numpy()[good], pref[good], equal_nan=True), (it, kind, k, n)
        nsrc_total += n
    assert nsrc_total > 2_000_000
