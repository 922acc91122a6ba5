"""Pins the CPU oracle (oracle/dff_oracle.c) -- runs on CPU, no GPU needed.

The oracle is the checker for every GPU parity test, so before it is trusted it
is checked here against
  * the reference AUTHOR's published gfortran outputs (KAT-A notebook print,
    KAT-B rays.dat), and
  * fixtures produced by running the reference source itself
    (tests/golden/make_golden.py; /root/reference/subroutineR-quiet.f90
    compiled unmodified with amdflang), and
  * where oracle/_ref was built (the build container), the live reference on
    fresh random inputs.
"""
import numpy as np
import pytest

from conftest import compare_times, golden_case_names
from oracle import oracle as O


def test_kat_a_notebook_print(kat):
    """raytracerR-export-data-to-MCMC.nb.html: print(timeP$timeP) for NSrc=5, 7 printed digits."""
    m = kat["model"]
    a = kat["kat_a"]
    t = O.oracle_dff(m["vels"], m["depths"], a["src_offset"], a["src_depth"])
    printed = np.array(a["printed_timeP"])
    assert np.all(np.abs(t - printed) < 0.5e-7)
    assert [f"{x:.7f}" for x in t] == [f"{x:.7f}" for x in printed]
    compare_times(t, kat["kat_a_ref_timeP"], tol=1e-12)


def test_kat_b_rays_dat(kat):
    """/root/reference/rays.dat: the author's keep_delta=10 dump of the 20 test_1 rays
    (subroutineR-quiet.f90:157-164): per ray one record of d_i, one of H_i."""
    m = kat["model"]
    off, dep = kat["test_1_src_offset"], kat["test_1_src_depth"]
    t, p, nl = O.oracle_dff(m["vels"], m["depths"], off, dep, want_p=True)
    ours = []
    for k in range(len(off)):
        d, h = O.oracle_ray_segments(m["vels"], m["depths"], off[k], dep[k], p[k])
        assert d.size == nl[k]
        ours += list(d) + list(h)
    ref = np.array(kat["kat_b"]["rays_dat_tokens"])
    ours = np.array(ours)
    assert ours.size == ref.size == 152
    rel = np.abs(ours - ref) / np.maximum(np.abs(ref), 1e-300)
    assert rel.max() < 5e-13, rel.max()
    # the reference stops at |f| < 0.1 m, not at the root: the horizontal miss is visible in its own dump
    pos = 0
    for k in range(len(off)):
        miss = abs(ref[pos:pos + nl[k]].sum() - off[k])
        assert miss < 0.1
        pos += 2 * nl[k]


def test_config1_test_1(kat):
    """BASELINE config 1: dff on test_1_src_data.txt with the test_1 model."""
    m = kat["model"]
    t = O.oracle_dff(m["vels"], m["depths"], kat["test_1_src_offset"], kat["test_1_src_depth"])
    err = compare_times(t, kat["config1_ref_timeP"], tol=1e-12)
    assert err == 0.0  # bit-exact against flang -O2 (plain left-to-right SUM, as gfortran)
    compare_times(t, kat["config1_ref_timeP_O0"], tol=1e-12)  # flang -O0 uses a compensated SUM()
    # KAT-C (statistical): the committed observations are forward + N(0, 0.016 s)
    res = np.array(kat["test_1_obs_RT"]) - t
    assert abs(res.std() - 0.016) < 0.006 and np.abs(res).max() < 0.05


def test_raytracer_r_example(kat):
    """rayTracerR.R:23-33 on eqdf.csv."""
    e = kat["rayTracerR_example"]
    t = O.oracle_dff(e["vels"], e["depths"], e["src_offset"], e["src_depth"])
    assert compare_times(t, e["ref_timeP"], tol=1e-12) == 0.0


@pytest.mark.parametrize("name", golden_case_names())
def test_reference_fixture(name, golden_batches):
    c = golden_batches[name]
    t = O.oracle_dff(c["vels"], c["depths"], c["off"], c["dep"])
    err = compare_times(t, c["timeP"], tol=1e-12)
    assert err == 0.0, f"{name}: oracle not bit-exact with the compiled reference ({err:.3e})"


def test_loglhood_rt_golden(kat):
    """loglhood.f90:165-199 on the test_1 MAP model (sd = 0.02): SURVEY.md section 4 value."""
    m = kat["model"]
    t = O.oracle_dff(m["vels"], m["depths"], kat["test_1_src_offset"], kat["test_1_src_depth"])
    g = kat["test_1_map_logL_survey"]
    ll = O.oracle_loglhood_rt(kat["test_1_obs_RT"], t, g["sd"])
    assert abs(ll - g["value"]) < 1e-10 * abs(g["value"])
    ll2 = O.oracle_loglhood_rt(kat["test_1_obs_RT"], t, g["sd"], stable=True)
    assert abs(ll2 - ll) < 1e-12 * abs(ll)


def test_branch_statistics_consistent(kat, golden_index):
    """The oracle's branch counters add up (every source takes exactly one of the four paths) and the
    committed index carries them for every fixture (used by DESIGN.md's flop estimate).  The table in
    SURVEY.md 8(a) came from a differently instrumented copy and is NOT reproduced: the bit-exact
    travel times above are the pin, not these counters."""
    m = kat["model"]
    t, st = O.oracle_dff_stats(m["vels"], m["depths"], kat["test_1_src_offset"], kat["test_1_src_depth"])
    assert st["top_layer"] + st["cf_negative"] + st["safe_jump"] + st["bisect"] == len(t)
    assert st["not_conv"] == int((t == -999.0).sum())
    assert st["fp_evals"] <= st["f_evals"]
    for c in golden_index:
        s = c["oracle_branch_stats"]
        assert s["top_layer"] + s["cf_negative"] + s["safe_jump"] + s["bisect"] == c["nsrc"]
        assert s["not_conv"] == c["n_sentinel"]


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built here (needs /root/reference + amdflang)")
def test_live_reference_random_models():
    """Fresh random models against the live compiled reference (build container only)."""
    rng = np.random.default_rng(20261004)
    worst = 0.0
    for it in range(40):
        k = int(rng.integers(2, 31))
        d = np.sort(rng.uniform(100.1, 10000.1, k - 1))
        v = rng.uniform(1500, 10000, k)
        off = rng.uniform(0, 12000, 3000)
        dep = rng.uniform(1, 11000, 3000)
        t = O.oracle_dff(v, d, off, dep)
        r = O.ref_dff(v, d, off, dep)
        worst = max(worst, compare_times(t, r, tol=1e-12))
    assert worst == 0.0


def test_fp32_oracle_tracks_fp64():
    """The single-precision build of the restatement (checker of the fp32 kernel path, BASELINE config 5) follows
    the double-precision one to single-precision accuracy on typical rays; the tail (near-critical rays, different
    iteration paths through the |f| < 0.1 m exit) is reported by the GPU test."""
    rng = np.random.default_rng(2025)
    d = np.cumsum(rng.uniform(200, 600, 20))
    v = np.sort(rng.uniform(1500, 6500, 21))
    off, dep = rng.uniform(10, 8000, 20000), rng.uniform(50, d[-1] + 500, 20000)
    t32 = O.oracle_dff_f32(v, d, off, dep)
    t64 = O.oracle_dff(v.astype(np.float32), d.astype(np.float32), off.astype(np.float32), dep.astype(np.float32))
    assert t32.dtype == np.float32
    both = (t32 != -999.0) & (t64 != -999.0)
    dt = np.abs(t32[both].astype(np.float64) - t64[both])
    assert both.mean() > 0.99 and np.median(dt) < 1e-5


def c4_fixture():
    """tests/golden/c4_models.npz + the seeded C4 workload it was generated from (digest-checked)."""
    import os
    from conftest import GOLDEN
    from raytracerfortran_amd import synthetic as S
    z = np.load(os.path.join(GOLDEN, "c4_models.npz"), allow_pickle=False)
    w = S.c4_workload(32)
    assert bytes(z["digest_sources"]).hex() == S.digest(w["off"], w["dep"]), "C4 source recipe drifted from the fixture"
    return z, w


@pytest.mark.parametrize("name", ["k30", "k2"])
def test_config4_full_size_fixture(name):
    """BASELINE config 4's source count (1e5) on its largest (k = 30) and smallest (k = 2) model: the oracle against
    travel times of the compiled reference."""
    z, w = c4_fixture()
    t = O.oracle_dff(z[f"{name}__vels"], z[f"{name}__depths"], w["off"], w["dep"])
    assert compare_times(t, z[f"{name}__timeP"], tol=1e-12) == 0.0


def test_pairs_where_the_fast_fp32_path_and_the_reference_differ():
    """tests/golden/f32fast_pairs.json (made by make_f32fast_golden.py from oracle/_ref = the reference itself): the oracle gives the
    reference's answer on every pair, bit for bit -- the -999 of GetPTime (subroutineR-quiet.f90:112-173), the grossly wrong iterate
    solve accepts through its inverted test (:327-330) -- and the converged solution of costFunc = 0 (:184-202) is the other column.
    The GPU side of the fixture is tests/test_gpu_f32fast.py::test_fast_path_on_the_pinned_pairs."""
    import json
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from f32fast_ref import converged_times
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "f32fast_pairs.json")))
    seen = {"sentinel": 0, "gross": 0, "agree": 0}
    for c in fx["cases"]:
        v, d, off, dep = (np.array(c[k], np.float64) for k in ("vels", "depths", "src_offset", "src_depth"))
        for a in (v, d, off, dep):
            assert (a.astype(np.float32).astype(np.float64) == a).all()          # what the kernel is handed is exactly this
        ref, conv = np.array(c["t_reference"]), np.array(c["t_converged"])
        got = O.oracle_dff(v, d, off, dep)
        assert (got == ref).all(), (c["kind"], got, ref)
        if O.have_ref():
            assert (O.ref_dff(v, d, off, dep) == ref).all()
        tc, _ = converged_times(v, d, off, dep)
        assert np.abs(tc - conv).max() < 1e-12
        if c["kind"] == "sentinel":
            assert (ref == -999.0).all() and np.isfinite(conv).all() and (conv > 0).all()
        elif c["kind"] == "gross":
            assert (ref != -999.0).all() and (np.abs(ref - conv) > 1e-3).all()
        else:
            assert (np.abs(ref - conv) < 6.7e-5).all()                          # the reference's stopping tolerance: 0.1 m / 1500 m/s
        seen[c["kind"]] += ref.size
    assert seen == {"sentinel": 24, "gross": 24, "agree": 16}
