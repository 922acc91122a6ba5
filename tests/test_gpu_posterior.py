"""What the device sampler SAMPLES ("posterior of k, sigma, profiles"), asserted.

The other sampler tests pin bookkeeping (logL of every chain equals the oracle's, engines and speculation bit-identical, k follows
the prior at beta = 0); none of them would notice a wrong Hastings ratio in birth / death, a mis-scaled Cauchy step that breaks
symmetry, or a temperature applied to the wrong factor.  Here the chains' stationary law itself is checked:
 (a) against exact posteriors of a small problem, computed by quadrature on the ORACLE's likelihood
     (tests/golden/make_posterior_golden.py -> tests/golden/posterior_small.json): the marginals of a fixed-dimension k = 2 model
     at beta = 1 and beta = 0.5 (prjmh_temper_rf.f90:721-759, :742-751);
 (b) the trans-dimensional odds P(k = 2) : P(k = 1) against the evidence ratio x Poisson prior (:669-713, :697-706, BIRTH_FULL
     :997-1095, DEATH_FULL :917-995), beta = 1 and 0.5;
 (c) BASELINE config 3's problem, shortened: sigma and the velocity profile of the generating model are recovered.
All runs use the library's defaults (speculative step, incremental likelihood, engine of the handle's choice) with the fused
kernel path forced for the 8-source problems, so that what is tested is the code that runs at full size.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from raytracerfortran_amd import tempering as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    with open(os.path.join(GOLDEN, "posterior_small.json")) as f:
        return json.load(f)


def _chains(small, kmin, kmax, beta, nch, seed, start_nodes):
    import raytracerfortran_amd as rt
    from raytracerfortran_amd import raymod
    rt.load_library(); rt.init(0)
    s = small["sources"]
    rt.set_sources(np.array(s["off"]), np.array(s["dep"]), np.array(s["obs"]))
    p = small["prior"]
    pr = T.Prior(kmin=kmin, kmax=kmax, lam=p["lam"], hmx=p["hmx"], hmin=p["hmin"], vmin=p["vmin"], vmax=p["vmax"],
                 sdmin=p["sdmin"], sdmax=p["sdmax"], pertsdsc=p["pertsdsc"], pertsdsdsc=p["pertsdsdsc"])
    h = raymod.NativeChains(nch, 4, len(s["off"]), pr, seed=seed)
    for i in range(nch):
        h.set_chain(i, start_nodes, small["sd"], beta)
    h.evaluate()
    return h


def _ks(samples, grid_mid, cdf):
    """sup |F_n - F| with F given on the upper edges of the quadrature cells."""
    x = np.sort(np.asarray(samples))
    edges = np.asarray(grid_mid) + 0.5 * (grid_mid[1] - grid_mid[0])
    fn = np.searchsorted(x, edges, side="right") / x.size
    return float(np.abs(fn - np.asarray(cdf)).max())


@pytest.mark.parametrize("beta", [1.0, 0.5])
def test_fixed_dimension_marginals_match_quadrature(small, beta):
    """k = 2 fixed (kmin = kmax = 2), sigma fixed: three parameters (v1, z2, v2).  64 chains x 6000 sweeps at the given beta,
    the first 1000 dropped, every 5th kept: the empirical marginals must lie within sup-distance 0.02 of the quadrature CDFs
    (64 000 correlated samples; the statistic of a correct sampler is ~0.006, a target tempered with the wrong exponent or a
    proposal that is not symmetric moves it past 0.05)."""
    import raytracerfortran_amd as rt
    rt.set_option("path", 2)
    try:
        nch = 64
        h = _chains(small, 2, 2, beta, nch, seed=31, start_nodes=np.array([[0.0, 4200.0], [3000.0, 4500.0]]))
        h.sweep(1000)
        zs, v1s, v2s = [], [], []
        for _ in range(1000):
            h.sweep(5)
            a = h.get_all()
            assert np.all(a["k"] == 2)
            zs.append(a["z"][:, 1].copy()); v1s.append(a["v"][:, 0].copy()); v2s.append(a["v"][:, 1].copy())
        info = h.speculation()
        assert info["rows"] > 1 and info["sorted_sources"]          # the code under test is the speculative step
        st = h.stats()
        assert 0.05 < st["acc_mh"] / st["prop_mh"] < 0.9
        q = small["fixed_k2"][str(beta)]
        d = {"z": _ks(np.concatenate(zs), small["z_mid"], q["cdf_z"]),
             "v1": _ks(np.concatenate(v1s), small["v_mid"], q["cdf_v1"]),
             "v2": _ks(np.concatenate(v2s), small["v_mid"], q["cdf_v2"])}
        print("KS distances beta=%g:" % beta, d, "acceptance", st["acc_mh"] / st["prop_mh"])
        assert max(d.values()) < 0.02, d
        assert rt.load_library().rtmi_device_status() == 0
    finally:
        rt.set_option("path", 0)


@pytest.mark.parametrize("beta", [1.0, 0.5])
def test_trans_dimensional_odds_match_the_evidence_ratio(small, beta):
    """k in {1, 2} with the likelihood ON: the fraction of time the chains spend at k = 2 must equal
    pk(2) Z2 / (pk(1) Z1 + pk(2) Z2) from the quadrature.  64 chains x 8000 sweeps (first 1000 dropped); tolerance = five
    standard errors of the mean over chains + 0.005.  This is the test a wrong birth / death acceptance ratio fails."""
    import raytracerfortran_amd as rt
    rt.set_option("path", 2)
    try:
        nch = 64
        h = _chains(small, 1, 2, beta, nch, seed=77, start_nodes=np.array([[0.0, 4200.0]]))
        h.sweep(1000)
        at2 = np.zeros(nch)
        nrec = 1400
        for _ in range(nrec):
            h.sweep(5)
            at2 += h.get_all()["k"] == 2
        f = at2 / nrec
        want = small["transd"][str(beta)]["p_k2"]
        se = f.std(ddof=1) / np.sqrt(nch)
        st = h.stats()
        print("P(k=2): sampled %.4f +- %.4f, quadrature %.4f; birth/death acceptance %.3f" % (f.mean(), se, want, st["acc_bd"] / st["prop_bd"]))
        assert st["acc_bd"] > 1000
        assert abs(f.mean() - want) < 5.0 * se + 0.005, (f.mean(), se, want)
        assert rt.load_library().rtmi_device_status() == 0
    finally:
        rt.set_option("path", 0)


def test_config3_recovers_sigma_and_the_velocity_profile():
    """BASELINE config 3's problem (32 tempered chains, 1e4 sources of the generating 7-layer model + N(0, 0.016 s)), 6000
    sweeps from a 2-layer start (the full-size run of profiles/config3_run.py takes 10 000): the T = 1 chain's sigma within
    5 % of 0.016 s; its velocity in the middle of every generating layer that the sources sample from top to bottom
    (1500 ... 3960 m; the sources lie between 1050 and 4200 m) within 4 posterior standard deviations + 1 % of the truth; and the
    vertical travel time through the 1500 m above them -- all the data can say about those two layers: their velocities and
    the interface between them trade off against each other -- within 0.5 %."""
    import torch
    import raytracerfortran_amd as rt
    from raytracerfortran_amd import synthetic as S
    rt.load_library(); rt.init(0)
    nsrc = 10_000
    rng = np.random.default_rng(2024)
    x, y = rng.uniform(10, 6500, nsrc), rng.uniform(10, 4500, nsrc)
    off, dep = np.sqrt(x * x + y * y), rng.uniform(1050, 4200, nsrc)
    f64 = torch.float64
    tt = rt.trace_device(torch.tensor(S.TRUE_V, device="cuda", dtype=f64), torch.tensor(S.TRUE_D, device="cuda", dtype=f64),
                         torch.tensor(off, device="cuda"), torch.tensor(dep, device="cuda")).cpu().numpy()
    obs = tt + rng.normal(0, 0.016, nsrc)
    ev = T.GpuEvaluator(off, dep, obs)
    s = T.NativeSampler(T.Prior(), ev, ndata=nsrc, nchains_total=32, nptchains1=1, dTlog=1.15, seed=11)
    s.init_from_model(np.array([[0.0, 3500.0], [2500.0, 5500.0]]), 0.05)
    samples = s.run(6000, burnin=4000, thin=4)
    assert len(samples) == 500
    sds = np.array([x[2] for x in samples])
    assert abs(sds.mean() - 0.016) < 0.05 * 0.016, sds.mean()
    zq = np.array([1750.0, 2250.0, 2900.0, 3600.0])            # the middle of the generating layers 3 ... 6
    vtrue = np.array(S.TRUE_V)[2:6]
    vq = np.array([nodes[np.searchsorted(nodes[:, 0], zq, side="right") - 1, 1] for _, nodes, _, _ in samples])
    err = np.abs(vq.mean(0) - vtrue)
    tol = 4.0 * vq.std(0) + 0.01 * vtrue
    def t_top(nodes, zmax=1500.0):                             # vertical travel time from the surface to zmax
        z = np.append(nodes[:, 0], np.inf)
        return sum((min(z[i + 1], zmax) - z[i]) / nodes[i, 1] for i in range(nodes.shape[0]) if z[i] < zmax)
    tts = np.array([t_top(nodes) for _, nodes, _, _ in samples])
    tt_true = 1000.0 / 3100.0 + 500.0 / 2400.0
    print("sigma %.5f; v mean %s std %s; vertical time to 1500 m %.5f +- %.5f (true %.5f)"
          % (sds.mean(), vq.mean(0).round(1), vq.std(0).round(1), tts.mean(), tts.std(), tt_true))
    assert np.all(err < tol), (err, tol)
    assert abs(tts.mean() - tt_true) < 0.005 * tt_true, (tts.mean(), tt_true)
    true_ll = T.gaussian_loglike(float(np.sum((obs - tt) ** 2)), nsrc, 0.016)
    assert np.median([x[3] for x in samples]) > true_ll - 40.0
