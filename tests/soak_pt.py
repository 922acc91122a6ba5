#!/usr/bin/env python3
"""tests/soak_pt.py [ncases] -- the sampler's sweep engines and its incremental likelihood against the plain form on random
shapes (not collected by pytest; run on the GPU box): chains x sources x kmax x seed drawn at random, sweeps with
exchanges and recorded samples; every chain / statistic / ladder / sample bit-identical between
  per-slot launches, every proposal evaluated over all sources      (RTMI_PT_PERSISTENT=0 RTMI_PT_INCREMENTAL=0)
  per-slot launches, unaffected sources taken from the residual cache (0, 1)
  the persistent launch with the cache                                 (1, 1)
and the logL of sampled final states equal to the oracle's (1e-9 relative).  Every other case has ONE CHAIN PER LEVEL (a random
permutation): there the persistent engine puts the whole run into multi-sweep launches with the exchange inside the kernel
(csrc/rtmi_kernels.hip::prop_exchange), in two rtmi_pt_run calls with different exchange and record intervals."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracerfortran_amd as rt
from raytracerfortran_amd import raymod, synthetic as S, tempering as T
from oracle import oracle as O

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rt.load_library(); rt.init(0)
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "99")))  # SOAK_SEED: other shapes than the committed runs saw
ran_persistent = 0
for case in range(ncases):
    nch = int(rng.choice([8, 16, 32, 40, 64]))
    n = int(rng.choice([5000, 10_000, 20_000, 33_333]))
    kmax = int(rng.choice([6, 10, 16, 30]))
    seed = int(rng.integers(1, 1 << 30))
    sweeps = int(rng.integers(9, 30))
    off, dep = S.sampler_sources(n, int(rng.integers(1, 1000)))
    obs = O.oracle_dff(S.TRUE_V, S.TRUE_D, off, dep) + rng.normal(0, 0.016, n)
    rt.set_sources(off, dep, obs)
    nodes = np.column_stack([[0.0] + list(S.TRUE_D), S.TRUE_V])[: min(7, kmax)]
    pr = T.Prior(kmax=kmax)
    perm = case % 2 == 1
    level_perm = rng.permutation(nch)
    swap2 = int(rng.choice([1, 2]))
    res = {}
    for engine, incr in (("0", "0"), ("0", "1"), ("1", "1")):
        os.environ["RTMI_PT_PERSISTENT"] = engine
        os.environ["RTMI_PT_INCREMENTAL"] = incr
        h = raymod.NativeChains(nch, kmax + 1, n, pr, seed=seed)
        for i in range(nch):
            h.set_chain(i, nodes, 0.016, 1.0)
        h.evaluate()
        nlev = nch if perm else max(2, nch // 2)
        h.set_ladder(nch, 0, level_perm if perm else np.arange(nch) % nlev, 1.15 ** -np.arange(float(nlev)))
        h.sample_buffer(8192)
        h.run(sweeps, 1, None, 3)
        if perm:
            h.run(7, swap2, None, 2)
        res[engine + incr] = (h.get_all(), h.stats(), h.get_ladder()[0], h.samples())
        del h
    (a0, s0, l0, m0) = res["00"]
    for other in ("01", "11"):
        (a1, s1, l1, m1) = res[other]
        live = np.arange(a0["z"].shape[1])[None, :] < a0["k"][:, None]
        ok = all(np.array_equal(a0[k], a1[k]) for k in ("k", "sd", "logL", "ss", "beta"))
        ok = ok and np.array_equal(a0["z"][live], a1["z"][live]) and np.array_equal(a0["v"][live], a1["v"][live])
        ok = ok and np.array_equal(l0, l1) and {k: v for k, v in s0.items() if k != "launches"} == {k: v for k, v in s1.items() if k != "launches"}
        ok = ok and all(np.array_equal(m0[k], m1[k]) for k in ("sweep", "chain", "k", "sd", "logL")) and m0["dropped"] == m1["dropped"]
        assert ok, ("variants differ", other, case, nch, n, kmax, seed)
    ran_persistent += s1["launches"] < s0["launches"]
    for i in rng.choice(nch, 3, replace=False):
        k = int(a1["k"][i])
        v, d = T.layering(np.column_stack([a1["z"][i, :k], a1["v"][i, :k]]))
        ref = O.oracle_loglhood_rt(obs, O.oracle_dff(v, d, off, dep), a1["sd"][i], stable=True)
        assert abs(a1["logL"][i] - ref) <= 1e-9 * max(1.0, abs(ref)), (case, i, a1["logL"][i], ref)
    print("case %2d ok: %2d chains x %5d sources, kmax %2d, %2d sweeps, models %d, persistent ran: %s%s" %
          (case, nch, n, kmax, sweeps, s1["models"], s1["launches"] < s0["launches"],
           ", one chain per level: %d launches for %d sweeps" % (s1["launches"], sweeps + 7) if perm else ""), flush=True)
assert rt.load_library().rtmi_device_status() == 0
print("soak ok: %d cases, persistent engine ran in %d" % (ncases, ran_persistent))
