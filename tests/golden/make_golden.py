#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

Runs ONLY in the build container, where /root/reference exists and
oracle/Makefile has compiled the reference's own forward model
(/root/reference/subroutineR-quiet.f90, unmodified) into oracle/_ref/.
The outputs are DATA: inputs + expected outputs as .npz (no pickles) and a
JSON index.  No reference source text is stored.

  python tests/golden/make_golden.py

Fixtures
  kat_author.json   numbers the reference AUTHOR published (gfortran runs):
                      KAT-A  notebook print of 5 travel times (7 digits)
                             raytracerR-export-data-to-MCMC.nb.html
                      KAT-B  /root/reference/rays.dat token stream (40 records)
                    plus the inputs needed to reproduce them
                    (test_1/test_1_src_data.txt, test_1/test_1_RT.txt,
                    eqdf.csv: data files the reference's example holds).
  ref_batches.npz   seeded synthetic models x sources with the travel times the
                    compiled reference returns for them (flang -O2; the -O0
                    build's max deviation is recorded in the index).
  c4_models.npz     BASELINE config 4 at its full source count: the k = 30 and k = 2 models of the seeded C4
                    workload x 1e5 shared sources (sources regenerated from the recipe; digest stored).
  index.json        per-case metadata (seed, sizes, what the case exercises,
                    reference branch statistics from the oracle's counters).
"""
from __future__ import annotations

import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF = "/root/reference"

V_TEST1 = [3100.0, 2400.0, 3730.0, 4200.0, 5800.0, 4420.0, 6200.0]  # raytracerR-export-data-to-MCMC.Rmd:61
D_TEST1 = [1000.0, 1500.0, 2000.0, 2500.0, 3260.0, 3960.0]          # :62


def author_kats():
    src = np.loadtxt(os.path.join(REF, "test_1/test_1_src_data.txt"))
    obs = np.loadtxt(os.path.join(REF, "test_1/test_1_RT.txt"))
    html = open(os.path.join(REF, "raytracerR-export-data-to-MCMC.nb.html"), encoding="utf-8", errors="ignore").read()
    m = re.search(r"\[1\] (0\.7445202 [0-9. ]+)", html)
    kat_a_print = [float(x) for x in m.group(1).split()]
    # R's set.seed(12) uniform stream is recoverable from the NSrc=20 depths
    # (Rmd:44-46: z = runif(N,1050,4200) is the first draw): u_i = (z_i-1050)/3150.
    u = (src[:, 1] - 1050.0) / 3150.0
    z5 = 1050.0 + 3150.0 * u[0:5]
    x5 = 10.0 + 6490.0 * u[5:10]
    y5 = 10.0 + 4490.0 * u[10:15]
    off5 = np.sqrt(x5 ** 2 + y5 ** 2)
    rays_tokens = [float(t) for t in open(os.path.join(REF, "rays.dat")).read().split()]
    eq = np.loadtxt(os.path.join(REF, "eqdf.csv"), delimiter=",", skiprows=1)
    eq_off = np.sqrt(eq[:, 0] ** 2 + eq[:, 1] ** 2)
    eq_dep = eq[:, 2]
    v_eq, d_eq = [3100.0, 4270.0, 6000.0], [2000.0, 4000.0]  # rayTracerR.R:25-26
    out = {
        "model": {"vels": V_TEST1, "depths": D_TEST1},
        "test_1_src_offset": src[:, 0].tolist(),
        "test_1_src_depth": src[:, 1].tolist(),
        "test_1_obs_RT": obs.tolist(),
        "kat_a": {"src_offset": off5.tolist(), "src_depth": z5.tolist(), "printed_timeP": kat_a_print,
                  "source": "raytracerR-export-data-to-MCMC.nb.html, print(timeP$timeP), NSrc=5, gfortran"},
        "kat_b": {"rays_dat_tokens": rays_tokens,
                  "source": "/root/reference/rays.dat: keep_delta=10 dump of the 20 test_1 rays, gfortran"},
        "config1_ref_timeP": O.ref_dff(V_TEST1, D_TEST1, src[:, 0], src[:, 1]).tolist(),
        "config1_ref_timeP_O0": O.ref_dff(V_TEST1, D_TEST1, src[:, 0], src[:, 1], opt="O0").tolist(),
        "kat_a_ref_timeP": O.ref_dff(V_TEST1, D_TEST1, off5, z5).tolist(),
        "rayTracerR_example": {"vels": v_eq, "depths": d_eq, "src_offset": eq_off.tolist(),
                               "src_depth": eq_dep.tolist(),
                               "ref_timeP": O.ref_dff(v_eq, d_eq, eq_off, eq_dep).tolist(),
                               "source": "rayTracerR.R:23-33 on eqdf.csv"},
        # logL of test_1_map.dat (k=7, sd=0.02) through the reference's own
        # LOGLHOOD_RT, measured during the survey; not regenerable here without MPI stand-ins.
        "test_1_map_logL_survey": {"value": 55.124129805485509, "sd": 0.02,
                                   "source": "SURVEY.md section 4 (survey-time run of loglhood.f90)"},
    }
    return out


def synth_cases():
    """Seeded (model, sources) batches covering the edge cases SURVEY.md section 4 lists."""
    cases = []

    def add(name, vels, depths, off, dep, note):
        cases.append({"name": name, "vels": np.asarray(vels, float), "depths": np.asarray(depths, float),
                      "off": np.asarray(off, float), "dep": np.asarray(dep, float), "note": note})

    # C2 recipe, first 4000 sources of the 1e6 stream.
    rng = np.random.default_rng(12345)
    d = np.cumsum(rng.uniform(300, 800, 8))
    v = np.sort(rng.uniform(1500, 6500, 9))
    v[[2, 3]] = v[[3, 2]]
    off = rng.uniform(10, 8000, 1000000)
    dep = rng.uniform(50, d[-1] + 500, 1000000)
    add("c2_head", v, d, off[:4000], dep[:4000], "BASELINE config 2 recipe, first 4000 of 1e6 (8 interfaces, one LVZ)")
    # the -999 sources of the full 1e6 C2 stream plus their neighbours
    t_full = O.ref_dff(v, d, off, dep)
    bad = np.flatnonzero(t_full == -999.0)
    idx = np.unique(np.clip(np.concatenate([bad, bad + 1, bad - 1]), 0, off.size - 1))
    add("c2_sentinels", v, d, off[idx], dep[idx], "every -999 source of the 1e6 C2 stream and its neighbours")

    # monotone velocities, 8 interfaces
    rng = np.random.default_rng(1)
    d = np.cumsum(rng.uniform(300, 800, 8))
    v = np.sort(rng.uniform(1500, 6500, 9))
    add("mono8", v, d, rng.uniform(10, 8000, 3000), rng.uniform(50, d[-1] + 500, 3000), "monotone v, 8 interfaces")

    # 20 interfaces (C5 recipe geometry, fp64)
    rng = np.random.default_rng(2025)
    d = np.cumsum(rng.uniform(200, 600, 20))
    v = np.sort(rng.uniform(1500, 6500, 21))
    add("c5_geom20", v, d, rng.uniform(10, 8000, 3000), rng.uniform(50, d[-1] + 500, 3000), "20 interfaces")

    # prior-like unsorted velocities, k = 30 (29 interfaces), sampler source geometry
    rng = np.random.default_rng(7)
    d = np.sort(rng.uniform(100.1, 10000.1, 29))
    v = rng.uniform(1500, 10000, 30)
    x = rng.uniform(10, 6500, 3000)
    y = rng.uniform(10, 4500, 3000)
    add("prior_k30", v, d, np.sqrt(x * x + y * y), rng.uniform(1050, 4200, 3000),
        "k=30 prior draw (read_input.f90:205-214 bounds), unsorted v")

    # many small prior models k in 2..10 on the test_1-like geometry -> one case each
    rng = np.random.default_rng(11)
    x = rng.uniform(10, 6500, 500)
    y = rng.uniform(10, 4500, 500)
    so, sd_ = np.sqrt(x * x + y * y), rng.uniform(1050, 4200, 500)
    for k in range(2, 11):
        d = np.sort(rng.uniform(100.1, 10000.1, k - 1))
        v = rng.uniform(1500, 10000, k)
        add(f"prior_k{k}", v, d, so, sd_, f"prior draw with k={k} nodes")

    # k = 1 as the sampler calls it: one fake interface at 9999.9 m, v repeated (loglhood.f90:137-144)
    add("k1_halfspace", [4321.0, 4321.0], [9999.9], so, sd_, "k=1 special case of LOGLHOOD_RT")

    # edge cases on the test_1 model
    v, d = np.array(V_TEST1), np.array(D_TEST1)
    e_dep = np.concatenate([d, d, d + 1e-9, d - 1e-9, [0.0, 1e-3, 999.999999, 3960.0, 3960.0000001, 12000.0, 500.0, 500.0]])
    e_off = np.concatenate([np.full(6, 1234.5), np.full(6, 0.0), np.full(6, 5000.0), np.full(6, 20000.0),
                            [0.0, 10.0, 0.0, 1e5, 1e5, 0.0, 0.0, 1e6]])
    add("edges_test1", v, d, e_off, e_dep,
        "sources on interfaces (incl. the last), epsilon above/below, offset 0, top layer, half-space, huge offsets")

    # near-critical: slow thick lid over fast layers, long offsets
    rng = np.random.default_rng(3)
    add("near_critical", [1500.0, 6500.0, 1600.0, 9900.0], [200.0, 1200.0, 1300.0],
        rng.uniform(3000, 60000, 2000), rng.uniform(205, 3000, 2000), "head-wave-like geometry, near-critical rays")

    # strong LVZ stack
    rng = np.random.default_rng(4)
    add("lvz", [5000.0, 1500.0, 6000.0, 1700.0, 7000.0, 2000.0], [400.0, 900.0, 1500.0, 2200.0, 3000.0],
        rng.uniform(1, 9000, 2000), rng.uniform(1, 4000, 2000), "alternating low-velocity zones")

    # 99 interfaces: the limit of the sampler copy's DIMENSION(100) work arrays
    rng = np.random.default_rng(5)
    d = np.cumsum(rng.uniform(20, 120, 99))
    v = rng.uniform(1500, 8000, 100)
    add("n99", v, d, rng.uniform(10, 9000, 1500), rng.uniform(10, d[-1] + 300, 1500), "99 interfaces")
    return cases


def c4_fixture():
    """BASELINE config 4 at full source count: the k = 30 and the k = 2 model of the seeded C4 workload
    (raytracerfortran_amd/synthetic.py) over all 1e5 shared sources, travel times from the compiled reference.
    The sources are regenerated from the recipe by the tests (their SHA-256 is stored), so the file holds the two
    models and 2 x 1e5 reference times only."""
    from raytracerfortran_amd import synthetic as S
    w = S.c4_workload(32)
    out = {"digest_sources": np.frombuffer(bytes.fromhex(S.digest(w["off"], w["dep"])), dtype=np.uint8)}
    for name, m in (("k30", 0), ("k2", 1)):
        nl = int(w["NL"][m])
        v, d = w["V"][m, :nl + 1], w["D"][m, :nl]
        t = O.ref_dff(v, d, w["off"], w["dep"])
        t0 = O.ref_dff(v, d, w["off"], w["dep"], opt="O0")
        both = (t != -999.0) & (t0 != -999.0)
        print(f"c4 {name}: nl={nl} sentinels={int((t == -999.0).sum())} O0-vs-O2 max {np.abs(t - t0)[both].max():.3e}")
        out[f"{name}__vels"], out[f"{name}__depths"], out[f"{name}__timeP"] = v, d, t
    np.savez_compressed(os.path.join(HERE, "c4_models.npz"), **out)


def main():
    assert O.have_ref(), "run `make -C oracle ref` first (needs /root/reference and amdflang)"
    if "--c4-only" in sys.argv:
        return c4_fixture()
    kat = author_kats()
    with open(os.path.join(HERE, "kat_author.json"), "w") as f:
        json.dump(kat, f, indent=1)
    arrays, index = {}, []
    for c in synth_cases():
        t_ref = O.ref_dff(c["vels"], c["depths"], c["off"], c["dep"])
        t_o0 = O.ref_dff(c["vels"], c["depths"], c["off"], c["dep"], opt="O0")
        _, st = O.oracle_dff_stats(c["vels"], c["depths"], c["off"], c["dep"])
        both = (t_ref != -999.0) & (t_o0 != -999.0)
        n = c["name"]
        arrays[f"{n}__vels"] = c["vels"]
        arrays[f"{n}__depths"] = c["depths"]
        arrays[f"{n}__off"] = c["off"]
        arrays[f"{n}__dep"] = c["dep"]
        arrays[f"{n}__timeP"] = t_ref
        index.append({"name": n, "note": c["note"], "nsrc": int(c["off"].size), "nlayers": int(c["depths"].size),
                      "n_sentinel": int((t_ref == -999.0).sum()),
                      "flang_O0_vs_O2_max_abs": float(np.abs(t_ref - t_o0)[both].max()) if both.any() else 0.0,
                      "flang_O0_vs_O2_sentinel_mismatch": int(((t_ref == -999.0) != (t_o0 == -999.0)).sum()),
                      "oracle_branch_stats": st})
    np.savez_compressed(os.path.join(HERE, "ref_batches.npz"), **arrays)
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "reference": "subroutineR-quiet.f90 compiled unmodified, amdflang 22 (ROCm 7.2) -O2",
                   "cases": index}, f, indent=1)
    tot = sum(c["nsrc"] for c in index)
    print(f"wrote {len(index)} cases, {tot} sources")
    c4_fixture()


if __name__ == "__main__":
    main()
