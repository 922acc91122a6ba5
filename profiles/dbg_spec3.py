import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
if os.environ.get("OLD") == "1":
    sys.path.insert(0, os.path.join(os.getcwd(), "profiles", "ab_prev"))  # the previous commit's package + library
import raytracerfortran_amd as rt
from raytracerfortran_amd import raymod, synthetic as S, tempering as T
from oracle import oracle as O
import test_tempering as tt
rt.load_library(); rt.init(0)
off, dep = S.sampler_sources(3000, 77)
rng = np.random.default_rng(78)
z = np.concatenate([[0.0], np.sort(rng.uniform(300.0, 9000.0, 11))])
z[1:] += np.arange(11) * 150.0
nodes12 = np.column_stack([z, rng.uniform(2000.0, 8000.0, 12)])
vv, dd = T.layering(nodes12)
obs = O.oracle_dff(vv, dd, off, dep) + rng.normal(0.0, 0.02, off.size)
rt.set_sources(off, dep, obs)
pr = T.Prior(kmin=2, kmax=30, lam=10.0)
n = 16
if os.environ.get("PRED", "1") == "1":
    nat = raymod.NativeChains(n, 31, off.size, pr, seed=9, stream_id=0)
    for i in range(n):
        nat.set_chain(i, nodes12, 0.02, 1.0)
    nat.evaluate()
    nat.set_ladder(n, 0, np.arange(n), 1.2 ** -np.arange(float(n)))
    nat.run(40, 1, None, 0)
    nat.get_all()
    del nat
if os.environ.get("T1", "1") != "0":
    import json
    kat = json.load(open("tests/golden/kat_author.json"))
    m = kat["model"]
    o1, d1, b1 = (np.asarray(kat[k], float) for k in ("test_1_src_offset", "test_1_src_depth", "test_1_obs_RT"))
    sh1 = {"name": "test_1", "off": o1, "dep": d1, "obs": b1, "nch": 8, "ld": 11, "prior": T.Prior(), "seed": 5,
           "nodes": [np.column_stack([[0.0] + m["depths"], m["vels"]])], "sd": 0.02, "sweeps": 60}
    rt.set_option("path", 2)
    rt.set_sources(o1, d1, b1)
    for cmb in os.environ.get("T1", "0,0 1,0 2,0 4,0 8,0 3,1 4,1 6,1 8,1 0,1").split():
        spec, pers = cmb.split(",")
        tt._run_native(sh1, {"RTMI_PT_SPECULATE": spec, "RTMI_PT_PERSISTENT": pers})
    rt.set_option("path", 0)
    rt.set_sources(off, dep, obs)
sh = {"name": "config-4 miniature", "off": off, "dep": dep, "obs": obs, "nch": 16, "ld": int(os.environ.get("LD", "31")),
      "prior": pr, "seed": 9, "nodes": [nodes12], "sd": 0.02, "sweeps": 30}
ref, ref_st, _ = tt._run_native(sh, {"RTMI_PT_SPECULATE": "0", "RTMI_PT_PERSISTENT": "0"})
for cmb in os.environ.get("COMBOS", "1,0 2,0 4,0 8,0 3,1").split():
    spec, pers = cmb.split(",")
    got, st, info = tt._run_native(sh, {"RTMI_PT_SPECULATE": spec, "RTMI_PT_PERSISTENT": pers})
    bad = [k for k in ref if not np.array_equal(ref[k], got[k])]
    print(cmb, "OK" if not bad and st == ref_st else "DIFF " + ",".join(bad), flush=True)
# ---- where is the bookkeeping wrong: SS of the final states recomputed in full
for inc in ("1", "0"):
    os.environ["RTMI_PT_INCREMENTAL"] = inc
    os.environ["RTMI_PT_SPECULATE"] = "3"; os.environ["RTMI_PT_PERSISTENT"] = "1"
    h = raymod.NativeChains(16, int(os.environ.get('LD', '31')), off.size, pr, seed=9)
    for i in range(16):
        h.set_chain(i, nodes12, 0.02, 1.0)
    h.evaluate()
    h.set_ladder(16, 0, np.arange(16) // 2, 1.15 ** -np.arange(8.0))
    first = None
    for sw in range(30):
        h.run(1, 1, None, 0)
        a = h.get_all()
        h2 = None
        ss_chain = a["ss"].copy()
        # recompute: a second handle evaluates the same states in full
        os.environ["RTMI_PT_SPECULATE"] = "0"; os.environ["RTMI_PT_PERSISTENT"] = "0"
        h2 = raymod.NativeChains(16, 31, off.size, pr, seed=1)
        os.environ["RTMI_PT_SPECULATE"] = "3"; os.environ["RTMI_PT_PERSISTENT"] = "1"
        for i in range(16):
            k = int(a["k"][i])
            h2.set_chain(i, np.column_stack([a["z"][i, :k], a["v"][i, :k]]), float(a["sd"][i]), 1.0)
        h2.evaluate()
        b = h2.get_all()
        del h2
        bad = np.argwhere(b["ss"] != ss_chain).ravel()
        if bad.size:
            print(f"incremental={inc}: after sweep {sw}: chains {bad.tolist()} carry an SS that is not their model's:",
                  [(float(ss_chain[i]), float(b['ss'][i]), float(ss_chain[i] - b['ss'][i])) for i in bad[:4]], flush=True)
            break
    else:
        print(f"incremental={inc}: SS of every chain equals a full evaluation of its model after each of 30 sweeps", flush=True)
    del h
