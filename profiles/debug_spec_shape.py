#!/usr/bin/env python3
"""profiles/debug_spec_shape.py -- which (speculation rows, engine, incremental, task size) leave the config-4 miniature's chains
different from the one-at-a-time per-slot reference, and from which sweep on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import raytracerfortran_amd as rt
from raytracerfortran_amd import raymod, synthetic as S, tempering as T
from oracle import oracle as O
rt.load_library(); rt.init(0)
off, dep = S.sampler_sources(3000, 77)
rng = np.random.default_rng(78)
z = np.concatenate([[0.0], np.sort(rng.uniform(300.0, 9000.0, 11))]); z[1:] += np.arange(11) * 150.0
nodes12 = np.column_stack([z, rng.uniform(2000.0, 8000.0, 12)])
vv, dd = T.layering(nodes12)
obs = O.oracle_dff(vv, dd, off, dep) + rng.normal(0.0, 0.02, off.size)
rt.set_sources(off, dep, obs)
pr = T.Prior(kmin=2, kmax=30, lam=10.0)
def run(env, sweeps, nch=16):
    old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
    try:
        h = raymod.NativeChains(nch, 31, off.size, pr, seed=9)
        for i in range(nch):
            h.set_chain(i, nodes12, 0.02, 1.0)
        h.evaluate()
        h.set_ladder(nch, 0, np.arange(nch) // 2, 1.15 ** -np.arange(float(nch // 2)))
        rec = int(os.environ.get('DBG_RECORD', '0'))
        if rec:
            h.sample_buffer(4096)
        h.run(sweeps, 1, None, rec)
        a = h.get_all(); ctr = h.stream_positions(); info = h.speculation()
        del h
        return a, ctr, info
    finally:
        for k, v in old.items():
            (os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v))
for sweeps in (2, 5, 10, 30):
    ref, rctr, _ = run({"RTMI_PT_SPECULATE": "0", "RTMI_PT_PERSISTENT": "0"}, sweeps)
    # oracle check of the reference's own logL
    bad = 0
    for i in range(16):
        k = int(ref["k"][i]); v2, d2 = T.layering(np.column_stack([ref["z"][i, :k], ref["v"][i, :k]]))
        want = O.oracle_loglhood_rt(obs, O.oracle_dff(v2, d2, off, dep), ref["sd"][i], stable=True)
        bad += abs(ref["logL"][i] - want) > 1e-9 * abs(want)
    print("sweeps", sweeps, "reference chains off the oracle:", bad)
    for env in ({"RTMI_PT_SPECULATE": "3", "RTMI_PT_PERSISTENT": "0"}, {"RTMI_PT_SPECULATE": "3", "RTMI_PT_PERSISTENT": "1"},
                {"RTMI_PT_SPECULATE": "2", "RTMI_PT_PERSISTENT": "1"}, {"RTMI_PT_SPECULATE": "4", "RTMI_PT_PERSISTENT": "1"},
                {"RTMI_PT_SPECULATE": "1", "RTMI_PT_PERSISTENT": "1"},
                {"RTMI_PT_SPECULATE": "3", "RTMI_PT_PERSISTENT": "1", "RTMI_PT_INCREMENTAL": "0"},
                {"RTMI_PT_SPECULATE": "3", "RTMI_PT_PERSISTENT": "1", "RTMI_PT_TASK_SRC": "100000"},
                {"RTMI_PT_SPECULATE": "3", "RTMI_PT_PERSISTENT": "1", "RTMI_PT_TASK_SRC": "640"}):
        got, gctr, info = run(env, sweeps)
        diff = [key for key in ("k", "sd", "logL", "ss") if not np.array_equal(ref[key], got[key])]
        if not np.array_equal(rctr, gctr):
            diff.append("ctr")
        off_oracle = 0
        for i in range(16):
            k = int(got["k"][i]); v2, d2 = T.layering(np.column_stack([got["z"][i, :k], got["v"][i, :k]]))
            want = O.oracle_loglhood_rt(obs, O.oracle_dff(v2, d2, off, dep), got["sd"][i], stable=True)
            off_oracle += abs(got["logL"][i] - want) > 1e-9 * abs(want)
        print("  ", {k.replace("RTMI_PT_", ""): v for k, v in env.items()}, "engine", info["engine"], "different:", diff,
              "chains off the oracle:", off_oracle, "status", rt.load_library().rtmi_device_status())
