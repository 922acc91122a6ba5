#!/usr/bin/env python3
"""profiles/debug_predecessor.py -- the speculative sweep over the task queue gave slightly wrong sums when a handle of the same
shape had run the queue engine before it in the same process (tests/test_tempering.py, round 4).  Which predecessor does it?"""
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
def run(env, sweeps=30, nch=16, keep=False):
    old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
    try:
        h = raymod.NativeChains(nch, 31, off.size, pr, seed=9)
        for i in range(nch):
            h.set_chain(i, nodes12, 0.02, 1.0)
        h.evaluate()
        h.set_ladder(nch, 0, np.arange(nch) // 2, 1.15 ** -np.arange(float(nch // 2)))
        h.run(sweeps, 1, None, 0)
        a = h.get_all(); ctr = h.stream_positions()
        if not keep:
            del h
        return a, ctr
    finally:
        for k, v in old.items():
            (os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v))
def check(tag, ref, env):
    got, gctr = run(env)
    diff = [key for key in ("k", "sd", "logL", "ss") if not np.array_equal(ref[0][key], got[key])]
    if not np.array_equal(ref[1], gctr):
        diff.append("ctr")
    print("%-60s different: %s  status %d" % (tag, diff, rt.load_library().rtmi_device_status()), flush=True)
Q3 = {"RTMI_PT_SPECULATE": "3", "RTMI_PT_PERSISTENT": "1"}
ref = run({"RTMI_PT_SPECULATE": "0", "RTMI_PT_PERSISTENT": "0"})
check("fresh process, queue 3 rows", ref, Q3)
run({"RTMI_PT_SPECULATE": "4", "RTMI_PT_PERSISTENT": "0"}, sweeps=40)
check("after a 4-row handle on per-slot launches", ref, Q3)
run({"RTMI_PT_SPECULATE": "4", "RTMI_PT_PERSISTENT": "1"}, sweeps=40)
check("after a 4-row handle on the queue", ref, Q3)
check("again", ref, Q3)
run({"RTMI_PT_SPECULATE": "4", "RTMI_PT_PERSISTENT": "1"}, sweeps=40)
check("after a 4-row queue handle, tickets from 1<<20", ref, {**Q3, "RTMI_PT_TICKET_BASE": str(1 << 20)})
run({"RTMI_PT_SPECULATE": "4", "RTMI_PT_PERSISTENT": "1"}, sweeps=40)
check("after a 4-row queue handle, one task per row", ref, {**Q3, "RTMI_PT_TASK_SRC": "100000"})
run({"RTMI_PT_SPECULATE": "4", "RTMI_PT_PERSISTENT": "1"}, sweeps=40)
check("after a 4-row queue handle, full likelihood", ref, {**Q3, "RTMI_PT_INCREMENTAL": "0"})
run({"RTMI_PT_SPECULATE": "4"}, sweeps=40)
check("after a 4-row handle with the engine trial", ref, Q3)
run({"RTMI_PT_SPECULATE": "3", "RTMI_PT_PERSISTENT": "1"}, sweeps=40)
check("after a 3-row queue handle", ref, Q3)
check("4 rows after that", ref, {"RTMI_PT_SPECULATE": "4", "RTMI_PT_PERSISTENT": "1"})
check("2 rows", ref, {"RTMI_PT_SPECULATE": "2", "RTMI_PT_PERSISTENT": "1"})
