#!/usr/bin/env python3
"""profiles/c5_loop.py -- 12 launches of the fp32 fast forward model at config 5's shape (4e6 sources x 20 interfaces) and
nothing else: the program the PMC passes of profiles/pmc_sq.sh profile (PROG=c5).  RTMI_AB_LIB selects another build."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd import raymod
if os.environ.get("RTMI_AB_LIB"):
    raymod.LIB_PATH = os.path.abspath(os.environ["RTMI_AB_LIB"])
rt.load_library(); rt.init(0)
n = 4_000_000
rng = np.random.default_rng(2025)
d = np.cumsum(rng.uniform(200, 600, 20)); v = np.sort(rng.uniform(1500, 6500, 21))
off, dep = rng.uniform(10, 8000, n), rng.uniform(50, d[-1] + 500, n)
ten = [torch.tensor(np.asarray(a, np.float32), device="cuda") for a in (v, d, off, dep)]
out = torch.empty(n, dtype=torch.float32, device="cuda")
for _ in range(12):
    rt.trace_device(*ten, timeP=out, fast=True)
torch.cuda.synchronize()
t = out.cpu().numpy()
bad = np.nonzero(t == -999.0)[0]
print("sentinels:", bad[:10], [(float(off[i]), float(dep[i])) for i in bad[:10]])
