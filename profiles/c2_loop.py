#!/usr/bin/env python3
"""profiles/c2_loop.py -- 12 launches of the C2 forward (1e6 sources x 8 interfaces) and nothing else: the program the PMC
passes of profiles/pmc_sq.sh profile.  RTMI_AB_LIB selects another build of librtmi.so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd import raymod
if os.environ.get("RTMI_AB_LIB"):
    raymod.LIB_PATH = os.path.abspath(os.environ["RTMI_AB_LIB"])
from raytracerfortran_amd.synthetic import c2_workload
rt.load_library(); rt.init(0)
v, d, off, dep = c2_workload(1_000_000)
dv, dd, do, dz = (torch.tensor(x, device="cuda") for x in (v, d, off, dep))
out = torch.empty(off.size, dtype=torch.float64, device="cuda")
for _ in range(12):
    rt.trace_device(dv, dd, do, dz, timeP=out)
torch.cuda.synchronize()
