#!/usr/bin/env python3
"""profiles/host_register_cost.py -- what pinning the caller's arrays for the duration of a drop-in call costs:
hipHostRegister / hipHostUnregister of the three 8 MB arrays of a 1e6-source call (warm pages), whole and in chunks."""
import time
import numpy as np
import torch
rt = torch.cuda.cudart()
torch.cuda.init()
n = 1_000_000
arrs = [np.random.rand(n) for _ in range(3)]
def reg_all(chunks):
    t0 = time.perf_counter()
    for a in arrs:
        step = a.nbytes // chunks
        for c in range(chunks):
            assert int(rt.cudaHostRegister(a.ctypes.data + c * step, step, 0)) == 0
    t1 = time.perf_counter()
    for a in arrs:
        step = a.nbytes // chunks
        for c in range(chunks):
            assert int(rt.cudaHostUnregister(a.ctypes.data + c * step)) == 0
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3
for chunks in (1, 4, 16):
    r = np.array([reg_all(chunks) for _ in range(15)])
    print("3 x 8 MB in %2d piece(s) each: register %.3f ms, unregister %.3f ms (medians)" % (chunks, np.median(r[:, 0]), np.median(r[:, 1])))
