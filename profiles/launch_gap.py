#!/usr/bin/env python3
"""profiles/launch_gap.py -- what separates back-to-back launches of the C2 forward: one stream, two streams
(independent steps writing two output buffers), and a captured graph of 10 launches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
import bench
rt.load_library(); rt.init(0)
v, d, off, dep = bench.c2_workload(1_000_000)
dev = "cuda"
dv, dd, do, dz = (torch.tensor(x, device=dev) for x in (v, d, off, dep))
outs = [torch.empty(off.size, dtype=torch.float64, device=dev) for _ in range(2)]
K = 50
def wall(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K)
    return best * 1e3
def one_stream():
    for i in range(K):
        rt.trace_device(dv, dd, do, dz, timeP=outs[0])
s = [torch.cuda.Stream(), torch.cuda.Stream()]
def two_streams():
    for i in range(K):
        with torch.cuda.stream(s[i & 1]):
            rt.trace_device(dv, dd, do, dz, timeP=outs[i & 1])
for _ in range(3):
    one_stream(); two_streams()
# single launches, one at a time
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
single = []
for _ in range(20):
    torch.cuda.synchronize(); ev[0].record(); rt.trace_device(dv, dd, do, dz, timeP=outs[0]); ev[1].record(); torch.cuda.synchronize()
    single.append(ev[0].elapsed_time(ev[1]))
print("single launch (events)      %.4f ms" % np.median(single))
print("one stream, per step        %.4f ms" % wall(one_stream))
print("two streams, per step       %.4f ms" % wall(two_streams))
g = torch.cuda.CUDAGraph()
cs = torch.cuda.Stream()
with torch.cuda.stream(cs):
    rt.trace_device(dv, dd, do, dz, timeP=outs[0])
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=cs):
        for i in range(10):
            rt.trace_device(dv, dd, do, dz, timeP=outs[0])
def graph():
    for i in range(K // 10):
        g.replay()
graph()
print("graph of 10, per step       %.4f ms" % wall(graph))
