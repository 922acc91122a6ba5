import sys, numpy as np, time
sys.path.insert(0,'/root/repo')
import torch, raytracerfortran_amd as rt
rt.load_library(); rt.init(0)
rng = np.random.default_rng(17)
for nlay, n in ((100, 200000), (300, 100000), (1024, 50000)):
    d = np.cumsum(rng.uniform(5, 40, nlay)); v = rng.uniform(1500, 8000, nlay + 1)
    off, dep = rng.uniform(10, 9000, n), rng.uniform(1, d[-1] + 100, n)
    a = [torch.tensor(x, device="cuda") for x in (v, d, off, dep)]
    for _ in range(2): rt.trace_device(*a)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): rt.trace_device(*a)
    e1.record(); torch.cuda.synchronize()
    print(nlay, n, "%.3f ms" % (e0.elapsed_time(e1)/5))
