import sys, numpy as np
sys.path.insert(0,'/root/repo')
import torch, raytracerfortran_amd as rt
from raytracerfortran_amd import raymod
rt.load_library(); rt.init(0)
lo = int(np.float32(1.0000001e-12).view(np.uint32)); hi = int(np.float32(1.0).view(np.uint32))
tot=[0,0]; step=1<<26
for start in range(lo, hi+1, step):
    bits = torch.arange(start, min(start+step, hi+1), dtype=torch.int64, device="cuda").to(torch.int32)
    w = bits.view(torch.float32); one = torch.ones_like(w)
    a,b = raymod.selftest_fastmath_f32(w.contiguous(), one, one); tot[0]+=a; tot[1]+=b
print("exhaustive unit numerators: mismatches", tot, "of", hi-lo+1)
g = torch.Generator(device="cuda").manual_seed(11); n=1<<26
for trial in range(3):
    u = torch.rand(n, dtype=torch.float32, device="cuda", generator=g); e = torch.rand(n, dtype=torch.float32, device="cuda", generator=g)
    w = torch.clamp(torch.exp(-27.0*u), 1.1e-12, 1.0); den = torch.exp(82.0*(e-0.5)); num = torch.exp(82.0*(u-0.5))
    print("random", trial, raymod.selftest_fastmath_f32(w.contiguous(), num.contiguous(), den.contiguous()))
