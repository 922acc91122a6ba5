#!/usr/bin/env python3
"""profiles/fastmath_soak.py [rounds] -- soak test of the guarded sqrt / divide / shared-reciprocal layer term against the
plain IEEE operators (rtmi_selftest_fastmath) on solver-shaped operands: rounds x 2^24 operands per distribution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd import raymod
rt.load_library(); rt.init(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << 24
g = torch.Generator(device="cuda").manual_seed(20261004)
tot = [0, 0, 0]; count = 0
R = lambda: torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
for r in range(rounds):
    for kind in range(3):
        u, e, q = R(), R(), R()
        if kind == 0:      # w = 1 - p^2 v^2 over the whole (0, 1] range
            pv = u * (1.0 - 1e-9); w = 1.0 - pv * pv
            hv = (1.0 + 3000.0 * e) * (1500.0 + 8500.0 * q); den = hv; num = hv * (pv / (1500.0 + 8500.0 * q))
        elif kind == 1:    # near-critical: w log-uniform down to 1e-14
            w = torch.exp(-32.0 * u); den = (0.01 + 1e4 * e) * (1500.0 + 8500.0 * q); num = den * 1e-4 * (1e-6 + q)
        else:              # p tiny (first bisection point 1e-10): w = 1 - O(1e-13)
            w = 1.0 - (1e-10 * (1500.0 + 8500.0 * u)) ** 2; den = (0.01 + 1e4 * e) * (1500.0 + 8500.0 * q); num = den * 1e-10
        bs, bd, bt = raymod.selftest_fastmath(w.contiguous(), num.contiguous(), den.contiguous())
        tot[0] += bs; tot[1] += bd; tot[2] += bt; count += n
print(f"operand triples checked: {count:.3e}; mismatches vs IEEE: sqrt {tot[0]}, divide {tot[1]}, layer term (2 quotients each) {tot[2]}")
