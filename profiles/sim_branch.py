#!/usr/bin/env python3
"""profiles/sim_branch.py -- census of GetPTime's three-way branch (subroutineR-quiet.f90:136-153) on the C2 stream, and how
many of the sources that take the bisection could be PROVEN to take it without evaluating costFunc / costFunc_Prime at p0
(numpy, CPU; the estimate behind DESIGN.md section 7b's "skip the evaluation at p0" item).

Bounds tried, with A = sum |H V| of the stack, s_min = sqrt(1 - p0^2 vmax^2) (the smallest cosine of the stack):
  crude   costFunc's sum <= p0 A / s_min, |costFunc_Prime| <= A / s_min^3            (everything needed is in the layer table)
  tables  slower layers at their value at 1/vmax, the fastest layers exactly          (SA / SC / HC of bisection_skip + one more table)
  scaled  as `tables`, the slower layers' f-terms scaled by p0 vmax (term / x grows with x)
A source is provable when  cf_lo > 0  and  p0 + cf_lo / P_up >= 1/vmax - 1e-10 (+ margin)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracerfortran_amd.synthetic import c2_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
v, d, off, dep = c2_workload(n)
v, d = np.asarray(v), np.asarray(d)
nl = np.searchsorted(d, dep, side="right")
H = np.diff(np.concatenate([[0], d]))
cnt = dict(straight=int((nl == 0).sum()), newton_cf_negative=0, newton_step_inside=0, bisection=0)
prov = dict(crude=0, tables=0, scaled=0)
for il in range(1, len(d) + 1):
    m = nl == il
    if not m.any():
        continue
    R, z = off[m], dep[m]
    hs = np.concatenate([H[:il], [0]])[None, :].repeat(m.sum(), 0)
    hs[:, il] = z - d[il - 1]
    vs = v[:il + 1][None, :]
    p0 = (z / np.sqrt(R * R + z * z)) / (z / (hs / vs).sum(1))
    vmax = vs.max(); vmi = 1 / vmax
    for _ in range(60):
        bad = (p0 * p0) * (vmax + 1) ** 2 > 1
        if not bad.any():
            break
        p0 = np.where(bad, p0 / 2, p0)
    s = np.sqrt(1 - (p0[:, None] * vs) ** 2)
    hv = hs * vs
    cf = R - (hv * p0[:, None] / s).sum(1); cfp = -(hv / s ** 3).sum(1)
    neg = cf < 0
    inside = (~neg) & (p0 - cf / cfp < vmi - 1e-10)
    bis = (~neg) & (~inside)
    cnt["newton_cf_negative"] += int(neg.sum()); cnt["newton_step_inside"] += int(inside.sum()); cnt["bisection"] += int(bis.sum())
    A = hv.sum(1); smin = np.sqrt(1 - (p0 * vmax) ** 2); eps = 1 + 1e-9
    def provable(s_up, p_up):
        lo = R - s_up * eps
        ok = (lo > 0) & (p0 + lo / (p_up * eps) >= vmi - 1e-10 + 1e-13)
        assert not (ok & ~bis).any()      # the bound never claims a source that does not take the bisection
        return int((ok & bis).sum())
    fast = (vs >= vmax)[0]; slow = ~fast
    sv = np.sqrt(1 - (vmi * vs[:, slow]) ** 2)
    SA = (hv[:, slow] * vmi / sv).sum(1); SP = (hv[:, slow] / sv ** 3).sum(1); HC = hv[:, fast].sum(1)
    prov["crude"] += provable(p0 * A / smin, A / smin ** 3)
    prov["tables"] += provable(SA + HC * p0 / smin, SP + HC / smin ** 3)
    prov["scaled"] += provable(SA * (p0 / vmi) + HC * p0 / smin, SP + HC / smin ** 3)
tot = sum(cnt.values())
print(f"C2 stream, {tot} sources:", {k: f"{x} ({x / tot:.1%})" for k, x in cnt.items()})
print("bisection sources whose branch is provable without the evaluation at p0:",
      {k: f"{x / cnt['bisection']:.1%} (= {x / tot:.1%} of all sources)" for k, x in prov.items()})
