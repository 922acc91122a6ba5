#!/usr/bin/env python3
"""tests/golden/make_f32fast_golden.py -- the pairs on which the single-precision FAST forward model
(rtmi_trace_device_f32_fast) and the reference give DIFFERENT answers by design, with both answers.
Writes tests/golden/f32fast_pairs.json.  Run in the build container (needs oracle/_ref, the reference compiled
unmodified by oracle/Makefile); a few seconds of CPU.

The reference (subroutineR-quiet.f90:112-173 GetPTime, :296-346 solve) returns -999 where its iteration gives up and,
through the inverted convergence test of solve (:327-330), sometimes accepts an iterate far from the root; the fast path
solves the same equation (costFunc = 0, :184-202) to single-precision roundoff and returns the time of the actual ray
(include/rtmi.h).  The fixture holds, for float-representable inputs (so that the kernel, the
reference and the converged solution all see exactly the same numbers):
  * `sentinel`: sources where the reference returns -999,
  * `gross`:    sources where the reference's accepted time is off by more than 1 ms,
  * `agree`:    ordinary sources (both within the reference's own stopping tolerance), as a control,
each with  t_reference  (oracle/_ref, i.e. the reference itself)  and  t_converged  (tests/f32fast_ref.py: the equations'
double-precision solution).  tests/test_oracle_golden.py checks the oracle against t_reference (bit-exact, CPU);
tests/test_gpu_f32fast.py checks the kernel against t_converged and that it returns NO sentinel on these pairs: a change of
the fast path's sentinel rule, or of the reference-side behaviour it is documented against, shows up here."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from f32fast_ref import converged_times

assert O.have_ref(), "oracle/_ref is not built: run oracle/Makefile in the build container"
rng = np.random.default_rng(405)
f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
cases, want = [], {"sentinel": 24, "gross": 24, "agree": 16}
have = {k: 0 for k in want}
for trial in range(400):
    nl = int(rng.choice([3, 8, 20, 31]))
    v = f32(np.sort(rng.uniform(1500, 6500, nl + 1)))
    if trial % 3 == 0:                      # a low-velocity zone, as BASELINE config 2 has one
        i = int(rng.integers(1, nl))
        v[i] = f32(v[i - 1] * 0.8)
    d = f32(np.cumsum(rng.uniform(150, 700, nl)))
    n = 20000
    off, dep = f32(rng.uniform(10, 9000, n)), f32(rng.uniform(50, d[-1] + 400, n))
    ref = O.ref_dff(v, d, off, dep)
    tc, _ = converged_times(v, d, off, dep)
    err = np.abs(ref - tc)
    pick = {"sentinel": np.flatnonzero(ref == -999.0), "gross": np.flatnonzero((ref != -999.0) & (err > 1e-3)),
            "agree": np.flatnonzero((ref != -999.0) & (err < 6.7e-5))}
    for kind, idx in pick.items():
        take = idx[: min(3, want[kind] - have[kind])]
        if take.size == 0:
            continue
        have[kind] += int(take.size)
        cases.append({"kind": kind, "vels": v.tolist(), "depths": d.tolist(), "src_offset": off[take].tolist(),
                      "src_depth": dep[take].tolist(), "t_reference": ref[take].tolist(), "t_converged": tc[take].tolist()})
    if all(have[k] >= want[k] for k in want):
        break
assert all(have[k] >= want[k] for k in want), have
out = {"generator": "tests/golden/make_f32fast_golden.py",
       "reference": "subroutineR-quiet.f90 compiled unmodified (oracle/Makefile, amdflang -O2) for t_reference; "
                    "tests/f32fast_ref.py (converged double-precision solution of costFunc = 0) for t_converged",
       "note": "all inputs are float-representable doubles", "counts": have, "cases": cases}
path = os.path.join(ROOT, "tests", "golden", "f32fast_pairs.json")
with open(path, "w") as f:
    json.dump(out, f)
print(path, have, os.path.getsize(path), "bytes;", "trials", trial + 1)
