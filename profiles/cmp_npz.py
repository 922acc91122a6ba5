#!/usr/bin/env python3
"""profiles/cmp_npz.py a.npz b.npz [launches] -- bitwise comparison of two dumps of profiles/pt_state_dump.py.
`launches`: ignore the launch counter in the statistics (the engines of the sweep count launches differently)."""
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
assert sorted(a.files) == sorted(b.files)
ignore_launches = len(sys.argv) > 3 and sys.argv[3] == "launches"
def same(k):
    x, y = a[k], b[k]
    if ignore_launches and k.endswith("_stats"):
        x, y = np.delete(x, 3), np.delete(y, 3)   # sorted keys: acc_bd, acc_mh, acc_sd, launches, models, prop_*
    return x.shape == y.shape and np.array_equal(x, y, equal_nan=True)
bad = [k for k in a.files if not same(k)]
print("arrays", len(a.files), "different:", bad)
for k in bad[:6]:
    print(k, a[k].ravel()[:8], b[k].ravel()[:8])
print("k10 stats", a["k10_stats"], "samples", a["k10_samples_k"].size, "k30 stats", a["k30_stats"], "max k", a["k30_state_k"].max())
sys.exit(1 if bad else 0)
