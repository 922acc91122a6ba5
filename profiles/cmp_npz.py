import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
assert sorted(a.files) == sorted(b.files)
bad = [k for k in a.files if not (a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True))]
print("arrays", len(a.files), "different:", bad)
for k in bad[:6]:
    print(k, a[k].ravel()[:8], b[k].ravel()[:8])
print("k10 stats", a["k10_stats"], "samples", a["k10_samples_k"].size, "k30 stats", a["k30_stats"], "max k", a["k30_state_k"].max())
sys.exit(1 if bad else 0)
