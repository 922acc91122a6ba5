"""profiles/sim_predict.py -- how many evaluations the 'predicted convergence' shortcut removes on the C2 workload.

A Newton step that moves LEFT (f(x) < 0: right of the root of the concave, decreasing f) lands at xnew with
|F(xnew)| <= 1.5 x (vmax^2 / (1 - x^2 vmax^2)) |f q|   (Taylor, |F''| <= 3 x vmax^2/w_min |F'| on [xnew, x]).
When that bound is below 0.08 the next evaluation would only confirm |f| < 0.1: it need not be made.
numpy restatement for the histogram only (the kernels are checked against the oracle)."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracerfortran_amd.synthetic import c2_workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
v, d, off, dep = c2_workload(n)
nl = np.searchsorted(d, dep, side='right') + 1
ev = np.zeros(n, int); evp = np.zeros(n, int); wrong = 0; pred_fail_small = 0; total_last = 0
a2_done = 0; a2_done_pred = 0
def fvals(x, H, V, R):
    s = np.sqrt(1 - (x * x) * (V * V))
    return R - np.sum((H * V) * x / s), -np.sum((H * V) / (s * s * s))
for k in range(n):
    L = nl[k]
    if L == 1: continue
    V = v[:L]; H = np.empty(L); H[0] = d[0]; H[1:L-1] = d[1:L-1] - d[:L-2]; H[L-1] = dep[k] - d[L-2]
    R = off[k]; z = dep[k]
    c = z / np.sum(H / V); ct = z / np.sqrt(R * R + z * z); p0 = ct / c
    vmax = V.max(); vmi = 1 / vmax
    while not ((p0 * p0) * ((vmax + 1) ** 2) <= 1): p0 /= 2
    trips = 1; tp = 1
    f, fp = fvals(p0, H, V, R); chk = p0 - f / fp
    def predict(x, f, q, xnew):
        if not (f < 0 and xnew < x and xnew <= vmi): return False
        w = 1 - (x * vmax) ** 2
        return w > 1e-8 and 1.5 * x * vmax * vmax * abs(f * q) <= 0.08 * w and abs(f) < 1e4 and abs(q) > 1e-9 * x
    def newton(x, f, fp, first_known):
        global wrong, pred_fail_small, total_last
        t = 0; tpred = 0; predicted = False
        for kk in range(1, 16):
            if not first_known or kk > 1:
                f, fp = fvals(x, H, V, R); t += 1
                if not predicted: tpred += 1
            if abs(f) < 0.1:
                total_last += 1
                if not predicted and kk > 1: pred_fail_small += 1
                return t, tpred
            if predicted: wrong += 1
            q = f / fp; xn = x - q
            clamp = xn > vmi
            if clamp: xn = vmi - 1e-9
            if not clamp and kk < 15 and predict(x, f, q, xn): predicted = True
            x = xn
        return t + 1, tpred + 1
    if f < 0 or chk < vmi - 1e-10:
        a, b = newton(p0, f, fp, True); trips += a; tp += b
        if a == 1: a2_done += 0
        if b == 0: a2_done_pred += 1
    else:
        x = vmi - 1e-12; dx = 1e-10 - x; needed = 0
        for kk in range(1, 21):
            dx *= 0.5; xm = x + dx; fm, fpm = fvals(xm, H, V, R)
            if not (fm >= 0.1001): needed += 1
            if fm < 0: x = xm
            if fm == 0: fused = False; break
            if x - fm / fpm < vmi - 1e-10: x = xm; fused = True; break
            if abs(fm) < 0.1: fused = False; break
        else: fused = False
        trips += max(needed, 1); tp += max(needed, 1)
        a, b = newton(x, fm, fpm, fused); trips += a; tp += b
    ev[k] = trips; evp[k] = tp
m = nl > 1
print("sources", m.sum(), "evaluations/source now", ev[m].mean(), "with predicted convergence", evp[m].mean())
print("wrong predictions (must be 0):", wrong, " final evaluations:", total_last, " not predicted:", pred_fail_small)
print("solve-loop evaluations/source (all but the first): now", ev[m].mean() - 1, "then", evp[m].mean() - 1)
print("sources finished by the first-evaluation pass with the prediction:", a2_done_pred)
print("hist now :", np.bincount(ev[m])[:20]); print("hist then:", np.bincount(evp[m])[:20])
