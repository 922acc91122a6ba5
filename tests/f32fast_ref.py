"""The checker of the single-precision fast forward model (rtmi_trace_device_f32_fast): a fully CONVERGED double-precision
solution of the reference's own equations -- costFunc = 0 (subroutineR-quiet.f90:184-202), the travel-time sum (:156-166), the
stack of layers by whichLayer (:9-32) / InsertLayer (:38-69) -- in numpy.  Test infrastructure only; pinned against the
oracle (= the reference, which stops 0.1 m short of the root) by tests/test_f32fast_ref_cpu.py."""
import numpy as np


def converged_times(v, d, off, dep):
    """t(p*) with f(p*) = 0 to double-precision roundoff: monotone Newton in s = tan(theta) of the stack's fastest layer
    (X(s) = s sum H r / sqrt(1 + c s^2) is concave increasing), sources grouped by layer.  Returns (t, p)."""
    v, d, R, z = (np.asarray(a, np.float64) for a in (v, d, off, dep))
    NL, n = d.size, R.size
    cnt = np.searchsorted(d, z, side="right")
    il = cnt.copy()
    il[(cnt == NL) & (z == d[-1])] = NL - 1          # whichLayer's special case (:22-30)
    t, p = np.empty(n), np.full(n, np.nan)
    for L in range(NL + 1):
        m = il == L
        if not m.any():
            continue
        Rm, zm = R[m], z[m]
        if L == 0:
            t[m] = np.sqrt(zm * zm + Rm * Rm) / v[0]
            continue
        V = v[:L + 1]
        vmax = V.max()
        H = np.concatenate([[d[0]], np.diff(d[:L])])
        c = (vmax - V) * (vmax + V) / (vmax * vmax)
        hl = zm - d[L - 1]
        A = np.concatenate([np.broadcast_to(H * V[:L] / vmax, (m.sum(), L)), (hl * V[L] / vmax)[:, None]], 1)
        B = np.concatenate([np.broadcast_to(H / V[:L], (m.sum(), L)), (hl / V[L])[:, None]], 1)
        s = Rm / A.sum(1)
        with np.errstate(all="ignore"):  # a ray that does not exist (see the kernel's -999) diverges here: inf / NaN, by design
          for _ in range(60):
            y = 1.0 / np.sqrt(1.0 + c[None, :] * (s * s)[:, None])
            s = s + (Rm - s * (A * y).sum(1)) / (A * y ** 3).sum(1)
          y = 1.0 / np.sqrt(1.0 + c[None, :] * (s * s)[:, None])
          t[m] = np.sqrt(1.0 + s * s) * (B * y).sum(1)
          p[m] = s / np.sqrt(1.0 + s * s) / vmax
    return t, p
