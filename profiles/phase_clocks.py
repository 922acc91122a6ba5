#!/usr/bin/env python3
"""Phase shares of the fused kernel from in-kernel clocks (diagnostic build path: rtmi_debug_stamps)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracerfortran_amd as rt
from raytracerfortran_amd import raymod
if os.environ.get("RTMI_AB_LIB"):  # another build of the library (A/B of the phase clocks on one box)
    raymod.LIB_PATH = os.path.abspath(os.environ["RTMI_AB_LIB"])
import bench
lib = rt.load_library(); rt.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
if len(sys.argv) > 2:
    rt.set_option("promote_age", int(sys.argv[2]))
fine = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # 0: phase and per-wave clocks only (undisturbed timeline)
rt.set_option("diag_fine", fine)
v, d, off, dep = bench.c2_workload(n)
dev = "cuda"
dv, dd, do, dz = (torch.tensor(x, device=dev) for x in (v, d, off, dep))
st = torch.zeros(200 * 512, dtype=torch.int64, device=dev)
for _ in range(3):
    rt.trace_device(dv, dd, do, dz)
lib.rtmi_debug_stamps(ctypes.c_void_p(st.data_ptr()))
rt.trace_device(dv, dd, do, dz)
torch.cuda.synchronize()
lib.rtmi_debug_stamps(None)
full = st.cpu().numpy().reshape(-1, 200)
full = full[full[:, 0] != 0]
s = full[:, :4]
t0 = s[:, 0].min()
print("workgroups", len(s))
for name, a, b in (("load+sort", 0, 1), ("solve", 1, 2), ("time", 2, 3), ("total", 0, 3)):
    dur = (s[:, b] - s[:, a]).astype(float)
    print(f"{name:10s} ticks: mean {dur.mean():10.0f}  min {dur.min():10.0f}  max {dur.max():10.0f}")
print("start skew ticks: max", (s[:, 0] - t0).max(), " end spread: min", (s[:, 3] - t0).min(), "max", (s[:, 3] - t0).max())
sol = (s[:, 2] - s[:, 1]).astype(float)
for x in range(8):
    g = sol[x::8]
    print(f"blockIdx%8={x}: solve mean {g.mean():9.0f} min {g.min():9.0f} max {g.max():9.0f}")
# per-block work proxy: sum over the block's sources of the layer index of the source (layers per evaluation)
nb = len(s)
nch = (n + 63) // 64
work = np.zeros(nb)
nl = 1 + np.searchsorted(d, dep, side="right")  # the model's interfaces are sorted: layer index of each source
for b in range(nb):
    idx = np.concatenate([np.arange(c * 64, min((c + 1) * 64, n)) for c in range(b, nch, nb)])
    work[b] = nl[idx].sum()
print("corr(solve ticks, sum nl) =", np.corrcoef(sol, work)[0, 1], " sum-nl cv =", work.std() / work.mean())

wv = full[:, 8:200].reshape(len(full), 16, 12).astype(np.int64)
used = wv[:, :, 1] > 0
dry = np.where(used, wv[:, :, 0], 0).astype(float) - full[:, 1:2]      # since solve start
fin = np.where(used, wv[:, :, 1], 0).astype(float) - full[:, 1:2]
dry[~used] = np.nan; fin[~used] = np.nan
print("waves per workgroup:", used.sum(1).mean())
print("queue seen empty (first wave)  mean %.0f   (last wave) mean %.0f" % (np.nanmin(dry, 1).mean(), np.nanmax(dry, 1).mean()))
print("wave finished        (first)   mean %.0f   (last)      mean %.0f" % (np.nanmin(fin, 1).mean(), np.nanmax(fin, 1).mean()))
print("mean over waves of (finish - empty): %.0f ticks" % np.nanmean(fin - dry))
if not fine:
    print("wave:   " + " ".join("%7d" % w for w in range(16)))
    print("dry:    " + " ".join("%7.0f" % x for x in np.nanmean(dry, 0)))
    print("finish: " + " ".join("%7.0f" % x for x in np.nanmean(fin, 0)))
    sub = full[:, 4:7].astype(float) - full[:, 0:1]
    print("phase a cumulative:", np.round(sub.mean(0)))
    sys.exit(0)
iters = wv[:, :, 2][used].sum(); busy = wv[:, :, 3][used].sum(); trips = wv[:, :, 4][used].sum()
terms = (wv[:, :, 5][used] & ((1 << 40) - 1)).sum(); refills = (wv[:, :, 5][used] >> 40).sum()
print("solve loop: %d wave-trips, live lanes per trip %.1f / 64, layer-loop length per trip %.2f" % (iters, busy / iters, trips / iters))
print("useful layer terms %.3e of %.3e executed lane-terms = %.1f %%; refill events per trip %.3f" % (terms, trips * 64, 100.0 * terms / (trips * 64), refills / iters))
for nm, sl in (("express waves 0..3", slice(0, 4)), ("local waves 4..", slice(4, 16))):
    u = used[:, sl]
    if u.sum() == 0:
        continue
    it = wv[:, sl, 2][u].sum(); bu = wv[:, sl, 3][u].sum(); tr = wv[:, sl, 4][u].sum(); te = (wv[:, sl, 5][u] & ((1 << 40) - 1)).sum()
    print("%-20s trips/wave %.1f  live lanes %.1f  loop length %.2f  useful %.1f %%  finish mean %.0f" % (nm, it / u.sum(), bu / it, tr / it, 100.0 * te / (tr * 64), np.nanmean(fin[:, sl])))
print("wave:   " + " ".join("%7d" % w for w in range(16)))
print("dry:    " + " ".join("%7.0f" % x for x in np.nanmean(dry, 0)))
print("finish: " + " ".join("%7.0f" % x for x in np.nanmean(fin, 0)))
print("trips:  " + " ".join("%7.1f" % x for x in wv[:, :, 2].mean(0)))
print("live:   " + " ".join("%7.1f" % x for x in (wv[:, :, 3].sum(0) / np.maximum(wv[:, :, 2].sum(0), 1))))
nt = np.maximum(wv[:, :, 2].sum(0), 1)
for i, nm in enumerate(("tail check", "hand-over", "adoption", "refill", "max+exit", "evaluation")):
    print("%-11s/trip: " % nm + " ".join("%7.0f" % x for x in (wv[:, :, 6 + i].sum(0) / nt)))
print("wall       /trip: " + " ".join("%7.0f" % x for x in (np.nansum(fin, 0) / nt)))
sub = full[:, 4:7].astype(float) - full[:, 0:1]
print("phase a: depths+count %.0f  scatter+p0 %.0f  first eval %.0f  order %.0f (ticks, cumulative from start: %s)" % (
    sub[:, 0].mean(), (sub[:, 1] - sub[:, 0]).mean(), (sub[:, 2] - sub[:, 1]).mean(), (full[:, 1] - full[:, 6]).mean(), np.round(sub.mean(0))))
