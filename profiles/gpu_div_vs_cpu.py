"""Is the GPU's own fp64 divide / sqrt (the compiler's expansion: v_rcp_f64 + refinement + v_div_fmas/fixup) correctly
rounded on the constructed hard cases?  Compared with numpy on the host (IEEE hardware divide)."""
import sys, os, numpy as np, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hardcases as H
N, D = H.fp64_divide_cases(nrandom=400)
N, D = np.array(N), np.array(D)
q_gpu = (torch.tensor(N, device="cuda") / torch.tensor(D, device="cuda")).cpu().numpy()
q_cpu = N / D
bad = np.flatnonzero(q_gpu != q_cpu)
print("divide: cases", N.size, "GPU != CPU:", bad.size)
for i in bad[:10]:
    print("   n=%s d=%s gpu=%s cpu=%s" % (float(N[i]).hex(), float(D[i]).hex(), float(q_gpu[i]).hex(), float(q_cpu[i]).hex()))
# the layer term's quotients: a / sqrt(w) with hard numerators for s
W, Nn, Dn = H.fp64_term_cases(nrandom=400)
W, Nn, Dn = np.array(W), np.array(Nn), np.array(Dn)
tw = torch.tensor(W, device="cuda")
s_gpu = torch.sqrt(tw)
print("sqrt: GPU != CPU:", int((s_gpu.cpu().numpy() != np.sqrt(W)).sum()))
tf_gpu = (torch.tensor(Nn, device="cuda") / s_gpu).cpu().numpy()
tf_cpu = Nn / np.sqrt(W)
s3g = (s_gpu * s_gpu) * s_gpu
tp_gpu = (torch.tensor(Dn, device="cuda") / s3g).cpu().numpy()
s = np.sqrt(W); tp_cpu = Dn / ((s * s) * s)
b1, b2 = np.flatnonzero(tf_gpu != tf_cpu), np.flatnonzero(tp_gpu != tp_cpu)
print("a/s: GPU != CPU:", b1.size, " hv/s3: GPU != CPU:", b2.size)
for i in b1[:10]:
    print("   w=%s s=%s a=%s gpu=%s cpu=%s" % (float(W[i]).hex(), float(s[i]).hex(), float(Nn[i]).hex(), float(tf_gpu[i]).hex(), float(tf_cpu[i]).hex()))
ws = np.array(H.fp64_sqrt_cases(nrandom=400))
print("sqrt hard cases", ws.size, "GPU != CPU:", int((torch.sqrt(torch.tensor(ws, device="cuda")).cpu().numpy() != np.sqrt(ws)).sum()))
