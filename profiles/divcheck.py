import numpy as np, torch
ks = np.arange(0, 40)
d = np.nextafter(2.0 ** (-ks.astype(np.float64)), 0.0)      # all-ones mantissa
spec = np.concatenate([2.0 ** np.arange(-20, 21), np.arange(1, 100, dtype=np.float64)])
D, A = np.meshgrid(d, spec)
D = D.ravel(); A = A.ravel()
cpu = A / D
gpu = (torch.tensor(A, device="cuda") / torch.tensor(D, device="cuda")).cpu().numpy()
print("GPU IEEE divide vs CPU divide on all-ones denominators: mismatches", int((cpu != gpu).sum()), "of", cpu.size)
w = np.nextafter(np.float64(1.0), 0.0)
print("sqrt:", np.sqrt(w) == torch.sqrt(torch.tensor([w], device="cuda")).item())
