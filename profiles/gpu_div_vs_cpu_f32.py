"""Is the GPU's own fp32 divide correctly rounded?  Every float divisor s = sqrtf(w), w in (1e-12, 1], with a numerator
constructed to put the quotient next to a rounding midpoint (tests/test_gpu_fastmath_hard.py: hard_num), against numpy
float32 on the host (IEEE hardware divide)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
src = open(os.path.join(ROOT, "tests", "test_gpu_fastmath_hard.py")).read()
start = src.index("    def hard_num(divisor):"); end = src.index("    lo = int(np.float32(1.0000001e-12)")
ns = {}
exec("import torch\n" + "\n".join(l[4:] for l in src[start:end].splitlines()), ns)
hard_num = ns["hard_num"]
lo = int(np.float32(1.0000001e-12).view(np.uint32)); hi = int(np.float32(1.0).view(np.uint32))
bad = tot = 0
step = 1 << 25
for startb in range(lo, hi + 1, step):
    b = torch.arange(startb, min(startb + step, hi + 1), dtype=torch.int64, device="cuda").to(torch.int32)
    s = torch.sqrt(b.view(torch.float32))
    num, _ = hard_num(s)
    q = (num / s).cpu().numpy()
    ref = num.cpu().numpy() / s.cpu().numpy()
    bad += int((q.view(np.uint32) != ref.view(np.uint32)).sum()); tot += q.size
    # all-ones neighbourhood explicitly: mantissa 2^24 - M, M = 1..64, every exponent of the range
print("fp32 divide on the GPU: %d of %d constructed quotients differ from the IEEE host result" % (bad, tot))
