#!/usr/bin/env python3
"""profiles/codegen_stats.py [file.s] -- register / spill / scratch figures of the hot kernels from the assembly hipcc emits
(build container, no GPU): the cheap check before a timing run (register allocation is a cliff)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracerfortran_amd", "csrc", sys.argv[2] if len(sys.argv) > 2 else "rtmi_kernels.hip")
out = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else os.path.join(tempfile.gettempdir(), "rtmi_k.s")
if not (len(sys.argv) > 1 and sys.argv[1] != "-" and os.path.exists(out)):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c++17",
                           "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
text = open(out).read()
meta = {}
for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+)\s+\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", text, re.S):
    meta[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
print("%-70s %5s %6s %5s %6s %8s %6s" % ("kernel", "sgpr", "sspill", "vgpr", "vspill", "scratch", "insts"))
for name in sorted(meta):
    if not any(k in name for k in ("fused_kernel", "pt_sweep", "pt_step")):
        continue
    body = re.search(r"^%s:.*?s_endpgm" % re.escape(name), text, re.S | re.M)
    nscr = len(re.findall(r"^\s*scratch_", body.group(0), re.M)) if body else -1
    nin = len(re.findall(r"^\s+[a-z_0-9]+ ", body.group(0), re.M)) if body else -1
    print("%-70s %5d %6d %5d %6d %8d %6d" % ((name[:70],) + meta[name] + (nscr, nin)))
