"""Code-generation guard (build container: needs hipcc; no GPU).  The fused kernels sit at the 128-register limit of four
waves per SIMD, and the sweep kernels -- the likelihood body inlined into a task loop -- are at the edge: a change that is
neutral in the source can make the compiler spill several times as many values to scratch inside the solve loop (round 3:
tempering sweep 0.92 -> 1.01 ms, found only by bisecting timings, profiles/r03g/pt_bisect.txt).  The static symptoms are cheap
to check: scratch instructions in the kernel's assembly, and the register / spill counts of its metadata (a scalar spill is a
v_writelane / v_readlane pair, i.e. vector issue slots)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
SRC = os.path.join(ROOT, "raytracerfortran_amd", "csrc", "rtmi_kernels.hip")

# kernel -> (scratch instructions, vgpr_count, sgpr_spill_count, vgpr_spill_count): upper limits = the figures of the build
# whose timings are in profiles/r04*/ plus a small margin (exact counts move by a few with any edit)
LIMITS = {
    # C2's kernel (travel times only, row tables): no scratch at all                         [round 4: 0, 118, 78, 0]
    "_ZN4rtmi12fused_kernelIdLb0ELb1ELb0ELb0ELi1ELb0EEE": (0, 120, 90, 0),
    # the batch API's kernel with the fused logL                                              [0, 121, 100, 0]
    "_ZN4rtmi12fused_kernelIdLb0ELb0ELb1ELb0ELi1ELb0EEE": (0, 124, 112, 0),
    # the sampler's likelihood kernel (exact sums, residual cache, sorted sources): a handful around the loop, none in it [8, 128, 112, 9]
    "_ZN4rtmi12fused_kernelIdLb0ELb0ELb1ELb0ELi1ELb1EEE": (16, 128, 125, 14),
    # the speculative sweep (workgroups dealt to the chains in proportion to their slots), k <= 21: the layout, the step and
    # the handshakes are real calls and the loop state sits in LDS; what is left are the body's own        [120, 128, 193, 70]
    "_ZN4rtmi20pt_sweep_prop_kernelILi1EEE": (140, 128, 215, 85),
    # ... and for deeper models (table form 0)                                                [192, 128, 183, 115]
    "_ZN4rtmi20pt_sweep_prop_kernelILi0EEE": (220, 128, 210, 132),
    # the one-at-a-time persistent sweep (RTMI_PT_SPECULATE=0), k <= 21                       [30, 128, 213, 25]
    "_ZN4rtmi15pt_sweep_kernelILi1ELb1EEE": (40, 128, 235, 32),
    # ... and for deeper models: more scratch than round 3's (60 instructions) with the sorted-source form of the body, and still
    # 17 % faster than round 3's on config 4's share (profiles/r04c/pt_timings.txt), where it is the default engine
    "_ZN4rtmi15pt_sweep_kernelILi0ELb1EEE": (150, 128, 210, 100),
    # small launches, a group of 16 lanes per source (the reference's own example size): no scratch
    "_ZN4rtmi18trace_small_kernelILi16ELb0EEE": (0, 128, 40, 0),
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_hot_kernels_keep_their_registers_and_stay_out_of_scratch(tmp_path):
    out = tmp_path / "k.s"
    from raytracerfortran_amd import build as B          # the SHIPPED build's flags (incl. -enable-ipra=false, see build.py)
    subprocess.check_call([HIPCC] + B.CFLAGS + ["-S", "--cuda-device-only", "-o", str(out), SRC], stderr=subprocess.DEVNULL)
    text = out.read_text()
    meta = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)",
                         text, re.S):
        meta[m.group(1)] = (int(m.group(3)), int(m.group(2)), int(m.group(4)))   # vgpr, sgpr spills, vgpr spills

    def kernel(prefix):
        m = re.search(r"^%s[^\n]*:.*?s_endpgm" % re.escape(prefix), text, re.S | re.M)
        assert m, prefix
        return m.group(0)

    report = []
    for prefix, (max_scratch, max_vgpr, max_sspill, max_vspill) in LIMITS.items():
        full = [n for n in meta if n.startswith(prefix)]
        assert len(full) == 1, (prefix, full)
        vgpr, sspill, vspill = meta[full[0]]
        nscr = len(re.findall(r"^\s*scratch_", kernel(prefix), re.M))
        report.append((prefix, nscr, vgpr, sspill, vspill))
        assert nscr <= max_scratch, f"{prefix}: {nscr} scratch instructions (limit {max_scratch}): the register allocation of this kernel regressed"
        assert vgpr <= max_vgpr, f"{prefix}: {vgpr} VGPRs (limit {max_vgpr}: four waves per SIMD need <= 128)"
        assert sspill <= max_sspill, f"{prefix}: {sspill} spilled SGPRs (limit {max_sspill})"
        assert vspill <= max_vspill, f"{prefix}: {vspill} spilled VGPRs (limit {max_vspill})"
    print("\n".join("%-60s scratch %3d vgpr %3d sgpr spills %3d vgpr spills %3d" % r for r in report))
