"""Code-generation guard (build container: needs hipcc; no GPU).  The fused kernels sit at the 128-register limit of four
waves per SIMD, and the persistent sweep kernel -- the likelihood body inlined into a slot loop -- is at the edge: a change that
is neutral in the source can make the compiler spill several times as many values to scratch inside the solve loop (round 3:
tempering sweep 0.92 -> 1.01 ms, found only by bisecting timings, profiles/r03g/pt_bisect.txt).  The static symptom is cheap to
check: the number of scratch instructions in the kernel's assembly."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
SRC = os.path.join(ROOT, "raytracerfortran_amd", "csrc", "rtmi_kernels.hip")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_hot_kernels_do_not_spill_to_scratch_in_their_loops(tmp_path):
    out = tmp_path / "k.s"
    subprocess.check_call([HIPCC, "-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c++17", "--offload-arch=gfx950",
                           "-S", "--cuda-device-only", "-o", str(out), SRC], stderr=subprocess.DEVNULL)
    text = out.read_text()

    def kernel(prefix):
        m = re.search(r"^%s[^\n]*:.*?s_endpgm" % re.escape(prefix), text, re.S | re.M)
        assert m, prefix
        return m.group(0)

    limits = {
        # C2's kernel (travel times only, row tables): no scratch at all
        "_ZN4rtmi12fused_kernelIdLb0ELb1ELb0ELb0ELi1ELb0EEE": 0,
        # the batch API's kernel with the fused logL
        "_ZN4rtmi12fused_kernelIdLb0ELb0ELb1ELb0ELi1ELb0EEE": 0,
        # the sampler's likelihood kernel (grid sums + residual cache): a handful around the loop, none in it
        "_ZN4rtmi12fused_kernelIdLb0ELb0ELb1ELb0ELi1ELb1EEE": 16,
        # the persistent sweep kernel, k <= 21 (prologue spills of its argument blocks: 18 when healthy, 82 when not)
        "_ZN4rtmi15pt_sweep_kernelILi1ELb1EEE": 40,
        # ... and for deeper models (form 0)
        "_ZN4rtmi15pt_sweep_kernelILi0ELb1EEE": 60,
    }
    for prefix, limit in limits.items():
        n = len(re.findall(r"^\s*scratch_", kernel(prefix), re.M))
        assert n <= limit, f"{prefix}: {n} scratch instructions (limit {limit}): the register allocation of this kernel regressed"
