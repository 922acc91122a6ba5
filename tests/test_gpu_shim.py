"""The Fortran caller path on the GPU box: shim/test_shim (amdflang-built in the build container, travels with
the snapshot) does `USE RAYMOD; CALL TraceRays(...)` like LOGLHOOD_RT (loglhood.f90:135) against shim/raymod.f90,
which forwards to librtmi.so through ISO_C_BINDING."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, compare_times

pytestmark = pytest.mark.gpu


def test_fortran_tracerays_caller(kat):
    exe = os.path.join(ROOT, "shim", "test_shim")
    if not os.path.exists(exe):
        pytest.skip("shim/test_shim not built (needs amdflang: make -C shim)")
    m = kat["model"]
    off, dep = kat["test_1_src_offset"], kat["test_1_src_depth"]
    lines = [f"{len(m['depths'])} {len(off)}", " ".join(repr(x) for x in m["vels"]),
             " ".join(repr(x) for x in m["depths"])] + [f"{o!r} {z!r}" for o, z in zip(off, dep)]
    p = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines_out = p.stdout.strip().splitlines()
    t = np.array([float(ln) for ln in lines_out[:len(off)]])      # the NSrc travel times come first (RCCL may chat later)
    assert compare_times(t, kat["config1_ref_timeP"], 1e-9) == 0.0
    # module rtmi_likelihood: rtmi_set_sources + rtmi_loglhood_batch from Fortran (obs = predictions -> SS = 0)
    tag, rc, logl, ss = [ln for ln in lines_out if ln.startswith("LOGL")][0].split()
    n = len(off)
    assert int(rc) == 0 and float(ss) == 0.0
    assert abs(float(logl) - (-0.5 * n * np.log(2 * np.pi) - n * np.log(0.02))) < 1e-10
    # module rtmi_sampler: the GPU-resident sweep + exchange through a one-rank RCCL communicator, driven from Fortran
    f = [ln for ln in lines_out if ln.startswith("PT")][0].split()
    assert int(f[1]) == 0, f
    assert abs(float(f[2]) - float(logl)) < 1e-9            # rtmi_pt_evaluate of the model = the batch likelihood above
    assert all(1 <= int(x) <= 10 for x in f[3:7])
    assert sorted(float(x) for x in f[7:11]) == [0.35, 0.5, 0.7, 1.0]   # the ladder is a permutation of itself
