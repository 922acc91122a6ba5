"""In-tree build of librtmi.so (HIP kernels + C ABI) for gfx950.

    python -m raytracerfortran_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical
contract (see csrc/rtmi_kernels.hip): the kernels must round every product and
sum separately, exactly like the reference's gfortran build.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librtmi.so")
LIB7 = os.path.join(HERE, "librtmi_dff7.so")
SOURCES = [os.path.join(CSRC, "rtmi_kernels.hip"), os.path.join(CSRC, "rtmi_capi.cpp"),
           os.path.join(CSRC, "rtmi_pt.cpp")]
SHIM7 = os.path.join(CSRC, "rtmi_dff7_shim.c")
HEADERS = [os.path.join(CSRC, "rtmi_kernels.h"), os.path.join(CSRC, "rtmi_device.hpp"), os.path.join(os.path.dirname(HERE), "include", "rtmi.h")]

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ARCH = "gfx950"
FLAGS = ["-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-std=c++17",
         f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-result"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    deps = SOURCES + HEADERS + [os.path.abspath(__file__)]
    if force or _stale(LIB, deps):
        if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
            raise RuntimeError("hipcc not found: librtmi.so cannot be built and there is no CPU fallback")
        cmd = [HIPCC] + FLAGS + ["-o", LIB] + SOURCES
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    if force or _stale(LIB7, [SHIM7, LIB]):
        cc = shutil.which("gcc") or shutil.which("cc")
        cmd = [cc, "-O2", "-fPIC", "-shared", "-o", LIB7, SHIM7, f"-L{HERE}", "-lrtmi", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
