#!/usr/bin/env python3
"""profiles/bench_with_lib.py <librtmi.so> <bench.py arguments...> -- bench.py against another build of the library."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracerfortran_amd import raymod
raymod.LIB_PATH = os.path.abspath(sys.argv[1])
import ctypes
import torch  # noqa: F401  (first: one HIP runtime per process, see raymod.load_library)
_probe = ctypes.CDLL(raymod.LIB_PATH)   # an older build may lack newer entry points: they are not what is being measured
for _name in [n for n in raymod.SIGNATURES if not hasattr(_probe, n)]:
    del raymod.SIGNATURES[_name]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
