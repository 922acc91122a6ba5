#!/usr/bin/env python3
"""profiles/bench_with_lib.py <librtmi.so> <bench.py arguments...> -- bench.py against another build of the library."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracerfortran_amd import raymod
raymod.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
