"""A/B of the drop-in host path (dff_ on pageable host arrays, 1e6 sources): chunk size of the upload/kernel/download
pipeline, caller-buffer registration, pipelines per device.  One child process per setting (the environment is read
once per process).  python profiles/dropin_modes.py > gpurun_out/<tag>/dropin_modes.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json, time, ctypes, numpy as np
sys.path.insert(0, %r)
import raytracerfortran_amd as rt
from raytracerfortran_amd.synthetic import c2_workload
rt.init(0)
lib = rt.load_library()
n = int(sys.argv[1])
v, d, off, dep = c2_workload(n)
t = np.zeros(n)
dp = ctypes.POINTER(ctypes.c_double)
cn, cl, kd = ctypes.c_int(n), ctypes.c_int(8), ctypes.c_int(-1)
a = (v.ctypes.data_as(dp), d.ctypes.data_as(dp), ctypes.byref(cl), off.ctypes.data_as(dp), dep.ctypes.data_as(dp),
     ctypes.byref(cn), t.ctypes.data_as(dp), ctypes.byref(kd))
for _ in range(4):
    lib.dff_(*a)
ts = []
for _ in range(25):
    t0 = time.perf_counter(); lib.dff_(*a); ts.append(time.perf_counter() - t0)
print(json.dumps({"median_ms": float(np.median(ts)) * 1e3, "min_ms": float(np.min(ts)) * 1e3}))
""" % ROOT


def run(n, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    p = subprocess.run([sys.executable, "-c", CHILD, str(n)], capture_output=True, text=True, env=e)
    try:
        return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception:
        return {"error": p.stderr[-300:]}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    for reg in (0, 1):
        for chunk in (1 << 21, 1 << 19, 1 << 18, 1 << 17, 1 << 16):
            r = run(n, RTMI_DROPIN_CHUNK=chunk, RTMI_DROPIN_REGISTER=reg)
            print(f"n={n} register={reg} chunk={chunk:8d}  {r}", flush=True)
    for chunk in (1 << 19, 1 << 18):
        r = run(n, RTMI_DROPIN_CHUNK=chunk, RTMI_DEVICES="0,0", RTMI_SHARD_MIN=1000)
        print(f"n={n} two pipelines on one device, chunk={chunk:8d}  {r}", flush=True)
