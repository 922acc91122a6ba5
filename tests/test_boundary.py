"""CPU-side checks of the drop-in boundary (no GPU compute): the C-ABI library loads and exports
every symbol include/rtmi.h declares, the Python mirror binds exactly those, the product never
touches the oracle, the host-only rays.dat writer reproduces the author's file, and the
multi-rank timing harness of bench.py works over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from raytracerfortran_amd import build
    build.build()
    return build


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dff_|rtmi_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    from raytracerfortran_amd import raymod
    syms = header_symbols()
    assert "dff_" in syms and "rtmi_trace_device" in syms and len(syms) >= 11
    lib = ctypes.CDLL(built.LIB)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rtmi.h but not exported by librtmi.so"
    assert sorted(raymod.SIGNATURES) == syms
    lib7 = ctypes.CDLL(built.LIB7)
    assert hasattr(lib7, "dff_")  # the 7-argument flavour (subroutineR.f90:405)
    assert raymod.load_library().rtmi_version() == 400


def test_gpu_code_object_is_gfx950(built):
    out = subprocess.run(["strings", "-a", built.LIB], capture_output=True, text=True).stdout
    assert "gfx950" in out
    assert "trace_kernel" in out


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under raytracerfortran_amd/, include/ or shim/ may
    import, link or mention it."""
    bad = []
    for base in ("raytracerfortran_amd", "include", "shim"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".c", ".h", ".f90")) or f == "Makefile":
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"\boracle\b|liboracle|_ref/|libdff_ref", txt):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    ldd = subprocess.run(["ldd", os.path.join(ROOT, "raytracerfortran_amd", "librtmi.so")],
                         capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "dff_ref" not in ldd


def test_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import raytracerfortran_amd as r
    with pytest.raises(r.RtmiError, match="no HIP device|failed"):
        r.dff([3000.0, 4000.0], [1000.0], 1, [100.0], [2000.0], 1)
    # and the void drop-in symbol itself: the reason on stderr and NaN in every output (no silent garbage; an R session that
    # loaded the library survives), or a hard stop with RTMI_ABORT_ON_ERROR=1
    code = ("import ctypes,numpy as np;from raytracerfortran_amd import raymod as m;"
            "r=m.load_library();"
            "a=np.array([3000.,4000.]);d=np.array([1000.]);o=np.array([1.]);z=np.array([2000.]);t=np.zeros(1);"
            "n=ctypes.c_int(1);dp=ctypes.POINTER(ctypes.c_double);"
            "r.dff_(a.ctypes.data_as(dp),d.ctypes.data_as(dp),ctypes.byref(n),o.ctypes.data_as(dp),"
            "z.ctypes.data_as(dp),ctypes.byref(n),t.ctypes.data_as(dp),ctypes.byref(n));print('survived', t[0])")
    env = {k: v for k, v in os.environ.items() if k != "RTMI_ABORT_ON_ERROR"}
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env)
    assert p.returncode == 0 and "survived nan" in p.stdout
    assert "librtmi: dff_ failed" in p.stderr
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env={**env, "RTMI_ABORT_ON_ERROR": "1"})
    assert p.returncode != 0 and "survived" not in p.stdout
    assert "librtmi: dff_ failed" in p.stderr


def test_missing_library_raises(tmp_path):
    from raytracerfortran_amd import raymod
    with pytest.raises(raymod.RtmiError, match="no CPU fallback"):
        raymod.load_library(str(tmp_path / "librtmi.so"))


def test_rays_dat_writer_matches_authors_file(built, kat, tmp_path):
    """rtmi_write_rays_dat is host-only formatting (subroutineR-quiet.f90:157-164).  Fed with the oracle's
    final ray parameters it must reproduce /root/reference/rays.dat (KAT-B) token for token, one record
    per line, in gfortran's list-directed layout."""
    from oracle import oracle as O
    import raytracerfortran_amd as r
    m = kat["model"]
    off, dep = kat["test_1_src_offset"], kat["test_1_src_depth"]
    _, p, nl = O.oracle_dff(m["vels"], m["depths"], off, dep, want_p=True)
    path = tmp_path / "rays.dat"
    r.write_rays_dat(str(path), m["vels"], m["depths"], off, dep, p)
    lines = path.read_text().splitlines()
    assert len(lines) == 40
    assert [len(ln.split()) for ln in lines] == [int(n) for n in nl for _ in (0, 1)]
    ours = np.array([float(x) for ln in lines for x in ln.split()])
    ref = np.array(kat["kat_b"]["rays_dat_tokens"])
    assert (np.abs(ours - ref) / np.abs(ref)).max() < 5e-13
    # layout of a gfortran list-directed REAL(8) record: 17 significant digits, fields 26 wide
    # (the author's file has "   1000.0000000000000        268.48688478348777     " here; the committed
    # test_1_src_data.txt carries 15 digits of the depth, hence the last digits of the second field)
    assert lines[1].startswith("   1000.0000000000000        268.486884783")
    assert re.fullmatch(r"(   \d{4}\.\d{13}     |   \d{3}\.\d{14}     )+", lines[1])
    assert all(len(ln) == 26 * len(ln.split()) for ln in lines)
    # a top-layer source: the reference writes the offset, then the source depth (:98-105)
    r.write_rays_dat(str(path), m["vels"], m["depths"], [123.5], [400.0], [float("nan")])
    a, b = path.read_text().splitlines()
    assert float(a) == 123.5 and float(b) == 400.0


def _gloo_worker(rank, world, port, q):
    import time
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.002 * (rank + 1))  # rank 1 is the slow one

    elapsed, kern = bench.timed_steps(step, steps=10, warmup=2, world=world, dist=dist, device="cpu",
                                      sync=lambda: None)
    v0, d0, off0, _ = bench.c2_workload(1000, shard=0)
    v1, d1, off1, _ = bench.c2_workload(1000, shard=rank)
    q.put((rank, elapsed, kern, len(calls), bool(np.array_equal(v0, v1)), bool(np.array_equal(off0, off1))))
    dist.destroy_process_group()


def test_bench_timing_harness_two_ranks_gloo():
    """bench.py's N > 1 path on CPU: barrier + MAX over ranks, same model and different source shards per rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, e0, k0, c0, samev0, sameoff0), (r1, e1, k1, c1, samev1, sameoff1) = res
    assert e0 == e1 and e0 >= 10 * 0.004  # every rank reports the slowest rank's time
    assert k0 is None and c0 == c1 == 12  # exactly warmup + steps calls
    assert samev0 and samev1               # one model on every rank
    assert sameoff0 and not sameoff1       # a different source shard on rank 1


def test_bench_gpus_n_self_launches_n_ranks(monkeypatch):
    """`python bench.py --gpus N` without torchrun becomes the launcher of N fresh ranks (never a 1-rank measurement
    labelled N): the command it builds, and its refusal when RCCL is asked for more ranks than devices."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert bench.launch_ranks(["--gpus", "4", "--steps", "7"], 4, "nccl") == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # fewer devices than ranks: RCCL refuses (exit code 2, nothing launched), gloo rehearsal is allowed
    seen.clear()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert bench.launch_ranks(["--gpus", "4"], 4, "nccl") == 2 and not seen
    assert bench.launch_ranks(["--gpus", "2"], 2, "gloo") == 0 and "--nproc-per-node=2" in seen["cmd"]


def test_bench_gpus_n_exits_nonzero_without_devices():
    """End to end on this CPU container: `python bench.py --gpus 2` must fail loudly, not print a 1-GPU line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RTMI_BENCH_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 2 and "refusing to measure fewer GPUs" in p.stderr and not p.stdout.strip()
    # under a launcher with the wrong world size it refuses as well
    env.update(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=3 but --gpus 2" in p.stderr


@pytest.mark.skipif(not (os.path.exists("/root/reference/ray_tracing_sampling/loglhood.f90") and
                         os.path.exists("/opt/rocm/bin/amdflang")),
                    reason="needs the reference's sampler sources and a Fortran compiler (build container only)")
def test_reference_loglhood_compiles_unchanged_against_the_shim(tmp_path):
    """the reference's only caller of TraceRays, ray_tracing_sampling/loglhood.f90 (USE RAYMOD,
    :42; CALL TraceRays :135,144), compiles UNCHANGED against shim/raymod.f90 -- and its object then asks for the shim's
    module procedure, which in turn asks for the C ABI.  The sources are compiled where they lie (nothing is copied);
    tests/fortran/mpi_stub.f90 stands in for `USE MPI` (constants only: this is a compile check, nothing is run)."""
    ref = "/root/reference/ray_tracing_sampling"
    fc = "/opt/rocm/bin/amdflang"
    units = [os.path.join(ROOT, "tests", "fortran", "mpi_stub.f90"), f"{ref}/data_type.f90", f"{ref}/rjmcmc_com.f90",
             f"{ref}/quicksort.f90", os.path.join(ROOT, "shim", "raymod.f90"), f"{ref}/loglhood.f90"]
    for u in units:
        obj = os.path.basename(u) + ".o"
        p = subprocess.run([fc, "-c", "-w", u, "-o", obj], cwd=tmp_path, capture_output=True, text=True)
        assert p.returncode == 0, f"{u}:\n{p.stderr[-1500:]}"
    nm = lambda o: subprocess.run(["nm", o], cwd=tmp_path, capture_output=True, text=True).stdout
    caller, shim = nm("loglhood.f90.o"), nm("raymod.f90.o")
    sym = [ln.split()[-1] for ln in caller.splitlines() if "tracerays" in ln.lower()]
    assert sym and all(f" T {s}" in shim for s in sym), (sym, shim)       # the caller's reference is the shim's definition
    assert " U rtmi_tracerays" in shim                                     # which forwards to the C ABI
