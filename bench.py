#!/usr/bin/env python3
"""bench.py -- travel-times/sec of the `dff` forward model on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nsrc 1000000] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md 8(d) "C2"): one `dff` forward over 1e6 synthetic
sources and an 8-interface / 9-velocity model with one low-velocity zone, fp64, seed 12345.
A "step" is one pass of the hot path (rtmi_trace_device through the C ABI) over that batch, with
src_offset / src_depth / the model already resident in HBM and timeP left in HBM.

Multi-GPU (SURVEY.md 8(e), row "single forward"): sources are independent, so every rank traces
its own 1e6-source shard of the same model -- weak scaling, no data-path collective; the only
torch.distributed traffic is the timing barrier and the MAX-over-ranks of the elapsed time.

Output: ONE JSON line on rank 0 (contract in the task statement) with
  roofline      HBM view the north_star asks for: 24 algorithmic bytes per travel time
                (8 B offset + 8 B depth read, 8 B timeP written) / average kernel duration
                measured with HIP events on the launch stream, against 8 TB/s.  The kernel is
                fp64-VALU bound, so this fraction is small by construction;
                `valu_fp64` next to it prices the same launch against the fp64 VALU issue rate.
  cpu_baseline  the CPU restatement (oracle/, "port") timed on this box's host cores on the
                same workload -- baseline only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz
BYTES_PER_TT = 24.0


from raytracerfortran_amd.synthetic import c2_workload, c4_workload, TRUE_V, TRUE_D  # noqa: E402  (inputs only)


def launch_ranks(argv, ngpus: int, backend: str) -> int:
    """`python bench.py --gpus N` without torchrun: start N fresh child ranks (one per GPU) with
    `python -m torch.distributed.run` and return its exit code.  Called BEFORE anything in this process touches
    the GPU (torch.cuda.device_count() does not initialise it on this image); the children are new processes,
    nothing is re-exec'ed.  RCCL needs one device per rank: fewer visible devices is an error, not a silent
    1-GPU measurement.  RTMI_BENCH_BACKEND=gloo lets several ranks rehearse the N > 1 path on fewer devices."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < ngpus:
        print(f"[bench] --gpus {ngpus} but only {ndev} device(s) visible: refusing to measure fewer GPUs than asked "
              f"(set RTMI_BENCH_BACKEND=gloo to rehearse {ngpus} ranks on {ndev} device(s))", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def host_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota if one is set
    (a 1-GPU box exposes all host CPUs in the mask but grants a 16-CPU share)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // p))
            break
        except Exception:
            continue
    return min(n, int(os.environ.get("RTMI_BENCH_MAX_THREADS", "64")))


def measured_counters(kernel_substr: str = "fused_kernel"):
    """Per-launch PMC figures of the dominant kernel from the COMMITTED rocprofv3 passes of this same command
    (profiles/<tag>/pmc_summary.json, newest tag) -- figures of another run on another box, possibly of an earlier build:
    returned with the file's own record of the build it measured (`build_commit`, written by profiles/summarize.py).
    (HBM bytes with FETCH_SIZE doubled per the gfx950 correction, VALU wave-instructions, source file, build commit).
    None if absent."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        commit = d.get("build_commit") if isinstance(d.get("build_commit"), str) else None
        for k, c in d.items():
            if isinstance(c, dict) and (k == kernel_substr or (kernel_substr in k and "(" not in kernel_substr)) and "hbm_bytes_per_launch" in c:
                valu = c.get("SQ_INSTS_VALU", {}).get("mean_per_launch")
                best = (c["hbm_bytes_per_launch"]["total"], valu, os.path.relpath(path, ROOT), commit)
    return best


def measured_traffic(kernel_substr: str = "fused_kernel"):
    m = measured_counters(kernel_substr)
    return None if m is None else (m[0], m[2], m[3])


def c2_roofline(n: int, kern_ms: float, kernel_name: str):
    """The `roofline` and `valu_fp64` objects of the C2 line.  Measured in this run: the kernel's average duration (HIP events on
    the launch stream) and its name (from the library's own dispatch).  NOT measured in this run, and labelled so: HBM traffic and
    VALU instruction counts, which come from the committed rocprofv3 --pmc passes of this same command (another box, possibly an
    earlier build -- the file's `build_commit` is quoted)."""
    achieved = BYTES_PER_TT * n / (kern_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "kernel": kernel_name, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": BYTES_PER_TT * n,
            "note": "fp64-VALU-bound root-find: 24 B per travel time cannot approach the HBM roof; see DESIGN.md section 4"}
    tr = measured_traffic(kernel_name)   # (the summary holds other instantiations too -- the sampler's -- : the exact name)
    if tr is not None and n == 1_000_000:
        roof["traffic"] = tr[0]
        roof["traffic_source"] = ("NOT MEASURED IN THIS RUN: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                  "command, %s, build %s" % (tr[1], tr[2] or "not recorded in that file"))
    # the roof that actually binds: the fp64 vector ALU.  One wave-instruction occupies a SIMD for 4 cycles (v_rsq/v_rcp_f64:
    # 16), i.e. 16 lanes per clock and SIMD = FP64_VALU_PEAK_TFLOPS / 2 lane-instructions per second over the chip.
    peak_lane_instr = FP64_VALU_PEAK_TFLOPS * 1e12 / 2.0
    valu = {"peak_tflops": FP64_VALU_PEAK_TFLOPS, "peak_lane_instr_per_s": peak_lane_instr,
            "note": "fp64 VALU issue is the binding roof (DESIGN.md section 4.2)"}
    mc = measured_counters(kernel_name)
    if mc is not None and mc[1] and n == 1_000_000 and kern_ms:
        valu["executed_valu_wave_instr_per_launch"] = mc[1]
        valu["executed_frac_of_peak"] = mc[1] * 64.0 / (kern_ms * 1e-3) / peak_lane_instr
        valu["source"] = "NOT MEASURED IN THIS RUN: %s, build %s (with this run's kernel time)" % (mc[2], mc[3] or "not recorded in that file")
    return roof, valu


def level_hash(level) -> int:
    """CRC-32 of a temperature-level vector (int32, global chain order): equal on every rank iff the ranks took the same
    swap decisions."""
    import zlib
    return int(zlib.crc32(np.ascontiguousarray(level, dtype=np.int32).tobytes()))


def hashes_agree(h: int, dist, world: int, device="cpu") -> bool:
    """All-reduce (MIN and MAX) of a rank's hash: True iff every rank holds the same value."""
    if dist is None or world == 1:
        return True
    import torch
    lo = torch.tensor([h], dtype=torch.int64, device=device)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return int(lo[0]) == int(hi[0])


def chain_parallel_block(level, rccl_ranks, world, elapsed_s, sweeps, models, nsrc, nloc, dist, red_device, rccl_info, engine, exchange):
    """The `chain_parallel` object of the bench line (BASELINE configs 3/4: chains sharded over ranks, one logL all-gather
    per swap round): what an N-GPU line must show for the north_star's chain-parallel scaling claim -- the communicator's size
    as the LIBRARY reports it, the time per sweep (MAX over ranks), and proof that every rank ended with the same ladder."""
    h = level_hash(level)
    return {"rccl_ranks": int(rccl_ranks), "ranks": int(world), "chains": int(nloc * world), "chains_per_gpu": int(nloc),
            "sources": int(nsrc), "sweeps": int(sweeps), "ms_per_sweep": elapsed_s / max(1, sweeps) * 1e3,
            "models_per_s": models / elapsed_s, "travel_times_per_s": models * nsrc / elapsed_s,
            "level_hash": h, "level_hash_equal_across_ranks": hashes_agree(h, dist, world, red_device),
            "swap_every": 1, "exchange": exchange, "engine": engine, **rccl_info}


def fenced(fn, limit_s, on_timeout):
    """Run fn() under a fence: -> (fn(), False), or ({"error": ...}, True) if it raised.  If it has not returned after limit_s
    seconds, a watchdog thread calls on_timeout({"error": ...}) (rank 0: write the line without the block) and ends THIS process
    with exit code 0 -- every rank carries the same fence, so a set of ranks waiting for each other all leave."""
    import threading
    import traceback
    done = threading.Event()

    def watch():
        if not done.wait(limit_s):
            if on_timeout is not None:
                on_timeout({"error": "no result after %.0f s (RTMI_BENCH_CP_TIMEOUT_S): block abandoned, ranks ended" % limit_s})
            sys.stdout.flush()
            os._exit(0)

    threading.Thread(target=watch, daemon=True).start()
    try:
        return fn(), False
    except BaseException as e:  # a rank that fails must still let rank 0 write the headline
        if isinstance(e, KeyboardInterrupt):
            raise
        sys.stderr.write("[bench] chain_parallel failed:\n" + traceback.format_exc())
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}, True
    finally:
        done.set()


def cpu_baseline(v, d, off, dep, budget_s: float = 12.0):
    """Time the CPU restatement on this host: all cores (threads over source slices; the C code
    holds no shared state and ctypes releases the GIL) and one core.  Bounded to ~budget_s."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O

    n = off.size
    cores = host_cores()
    O.oracle_dff(v, d, off[:1000], dep[:1000])  # load + warm
    # single thread, on a sample sized to ~3 s
    n1 = min(n, 1_000_000)
    t0 = time.perf_counter()
    O.oracle_dff(v, d, off[:n1], dep[:n1])
    t_single = time.perf_counter() - t0
    single = n1 / t_single
    # all cores: repeat the full workload until the budget is used
    nthreads = max(1, cores)
    bounds = np.linspace(0, n, nthreads * 4 + 1).astype(int)

    def work(i):
        O.oracle_dff(v, d, off[bounds[i]:bounds[i + 1]], dep[bounds[i]:bounds[i + 1]])

    reps = 0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthreads) as ex:
        while True:
            list(ex.map(work, range(nthreads * 4)))
            reps += 1
            if time.perf_counter() - t0 > budget_s - t_single or reps >= 200:
                break
    t_all = time.perf_counter() - t0
    out = {"value": reps * n / t_all, "unit": "travel-times/s", "cores": nthreads, "kind": "port",
           "sample": f"{reps} x the full {n}-source C2 workload, {nthreads} threads over source slices "
                     f"(oracle/dff_oracle.c, gcc -O2 -ffp-contract=off)",
           "single_thread_value": single}
    # the reference's own Fortran (oracle/_ref, built in the build container), one core
    if O.have_ref():
        try:
            t0 = time.perf_counter()
            O.ref_dff(v, d, off[:n1], dep[:n1])
            out["reference_single_thread_value"] = n1 / (time.perf_counter() - t0)
            out["reference_build"] = "subroutineR-quiet.f90, amdflang -O2 (oracle/Makefile)"
        except Exception as e:  # pragma: no cover
            out["reference_error"] = str(e)
    return out


def dropin_figures(rt, v, d, off, dep, ref_t=None):
    """PCIe-inclusive cost of the drop-in symbol itself: `dff_` called through the C ABI on pageable host arrays, the
    way R's .Fortran or a Fortran caller does (no Python-side copies: the arrays are allocated once, ctypes passes
    their addresses).  Reported next to the device-resident headline, never as `value`."""
    import ctypes
    lib = rt.load_library()
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)

    def caller(nl, off_a, dep_a):
        n = off_a.size
        t = np.zeros(n)
        cn, cl, kd = ctypes.c_int(n), ctypes.c_int(nl), ctypes.c_int(-1)
        a = (v.ctypes.data_as(dp), d.ctypes.data_as(dp), ctypes.byref(cl), off_a.ctypes.data_as(dp),
             dep_a.ctypes.data_as(dp), ctypes.byref(cn), t.ctypes.data_as(dp), ctypes.byref(kd))
        return (lambda: lib.dff_(*a)), t

    def median_ms(fn, reps, warm):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3, float(np.min(ts)) * 1e3

    out = {}
    n = off.size
    big, t_big = caller(d.size, off, dep)
    med, best = median_ms(big, 15, 3)
    out["dff_1e6_ms" if n == 1_000_000 else f"dff_{n}_ms"] = med
    out["dff_large_best_ms"] = best
    out["dff_large_travel_times_per_s"] = n / (med * 1e-3)
    out["pcie_floor_ms"] = 24.0 * n / 63e9 * 1e3
    if ref_t is not None:
        out["dff_large_bit_equal_to_device_path"] = bool(np.array_equal(t_big, ref_t))
    # fresh arrays on every call (R's .Fortran hands over newly allocated copies): the pages have to be pinned anew
    fresh, keep = [], []
    for _ in range(8):
        o2, z2 = off.copy(), dep.copy()
        f, t2 = caller(d.size, o2, z2)
        keep.append((o2, z2, t2))
        t0 = time.perf_counter()
        f()
        fresh.append(time.perf_counter() - t0)
    out["dff_large_fresh_arrays_ms"] = float(np.median(fresh)) * 1e3
    del keep
    small, _ = caller(d.size, np.ascontiguousarray(off[:20]), np.ascontiguousarray(dep[:20]))
    med20, best20 = median_ms(small, 300, 20)
    out["dff_nsrc20_us"] = med20 * 1e3
    out["dff_nsrc20_best_us"] = best20 * 1e3
    if off.size >= 10_000:
        # the size of the reference sampler's own TraceRays calls (loglhood.f90:135 with NSRC = 1e4): the mapped-buffer path
        mid, _ = caller(d.size, np.ascontiguousarray(off[:10_000]), np.ascontiguousarray(dep[:10_000]))
        med1e4, _ = median_ms(mid, 100, 10)
        out["dff_nsrc1e4_us"] = med1e4 * 1e3
    out["devices"] = os.environ.get("RTMI_DEVICES", "bound device only")
    out["note"] = ("host buffers in, host buffers out through the C symbol dff_ (pageable memory, PCIe inclusive); "
                   "median of 15 calls at the bench size on the same arrays (a Fortran caller's static buffers), of 8 calls on "
                   "newly allocated arrays (R's .Fortran copies), of 300 calls at the reference's example size NSrc = 20, of 100 calls at "
                   "NSrc = 1e4")
    return out


def timed_steps(step, steps, warmup, world, dist, device, sync, make_event=None, stream=None, run_steps=None):
    """The timing contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by a barrier and a
    device synchronise on both sides; returns (max-over-ranks wall seconds, max-over-ranks average
    per-launch milliseconds from HIP events recorded on the launch stream, or None without events).
    `dist` is torch.distributed (initialised) or None; works with gloo on CPU for the tests.
    `run_steps(n)`, when given, enqueues n steps in one call (a device-resident loop) instead of n calls of `step`."""
    import torch

    def barrier():
        if dist is not None and world > 1:
            dist.barrier()

    if run_steps is not None:
        if warmup:
            run_steps(warmup)
        step = None
    for _ in range(warmup if step is not None else 0):
        step()
    sync()
    barrier()
    ev0 = ev1 = None
    if make_event is not None:
        ev0, ev1 = make_event(), make_event()
    sync()
    t0 = time.perf_counter()
    if ev0 is not None:
        ev0.record(stream)
    if run_steps is not None:
        run_steps(steps)
    else:
        for _ in range(steps):
            step()
    if ev1 is not None:
        ev1.record(stream)
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / steps if ev0 is not None else -1.0
    if dist is not None and world > 1:
        red_dev = device if dist.get_backend() == "nccl" else "cpu"
        tt = torch.tensor([elapsed, kern_ms], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(tt[0]), float(tt[1])
    return elapsed, (kern_ms if kern_ms >= 0 else None)


def bench_c4(args, rt, torch, dist, rank, world, dev):
    """BASELINE config 4's per-GPU share as a driver-runnable number: 32 models drawn from the prior with
    k in [2, 30] over 1e5 shared sources, one rtmi_trace_batch_device launch with the fused Gaussian logL per step
    (what one proposal slot of the 256-chain ladder costs each GPU; travel times are not materialised, as in the
    sampler).  Models differ per rank (seed + rank), sources are replicated: chains shard, no collective."""
    nsrc = 100_000 if args.nsrc == 1_000_000 else args.nsrc
    nm = 32
    w = c4_workload(nm, nsrc, seed=404 + 1000 * rank)
    f64 = torch.float64
    t = lambda a: torch.tensor(a, device=dev)
    dV, dD, dNL, do, dz, dsd = t(w["V"]), t(w["D"]), t(w["NL"]), t(w["off"]), t(w["dep"]), t(w["sd"])
    true_t = rt.trace_device(torch.tensor(TRUE_V, device=dev, dtype=f64), torch.tensor(TRUE_D, device=dev, dtype=f64), do, dz)
    dobs = true_t + t(w["obs_noise"])
    stream = torch.cuda.current_stream()
    out = {}

    def step():
        out["L"] = rt.trace_batch_device(dV, dD, dNL, do, dz, obs=dobs, sd=dsd, want_timeP=False, stream=stream)[1]

    elapsed, kern_ms = timed_steps(step, args.steps, args.warmup, world, dist if world > 1 else None, dev,
                                   sync=torch.cuda.synchronize,
                                   make_event=lambda: torch.cuda.Event(enable_timing=True), stream=stream)
    kernel_name = rt.last_kernel_name()
    if rank == 0:
        value = world * nm * nsrc * args.steps / elapsed
        algo = 16.0 * nm * nsrc + 8.0 * nm   # shared sources re-read per model, one logL per model
        achieved = algo / (kern_ms * 1e-3) / 1e9
        res = {"metric": "travel-times/sec", "value": value, "unit": "travel-times/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "ranks": world, "devices": args.devices, "collective_backend": args.backend,
               "config": {"workload": "C4 per-GPU share: 32 models (prior draws, k in [2,30]) x %d shared sources, travel "
                                      "times + fused logL in one launch; BASELINE.json configs[3]" % nsrc,
                          "models_per_gpu": nm, "nsrc": nsrc, "k_range": [int(w["NL"].min()) + 1, int(w["NL"].max()) + 1],
                          "sharding": "chains (models) per rank, sources replicated, no collective in the step"},
               "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms,
                            "kernel": kernel_name + " + rtmi::loglike_kernel",
                            "algorithmic_bytes_per_launch": algo,
                            "note": "fp64-VALU-bound root-find (DESIGN.md section 4)"},
               "logL_finite": bool(torch.isfinite(out["L"]).all().item())}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def bench_c5(args, rt, torch, dist, rank, world, dev):
    """BASELINE config 5 (fp32 kernel path: 1e7 sources x 20 interfaces over 8 GPUs, delta-t vs fp64 reported) as a
    driver-runnable number: every rank traces its own 1.25e6-source shard (1e7 / 8; weak scaling, no collective) of the
    20-interface model with the single-precision FAST forward model (rtmi_trace_device_f32_fast); the line also carries the
    fp32 path that repeats the reference's iteration in float, the fp64 path on the same (float-rounded) inputs, and the
    deviation of both fp32 paths from the fp64 result."""
    n = 1_250_000 if args.nsrc == 1_000_000 else args.nsrc
    rng = np.random.default_rng(2025)
    d = np.cumsum(rng.uniform(200, 600, 20))
    v = np.sort(rng.uniform(1500, 6500, 21))
    rng = np.random.default_rng([2025, rank])
    off, dep = rng.uniform(10, 8000, n), rng.uniform(50, d[-1] + 500, n)
    f32 = [torch.tensor(np.asarray(a, np.float32), device=dev) for a in (v, d, off, dep)]
    f64 = [t.double() for t in f32]
    out = torch.empty(n, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        rt.trace_device(*f32, timeP=out, stream=stream, fast=True)

    elapsed, kern_ms = timed_steps(step, args.steps, args.warmup, world, dist if world > 1 else None, dev,
                                   sync=torch.cuda.synchronize,
                                   make_event=lambda: torch.cuda.Event(enable_timing=True), stream=stream)
    if rank == 0:
        def ms_of(fn, reps=10):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                r = fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps, r
        ms64, t64 = ms_of(lambda: rt.trace_device(*f64))
        ms32, t32 = ms_of(lambda: rt.trace_device(*f32))
        t64 = t64.cpu().numpy()

        def dev_of(t):
            t = t.cpu().numpy().astype(np.float64)
            both = (t64 != -999.0) & (t != -999.0)
            e = np.abs(t[both] - t64[both])
            return {"median_s": float(np.median(e)), "p999_s": float(np.percentile(e, 99.9)), "max_s": float(e.max()),
                    "sentinels": int((t == -999.0).sum())}
        value = world * n * args.steps / elapsed
        algo = 12.0 * n
        achieved = algo / (kern_ms * 1e-3) / 1e9
        res = {"metric": "travel-times/sec", "value": value, "unit": "travel-times/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "ranks": world, "devices": args.devices, "collective_backend": args.backend,
               "config": {"workload": "C5: fp32 fast forward model, %d sources per GPU x 20 interfaces (1e7 over 8 GPUs), seed 2025; "
                                      "BASELINE.json configs[4]" % n,
                          "nsrc_per_gpu": n, "nlayers": 20, "sharding": "sources per rank, no collective"},
               "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms,
                            "kernel": "rtmi::f32fast_kernel", "algorithmic_bytes_per_launch": algo,
                            "note": "12 B per travel time; fp32-VALU-bound (about 900 instructions per 64 sources), DESIGN.md section 4.3"},
               "delta_t_vs_fp64": {"fp32_fast": dev_of(out), "fp32_reference_iteration": dev_of(t32),
                                   "fp64_sentinels": int((t64 == -999.0).sum()),
                                   "note": "same float-rounded inputs; the reference stops 0.1 m short of the root (up to 6.7e-5 s), "
                                           "the fast path solves to single-precision roundoff (1e-6 s of the converged solution)"},
               "other_paths": {"fp64_ms": ms64, "fp64_travel_times_per_s": n / ms64 * 1e3,
                               "fp32_reference_iteration_ms": ms32, "fp32_reference_iteration_travel_times_per_s": n / ms32 * 1e3}}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def rccl_info(rt):
    """Which RCCL the library's communicator binds to (a process holds torch's bundled copy too): version and the path of the
    shared object that ncclAllGather resolves to INSIDE librtmi.so."""
    try:
        return {"rccl_version": rt.comm_rccl_version(), "rccl_library": rt.comm_rccl_library()}
    except Exception as e:  # pragma: no cover
        return {"rccl_version": None, "rccl_library": "unknown (%s)" % e}


def pt_problem(torch, rt, dev, nsrc, seed=2024):
    """Synthetic data of the sampler workloads: the reference's example geometry and 7-layer model + N(0, 0.016 s)."""
    rng = np.random.default_rng(seed)
    x, y = rng.uniform(10, 6500, nsrc), rng.uniform(10, 4500, nsrc)
    off, dep = np.sqrt(x * x + y * y), rng.uniform(1050, 4200, nsrc)
    f64 = torch.float64
    # observations = forward(true model) + N(0, 0.016 s), generated with the HIP path itself (synthetic data)
    tt = rt.trace_device(torch.tensor(TRUE_V, device=dev, dtype=f64), torch.tensor(TRUE_D, device=dev, dtype=f64),
                         torch.tensor(off, device=dev), torch.tensor(dep, device=dev)).cpu().numpy()
    return off, dep, tt + rng.normal(0, 0.016, nsrc)


def run_pt(rt, torch, dist, rank, world, dev, nsrc, kmax, steps, warmup):
    """`steps` timed sweeps of 32 tempered chains per rank over nsrc shared sources, a swap round after every sweep.  Returns
    a dict with the wall time (MAX over ranks), the sampler's counters over the timed region and the final ladder."""
    from raytracerfortran_amd import tempering as T
    off, dep, obs = pt_problem(torch, rt, dev, nsrc)
    ev = T.GpuEvaluator(off, dep, obs)
    nloc = 32
    s = T.NativeSampler(T.Prior(kmax=kmax), ev, ndata=nsrc, nchains_total=nloc * world, nptchains1=world, dTlog=1.15, seed=7,
                        rank=rank, world=world, dist=dist if world > 1 else None,
                        dist_device=str(dev) if (world > 1 and dist.get_backend() == "nccl") else "cpu")
    s.init_from_model(np.column_stack([[0.0] + TRUE_D, TRUE_V]), 0.016)
    device_loop = world == 1 or s.comm is not None
    if device_loop:
        # sweeps + swap rounds enqueued back to back by ONE library call per timed region: no host synchronisation
        # between the first launch and the last (rtmi_pt_run); the exchange is the library's RCCL all-gather
        def run_steps(n):
            s.native.run(n, 1, s.comm, 0)
    else:
        # rehearsal of N ranks without RCCL (gloo): the gather of every swap round goes through the host
        def run_steps(n):
            for _ in range(n):
                s.sweep()
                s.swap_round()
    # Setup, untimed: a handle chooses its sweep engine by a timed trial in its first long call (ten sweeps, half of them on the
    # engine it may then discard, once the sweeps have their full length: rtmi_pt_run) -- that belongs to creating the handle,
    # like capturing a graph, not to the K steps this bench times.
    settle = kmax + 22
    if device_loop:
        run_steps(settle)
    run_steps(warmup)
    st0 = s.native.stats()
    tr0, ca0 = s.native.likelihood_counts()
    sp0 = s.native.speculation()
    elapsed, _ = timed_steps(None, steps, 0, world, dist if world > 1 else None, dev, sync=torch.cuda.synchronize, run_steps=run_steps)
    st1 = s.native.stats()
    tr1, ca1 = s.native.likelihood_counts()
    sp1 = s.native.speculation()
    s.sync()
    red = dev if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    tot = np.array([st1["models"] - st0["models"], tr1 - tr0, ca1 - ca0, sp1["discarded"] - sp0["discarded"]], dtype=np.float64)
    if world > 1:
        tm = torch.tensor(tot, dtype=torch.float64, device=red)
        dist.all_reduce(tm)
        tot = tm.cpu().numpy()
    launches = st1["launches"] - st0["launches"]
    eng = sp1["engine"]   # what the handle runs: 1 persistent, 0 per-slot launches, -1 still undecided (no long call yet)
    multi = eng == 1 and launches * 4 < max(1, steps)   # the library put many sweeps into one launch (every chain local, one chain per level)
    engine = ("persistent cooperative launches of many sweeps each, the temperature exchange inside the kernel (chain meets partner)" if multi else
              "one persistent cooperative launch per sweep" if eng == 1 else
              ("two launches per proposal slot / pass, replayed as a hipGraph" if eng == 0 or launches > 2 * max(1, steps)
               else "undecided (per-slot launches so far)"))
    exchange = ("RCCL all-gather of 8 B per chain (rtmi_pt_swap_round)" if (s.comm is not None and world > 1)
                else ("every chain local: no collective" if world == 1 else "host gather (gloo rehearsal)"))
    return {"elapsed": elapsed, "models": float(tot[0]), "traced": float(tot[1]), "cached": float(tot[2]), "discarded": float(tot[3]),
            "launches": launches, "engine": engine, "engine_trial_sweeps": settle if device_loop else 0, "exchange": exchange, "level": s.level, "nloc": nloc, "speculation_rows": sp1["rows"],
            "rccl_ranks": s.comm.size() if s.comm is not None else 1, "swap_accept_rate": s.stats["acc_swap"] / max(1, s.stats["prop_swap"]),
            "red_device": red}


def bench_pt(args, rt, torch, dist, rank, world, dev):
    """Parallel-tempering sweeps (raytracerfortran_amd/tempering.py, device-resident sweep csrc/rtmi_pt.hip): 32 chains per GPU over 1e4 shared sources
    (BASELINE config 3 on one GPU; config 4's sharding with N GPUs), k in [1, 10], swap round every sweep with the
    logL all-gather over RCCL.  A "step" is one sweep of every chain; value = travel times TRACED per second (sources a
    proposal does not affect are served from the chain's residual cache and are not counted; `delivered` counts them)."""
    nsrc = 10_000 if args.nsrc == 1_000_000 else args.nsrc
    r = run_pt(rt, torch, dist, rank, world, dev, nsrc, args.kmax, args.steps, args.warmup)
    if rank == 0:
        elapsed, models = r["elapsed"], r["models"]
        seen = r["traced"] + r["cached"]
        traced_tt = r["traced"] if seen > 0 else models * nsrc   # (incremental likelihood off: every source of every model is traced)
        res = {"metric": "travel-times/sec", "value": traced_tt / elapsed, "unit": "travel-times/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "ranks": world, "devices": args.devices, "collective_backend": args.backend,
               "delivered_travel_times_per_s": models * nsrc / elapsed,
               "config": {"workload": "PT-RJMCMC sweeps: 32 tempered chains per GPU x %d shared sources, k in [1,%d], "
                                      "swap round every sweep (BASELINE configs 3/4 shape)" % (nsrc, args.kmax),
                          "chains": r["nloc"] * world, "models_decided": models, "models_evaluated_and_discarded": r["discarded"],
                          "speculation_rows_per_chain": r["speculation_rows"],
                          "launches_per_step": r["launches"] / max(1, args.steps),
                          "engine_trial_sweeps_before_warmup": r["engine_trial_sweeps"],
                          "engine": "device-resident sweep (rtmi_pt_run), " + r["engine"] + ", swap exchange: " + r["exchange"],
                          "swap_accept_rate": r["swap_accept_rate"],
                          "sources_retraced_fraction": (r["traced"] / seen) if seen > 0 else 1.0,
                          "likelihood": ("full: every proposal traced over all sources" if os.environ.get("RTMI_PT_INCREMENTAL") == "0"
                                         else "incremental: sources above the depth a proposal changes keep the residuals of the "
                                              "chain's current model (bit-identical).  `value` counts the travel times actually "
                                              "traced (including those of speculated proposals that were discarded); "
                                              "`delivered_travel_times_per_s` = decided models x sources, what a sampler that "
                                              "evaluates every proposal in full has to trace for the same chain")},
               "chain_parallel": chain_parallel_block(r["level"], r["rccl_ranks"], world, elapsed, args.steps, models, nsrc, r["nloc"],
                                                      dist if world > 1 else None, r["red_device"], rccl_info(rt), r["engine"], r["exchange"])}
        print(json.dumps(res), flush=True)
    else:
        chain_parallel_block(r["level"], r["rccl_ranks"], world, r["elapsed"], args.steps, r["models"], nsrc, r["nloc"],
                             dist if world > 1 else None, r["red_device"], {}, r["engine"], r["exchange"])  # (takes part in the all-reduce)
    if world > 1:
        dist.destroy_process_group()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nsrc", type=int, default=1_000_000)
    ap.add_argument("--kmax", type=int, default=10, help="pt: largest number of nodes of a chain (config 3: 10, config 4: 30)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the PCIe-inclusive drop-in figures (c2 only)")
    ap.add_argument("--no-chain-parallel", action="store_true",
                    help="c2: skip the chain_parallel block (100 sweeps of the tempering sampler with the RCCL swap exchange)")
    ap.add_argument("--check", action="store_true", help="also compare one step with the oracle")
    ap.add_argument("--workload", choices=["c2", "pt", "c4", "c5"], default="c2",
                    help="c2: single dff forward, 1e6 sources (the bench metric's configuration, default); "
                         "pt: parallel-tempering sweeps, 32 chains per GPU x 1e4 sources (BASELINE config 3 shape); "
                         "c4: batched likelihood of config 4's per-GPU share, 32 models (k in [2,30]) x 1e5 sources; "
                         "c5: fp32 fast forward model, 1.25e6 sources per GPU x 20 interfaces (config 5's share), with the "
                         "deviation from fp64")
    ap.add_argument("--path", choices=["auto", "plain", "fused"], default="auto",
                    help="kernel path of rtmi_trace_device (A/B runs; auto = what the library picks)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.path != "auto":
        os.environ["RTMI_FORCE_PATH"] = args.path  # read once by librtmi.so
    backend = os.environ.get("RTMI_BENCH_BACKEND", "nccl")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher of N ranks; this process never touches the GPU
        raise SystemExit(launch_ranks(sys.argv[1:], args.gpus, backend))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"[bench] {world} ranks but {ndev} device(s): RCCL needs one device per rank")
    local_rank = local_rank % max(1, ndev)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL by default; RTMI_BENCH_BACKEND=gloo lets several ranks rehearse the N > 1 path on ONE GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world
    args.devices = min(world, ndev)
    args.backend = ("rccl" if backend == "nccl" else backend) if world > 1 else None

    import raytracerfortran_amd as rt
    rt.load_library()
    rt.init(local_rank)

    if args.workload == "pt":
        return bench_pt(args, rt, torch, dist, rank, world, dev)
    if args.workload == "c4":
        return bench_c4(args, rt, torch, dist, rank, world, dev)
    if args.workload == "c5":
        return bench_c5(args, rt, torch, dist, rank, world, dev)

    n = args.nsrc
    v, d, off, dep = c2_workload(n, shard=rank)
    dv, dd = torch.tensor(v, device=dev), torch.tensor(d, device=dev)
    do, dz = torch.tensor(off, device=dev), torch.tensor(dep, device=dev)
    dt = torch.empty(n, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        rt.trace_device(dv, dd, do, dz, timeP=dt, stream=stream)

    elapsed, kern_ms = timed_steps(step, args.steps, args.warmup, world, dist if world > 1 else None, dev,
                                   sync=torch.cuda.synchronize,
                                   make_event=lambda: torch.cuda.Event(enable_timing=True), stream=stream)

    kernel_name = rt.last_kernel_name()  # what the timed launches dispatched (before anything else launches)
    check = None
    if args.check and rank == 0:
        from oracle import oracle as O
        ref = O.oracle_dff(v, d, off, dep)
        got = dt.cpu().numpy()
        okm = ref != -999.0
        check = {"max_abs_dt": float(np.abs(got[okm] - ref[okm]).max()),
                 "sentinel_mismatch": int(((got == -999.0) != (ref == -999.0)).sum())}

    # The path's one exchange step, made visible to the scaling run (BASELINE configs 3/4): after the C2
    # measurement every rank runs 100 sweeps of its 32 tempered chains over 1e4 shared sources through rtmi_pt_run with the
    # LIBRARY's communicator -- one RCCL all-gather of 8 B per chain per swap round -- and the line carries what came of it.
    # The block is fenced: the headline above is already measured when it starts, and the library's RCCL exchange has never run
    # on more than one GPU before the driver's scaling run -- an error in it, or ranks that wait for each other for ever, must
    # cost the line its `chain_parallel` object (which then says what happened), not the line.
    def line(cp, full=True):
        """Rank 0's one JSON line.  `full` = with the drop-in figures and the CPU baseline (left out when the fence writes the
        line: the card may be the thing that hangs)."""
        value = world * n * args.steps / elapsed
        res = {
            "metric": "travel-times/sec", "value": value, "unit": "travel-times/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "ranks": world, "devices": args.devices, "collective_backend": args.backend,
            "config": {"workload": "C2: single dff forward, 1e6 sources x 8 interfaces (9 velocities, one LVZ), "
                                   "seed 12345; BASELINE.json configs[1]",
                       "nsrc_per_gpu": n, "nlayers": 8, "sharding": "sources per rank, no collective"},
        }
        res["roofline"], res["valu_fp64"] = c2_roofline(n, kern_ms, kernel_name)
        if check is not None:
            res["check_vs_oracle"] = check
        if cp is not None:
            res["chain_parallel"] = cp
        if full and world == 1 and not args.no_dropin:
            res["dropin"] = dropin_figures(rt, v, d, off, dep, dt.cpu().numpy())
        if full and world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(v, d, off, dep)
        return json.dumps(res)

    def chain_parallel():
        r = run_pt(rt, torch, dist, rank, world, dev, 10_000, 10, 100, 20)
        cp = chain_parallel_block(r["level"], r["rccl_ranks"], world, r["elapsed"], 100, r["models"], 10_000, r["nloc"],
                                  dist if world > 1 else None, r["red_device"], rccl_info(rt) if rank == 0 else {}, r["engine"], r["exchange"])
        if world == 1 and rank == 0 and cp is not None:
            # One rank runs many sweeps per launch with the exchange inside the kernel; several ranks cannot (a pair may straddle
            # two GPUs): they take one launch per sweep and the all-gather between launches.  The single-rank cost of THAT road,
            # for whoever compares an N-GPU line with this one:
            old = os.environ.get("RTMI_PT_MULTI")
            os.environ["RTMI_PT_MULTI"] = "0"
            try:
                r1 = run_pt(rt, torch, dist, rank, world, dev, 10_000, 10, 100, 20)
                cp["ms_per_sweep_one_launch_per_sweep"] = r1["elapsed"] / 100 * 1e3
            finally:
                if old is None:
                    os.environ.pop("RTMI_PT_MULTI", None)
                else:
                    os.environ["RTMI_PT_MULTI"] = old
        return cp

    cp, broken = None, False
    if not args.no_chain_parallel:
        cp, broken = fenced(chain_parallel, float(os.environ.get("RTMI_BENCH_CP_TIMEOUT_S", "120")),
                            (lambda c: print(line(c, full=False), flush=True)) if rank == 0 else None)
    if rank == 0:
        print(line(cp, full=not broken), flush=True)
    if broken:
        # peers may still sit in the exchange this rank left: no collective tear-down, every rank leaves on its own
        sys.stdout.flush()
        os._exit(0)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
