"""Host-side mirror of the reference's `module raymod` call boundary, over librtmi.so.

The reference is called in two ways:

  R        .Fortran("dff", vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, keep_delta)
           (/root/reference/rayTracerR.R:31-33) -> C symbol `dff_`,
           /root/reference/subroutineR-quiet.f90:408
  Fortran  CALL TraceRays(vels, thickness, k-1, src_offset, src_depth, NSRC, DpredRT, -1)
           (/root/reference/ray_tracing_sampling/loglhood.f90:135) ->
           /root/reference/subroutineR-quiet.f90:467

`dff` and `TraceRays` below take the same arguments with the same meaning and go
through exactly those C symbols of librtmi.so (host buffers in and out), so the
parity tests read like the reference's own calls.  `trace_device` /
`trace_batch_device` are the device-resident entry points (HBM in, HBM out; torch
tensors are used only as owners of device memory and streams).

There is no CPU fallback: if librtmi.so is missing or the GPU path fails, these
functions raise.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtmi.so")
LIB7_PATH = os.path.join(_HERE, "librtmi_dff7.so")

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p

# Every symbol include/rtmi.h declares: (restype, argtypes).  tests/test_boundary.py
# checks this table against the header and the built library.
SIGNATURES = {
    "dff_": (None, [_dp, _dp, _ip, _dp, _dp, _ip, _dp, _ip]),
    "rtmi_tracerays": (None, [_dp, _dp, _ip, _dp, _dp, _ip, _dp, _ip]),
    "rtmi_dff7": (None, [_dp, _dp, _ip, _dp, _dp, _ip, _dp]),
    "rtmi_set_sources": (ctypes.c_int, [_dp, _dp, _dp, ctypes.c_int64]),
    "rtmi_sources_generation": (ctypes.c_uint64, []),
    "rtmi_loglhood_batch": (ctypes.c_int, [_dp, _dp, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int, _dp,
                                            _dp, _dp]),
    "rtmi_pt_create": (_vp, [ctypes.c_int, ctypes.c_int, ctypes.c_int64, _vp, ctypes.c_uint64, ctypes.c_uint64]),
    "rtmi_pt_destroy": (None, [_vp]),
    "rtmi_pt_set_chain": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_double]),
    "rtmi_pt_set_beta": (ctypes.c_int, [_vp, _dp]),
    "rtmi_pt_get_chain": (ctypes.c_int, [_vp, ctypes.c_int, _ip, _dp, _dp, _dp, _dp, _dp]),
    "rtmi_pt_get_all": (ctypes.c_int, [_vp, _ip, _dp, _dp, _dp, _dp, _dp, _dp]),
    "rtmi_pt_evaluate": (ctypes.c_int, [_vp]),
    "rtmi_pt_sweep": (ctypes.c_int, [_vp, ctypes.c_int]),
    "rtmi_pt_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "rtmi_pt_chain_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "rtmi_pt_likelihood_counts": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "rtmi_pt_stream_positions": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "rtmi_pt_speculation": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "rtmi_pt_set_ladder": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _ip, _dp, ctypes.c_int]),
    "rtmi_pt_set_ladder_betas": (ctypes.c_int, [_vp, _dp, ctypes.c_int]),
    "rtmi_pt_get_ladder": (ctypes.c_int, [_vp, _ip, ctypes.POINTER(ctypes.c_int64)]),
    "rtmi_pt_swap_round": (ctypes.c_int, [_vp, _vp]),
    "rtmi_pt_swap_round_host": (ctypes.c_int, [_vp, _dp]),
    "rtmi_pt_swap_decide": (ctypes.c_int, [ctypes.c_int, _dp, _ip, _dp, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64,
                                           ctypes.POINTER(ctypes.c_int64)]),
    "rtmi_pt_run": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int]),
    "rtmi_pt_sample_buffer": (ctypes.c_int, [_vp, ctypes.c_int]),
    "rtmi_pt_samples": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), _ip, _ip, _dp, _dp, _dp, _dp, _ip]),
    "rtmi_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "rtmi_comm_create": (_vp, [ctypes.c_int, ctypes.c_int, ctypes.c_char_p]),
    "rtmi_comm_destroy": (None, [_vp]),
    "rtmi_comm_size": (ctypes.c_int, [_vp]),
    "rtmi_comm_rccl_version": (ctypes.c_int, []),
    "rtmi_comm_rccl_library": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "rtmi_last_kernel_name": (ctypes.c_char_p, []),
    "rtmi_comm_rank": (ctypes.c_int, [_vp]),
    "rtmi_comm_allgather_f64": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, _vp]),
    "rtmi_init": (ctypes.c_int, [ctypes.c_int]),
    "rtmi_last_error": (ctypes.c_char_p, []),
    "rtmi_version": (ctypes.c_int, []),
    "rtmi_device_cus": (ctypes.c_int, []),
    "rtmi_device_status": (ctypes.c_int, []),
    "rtmi_clear_device_status": (ctypes.c_int, []),
    "rtmi_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "rtmi_trace_device": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp]),
    "rtmi_trace_device_f32": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp]),
    "rtmi_trace_device_f32_fast": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp]),
    "rtmi_trace_batch_device": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int64,
                                               _vp, _vp, _vp, _vp, _vp, _vp]),
    "rtmi_trace_batch_work_bytes": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int64]),
    "rtmi_debug_stamps": (ctypes.c_int, [_vp]),
    "rtmi_selftest_fastmath": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, _vp, _vp]),
    "rtmi_selftest_fastmath_ex": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, _vp, _vp]),
    "rtmi_selftest_values": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rtmi_selftest_values_f32": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    "rtmi_selftest_fastmath_f32": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, _vp, _vp]),
    "rtmi_write_rays_dat": (ctypes.c_int, [ctypes.c_char_p, _dp, _dp, ctypes.c_int, _dp, _dp, _dp, ctypes.c_int64]),
}

_lib = None


class RtmiError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen librtmi.so and bind the C ABI.  Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as the
    # system one, libamdhip64.so.7) and pull it in by file name; if librtmi.so were loaded first it would
    # bind the system runtime and torch would then bring a SECOND runtime into the process, which cannot
    # see the GPU ("No HIP GPUs are available").  Importing torch first makes the loader satisfy
    # librtmi.so's DT_NEEDED libamdhip64.so.7 with the runtime torch already loaded, so device pointers
    # and streams are shared.  Without torch the system runtime (/opt/rocm) is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise RtmiError(f"{p} not found: build it with `python -m raytracerfortran_amd.build` "
                        "(hipcc, gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        msg = load_library().rtmi_last_error().decode("utf-8", "replace")
        raise RtmiError(f"librtmi error {rc}: {msg}")


def init(device: int = 0) -> None:
    _check(load_library().rtmi_init(int(device)))


def set_option(name: str, value: int) -> None:
    """Scheduling knobs of trace_device ("path", "waves_per_simd", "refill_min"); see include/rtmi.h."""
    _check(load_library().rtmi_set_option(name.encode(), int(value)))


def _as_f64(a, n=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None and a.size < n:
        raise ValueError(f"{name} has {a.size} elements, needs {n}")
    return a


def _host_call(symbol, vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, keep_delta):
    lib = load_library()
    NLayers = int(NLayers)
    NSrc = int(NSrc)
    vels = _as_f64(vels, NLayers + 1, "vels")
    depths = _as_f64(depths, NLayers, "depths")
    # .Fortran hands the callee copies of every vector and returns the (modified) copies
    off = _as_f64(src_offset, NSrc, "src_offset").copy()
    dep = _as_f64(src_depth, NSrc, "src_depth").copy()
    t = _as_f64(timeP if timeP is not None else np.zeros(NSrc), NSrc, "timeP").copy()
    nl, ns = ctypes.c_int(NLayers), ctypes.c_int(NSrc)
    args = [vels.ctypes.data_as(_dp), depths.ctypes.data_as(_dp), ctypes.byref(nl), off.ctypes.data_as(_dp),
            dep.ctypes.data_as(_dp), ctypes.byref(ns), t.ctypes.data_as(_dp)]
    if keep_delta is not None:
        kd = ctypes.c_int(int(keep_delta))
        args.append(ctypes.byref(kd))
    # a device failure inside the void drop-in symbols NaN-fills the output (and aborts only with RTMI_ABORT_ON_ERROR=1):
    # make sure of the NaN-fill here so that Python can raise
    old = os.environ.get("RTMI_ABORT_ON_ERROR")
    os.environ["RTMI_ABORT_ON_ERROR"] = "0"
    try:
        getattr(lib, symbol)(*args)
    finally:
        if old is None:
            os.environ.pop("RTMI_ABORT_ON_ERROR", None)
        else:
            os.environ["RTMI_ABORT_ON_ERROR"] = old
    if NSrc > 0 and np.isnan(t[:NSrc]).all():
        msg = lib.rtmi_last_error().decode("utf-8", "replace")
        if msg:
            raise RtmiError(f"{symbol} failed: {msg}")
    out = {"vels": vels, "depths": depths, "NLayers": NLayers, "src_offset": off, "src_depth": dep, "NSrc": NSrc,
           "timeP": t}
    if keep_delta is not None:
        out["keep_delta"] = int(keep_delta)
    return out


def dff(vels, depths, NLayers, src_offset, src_depth, NSrc, timeP=None, keep_delta=-1):
    """R's `.Fortran("dff", vels=, depths=, NLayers=, src_offset=, src_depth=, NSrc=, timeP=, keep_delta=)`
    (rayTracerR.R:31-33).  Returns the same named list (a dict); `["timeP"]` holds the travel times.
    keep_delta > 0 also writes ./rays.dat like the reference."""
    return _host_call("dff_", vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, keep_delta)


def dff7(vels, depths, NLayers, src_offset, src_depth, NSrc, timeP=None):
    """The 7-argument `dff` of subroutineR.f90:405 (README.md:16-21, adaptMCMC.Rmd:74-78)."""
    return _host_call("rtmi_dff7", vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, None)


def TraceRays(vels, depths, NLayers, src_offset, src_depth, NSrc, timeP=None, keep_delta=-1):
    """`CALL TraceRays(vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, keep_delta)`
    (subroutineR-quiet.f90:467; loglhood.f90:135).  Returns timeP (the INTENT(OUT) argument)."""
    return _host_call("rtmi_tracerays", vels, depths, NLayers, src_offset, src_depth, NSrc, timeP,
                      keep_delta)["timeP"]


dofullforwardproblem = dff  # the Fortran-side name of `dff_` (subroutineR-quiet.f90:408)


def set_sources(src_offset, src_depth, obs=None) -> None:
    """READDATA's role (read_input.f90:302-387): sources and observed times go to HBM once (rtmi_set_sources)."""
    lib = load_library()
    off, dep = _as_f64(src_offset), _as_f64(src_depth)
    if off.size != dep.size:
        raise ValueError("src_offset / src_depth size mismatch")
    o = _as_f64(obs, off.size, "obs") if obs is not None else None
    _check(lib.rtmi_set_sources(off.ctypes.data_as(_dp), dep.ctypes.data_as(_dp),
                                o.ctypes.data_as(_dp) if o is not None else None, off.size))


def loglhood_batch(vels, depths, nlayers, sd, want_ss: bool = False):
    """LOGLHOOD for a batch of models over the resident sources (rtmi_loglhood_batch).
    vels, depths: [nmodels, ld] float64 host arrays; nlayers: [nmodels] int32; sd: [nmodels].
    Returns logL[nmodels] (and SS[nmodels] with want_ss)."""
    lib = load_library()
    vels = np.ascontiguousarray(vels, dtype=np.float64)
    depths = np.ascontiguousarray(depths, dtype=np.float64)
    nl = np.ascontiguousarray(nlayers, dtype=np.int32)
    sd = np.ascontiguousarray(sd, dtype=np.float64)
    nm, ld = vels.shape
    if depths.shape != (nm, ld) or nl.size != nm or sd.size != nm:
        raise ValueError("shape mismatch")
    logl = np.empty(nm)
    ss = np.empty(nm) if want_ss else None
    _check(lib.rtmi_loglhood_batch(vels.ctypes.data_as(_dp), depths.ctypes.data_as(_dp),
                                   nl.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ld, nm, sd.ctypes.data_as(_dp),
                                   logl.ctypes.data_as(_dp), ss.ctypes.data_as(_dp) if want_ss else None))
    return (logl, ss) if want_ss else logl


class PtPrior(ctypes.Structure):
    """rtmi_pt_prior of include/rtmi.h."""
    _fields_ = [("kmin", ctypes.c_int), ("kmax", ctypes.c_int), ("lam", ctypes.c_double), ("hmx", ctypes.c_double),
                ("hmin", ctypes.c_double), ("vmin", ctypes.c_double), ("vmax", ctypes.c_double),
                ("sdmin", ctypes.c_double), ("sdmax", ctypes.c_double), ("pertsdsc", ctypes.c_double),
                ("pertsdsdsc", ctypes.c_double)]


class Comm:
    """The RCCL communicator of the temperature exchange (rtmi_comm_*): one rank per GPU.  `bcast(bytes_or_None)`
    must return rank 0's bytes on every rank (torch.distributed.broadcast_object_list, MPI_Bcast ...)."""

    def __init__(self, nranks: int, rank: int, bcast):
        self.lib = load_library()
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            _check(self.lib.rtmi_comm_unique_id(buf))
        ident = bcast(bytes(buf.raw) if rank == 0 else None)
        self.h = self.lib.rtmi_comm_create(int(nranks), int(rank), ctypes.create_string_buffer(ident, 128))
        if not self.h:
            raise RtmiError("rtmi_comm_create failed: " + self.lib.rtmi_last_error().decode("utf-8", "replace"))
        self.nranks, self.rank = int(nranks), int(rank)

    def size(self) -> int:
        """Ranks of the communicator as the library reports it (rtmi_comm_size)."""
        return int(self.lib.rtmi_comm_size(self.h))

    def close(self):
        if self.h:
            self.lib.rtmi_comm_destroy(self.h)
            self.h = None


def comm_rccl_version() -> int:
    """ncclGetVersion of the RCCL librtmi.so's communicator binds to."""
    return int(load_library().rtmi_comm_rccl_version())


def comm_rccl_library() -> str:
    """Path of the shared object ncclAllGather resolves to inside librtmi.so."""
    buf = ctypes.create_string_buffer(1024)
    _check(load_library().rtmi_comm_rccl_library(buf, 1024))
    return buf.value.decode("utf-8", "replace")


def last_kernel_name() -> str:
    """Name (as rocprofv3 prints it) of the kernel this thread's last trace / batch launch dispatched."""
    return load_library().rtmi_last_kernel_name().decode("utf-8", "replace")


def swap_decide(logl_all, level, beta_of_level, seed: int, round_: int):
    """The exchange's pairing + acceptance rule on the host (rtmi_pt_swap_decide; no GPU): returns
    (new level array, accepted, proposed)."""
    lib = load_library()
    ll = _as_f64(logl_all)
    lev = np.ascontiguousarray(level, dtype=np.int32).copy()
    b = _as_f64(beta_of_level)
    prop = ctypes.c_int64(0)
    acc = lib.rtmi_pt_swap_decide(ll.size, ll.ctypes.data_as(_dp), lev.ctypes.data_as(_ip), b.ctypes.data_as(_dp), b.size,
                                  int(seed), int(round_), ctypes.byref(prop))
    if acc < 0:
        raise RtmiError(f"rtmi_pt_swap_decide: bad arguments ({acc})")
    return lev, int(acc), int(prop.value)


class NativeChains:
    """The chains of one process inside librtmi.so (rtmi_pt_*): Metropolis-Hastings sweeps resident on the GPU over a
    copy of the resident sources (set_sources with observations first)."""

    def __init__(self, nchains, ld, ndata, prior, seed=1, stream_id=0):
        self.lib = load_library()
        self.n, self.ld, self.kmax = int(nchains), int(ld), int(prior.kmax)
        pr = PtPrior(prior.kmin, prior.kmax, prior.lam, prior.hmx, prior.hmin, prior.vmin, prior.vmax, prior.sdmin,
                     prior.sdmax, prior.pertsdsc, prior.pertsdsdsc)
        self.h = self.lib.rtmi_pt_create(self.n, self.ld, int(ndata), ctypes.byref(pr), int(seed), int(stream_id))
        if not self.h:
            raise RtmiError("rtmi_pt_create: " + self.lib.rtmi_last_error().decode("utf-8", "replace"))
        self.nct = self.n

    def close(self):
        if self.h:
            self.lib.rtmi_pt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_chain(self, i, nodes, sd, beta):
        z, v = _as_f64(nodes[:, 0]), _as_f64(nodes[:, 1])
        _check(self.lib.rtmi_pt_set_chain(self.h, i, z.size, z.ctypes.data_as(_dp), v.ctypes.data_as(_dp), float(sd),
                                          float(beta)))

    def set_beta(self, beta):
        b = _as_f64(beta, self.n, "beta")
        _check(self.lib.rtmi_pt_set_beta(self.h, b.ctypes.data_as(_dp)))

    def get_chain(self, i):
        k = ctypes.c_int()
        z, v = np.empty(self.kmax), np.empty(self.kmax)
        sd, ll, ss = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(self.lib.rtmi_pt_get_chain(self.h, i, ctypes.byref(k), z.ctypes.data_as(_dp), v.ctypes.data_as(_dp),
                                          ctypes.byref(sd), ctypes.byref(ll), ctypes.byref(ss)))
        return np.column_stack([z[:k.value], v[:k.value]]), sd.value, ll.value, ss.value

    def get_all(self):
        """Every chain in one read-back: dict of k [n], z/v [n, kmax], sd, logL, ss, beta [n]."""
        n, K = self.n, self.kmax
        k = np.empty(n, np.int32)
        z, v = np.empty((n, K)), np.empty((n, K))
        sd, ll, ss, beta = np.empty(n), np.empty(n), np.empty(n), np.empty(n)
        _check(self.lib.rtmi_pt_get_all(self.h, k.ctypes.data_as(_ip), z.ctypes.data_as(_dp), v.ctypes.data_as(_dp),
                                        sd.ctypes.data_as(_dp), ll.ctypes.data_as(_dp), ss.ctypes.data_as(_dp),
                                        beta.ctypes.data_as(_dp)))
        return {"k": k, "z": z, "v": v, "sd": sd, "logL": ll, "ss": ss, "beta": beta}

    def logl(self):
        out = np.empty(self.n)
        _check(self.lib.rtmi_pt_get_all(self.h, None, None, None, None, out.ctypes.data_as(_dp), None, None))
        return out

    def evaluate(self):
        _check(self.lib.rtmi_pt_evaluate(self.h))

    def sweep(self, nsweeps=1):
        _check(self.lib.rtmi_pt_sweep(self.h, int(nsweeps)))

    def run(self, nsweeps, swap_every=0, comm=None, record_every=0):
        """nsweeps sweeps (+ swap rounds, + T = 1 samples into the ring) enqueued without host synchronisation."""
        _check(self.lib.rtmi_pt_run(self.h, int(nsweeps), int(swap_every), comm.h if comm is not None else None,
                                    int(record_every)))

    def set_ladder(self, nchains_total, first_chain, level, beta_of_level):
        lev = np.ascontiguousarray(level, dtype=np.int32)
        b = _as_f64(beta_of_level)
        _check(self.lib.rtmi_pt_set_ladder(self.h, int(nchains_total), int(first_chain), lev.ctypes.data_as(_ip),
                                           b.ctypes.data_as(_dp), b.size))
        self.nct = int(nchains_total)

    def set_ladder_betas(self, beta_of_level):
        """New inverse temperatures for the ladder's levels in place (levels, exchange counters and random stream kept)."""
        b = _as_f64(beta_of_level)
        _check(self.lib.rtmi_pt_set_ladder_betas(self.h, b.ctypes.data_as(_dp), b.size))

    def get_ladder(self):
        lev = np.empty(self.nct, np.int32)
        st = (ctypes.c_int64 * 2)()
        _check(self.lib.rtmi_pt_get_ladder(self.h, lev.ctypes.data_as(_ip), st))
        return lev, int(st[0]), int(st[1])

    def swap_round(self, comm=None):
        _check(self.lib.rtmi_pt_swap_round(self.h, comm.h if comm is not None else None))

    def swap_round_host(self, logl_all):
        ll = _as_f64(logl_all, self.nct, "logl_all")
        _check(self.lib.rtmi_pt_swap_round_host(self.h, ll.ctypes.data_as(_dp)))

    def sample_buffer(self, capacity):
        _check(self.lib.rtmi_pt_sample_buffer(self.h, int(capacity)))
        self._cap = int(capacity)

    def samples(self):
        """Drains the T = 1 sample ring: dict of sweep, chain, k, sd, logL [m], z / v [m, kmax], dropped."""
        cap, K = getattr(self, "_cap", 0), self.kmax
        sweep, chain, k = np.empty(cap, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int32)
        sd, ll = np.empty(cap), np.empty(cap)
        z, v = np.empty((cap, K)), np.empty((cap, K))
        dropped = ctypes.c_int(0)
        m = self.lib.rtmi_pt_samples(self.h, cap, sweep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                     chain.ctypes.data_as(_ip), k.ctypes.data_as(_ip), sd.ctypes.data_as(_dp),
                                     ll.ctypes.data_as(_dp), z.ctypes.data_as(_dp), v.ctypes.data_as(_dp),
                                     ctypes.byref(dropped))
        if m < 0:
            _check(-m)
        return {"sweep": sweep[:m], "chain": chain[:m], "k": k[:m], "sd": sd[:m], "logL": ll[:m], "z": z[:m], "v": v[:m],
                "dropped": int(dropped.value)}

    def likelihood_counts(self):
        """(sources of proposals traced, sources taken from the residual cache) since the handle was created."""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self.lib.rtmi_pt_likelihood_counts(self.h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def stream_positions(self):
        """Draws every chain's counter-based random stream has taken so far (part of the chain's state)."""
        out = (ctypes.c_uint64 * self.n)()
        _check(self.lib.rtmi_pt_stream_positions(self.h, out))
        return np.array(out[:], dtype=np.uint64)

    def speculation(self):
        """{rows: model rows per chain of the speculative step (1 = off), discarded: models evaluated for nothing,
        engine: -1 undecided / 0 per-slot launches / 1 persistent launch, sorted_sources}."""
        out = (ctypes.c_int64 * 4)()
        _check(self.lib.rtmi_pt_speculation(self.h, out))
        return {"rows": int(out[0]), "discarded": int(out[1]), "engine": int(out[2]), "sorted_sources": bool(out[3])}

    def stats(self):
        out = (ctypes.c_int64 * 8)()
        _check(self.lib.rtmi_pt_stats(self.h, out))
        names = ("acc_bd", "prop_bd", "acc_mh", "prop_mh", "acc_sd", "prop_sd", "launches", "models")
        return dict(zip(names, [int(x) for x in out]))


    def chain_stats(self):
        """[nchains][8]: the counters of stats() chain by chain (columns acc_bd, prop_bd, acc_mh, prop_mh, acc_sd, prop_sd,
        models decided, speculated models discarded)."""
        out = np.zeros((self.n, 8), dtype=np.int64)
        _check(self.lib.rtmi_pt_chain_stats(self.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        return out


def write_rays_dat(path, vels, depths, src_offset, src_depth, p_final) -> None:
    lib = load_library()
    vels, depths = _as_f64(vels), _as_f64(depths)
    off, dep, p = _as_f64(src_offset), _as_f64(src_depth), _as_f64(p_final)
    _check(lib.rtmi_write_rays_dat(os.fsencode(path), vels.ctypes.data_as(_dp), depths.ctypes.data_as(_dp),
                                   depths.size, off.ctypes.data_as(_dp), dep.ctypes.data_as(_dp),
                                   p.ctypes.data_as(_dp), off.size))


# ---------------------------------------------------------------------------------------------
# Device-resident entry points.  Tensors are torch CUDA(=HIP) float64 tensors; only their
# data_ptr() crosses the boundary.
# ---------------------------------------------------------------------------------------------
def _ptr(t, dtype=None, name="tensor"):
    if t is None:
        return None
    import torch
    if not t.is_cuda:
        raise ValueError(f"{name} must live in HBM")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def _stream_ptr(stream):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def trace_device(vels, depths, src_offset, src_depth, timeP=None, p_final=None, stream=None, fast=False):
    """One forward evaluation on resident data (rtmi_trace_device).  Asynchronous on `stream`
    (default: torch's current stream).  Returns timeP.  float32 tensors take rtmi_trace_device_f32 (the reference's
    iteration in single precision) or, with fast=True, rtmi_trace_device_f32_fast (the same equations solved to
    single-precision roundoff; not the reference's iterates -- see include/rtmi.h)."""
    import torch
    lib = load_library()
    nlayers = int(depths.numel())
    if vels.numel() != nlayers + 1:
        raise ValueError("vels must have NLayers+1 elements")
    n = int(src_offset.numel())
    if src_depth.numel() != n:
        raise ValueError("src_offset / src_depth size mismatch")
    f64 = src_offset.dtype  # torch.float64 (the reference's precision) or torch.float32 (rtmi_trace_device_f32)
    if f64 not in (torch.float64, torch.float32):
        raise ValueError("src_offset must be float64 or float32")
    if timeP is None:
        timeP = torch.empty(n, dtype=f64, device=src_offset.device)
    if timeP.numel() < n or (p_final is not None and p_final.numel() < n):
        raise ValueError("output too small")
    if fast and f64 != torch.float32:
        raise ValueError("fast=True is the single-precision fast path: pass float32 tensors")
    fn = lib.rtmi_trace_device if f64 == torch.float64 else (lib.rtmi_trace_device_f32_fast if fast else lib.rtmi_trace_device_f32)
    _check(fn(_ptr(vels, f64, "vels"), _ptr(depths, f64, "depths"), nlayers, _ptr(src_offset, f64, "src_offset"),
              _ptr(src_depth, f64, "src_depth"), n, _ptr(timeP, f64, "timeP"), _ptr(p_final, f64, "p_final"),
              _stream_ptr(stream)))
    return timeP


def trace_batch_device(vels, depths, nlayers, src_offset, src_depth, timeP=None, obs=None, sd=None,
                       want_timeP=True, stream=None):
    """Many models over shared sources (rtmi_trace_batch_device).

    vels, depths: [nmodels, ld] float64 (row m: nlayers[m]+1 velocities / nlayers[m] depths);
    nlayers: [nmodels] int32.  With obs [nsrc] and sd [nmodels] also returns logL [nmodels]
    (loglhood.f90:165-199).  Returns (timeP or None, logL or None).
    """
    import torch
    lib = load_library()
    nmodels, ld = int(vels.shape[0]), int(vels.shape[1])
    if tuple(depths.shape) != (nmodels, ld) or nlayers.numel() != nmodels:
        raise ValueError("vels/depths/nlayers shape mismatch")
    n = int(src_offset.numel())
    dev = src_offset.device
    f64 = torch.float64
    if want_timeP and timeP is None:
        timeP = torch.empty((nmodels, n), dtype=f64, device=dev)
    if timeP is not None and timeP.numel() < nmodels * n:
        raise ValueError("timeP too small")
    logL = work = None
    if obs is not None:
        if sd is None or sd.numel() != nmodels or obs.numel() != n:
            raise ValueError("obs/sd shape mismatch")
        logL = torch.empty(nmodels, dtype=f64, device=dev)
        wb = int(lib.rtmi_trace_batch_work_bytes(nmodels, n))
        work = torch.empty(max(wb // 8, 1), dtype=f64, device=dev)
        if stream is not None:
            work.record_stream(stream)
    _check(lib.rtmi_trace_batch_device(_ptr(vels, f64, "vels"), _ptr(depths, f64, "depths"),
                                       _ptr(nlayers, torch.int32, "nlayers"), ld, nmodels,
                                       _ptr(src_offset, f64, "src_offset"), _ptr(src_depth, f64, "src_depth"), n,
                                       _ptr(timeP, f64, "timeP"), _ptr(obs, f64, "obs"), _ptr(sd, f64, "sd"),
                                       _ptr(logL, f64, "logL"), _ptr(work, f64, "work"), _stream_ptr(stream)))
    return timeP, logL


def selftest_fastmath_f32(w, num, den):
    """(num/sqrt(w), den/sqrt(w)^3) mismatch counts of the single-precision guarded layer term vs sqrtf and the plain
    divides on the given float32 CUDA tensors (rtmi_selftest_fastmath_f32)."""
    import torch
    lib = load_library()
    n = int(w.numel())
    if num.numel() != n or den.numel() != n:
        raise ValueError("size mismatch")
    out = torch.zeros(2, dtype=torch.int64, device=w.device)
    f32 = torch.float32
    _check(lib.rtmi_selftest_fastmath_f32(_ptr(w, f32, "w"), _ptr(num, f32, "num"), _ptr(den, f32, "den"), n,
                                          _ptr(out, torch.int64, "out"), _stream_ptr(None)))
    torch.cuda.synchronize()
    return int(out[0]), int(out[1])


def selftest_values(w, num, den):
    """Values of the kernels' divide / guarded divide / guarded sqrt / layer term on the given CUDA tensors (float64:
    rtmi_selftest_values -> dict div, div_guarded, sqrt, tf, tp; float32: rtmi_selftest_values_f32 -> div, tf, tp).
    NaN marks operands a guard rejects (the kernels then take the plain operators)."""
    import torch
    lib = load_library()
    n = int(w.numel())
    if num.numel() != n or den.numel() != n:
        raise ValueError("size mismatch")
    dt = w.dtype
    names = ("div", "div_guarded", "sqrt", "tf", "tp") if dt == torch.float64 else ("div", "tf", "tp")
    out = {k: torch.empty(n, dtype=dt, device=w.device) for k in names}
    if dt == torch.float64:
        _check(lib.rtmi_selftest_values(_ptr(w, dt, "w"), _ptr(num, dt, "num"), _ptr(den, dt, "den"), n,
                                        *[_ptr(out[k], dt, k) for k in names], _stream_ptr(None)))
    else:
        _check(lib.rtmi_selftest_values_f32(_ptr(w, dt, "w"), _ptr(num, dt, "num"), _ptr(den, dt, "den"), n,
                                            *[_ptr(out[k], dt, k) for k in names], _stream_ptr(None)))
    torch.cuda.synchronize()
    return out


def selftest_fastmath_ex(w, num, den):
    """selftest_fastmath plus (term mismatches with the all-ones guard ignored, operands that guard rejected)
    (rtmi_selftest_fastmath_ex): 5 counts."""
    import torch
    lib = load_library()
    n = int(w.numel())
    if num.numel() != n or den.numel() != n:
        raise ValueError("size mismatch")
    out = torch.zeros(64, dtype=torch.int64, device=w.device)
    f64 = torch.float64
    _check(lib.rtmi_selftest_fastmath_ex(_ptr(w, f64, "w"), _ptr(num, f64, "num"), _ptr(den, f64, "den"), n,
                                         _ptr(out, torch.int64, "out"), _stream_ptr(None)))
    torch.cuda.synchronize()
    o = [int(x) for x in out]
    # (index, accepted by the guard) of the first operands whose fast-form term differs from the plain operators
    selftest_fastmath_ex.last_offenders = [(x & ((1 << 62) - 1), x < 0) for x in o[6:6 + min(o[5], 58)]]
    return tuple(o[:5])


def selftest_fastmath(w, num, den):
    """(sqrt, divide, shared-reciprocal layer-term) mismatch counts of the guarded fast forms vs the plain
    operators on the given float64 CUDA tensors (rtmi_selftest_fastmath)."""
    import torch
    lib = load_library()
    n = int(w.numel())
    if num.numel() != n or den.numel() != n:
        raise ValueError("size mismatch")
    out = torch.zeros(3, dtype=torch.int64, device=w.device)
    f64 = torch.float64
    _check(lib.rtmi_selftest_fastmath(_ptr(w, f64, "w"), _ptr(num, f64, "num"), _ptr(den, f64, "den"), n,
                                      _ptr(out, torch.int64, "out"), _stream_ptr(None)))
    torch.cuda.synchronize()
    return int(out[0]), int(out[1]), int(out[2])
