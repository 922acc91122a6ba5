"""ctypes loaders for the CHECKERS under oracle/ -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under raytracerfortran_amd/ does (tests/test_boundary.py
greps for it).

  oracle_dff(...)       this repo's scalar C restatement (oracle/dff_oracle.c)
  ref_dff(...)          the reference's own Fortran `dff_`
                        (/root/reference/subroutineR-quiet.f90:408) compiled by
                        oracle/Makefile into oracle/_ref/ -- present only where
                        that build ran (the build container) or travelled to.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "liboracle_dff.so")
_ORACLE_F32_SO = os.path.join(_HERE, "liboracle_dff_f32.so")
_REF_SO = os.path.join(_HERE, "_ref", "libdff_ref.so")
_REF_O0_SO = os.path.join(_HERE, "_ref", "libdff_ref_O0.so")

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int_p = ctypes.POINTER(ctypes.c_int)


class OracleStats(ctypes.Structure):
    """Mirror of `oracle_stats` in dff_oracle.c (SURVEY.md section 8(a) branch table)."""

    _fields_ = [(n, ctypes.c_int64) for n in (
        "top_layer", "p0_halvings", "cf_negative", "safe_jump", "bisect", "newton_iters",
        "bisect_trips", "clamps", "not_conv", "f_evals", "fp_evals")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force: bool = False) -> None:
    """Compile the C restatement (always) and oracle/_ref (only where possible)."""
    src_m = os.path.getmtime(os.path.join(_HERE, "dff_oracle.c"))
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < src_m for so in (_ORACLE_SO, _ORACLE_F32_SO)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if os.path.exists("/root/reference/subroutineR-quiet.f90") and (force or not os.path.exists(_REF_SO)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None


def _oracle_lib():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_ORACLE_SO)
        lib.oracle_dff.argtypes = [_c_double_p, _c_double_p, ctypes.c_int, _c_double_p, _c_double_p,
                                   ctypes.c_int64, _c_double_p, _c_double_p, ctypes.POINTER(ctypes.c_int32)]
        lib.oracle_dff.restype = None
        lib.oracle_dff_stats.argtypes = [_c_double_p, _c_double_p, ctypes.c_int, _c_double_p, _c_double_p,
                                         ctypes.c_int64, _c_double_p, ctypes.POINTER(OracleStats)]
        lib.oracle_dff_stats.restype = None
        lib.oracle_ray_segments.argtypes = [_c_double_p, _c_double_p, ctypes.c_int, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_double, _c_double_p, _c_double_p]
        lib.oracle_ray_segments.restype = ctypes.c_int
        lib.oracle_loglhood_rt.argtypes = [_c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int]
        lib.oracle_loglhood_rt.restype = ctypes.c_double
        lib.oracle_ss_batch.argtypes = [_c_double_p, _c_double_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int64,
                                        _c_double_p, _c_double_p, _c_double_p, ctypes.c_int64, _c_double_p]
        lib.oracle_ss_batch.restype = None
        _lib = lib
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_c_double_p)


def oracle_dff(vels, depths, src_offset, src_depth, want_p: bool = False):
    """Travel times of the C restatement.  Returns timeP or (timeP, p_final, nl)."""
    lib = _oracle_lib()
    vels, depths, off, dep = _f64(vels), _f64(depths), _f64(src_offset), _f64(src_depth)
    assert vels.size == depths.size + 1 and off.size == dep.size
    n = off.size
    t = np.empty(n, dtype=np.float64)
    if want_p:
        p = np.empty(n, dtype=np.float64)
        nl = np.empty(n, dtype=np.int32)
        lib.oracle_dff(_p(vels), _p(depths), depths.size, _p(off), _p(dep), n, _p(t), _p(p),
                       nl.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        return t, p, nl
    lib.oracle_dff(_p(vels), _p(depths), depths.size, _p(off), _p(dep), n, _p(t), None, None)
    return t


def oracle_dff_stats(vels, depths, src_offset, src_depth):
    lib = _oracle_lib()
    vels, depths, off, dep = _f64(vels), _f64(depths), _f64(src_offset), _f64(src_depth)
    t = np.empty(off.size, dtype=np.float64)
    st = OracleStats()
    lib.oracle_dff_stats(_p(vels), _p(depths), depths.size, _p(off), _p(dep), off.size, _p(t), ctypes.byref(st))
    return t, st.as_dict()


def oracle_ray_segments(vels, depths, src_offset, src_depth, p_final):
    """(d, h) records of one ray as the reference writes them to rays.dat."""
    lib = _oracle_lib()
    vels, depths = _f64(vels), _f64(depths)
    d = np.empty(depths.size + 1)
    h = np.empty(depths.size + 1)
    n = lib.oracle_ray_segments(_p(vels), _p(depths), depths.size, float(src_offset), float(src_depth),
                                float(p_final), _p(d), _p(h))
    return d[:n].copy(), h[:n].copy()


def oracle_loglhood_rt(obs, pred, sd, stable: bool = False) -> float:
    lib = _oracle_lib()
    obs, pred = _f64(obs), _f64(pred)
    return float(lib.oracle_loglhood_rt(_p(obs), _p(pred), obs.size, float(sd), int(stable)))


_lib32 = None


def oracle_ss_batch(V, D, NL, src_offset, src_depth, obs):
    """SUM(DresRT**2) of every model row over one source set (one LOGLHOOD_RT call per model, loglhood.f90:128-199)."""
    V, D = _f64(V), _f64(D)
    NL = np.ascontiguousarray(NL, dtype=np.int32)
    off, dep, obs = _f64(src_offset), _f64(src_depth), _f64(obs)
    assert V.ndim == 2 and V.shape == D.shape and NL.size == V.shape[0] and off.size == dep.size == obs.size
    ss = np.empty(V.shape[0])
    _oracle_lib().oracle_ss_batch(_p(V), _p(D), NL.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), V.shape[1], V.shape[0],
                                  _p(off), _p(dep), _p(obs), off.size, _p(ss))
    return ss


def oracle_dff_f32(vels, depths, src_offset, src_depth, want_p: bool = False):
    """The same restatement in IEEE single precision (dff_oracle.c built with -DORACLE_F32): the checker of the
    fp32 kernel path.  float32 in, float32 out."""
    global _lib32
    if _lib32 is None:
        build()
        lib = ctypes.CDLL(_ORACLE_F32_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        lib.oracle_dff_f32.argtypes = [fp, fp, ctypes.c_int, fp, fp, ctypes.c_int64, fp, fp,
                                       ctypes.POINTER(ctypes.c_int32)]
        lib.oracle_dff_f32.restype = None
        _lib32 = lib
    fp = ctypes.POINTER(ctypes.c_float)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    vels, depths, off, dep = f32(vels), f32(depths), f32(src_offset), f32(src_depth)
    assert vels.size == depths.size + 1 and off.size == dep.size
    n = off.size
    t = np.empty(n, dtype=np.float32)
    p = np.empty(n, dtype=np.float32) if want_p else None
    nl = np.empty(n, dtype=np.int32) if want_p else None
    _lib32.oracle_dff_f32(vels.ctypes.data_as(fp), depths.ctypes.data_as(fp), depths.size, off.ctypes.data_as(fp),
                          dep.ctypes.data_as(fp), n, t.ctypes.data_as(fp),
                          p.ctypes.data_as(fp) if want_p else None,
                          nl.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if want_p else None)
    return (t, p, nl) if want_p else t


# ----------------------------------------------------------------------------------------------
# The reference itself (oracle/_ref), where it was built.
# ----------------------------------------------------------------------------------------------
_ref_libs = {}


def have_ref() -> bool:
    return os.path.exists(_REF_SO)


def ref_dff(vels, depths, src_offset, src_depth, keep_delta: int = -1, opt: str = "O2", cwd: str | None = None):
    """Call the reference's own `dff_` (8-argument, all by reference).

    The reference truncates ./rays.dat on every call, so the call runs inside a
    scratch directory (or `cwd`); with keep_delta > 0 the caller reads rays.dat there.
    """
    path = _REF_SO if opt == "O2" else _REF_O0_SO
    if path not in _ref_libs:
        lib = ctypes.CDLL(path)
        lib.dff_.argtypes = [_c_double_p, _c_double_p, _c_int_p, _c_double_p, _c_double_p, _c_int_p,
                             _c_double_p, _c_int_p]
        lib.dff_.restype = None
        _ref_libs[path] = lib
    lib = _ref_libs[path]
    vels, depths, off, dep = _f64(vels), _f64(depths), _f64(src_offset).copy(), _f64(src_depth).copy()
    t = np.zeros(off.size, dtype=np.float64)
    nl = ctypes.c_int(depths.size)
    ns = ctypes.c_int(off.size)
    kd = ctypes.c_int(keep_delta)
    old = os.getcwd()
    tmp = None
    if cwd is None:
        tmp = tempfile.TemporaryDirectory()
        cwd = tmp.name
    try:
        os.chdir(cwd)
        lib.dff_(_p(vels), _p(depths), ctypes.byref(nl), _p(off), _p(dep), ctypes.byref(ns), _p(t),
                 ctypes.byref(kd))
    finally:
        os.chdir(old)
        if tmp is not None:
            tmp.cleanup()
    return t
