"""Boundary behaviour of librtmi.so on the GPU: thread safety of the drop-in symbols, the chunked / sharded drop-in
pipeline, device-side validation of batch rows, the capture rule of the fault word."""
import ctypes
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import ROOT, compare_times

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope="module")
def rt():
    import raytracerfortran_amd as r
    r.load_library()
    r.init(0)
    return r


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def random_model(rng, nlay):
    d = np.cumsum(rng.uniform(150, 700, nlay))
    v = rng.uniform(1500, 7500, nlay + 1)
    return v, d


def test_two_threads_call_dff_concurrently(rt, O):
    """The reference is not re-entrant (SAVE'd locals, shared rays.dat); the replacement must be: two threads, different
    models and source sets (one small-call path, one chunked path), many calls each, every result bit-equal."""
    rng = np.random.default_rng(12)
    jobs = []
    for nlay, n in ((6, 20), (9, 300_000), (3, 5000), (14, 70_000)):
        v, d = random_model(rng, nlay)
        off, dep = rng.uniform(10, 8000, n), rng.uniform(20, d[-1] + 400, n)
        jobs.append((v, d, off, dep, O.oracle_dff(v, d, off, dep)))
    errors = []

    def worker(idx):
        try:
            for it in range(12):
                v, d, off, dep, ref = jobs[(idx + 2 * it) % 4]
                t = rt.dff(v, d, d.size, off, dep, off.size)["timeP"]
                if compare_times(t, ref, TOL) != 0.0:
                    errors.append((idx, it, "not bit-identical"))
        except Exception as e:  # pragma: no cover
            errors.append((idx, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    # options may be set from another thread while launches run (synchronised process-wide knobs)
    stop = threading.Event()

    def knobs():
        while not stop.is_set():
            rt.set_option("refill_min", 8)
            rt.set_option("refill_min", 32)

    k = threading.Thread(target=knobs)
    k.start()
    try:
        v, d, off, dep, ref = jobs[1]
        for _ in range(5):
            assert compare_times(rt.dff(v, d, d.size, off, dep, off.size)["timeP"], ref, TOL) == 0.0
    finally:
        stop.set()
        k.join()
        rt.set_option("refill_min", 32)


@pytest.mark.parametrize("n", [4097, 131072, 131073, 262145, 1_000_000, 2_500_000])
def test_chunked_dropin_pipeline_sizes(rt, O, n):
    """dff_ around and above the mapped-buffer limit (131072 sources): one chunk, then chunks of an eighth of the call
    (at least 262144 sources) through the upload / kernel / download queues, ragged last chunk."""
    rng = np.random.default_rng(n)
    v, d = random_model(rng, 8)
    off, dep = rng.uniform(10, 8000, n), rng.uniform(20, d[-1] + 400, n)
    t = rt.dff(v, d, 8, off, dep, n)["timeP"]
    assert compare_times(t, O.oracle_dff(v, d, off, dep), TOL) == 0.0


def test_sharded_dropin_two_pipelines(tmp_path):
    """RTMI_DEVICES: contiguous source slices over several devices, model replicated, host gathers timeP (SURVEY.md
    8(e), single forward -- no collective).  On the 1-GPU box the list names the bound device twice: two pipelines,
    two host threads, one card."""
    code = (
        "import numpy as np, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import raytracerfortran_amd as rt\n"
        "from oracle import oracle as O\n"
        "rt.init(0)\n"
        "rng = np.random.default_rng(4)\n"
        "d = np.cumsum(rng.uniform(150, 700, 11)); v = rng.uniform(1500, 7500, 12)\n"
        "for n in (9000, 100001, 700000):\n"
        "    off, dep = rng.uniform(10, 8000, n), rng.uniform(20, d[-1] + 400, n)\n"
        "    t = rt.dff(v, d, 11, off, dep, n, keep_delta=-1)['timeP']\n"
        "    ref = O.oracle_dff(v, d, off, dep)\n"
        "    assert np.array_equal(t, ref), n\n"
        "print('sharded ok')\n")
    env = dict(os.environ, RTMI_DEVICES="0,0,0", RTMI_SHARD_MIN="4000", RTMI_DROPIN_CHUNK="65536", RTMI_SMALL_CALL_NSRC="4096")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert p.returncode == 0 and "sharded ok" in p.stdout, p.stderr[-2000:]


def test_batch_rows_with_bad_nlayers_are_reported(rt, O):
    """rtmi_trace_batch_device reads d_nlayers on the device: a row outside [1, ld_model - 1] would overrun the LDS
    layer table.  It must be reported (fault bit 8), its outputs NaN, the other rows untouched -- both kernels."""
    import torch
    lib = rt.load_library()
    rng = np.random.default_rng(1)
    nm, ld, n = 6, 12, 9000
    off, dep = rng.uniform(10, 8000, n), rng.uniform(20, 6000, n)
    V, D, NL = np.zeros((nm, ld)), np.zeros((nm, ld)), np.zeros(nm, np.int32)
    for m in range(nm):
        k = 3 + m
        D[m, :k - 1] = np.sort(rng.uniform(100, 5000, k - 1))
        V[m, :k] = rng.uniform(1500, 8000, k)
        NL[m] = k - 1
    obs, sd = rng.uniform(0.1, 2.0, n), np.full(nm, 0.05)
    dev = "cuda"
    for path in (1, 2):
        for bad_value in (0, -3, ld, 1000):
            nl = NL.copy()
            nl[2] = bad_value
            args = [torch.tensor(a, device=dev) for a in (V, D, nl, off, dep)]
            try:
                rt.set_option("path", path)
                Tm, L = rt.trace_batch_device(*args, obs=torch.tensor(obs, device=dev), sd=torch.tensor(sd, device=dev))
                torch.cuda.synchronize()
            finally:
                rt.set_option("path", 0)
            assert lib.rtmi_device_status() == 8, (path, bad_value)
            assert lib.rtmi_clear_device_status() == 8 and lib.rtmi_device_status() == 0
            Tm, L = Tm.cpu().numpy(), L.cpu().numpy()
            assert np.isnan(Tm[2]).all() and np.isnan(L[2])
            for m in (0, 1, 3, 4, 5):
                ref = O.oracle_dff(V[m, :NL[m] + 1], D[m, :NL[m]], off, dep)
                assert compare_times(Tm[m], ref, TOL) == 0.0
                assert np.isfinite(L[m])
    # the host entry point validates before launching
    rt.set_sources(off, dep, obs)
    nl = NL.copy()
    nl[0] = ld
    with pytest.raises(rt.RtmiError, match="nlayers"):
        rt.loglhood_batch(V, D, nl, sd)
    assert lib.rtmi_device_status() == 0


def test_first_call_inside_a_capture_is_refused(tmp_path):
    """The fault word is a pinned allocation, illegal inside a stream capture: a process whose FIRST librtmi call is a
    captured launch gets RTMI_ERR_BAD_ARG with a clear message; after rtmi_init the same capture works and replays."""
    code = (
        "import numpy as np, sys, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import raytracerfortran_amd as rt\n"
        "from oracle import oracle as O\n"
        "rng = np.random.default_rng(2)\n"
        "d = np.cumsum(rng.uniform(300, 800, 8)); v = np.sort(rng.uniform(1500, 6500, 9)); n = 50000\n"
        "off, dep = rng.uniform(10, 8000, n), rng.uniform(50, d[-1] + 500, n)\n"
        "a = [torch.tensor(x, device='cuda') for x in (v, d, off, dep)]\n"
        "out = torch.empty(n, dtype=torch.float64, device='cuda')\n"
        "s = torch.cuda.Stream()\n"
        "g = torch.cuda.CUDAGraph()\n"
        "refused = False\n"
        "with torch.cuda.stream(s):\n"
        "    g.capture_begin()\n"
        "    try:\n"
        "        rt.trace_device(*a, timeP=out, stream=s)\n"
        "    except rt.RtmiError as e:\n"
        "        refused = 'rtmi_init' in str(e)\n"
        "    g.capture_end()\n"
        "assert refused, 'first captured call was not refused'\n"
        "rt.init(0)\n"
        "g2 = torch.cuda.CUDAGraph()\n"
        "with torch.cuda.stream(s):\n"
        "    g2.capture_begin()\n"
        "    rt.trace_device(*a, timeP=out, stream=s)\n"
        "    g2.capture_end()\n"
        "out.zero_(); torch.cuda.synchronize(); g2.replay(); torch.cuda.synchronize()\n"
        "assert np.array_equal(out.cpu().numpy(), O.oracle_dff(v, d, off, dep))\n"
        "print('capture ok')\n")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert p.returncode == 0 and "capture ok" in p.stdout, p.stderr[-2000:]
