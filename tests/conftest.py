import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def free_port() -> int:
    """An ephemeral port for a torch.distributed rendezvous on 127.0.0.1 (two test processes on one box must not collide)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "kat_author.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_index():
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def golden_batches():
    z = np.load(os.path.join(GOLDEN, "ref_batches.npz"), allow_pickle=False)
    names = sorted({k.split("__")[0] for k in z.files})
    return {n: {f: z[f"{n}__{f}"] for f in ("vels", "depths", "off", "dep", "timeP")} for n in names}


def golden_case_names():
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return [c["name"] for c in json.load(f)["cases"]]


def compare_times(t, t_ref, tol=1e-9):
    """The parity bar of BASELINE.json: sentinel positions identical, |dt| < tol elsewhere.

    Returns max |dt| over positions where neither side is the -999 sentinel.
    """
    t = np.asarray(t)
    t_ref = np.asarray(t_ref)
    assert t.shape == t_ref.shape
    s, s_ref = (t == -999.0), (t_ref == -999.0)
    assert int((s != s_ref).sum()) == 0, f"sentinel mismatch at {np.flatnonzero(s != s_ref)[:10]}"
    assert not np.isnan(t).any()
    both = ~s
    if not both.any():
        return 0.0
    err = float(np.abs(t[both] - t_ref[both]).max())
    assert err < tol, f"max |dt| = {err:.3e} s >= {tol}"
    return err
