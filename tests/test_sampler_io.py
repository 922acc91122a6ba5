"""The reference sampler's file formats (raytracerfortran_amd/sampler_io.py) on copies of the reference's own test_1
inputs (tests/golden/test_1/: data files of the reference's example, unmodified)."""
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN
from raytracerfortran_amd import sampler_io as S

T1 = os.path.join(GOLDEN, "test_1")


def test_parameter_and_map_files(kat):
    assert S.read_filebase(T1) == "test_1"
    p = S.read_parameter_file(os.path.join(T1, "test_1_parameter.dat"))
    assert (p["IMAP"], p["NPL"], p["NDAT_RT"], p["NSRC"], p["NLMN"], p["NLMX"], p["ICHAINTHIN"], p["NKEEP"]) == \
           (0, 2, 20, 20, 1, 10, 5, 100)
    assert (p["dTlog"], p["lambda"], p["hmx"], p["hmin"]) == (1.4, 3.01, 10000.1, 100.1)
    assert p["sdmn"] == [0.001, 0.001] and p["sdmx"] == [0.07, 0.07] and p["NPTCHAINS1"] == 1
    k, nodes, sd = S.read_map_file(os.path.join(T1, "test_1_map.dat"), p["NLMX"])
    m = kat["model"]
    assert k == 7 and sd == 0.02
    assert nodes[:, 1].tolist() == m["vels"] and nodes[:, 0].tolist() == [0.0] + m["depths"]
    obs, off, dep = S.read_data(os.path.join(T1, "test_1"), 20, 20)
    assert np.array_equal(obs, kat["test_1_obs_RT"]) and np.array_equal(off, kat["test_1_src_offset"])
    assert np.array_equal(dep, kat["test_1_src_depth"])


def test_map_file_round_trip(tmp_path):
    nodes = np.array([[0.0, 3100.0], [1000.0, 2400.0], [1500.0, 3730.0]])
    S.write_map_file(str(tmp_path / "x_map.dat"), nodes, 0.02, nlmx=10)
    k, n2, sd = S.read_map_file(str(tmp_path / "x_map.dat"), 10)
    assert k == 3 and sd == 0.02 and np.array_equal(n2, nodes)
    assert len(open(tmp_path / "x_map.dat").read().split()) == 80  # as test_1_map.dat


def test_es_format_and_sample_rows(tmp_path):
    assert S.format_es([1.5580056842, 0.0, -12345.678], 20, 10) == \
        "    1.5580056842E+00    0.0000000000E+00   -1.2345678000E+04"  # FORMAT(5000ES20.10), cf. test_1_obsRT.dat
    nodes = np.array([[0.0, 3100.0], [1000.0, 2400.0]])
    row = S.sample_row(55.12, 0.0, 0.0, nodes, 0.02, nlmx=10)
    assert len(row) == 18 * (4 + 20 + 9)  # FORMAT(500ES18.8): 33 columns, as the Rmd's all.names
    p = tmp_path / "t_voro_sample.txt"
    p.write_text(row + "\n")
    r = S.read_sample_file(str(p), 10)[0]
    assert r["k"] == 2 and abs(r["logL"] - 55.12) < 1e-7 and np.allclose(r["nodes"], nodes) and abs(r["sd"] - 0.02) < 1e-9


class _Flat:
    def __call__(self, node_sets):
        return np.full(len(node_sets), 0.004)


def test_run_from_files_with_injected_likelihood(tmp_path):
    """The file plumbing end to end on CPU (constant likelihood): samples land in <base>_voro_sample.txt."""
    r = S.run_from_files(T1, nsweeps=60, nchains=4, seed=3, out_dir=str(tmp_path), evaluator=_Flat())
    rows = S.read_sample_file(str(tmp_path / "test_1_voro_sample.txt"), 10)
    assert len(rows) == len(r["samples"]) == 12  # ICHAINTHIN = 5, one chain at T = 1
    assert all(1 <= x["k"] <= 10 for x in rows)


@pytest.mark.gpu
def test_imap_mode_and_sampling_from_reference_files(tmp_path, kat):
    """IMAP = 1 (prjmh_temper_rf.f90:246-268): evaluate the MAP model of test_1 and write the replica files; then a
    short tempered run from the same files."""
    d = tmp_path / "run"
    d.mkdir()
    for f in os.listdir(T1):  # plain copies: the snapshot's files may be read-only
        shutil.copyfile(os.path.join(T1, f), d / f)
    par = (d / "test_1_parameter.dat").read_text().splitlines()
    par[0] = " 1   ! IMAP"
    (d / "test_1_parameter.dat").write_text("\n".join(par) + "\n")
    r = S.run_from_files(str(d), 0)
    assert abs(r["logL"] - kat["test_1_map_logL_survey"]["value"]) < 1e-8
    pred = np.array((d / "test_1_mappredRT.dat").read_text().split(), dtype=float)
    assert np.abs(pred - np.array(kat["config1_ref_timeP"])).max() < 1e-10  # ES20.10 rounding
    par[0] = " 0   ! IMAP"
    (d / "test_1_parameter.dat").write_text("\n".join(par) + "\n")
    r = S.run_from_files(str(d), 100, nchains=4)
    rows = S.read_sample_file(str(d / "test_1_voro_sample.txt"), 10)
    assert len(rows) == 20 and max(x["logL"] for x in rows) > 40.0
