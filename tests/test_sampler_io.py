"""The reference sampler's file formats (raytracerfortran_amd/sampler_io.py) on the reference's test_1 problem: the
directory is rebuilt from the values of tests/golden/kat_author.json (sources, observations, model) and the settings of
/root/reference/test_1/test_1_parameter.dat, written through the module's own writers."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from raytracerfortran_amd import sampler_io as S

# /root/reference/test_1/test_1_parameter.dat, value by value (read_input.f90:22-54 order)
TEST_1_PARAMETERS = {"IMAP": 0, "ICOV": 1, "ENOS": 0, "IPOIPR": 1, "IAR": 0, "NPL": 2, "I_VARPAR": 0, "IBD_SINGLE": 0,
                     "I_RT": 1, "I_VREF": 0, "I_VPVS": 0, "ISMPPRIOR": 0, "ISETSEED": 0, "IEXCHANGE": 1, "NDAT_RT": 20,
                     "NMODE": 1, "NSRC": 20, "NLMN": 1, "NLMX": 10, "ICHAINTHIN": 5, "NKEEP": 100, "NPTCHAINS1": 1,
                     "dTlog": 1.4, "lambda": 3.01, "hmx": 10000.1, "hmin": 100.1, "TCHCKPT": 1000.0, "dVs": 0.1,
                     "dVpVs": 0.1, "sdmn": [0.001, 0.001], "sdmx": [0.07, 0.07], "VpVsmin": 1.5, "VpVsmax": 1.8}


@pytest.fixture()
def T1(tmp_path, kat):
    """A test_1 directory as the reference's sampler expects it (filebase.txt + <base>_*.dat/.txt)."""
    d = tmp_path / "test_1"
    d.mkdir()
    (d / "filebase.txt").write_text("6\ntest_1\n")
    S.write_parameter_file(str(d / "test_1_parameter.dat"), TEST_1_PARAMETERS)
    m = kat["model"]
    S.write_map_file(str(d / "test_1_map.dat"), np.column_stack([[0.0] + m["depths"], m["vels"]]), 0.02, nlmx=10)
    np.savetxt(d / "test_1_RT.txt", np.array(kat["test_1_obs_RT"]), fmt="%.17g")
    np.savetxt(d / "test_1_src_data.txt", np.column_stack([kat["test_1_src_offset"], kat["test_1_src_depth"]]), fmt="%.17g")
    return str(d)


def test_parameter_and_map_files(kat, T1):
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


def test_parameter_file_in_the_reference_layout(tmp_path):
    """The reference's files carry "value ! NAME !! free comment" records, two values on the sdmn / sdmx lines and no
    newline at the end (test_1_parameter.dat): list-directed READ semantics."""
    lines = []
    for name in S.PARAMETER_NAMES:
        v = TEST_1_PARAMETERS[name]
        txt = " ".join(str(x) for x in v) if isinstance(v, list) else str(v)
        lines.append(f" {txt}   ! {name:10s} !! some comment, with commas and 1.5 numbers")
    path = tmp_path / "x_parameter.dat"
    path.write_text("\n".join(lines))  # no trailing newline
    p = S.read_parameter_file(str(path))
    assert p == TEST_1_PARAMETERS


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


def test_run_from_files_with_injected_likelihood(tmp_path, T1):
    """The file plumbing end to end on CPU (constant likelihood): samples land in <base>_voro_sample.txt."""
    r = S.run_from_files(T1, nsweeps=60, nchains=4, seed=3, out_dir=str(tmp_path), evaluator=_Flat())
    rows = S.read_sample_file(str(tmp_path / "test_1_voro_sample.txt"), 10)
    assert len(rows) == len(r["samples"]) == 12  # ICHAINTHIN = 5, one chain at T = 1
    assert all(1 <= x["k"] <= 10 for x in rows)
    # SAVESAMPLE's bookkeeping columns (prjmh_temper_rf.f90:1880-1883; the R notebook names them, Rmd:135-146): LogPrior, tcmp
    # and, behind the 2 x NLMX node columns, SdRT, ArPar, ArParRT, acc_ratio, iacc_bd, irej_bd, iacc_bds, ic, rank
    st = r["sampler"].stats
    pr = r["sampler"].prior
    for ln, x in zip(open(tmp_path / "test_1_voro_sample.txt"), rows):
        t = np.array(ln.split(), dtype=float)
        assert t.size == 4 + 2 * 10 + 9
        assert abs(t[1] - pr.log_pk(x["k"])) < 1e-7 and t[2] > 0.0
        tail = t[4 + 2 * 10:]
        assert tail[1] == 0.0 and tail[2] == 0.0
        assert abs(tail[3] - st["acc_mh"] / max(1, st["prop_mh"])) < 1e-7 and 0.0 < tail[3] <= 1.0
        assert tail[4] == st["acc_bd"] and tail[5] == st["prop_bd"] - st["acc_bd"] and tail[6] == st["acc_sd"]
        assert tail[7] == 1.0 and tail[8] == 1.0


@pytest.mark.gpu
def test_imap_mode_and_sampling_from_reference_files(tmp_path, kat, T1):
    """IMAP = 1 (prjmh_temper_rf.f90:246-268): evaluate the MAP model of test_1 and write the replica files; then a
    short tempered run from the same files."""
    import pathlib
    d = pathlib.Path(T1)
    S.write_parameter_file(str(d / "test_1_parameter.dat"), dict(TEST_1_PARAMETERS, IMAP=1))
    r = S.run_from_files(str(d), 0)
    assert abs(r["logL"] - kat["test_1_map_logL_survey"]["value"]) < 1e-8
    pred = np.array((d / "test_1_mappredRT.dat").read_text().split(), dtype=float)
    assert np.abs(pred - np.array(kat["config1_ref_timeP"])).max() < 1e-10  # ES20.10 rounding
    S.write_parameter_file(str(d / "test_1_parameter.dat"), TEST_1_PARAMETERS)
    r = S.run_from_files(str(d), 100, nchains=4)
    rows = S.read_sample_file(str(d / "test_1_voro_sample.txt"), 10)
    assert len(rows) == 20 and max(x["logL"] for x in rows) > 40.0


def test_readers_on_the_references_own_files(kat):
    """tests/golden/test_1_files/ holds the reference's test_1 input files byte for byte (data files of the reference's
    example: /root/reference/test_1/{filebase.txt, test_1_parameter.dat, test_1_map.dat}) -- "value ! NAME !! comment"
    records, two values on the sdmn / sdmx lines, no newline at the end of the parameter file, an 80-token map row.
    The readers must return exactly the values READPARFILE (read_input.f90:22-57) and the MAP reader
    (prjmh_temper_rf.f90:152-189) would."""
    d = os.path.join(GOLDEN, "test_1_files")
    raw = open(os.path.join(d, "test_1_parameter.dat")).read()
    assert not raw.endswith("\n") and "!!" in raw and raw.count("\n") == 32    # the layout the readers must survive
    assert S.read_filebase(d) == "test_1"
    p = S.read_parameter_file(os.path.join(d, "test_1_parameter.dat"))
    assert p == TEST_1_PARAMETERS
    assert len(open(os.path.join(d, "test_1_map.dat")).read().split()) == 80
    k, nodes, sd = S.read_map_file(os.path.join(d, "test_1_map.dat"), p["NLMX"])
    m = kat["model"]
    assert k == 7 and sd == 0.02
    assert nodes[:, 0].tolist() == [0.0] + m["depths"] and nodes[:, 1].tolist() == m["vels"]
