"""The reference sampler's file formats, so that the tempering driver can be started
from -- and feed -- the same files as /root/reference/ray_tracing_sampling/prjmh_temper_rf.f90.

  filebase.txt            line 1: length, line 2: prefix <base>                      prjmh_temper_rf.f90:56-59
  <base>_parameter.dat    33 positional lines "value(s) ! NAME comment"              read_input.f90:22-57
  <base>_map.dat          one row: k, (depth, Vp) x NLMX zero padded, sigma at column NLMX*NPL+2
                                                                                     prjmh_temper_rf.f90:152-189, README.md:29-40
  <base>_RT.txt           observed travel times, one per line                        read_input.f90:350-364
  <base>_src_data.txt     "offset depth" per line                                    read_input.f90:366-378
  <base>_voro_sample.txt  one row per kept T = 1 sample, FORMAT(500ES18.8):
                          logL, logPr, tcmp, k, (depth, Vp) x NLMX, SdRT, ArPar, ArParRT, acc_ratio,
                          iacc_bd, irej_bd, iacc_bds, ic, rank                        prjmh_temper_rf.f90:1870-1883,1912;
                                                                                     raytracerR-export-data-to-MCMC.Rmd:135-146
  <base>_mappredRT.dat / <base>_obsRT.dat   FORMAT(5000ES20.10), written in IMAP = 1 mode  prjmh_temper_rf.f90:246-268,1923-1960
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np

PARAMETER_NAMES = ["IMAP", "ICOV", "ENOS", "IPOIPR", "IAR", "NPL", "I_VARPAR", "IBD_SINGLE", "I_RT", "I_VREF", "I_VPVS",
                   "ISMPPRIOR", "ISETSEED", "IEXCHANGE", "NDAT_RT", "NMODE", "NSRC", "NLMN", "NLMX", "ICHAINTHIN",
                   "NKEEP", "NPTCHAINS1", "dTlog", "lambda", "hmx", "hmin", "TCHCKPT", "dVs", "dVpVs", "sdmn", "sdmx",
                   "VpVsmin", "VpVsmax"]  # read_input.f90:22-54, in file order
_FLOAT_FIELDS = {"dTlog", "lambda", "hmx", "hmin", "TCHCKPT", "dVs", "dVpVs", "sdmn", "sdmx", "VpVsmin", "VpVsmax"}


def read_filebase(directory: str) -> str:
    with open(os.path.join(directory, "filebase.txt")) as f:
        n = int(f.readline().split()[0])
        return f.readline().strip()[:n]


def read_parameter_file(path: str) -> Dict[str, object]:
    """List-directed READs of one record each: the value(s) before the first '!' of every line."""
    out: Dict[str, object] = {}
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    if len(lines) < len(PARAMETER_NAMES):
        raise ValueError(f"{path}: {len(lines)} records, {len(PARAMETER_NAMES)} expected")
    for name, ln in zip(PARAMETER_NAMES, lines):
        toks = ln.split("!")[0].replace(",", " ").split()
        if name in ("sdmn", "sdmx"):      # two-element arrays (read_input.f90:50-51)
            out[name] = [float(t) for t in toks[:2]]
        elif name in _FLOAT_FIELDS:
            out[name] = float(toks[0])
        else:
            out[name] = int(float(toks[0]))
    return out


def write_parameter_file(path: str, par: Dict[str, object]) -> None:
    """makeParameterFile's job (MiscFunctions.R:68-103): one "value ! NAME" record per parameter, in READPARFILE's order."""
    with open(path, "w") as f:
        for name in PARAMETER_NAMES:
            v = par[name]
            txt = " ".join(repr(float(x)) for x in v) if isinstance(v, (list, tuple)) else (
                repr(float(v)) if name in _FLOAT_FIELDS else str(int(v)))
            f.write(f" {txt} ! {name}\n")


def read_map_file(path: str, nlmx: int, npl: int = 2, nmode: int = 1) -> Tuple[int, np.ndarray, float]:
    """(k, nodes[k, 2] = (depth, Vp), sigma) from the start-model row (prjmh_temper_rf.f90:152-189)."""
    tok = np.array(open(path).read().split(), dtype=float)
    k = int(tok[0])
    nodes = tok[1:1 + k * npl].reshape(k, npl)[:, :2].copy()
    sd = float(tok[nlmx * npl + 1])  # tmpmap(NFPMX+1+1), 1-based
    return k, nodes, sd


def write_map_file(path: str, nodes: np.ndarray, sd: float, nlmx: int, total: int = 80) -> None:
    """makeMAPFile's layout (MiscFunctions.R:105-148): k, nodes, zero padding, sigma at NLMX*NPL+2, zeros."""
    row = np.zeros(total)
    k = nodes.shape[0]
    row[0] = k
    row[1:1 + 2 * k] = nodes.reshape(-1)
    row[nlmx * 2 + 1] = sd
    with open(path, "w") as f:
        f.write(" ".join(f"{x:.2f}" for x in row))


def read_data(base: str, ndat: int, nsrc: int):
    """READDATA (read_input.f90:302-387): observed times and "offset depth" pairs."""
    obs = np.loadtxt(base + "_RT.txt", ndmin=1)[:ndat]
    src = np.loadtxt(base + "_src_data.txt", ndmin=2)[:nsrc]
    return obs, src[:, 0].copy(), src[:, 1].copy()


def format_es(values, width: int, digits: int) -> str:
    """Fortran ESw.d edit descriptor for every value of a record."""
    out = []
    for v in values:
        s = f"{float(v):.{digits}E}"           # d.ddddE+XX
        mant, exp = s.split("E")
        out.append(f"{mant}E{exp[0]}{int(exp[1:]):02d}".rjust(width))
    return "".join(out)


def sample_row(logl: float, logpr: float, tcmp: float, nodes_sorted: np.ndarray, sd: float, nlmx: int,
               acc_ratio: float = 0.0, iacc_bd: int = 0, irej_bd: int = 0, ic: int = 1, rank: int = 1, iacc_bds: int = 0) -> str:
    """One `<base>_voro_sample.txt` record (SAVESAMPLE, prjmh_temper_rf.f90:1870-1883, FORMAT(500ES18.8))."""
    voro = np.zeros(nlmx * 2)
    k = nodes_sorted.shape[0]
    voro[:2 * k] = nodes_sorted.reshape(-1)
    vals = [logl, logpr, tcmp, float(k)] + voro.tolist() + [sd, 0.0, 0.0, acc_ratio, float(iacc_bd), float(irej_bd),
                                                            float(iacc_bds), float(ic), float(rank)]   # SdRT, ArPar, ArParRT, ...
    return format_es(vals, 18, 8)


def read_sample_file(path: str, nlmx: int) -> List[dict]:
    rows = []
    for ln in open(path):
        t = np.array(ln.split(), dtype=float)
        if t.size == 0:
            continue
        k = int(t[3])
        rows.append({"logL": t[0], "logPr": t[1], "k": k, "nodes": t[4:4 + 2 * k].reshape(k, 2), "sd": t[4 + 2 * nlmx]})
    return rows


def run_from_files(directory: str, nsweeps: int, nchains: int = 8, seed: int = 1, burnin: int = 0, out_dir: str = None,
                   evaluator=None, adapt_ladder: bool = False):
    """Start the tempering driver the way `mpirun prjmh_temper_rf` is started in the reference's test_1 directory:
    filebase.txt -> <base>_parameter.dat, <base>_map.dat, data files; T = 1 samples go to <base>_voro_sample.txt.
    IMAP = 1: evaluate the start model only and write <base>_mappredRT.dat / <base>_obsRT.dat (:246-268).
    adapt_ladder: the burn-in re-spacing of the ladder (:363-383) with the chains actually following it (tempering.Sampler)."""
    from . import tempering as T
    from . import raymod
    base_name = read_filebase(directory)
    base = os.path.join(directory, base_name)
    out_base = os.path.join(out_dir or directory, base_name)
    par = read_parameter_file(base + "_parameter.dat")
    if par["NPL"] != 2 or par["I_VARPAR"] != 0 or par["ENOS"] != 0 or par["IAR"] != 0:
        raise NotImplementedError("only the NPL = 2, I_VARPAR = 0, ENOS = 0, IAR = 0 configuration is supported")
    k, nodes, sd = read_map_file(base + "_map.dat", par["NLMX"])
    obs, off, dep = read_data(base, par["NDAT_RT"], par["NSRC"])
    if par["IMAP"] == 1:
        v, z = T.layering(nodes)
        pred = raymod.TraceRays(v, z, z.size, off, dep, off.size, None, -1)
        with open(out_base + "_mappredRT.dat", "w") as f:
            f.write(format_es(pred, 20, 10) + "\n")
        with open(out_base + "_obsRT.dat", "w") as f:
            f.write(format_es(obs, 20, 10) + "\n")
        ss = float(np.sum((obs - pred) ** 2))
        return {"logL": T.gaussian_loglike(ss, obs.size, sd), "pred": pred}
    prior = T.Prior(kmin=par["NLMN"], kmax=par["NLMX"], lam=par["lambda"], hmx=par["hmx"], hmin=par["hmin"],
                    sdmin=par["sdmn"][1], sdmax=par["sdmx"][1])  # RT data use the second entry (read_input.f90:234-236)
    # without an injected likelihood the sweeps, exchanges and the sample bookkeeping run inside librtmi.so on the GPU
    # (tempering.NativeSampler -> rtmi_pt_run); an injected evaluator (CPU tests) drives the per-chain Python engine
    ev = evaluator or T.GpuEvaluator(off, dep, obs, ld=max(32, par["NLMX"] + 2))
    engine = T.Sampler if evaluator is not None else T.NativeSampler
    s = engine(prior, ev, ndata=obs.size, nchains_total=nchains, nptchains1=par["NPTCHAINS1"], dTlog=par["dTlog"],
               seed=seed, adapt_ladder=adapt_ladder)
    s.init_from_model(nodes, sd)
    import time
    t0 = time.perf_counter()
    samples = s.run(nsweeps, burnin=burnin, thin=par["ICHAINTHIN"])
    tcmp = (time.perf_counter() - t0) / max(1, nsweeps)
    # SAVESAMPLE's bookkeeping columns (prjmh_temper_rf.f90:1880-1883; names in raytracerR-export-data-to-MCMC.Rmd:135-146):
    # logPr = the Poisson prior of k (read_input.f90:78-81), tcmp = seconds of computation per step (obj%tcmp, :453), acc_ratio =
    # accepted / proposed perturbations, iacc_bd / irej_bd / iacc_bds = accepted and rejected birth-death moves and accepted
    # sigma moves, ic / rank = the chain that wrote the row.  The reference keeps these counters per chain object and writes
    # their values at the time of the sample; the device sampler keeps them per chain too (rtmi_pt_chain_stats) and they are
    # read once, after the run: every row of a chain carries that chain's totals of the run.  (The Python engine of the CPU
    # tests keeps one set of counters for all its chains.)
    st = s.stats
    per_chain = None
    if evaluator is None and getattr(s, "sample_chains", None) is not None and len(s.sample_chains) == len(samples):
        per_chain = s.native.chain_stats()          # [nloc][8]: acc_bd, prop_bd, acc_mh, prop_mh, acc_sd, prop_sd, ...
    with open(out_base + "_voro_sample.txt", "w") as f:
        for i, (kk, nd, sdv, ll) in enumerate(samples):
            if per_chain is not None:
                c = s.sample_chains[i]
                q = per_chain[c - getattr(s, "first", 0)]
                cols = dict(acc_ratio=q[2] / max(1, q[3]), iacc_bd=int(q[0]), irej_bd=int(q[1] - q[0]), iacc_bds=int(q[4]), ic=c + 1)
            else:
                cols = dict(acc_ratio=st["acc_mh"] / max(1, st["prop_mh"]), iacc_bd=st["acc_bd"], irej_bd=st["prop_bd"] - st["acc_bd"],
                            iacc_bds=st["acc_sd"], ic=1)
            f.write(sample_row(ll, prior.log_pk(int(kk)), tcmp, nd, sdv, par["NLMX"], rank=getattr(s, "rank", 0) + 1, **cols) + "\n")
    return {"samples": samples, "sampler": s, "parameters": par}


if __name__ == "__main__":  # python -m raytracerfortran_amd.sampler_io <dir> [nsweeps] [nchains]
    import sys
    d = sys.argv[1]
    r = run_from_files(d, int(sys.argv[2]) if len(sys.argv) > 2 else 1000, int(sys.argv[3]) if len(sys.argv) > 3 else 8)
    print({k: v for k, v in r.items() if k in ("logL",)} or f"{len(r['samples'])} samples written")
