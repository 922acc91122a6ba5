"""Parallel-tempering reversible-jump sampler over the GPU travel-time likelihood (host-driven).

The reference runs one MPI rank per tempered chain; every Metropolis-Hastings decision re-traces all rays
through `LOGLHOOD` -> `TraceRays` (/root/reference/ray_tracing_sampling/prjmh_temper_rf.f90:416-497,
loglhood.f90:128-199).  Here the chains of a rank advance in lock-step: for every proposal slot of a sweep the
proposed models of ALL local chains are evaluated by ONE `rtmi_trace_batch_device` launch (travel times + fused
Gaussian logL on the GPU), the accept/reject decisions are taken on the host, and chains are sharded over ranks
(one process per GPU).  A swap round all-gathers the chains' logL (`torch.distributed`: RCCL over xGMI for GPU
tensors, gloo on CPU), after which every rank takes the SAME swap decisions from a shared random stream and swaps
TEMPERATURES in place -- equivalent to the reference's "chain states swap, beta stays with the worker"
(prjmh_temper_rf.f90:1351-1357) with no model state moving between GPUs.

Proposal / prior semantics follow the reference configuration of test_1 (NPL = 2, ENOS = 0, IPOIPR = 1, ICOV = 1,
I_VARPAR = 0):
  state            k nodes (depth, Vp), node 1 pinned at depth 0; sigma               rjmcmc_com.f90:214-243
  layering         nodes sorted by depth; vels = Vp of nodes, interfaces = depths of nodes 2..k
                                                                                      loglhood.f90:214-260, :130-135
  birth / death    1/3 each (none at the k limits); new node ~ U(0, hmx-hmin) x U(1500, 10000);
                   logPr = log pk(k') - log pk(k), pk Poisson(lambda)                 prjmh_temper_rf.f90:669-713, :917-1095
  perturbation     every (node, parameter) in turn, Cauchy step of scale (max-min)/30 prjmh_temper_rf.f90:721-759, :1387-1420
  bounds           layer thickness >= hmin, interfaces <= hmx, depths in [0, hmx], Vp in [1500, 10000]
                                                                                      prjmh_temper_rf.f90:1639-1716
  sigma            with p = 0.9 a Gaussian step of (sdmx-sdmn)/10, bounds [sdmn, sdmx]  prjmh_temper_rf.f90:521-575, :1616-1630
  acceptance       u < exp(logPr + beta (logL' - logL))                               prjmh_temper_rf.f90:697-706
  ladder           beta_i = dTlog^-(i-1), NPTCHAINS1 chains at T = 1                   prjmh_temper_rf.f90:115-135
  swap             u <= exp((beta2 - beta1)(logL1 - logL2))                           prjmh_temper_rf.f90:1329-1357
Deliberate differences: the sigma step reuses the chain's residual sum instead of re-tracing identical rays; swap
partners are adjacent temperature levels (alternating even/odd pairs) instead of "the first two workers that
report to rank 0" (prjmh_temper_rf.f90:331-349), which depends on MPI timing.  Chain-level parity with the
reference is therefore statistical, not bitwise (its RANDOM_NUMBER streams are compiler specific anyway).
"""
from __future__ import annotations

import math
import sys
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)


@dataclass
class Prior:
    """read_input.f90:205-214 + test_1_parameter.dat defaults."""
    kmin: int = 1
    kmax: int = 10
    lam: float = 3.01          # Poisson prior on k
    hmx: float = 10000.1       # max partition depth
    hmin: float = 100.1        # min layer thickness
    vmin: float = 1500.0
    vmax: float = 10000.0
    sdmin: float = 0.001
    sdmax: float = 0.07
    pertsdsc: float = 30.0     # read_input.f90:169
    pertsdsdsc: float = 10.0   # read_input.f90:237

    def log_pk(self, k: int) -> float:
        return -self.lam + k * math.log(self.lam) - math.lgamma(k + 1.0)  # read_input.f90:78-81


@dataclass
class Chain:
    nodes: np.ndarray          # [k, 2] (depth, Vp); row 0 is the surface node (depth pinned)
    sd: float
    logL: float = -np.inf
    ss: float = np.inf         # sum of squared residuals of the current model

    @property
    def k(self) -> int:
        return self.nodes.shape[0]


def layering(nodes: np.ndarray):
    """INTERPLAYER_novar (loglhood.f90:214-260) + LOGLHOOD_RT's call arguments (:130-144):
    returns (vels[k], interfaces[k-1]) -- or the k = 1 special case (v, v), (9999.9,)."""
    order = np.argsort(nodes[:, 0], kind="stable")
    s = nodes[order]
    if s.shape[0] == 1:
        return np.array([s[0, 1], s[0, 1]]), np.array([9999.9])
    return s[:, 1].copy(), s[1:, 0].copy()


def inside_bounds(nodes: np.ndarray, pr: Prior) -> bool:
    """CHECKBOUNDS (prjmh_temper_rf.f90:1639-1679)."""
    k = nodes.shape[0]
    if np.any(nodes[:, 1] < pr.vmin) or np.any(nodes[:, 1] > pr.vmax):
        return False
    if k > 1:
        z = np.sort(nodes[:, 0])[1:]
        if np.any(nodes[1:, 0] < 0.0) or np.any(nodes[1:, 0] > pr.hmx):
            return False
        h = np.diff(np.concatenate([[0.0], z]))
        if np.any(h < pr.hmin) or np.any(z > pr.hmx):
            return False
    return True


def gaussian_loglike(ss: float, n: int, sd: float) -> float:
    """loglhood.f90:194-196 with the overflow-free normalisation; a NaN likelihood is the
    reference's -HUGE(1._RP) (loglhood.f90:197-200): such a model is never accepted."""
    ll = -0.5 * n * LOG_2PI - (ss / (2.0 * sd * sd) + n * math.log(sd))
    return -sys.float_info.max if ll != ll else ll


def mh_accept(u: float, log_ratio: float) -> bool:
    """accept unless u >= exp(log ratio) (prjmh_temper_rf.f90:697-706); a NaN ratio (-HUGE against -HUGE) rejects."""
    return log_ratio == log_ratio and not (u >= math.exp(min(log_ratio, 50.0)))


class GpuEvaluator:
    """Residual sums of many models over the resident sources: one rtmi_loglhood_batch call (one launch) per
    proposal slot.  Sources and observations are uploaded once (rtmi_set_sources), like READDATA."""

    def __init__(self, src_offset, src_depth, obs, ld: int = 32):
        from . import raymod
        self.raymod, self.ld = raymod, ld
        self.n = int(np.asarray(src_offset).size)
        raymod.set_sources(src_offset, src_depth, obs)
        self.generation = raymod.load_library().rtmi_sources_generation()
        self.launches = 0
        self.models = 0

    def set_observations(self, src_offset, src_depth, obs):
        self.raymod.set_sources(src_offset, src_depth, obs)
        self.generation = self.raymod.load_library().rtmi_sources_generation()
        self.n = int(np.asarray(src_offset).size)

    def __call__(self, node_sets: List[np.ndarray]) -> np.ndarray:
        nm = len(node_sets)
        if nm == 0:
            return np.zeros(0)
        if self.raymod.load_library().rtmi_sources_generation() != self.generation:
            # the resident set is ONE per process (like READDATA's module data): somebody replaced it under this evaluator
            raise RuntimeError("the resident source set was replaced by another rtmi_set_sources call: this evaluator's "
                               "sources are no longer on the GPU")
        V = np.zeros((nm, self.ld))
        D = np.zeros((nm, self.ld))
        NL = np.zeros(nm, np.int32)
        for m, nodes in enumerate(node_sets):
            v, z = layering(nodes)
            if v.size > self.ld:
                raise ValueError("model larger than ld")
            V[m, :v.size], D[m, :z.size], NL[m] = v, z, z.size
        _, ss = self.raymod.loglhood_batch(V, D, NL, np.ones(nm), want_ss=True)
        self.launches += 1
        self.models += nm
        return ss


@dataclass
class Sampler:
    prior: Prior
    evaluator: object                  # callable: list of node arrays -> SS per model
    ndata: int
    nchains_total: int
    nptchains1: int = 1                # chains at T = 1
    dTlog: float = 1.4
    seed: int = 1
    rank: int = 0
    world: int = 1
    dist: object = None                # torch.distributed (initialised) or None
    dist_device: str = "cpu"           # device of the all-gather tensors ("cuda:<n>" -> RCCL)
    adapt_ladder: bool = False         # burn-in re-spacing of the ladder (prjmh_temper_rf.f90:363-383); see adapt_after_window
    acceptance_window: int = 150       # exchange proposals per acceptance estimate (rjmcmc_com.f90:159)
    chains: List[Chain] = field(default_factory=list)

    def __post_init__(self):
        assert self.nchains_total % self.world == 0, "chains must divide evenly over ranks"
        self.nloc = self.nchains_total // self.world
        self.first = self.rank * self.nloc
        # temperature ladder (prjmh_temper_rf.f90:115-135): NPTCHAINS1 chains at T = 1, then one per level
        levels = [0] * self.nptchains1 + list(range(1, self.nchains_total - self.nptchains1 + 1))
        self.beta_of_level = np.array([self.dTlog ** (-lv) for lv in range(max(levels) + 1)])
        self.level = np.array(levels)                      # level[c] of GLOBAL chain c, replicated on every rank
        self.rng = np.random.default_rng([self.seed, self.rank])   # proposals: per-rank stream
        self.rng_swap = np.random.default_rng([self.seed, 987654])  # swaps: SAME stream on every rank
        self.stats = {"acc_bd": 0, "prop_bd": 0, "acc_mh": 0, "prop_mh": 0, "acc_sd": 0, "prop_sd": 0,
                      "acc_swap": 0, "prop_swap": 0}
        self.swap_parity = 0
        self.acceptance_rate = 0.25                        # prjmh_temper_rf.f90:317
        self._win_acc = self._win_prop = 0                 # ncswap / ncswapprop of the running window (:320-321)
        self.ladder_history = []                           # (sweep, acceptance rate, dTlog) at every window end in burn-in

    # ----- burn-in ladder adaptation (opt-in) --------------------------------------------------------
    def _window(self, acc: int, prop: int) -> bool:
        """TEMPSWP_MH's rolling estimate (prjmh_temper_rf.f90:1371-1380): once MORE than acceptance_window exchanges were
        proposed, the rate of that window becomes the estimate and the window restarts.  True when that happened."""
        self._win_acc += acc
        self._win_prop += prop
        if self._win_prop > self.acceptance_window:
            self.acceptance_rate = self._win_acc / self._win_prop
            self._win_acc = self._win_prop = 0
            return True
        return False

    def adapt_after_window(self, sweep: int) -> bool:
        """The master's burn-in rule (prjmh_temper_rf.f90:363-383), applied when a window has just closed: an estimate
        below 0.2 multiplies dTlog by 1.02, one above 0.5 by 0.98 -- the reference's own directions and thresholds
        (its comment says 0.2-0.3) -- and beta of level l becomes dTlog**-l.  In the reference the workers keep the
        betas they were started with (only the master's table changes), so the rule is nearly inert there; here the
        chains DO get the new ladder, which is why this is an option and not the default.  Returns whether dTlog moved."""
        old = self.dTlog
        if self.acceptance_rate < 0.2:
            self.dTlog *= 1.02
        if self.acceptance_rate > 0.5:
            self.dTlog *= 0.98
        self.ladder_history.append((sweep, self.acceptance_rate, self.dTlog))
        if self.dTlog == old:
            return False
        self.beta_of_level = np.array([self.dTlog ** (-lv) for lv in range(len(self.beta_of_level))])
        return True

    # ----- initialisation -----------------------------------------------------------------------
    def init_from_prior(self):
        pr = self.prior
        self.chains = []
        while len(self.chains) < self.nloc:
            k = int(self.rng.integers(max(pr.kmin, 1), min(pr.kmax, 4) + 1))
            nodes = np.zeros((k, 2))
            nodes[:, 1] = self.rng.uniform(pr.vmin, pr.vmax, k)
            nodes[1:, 0] = self.rng.uniform(0.0, pr.hmx - pr.hmin, k - 1)
            if inside_bounds(nodes, pr):
                self.chains.append(Chain(nodes, float(self.rng.uniform(pr.sdmin, pr.sdmax))))
        self._evaluate_current()

    def init_from_model(self, nodes: np.ndarray, sd: float):
        """Every chain starts at the <base>_map.dat model, as the reference does (prjmh_temper_rf.f90:152-189)."""
        self.chains = [Chain(np.array(nodes, dtype=float).copy(), float(sd)) for _ in range(self.nloc)]
        self._evaluate_current()

    def _evaluate_current(self):
        ss = self.evaluator([c.nodes for c in self.chains])
        for c, s in zip(self.chains, ss):
            c.ss = float(s)
            c.logL = gaussian_loglike(c.ss, self.ndata, c.sd)

    def beta(self, iloc: int) -> float:
        return float(self.beta_of_level[self.level[self.first + iloc]])

    # ----- one sweep of every local chain (EXPLORE_MH_NOVARPAR + EXPLORE_MH) ---------------------
    def _decide(self, idx, proposals, logpr, kind):
        """Evaluate the proposals of chains `idx` in one launch and apply the MH rule (:697-706, :742-751)."""
        if not idx:
            return
        ss_new = self.evaluator(proposals)
        for i, nodes, lp, s in zip(idx, proposals, logpr, ss_new):
            c = self.chains[i]
            logl_new = gaussian_loglike(float(s), self.ndata, c.sd)
            log_ratio = lp + (logl_new - c.logL) * self.beta(i)
            self.stats["prop_" + kind] += 1
            if mh_accept(self.rng.random(), log_ratio):
                c.nodes, c.ss, c.logL = nodes, float(s), logl_new
                self.stats["acc_" + kind] += 1

    def sweep(self):
        pr = self.prior
        rng = self.rng
        # --- birth / death (:669-713) ---
        idx, props, lps = [], [], []
        if pr.kmin != pr.kmax:
            for i, c in enumerate(self.chains):
                u = rng.random()
                k = c.k
                move = 0
                if k == pr.kmax:
                    move = 2 if u <= 0.3333 else 0
                elif k == pr.kmin:
                    move = 1 if u <= 0.3333 else 0
                else:
                    move = 1 if u <= 0.3333 else (2 if u > 0.6666 else 0)
                if move == 1:    # BIRTH_FULL (:997-1095)
                    new = np.array([[(pr.hmx - pr.hmin) * rng.random(), pr.vmin + (pr.vmax - pr.vmin) * rng.random()]])
                    nodes = np.vstack([c.nodes, new])
                    lp = pr.log_pk(k + 1) - pr.log_pk(k)
                elif move == 2 and k > 1:  # DEATH_FULL (:917-995): one of nodes 2..k
                    idel = 1 + int(rng.integers(0, k - 1))
                    nodes = np.delete(c.nodes, idel, axis=0)
                    lp = pr.log_pk(k - 1) - pr.log_pk(k)
                else:
                    continue
                if inside_bounds(nodes, pr):
                    idx.append(i), props.append(nodes), lps.append(lp)
                else:
                    self.stats["prop_bd"] += 1
        self._decide(idx, props, lps, "bd")
        # --- perturbation of every (node, parameter) in turn (:721-759) ---
        pert = np.array([(pr.hmx - pr.hmin) / pr.pertsdsc, (pr.vmax - pr.vmin) / pr.pertsdsc])
        kmax_now = max(c.k for c in self.chains)
        for ivo in range(kmax_now):
            for ipar in (0, 1):
                if ivo == 0 and ipar == 0:
                    continue  # the surface node's depth is not a parameter (:727)
                idx, props, lps = [], [], []
                for i, c in enumerate(self.chains):
                    if ivo >= c.k:
                        continue
                    nodes = c.nodes.copy()
                    nodes[ivo, ipar] += pert[ipar] * math.tan(math.pi * (rng.random() - 0.5))  # PROPOSAL (:1405)
                    if inside_bounds(nodes, pr):
                        idx.append(i), props.append(nodes), lps.append(0.0)
                    else:
                        self.stats["prop_mh"] += 1
                self._decide(idx, props, lps, "mh")
        # --- sigma (EXPLORE_MH :521-575): same rays, so the residual sum is reused ---
        step = (pr.sdmax - pr.sdmin) / pr.pertsdsdsc
        for i, c in enumerate(self.chains):
            if rng.random() >= 0.10:
                sd_new = c.sd + step * rng.standard_normal()
                self.stats["prop_sd"] += 1
                if pr.sdmin <= sd_new <= pr.sdmax:
                    logl_new = gaussian_loglike(c.ss, self.ndata, sd_new)
                    if mh_accept(rng.random(), (logl_new - c.logL) * self.beta(i)):
                        c.sd, c.logL = sd_new, logl_new
                        self.stats["acc_sd"] += 1

    # ----- temperature swaps (TEMPSWP_MH :1329-1357) ----------------------------------------------
    def gather_logl(self) -> np.ndarray:
        """logL of ALL chains, global order: the only data-path collective (one tiny all-gather per swap round)."""
        local = np.array([c.logL for c in self.chains])
        if self.dist is None or self.world == 1:
            return local
        import torch
        t_loc = torch.as_tensor(local, dtype=torch.float64, device=self.dist_device)
        t_all = torch.empty(self.nchains_total, dtype=torch.float64, device=self.dist_device)
        self.dist.all_gather_into_tensor(t_all, t_loc)
        return t_all.cpu().numpy()

    def swap_round(self):
        logl = self.gather_logl()
        # chains ordered by level; partners = neighbours in that order, alternating even / odd pairs
        order = np.argsort(self.level, kind="stable")
        start = self.swap_parity
        self.swap_parity ^= 1
        for a in range(start, self.nchains_total - 1, 2):
            c1, c2 = order[a], order[a + 1]
            b1, b2 = self.beta_of_level[self.level[c1]], self.beta_of_level[self.level[c2]]
            u = self.rng_swap.random()  # drawn on every rank, for every pair: identical streams
            self.stats["prop_swap"] += 1
            if b1 == b2:
                continue
            if u <= math.exp(min((b2 - b1) * (logl[c1] - logl[c2]), 50.0)):
                self.level[c1], self.level[c2] = self.level[c2], self.level[c1]
                self.stats["acc_swap"] += 1

    # ----- driver ---------------------------------------------------------------------------------
    def run(self, nsweeps: int, swap_every: int = 1, burnin: int = 0, thin: int = 1):
        """Returns the T = 1 samples of the LOCAL chains: list of (k, nodes sorted by depth, sd, logL)."""
        samples = []
        for it in range(nsweeps):
            self.sweep()
            if (it + 1) % swap_every == 0 and self.nchains_total > 1:
                a0, p0 = self.stats["acc_swap"], self.stats["prop_swap"]
                self.swap_round()
                if self._window(self.stats["acc_swap"] - a0, self.stats["prop_swap"] - p0) and self.adapt_ladder and it < burnin:
                    self.adapt_after_window(it)
            if it >= burnin and (it - burnin + 1) % thin == 0:  # after every thin-th sweep (prjmh_temper_rf.f90:329-389)
                for i, c in enumerate(self.chains):
                    if self.level[self.first + i] == 0:
                        s = c.nodes[np.argsort(c.nodes[:, 0], kind="stable")]
                        samples.append((c.k, s.copy(), c.sd, c.logL))
        return samples


class NativeSampler(Sampler):
    """Sampler whose sweeps, temperature exchange and sample bookkeeping run on the GPU inside librtmi.so
    (rtmi_pt_run, csrc/rtmi_pt.hip): chain states never leave HBM between read-outs.  `evaluator` must be a
    GpuEvaluator (it uploads the sources the handle then copies); proposals use one counter-based random stream per
    chain.  The exchange is one RCCL all-gather per round when a communicator is given (`comm`), the library's own
    single-process path when every chain is local, or -- rehearsal of N ranks on fewer GPUs, gloo -- a host gather
    handed to rtmi_pt_swap_round_host; the decisions are the same function in all three."""

    comm: object = None

    def __post_init__(self):
        super().__post_init__()
        from . import raymod
        self.raymod = raymod
        self.native = raymod.NativeChains(self.nloc, self.evaluator.ld, self.ndata, self.prior, seed=self.seed,
                                          stream_id=self.rank)
        self.native.set_ladder(self.nchains_total, self.first, self.level, self.beta_of_level)
        self.comm = None
        if self.world > 1 and self.dist is not None and self.dist.get_backend() == "nccl":
            def bcast(b):
                obj = [b]
                self.dist.broadcast_object_list(obj, src=0)
                return obj[0]
            self.comm = raymod.Comm(self.world, self.rank, bcast)

    def _evaluate_current(self):
        for i, c in enumerate(self.chains):
            self.native.set_chain(i, c.nodes, c.sd, self.beta(i))
        self.native.evaluate()
        self._pull(range(self.nloc))

    def _pull(self, which):
        a = self.native.get_all()
        for i in which:
            k = int(a["k"][i])
            self.chains[i] = Chain(np.column_stack([a["z"][i, :k], a["v"][i, :k]]), float(a["sd"][i]),
                                   float(a["logL"][i]), float(a["ss"][i]))

    def _refresh_stats(self):
        st = self.native.stats()
        for k in ("acc_bd", "prop_bd", "acc_mh", "prop_mh", "acc_sd", "prop_sd"):
            self.stats[k] = st[k]
        self.evaluator.launches, self.evaluator.models = st["launches"], st["models"]
        lev, acc, prop = self.native.get_ladder()
        self.level = lev.astype(int)
        self.stats["acc_swap"], self.stats["prop_swap"] = acc, prop

    def sweep(self):
        self.native.sweep(1)

    def swap_round(self):
        if self.world == 1 or self.comm is not None:
            self.native.swap_round(self.comm)
        else:
            # N ranks rehearsing without RCCL (gloo): the gather is done here, the decision on the device as always
            self.native.swap_round_host(self.gather_logl())

    def gather_logl(self) -> np.ndarray:
        local = self.native.logl()
        if self.dist is None or self.world == 1:
            return local
        import torch
        t_loc = torch.as_tensor(local, dtype=torch.float64)
        t_all = torch.empty(self.nchains_total, dtype=torch.float64)
        self.dist.all_gather_into_tensor(t_all, t_loc)
        return t_all.numpy()

    def run(self, nsweeps: int, swap_every: int = 1, burnin: int = 0, thin: int = 1):
        """Returns the T = 1 samples of the LOCAL chains: list of (k, nodes sorted by depth, sd, logL).  After
        `burnin` sweeps the chains at T = 1 are recorded after every `thin`-th sweep, on the device."""
        device_loop = self.world == 1 or self.comm is not None
        swap_every = swap_every if self.nchains_total > 1 else 0
        nrec = max(0, nsweeps - burnin)
        self.native.sample_buffer(max(1, (nrec // max(1, thin) + 1) * self.nloc))
        if device_loop and self.adapt_ladder and swap_every:
            # burn-in in pieces of one acceptance window each: the counters come back, the rule is applied, the new betas go
            # down (rtmi_pt_set_ladder_betas: levels, counters and the exchange's random stream stay as they are)
            per_round = max(1, (self.nchains_total - 1) // 2)
            piece = swap_every * (self.acceptance_window // per_round + 1)
            done, nb = 0, min(burnin, nsweeps)
            _, a0, p0 = self.native.get_ladder()
            while done < nb:
                n = min(piece, nb - done)
                self.native.run(n, swap_every, self.comm, 0)
                done += n
                _, a1, p1 = self.native.get_ladder()
                if self._window(a1 - a0, p1 - p0) and done < nb and self.adapt_after_window(done):
                    self.native.set_ladder_betas(self.beta_of_level)
                a0, p0 = a1, p1
            self.native.run(nrec, swap_every, self.comm, thin)
        elif device_loop:
            self.native.run(min(burnin, nsweeps), swap_every, self.comm, 0)
            self.native.run(nrec, swap_every, self.comm, thin)
        else:
            for it in range(nsweeps):
                self.native.run(1, 0, None, 1 if (it >= burnin and (it - burnin + 1) % thin == 0) else 0)
                if swap_every and (it + 1) % swap_every == 0:
                    _, a0, p0 = self.native.get_ladder()
                    self.swap_round()
                    _, a1, p1 = self.native.get_ladder()
                    if self._window(a1 - a0, p1 - p0) and self.adapt_ladder and it < burnin and self.adapt_after_window(it):
                        self.native.set_ladder_betas(self.beta_of_level)
        s = self.native.samples()
        out = []
        self.sample_chains, self.sample_sweeps = [], []   # parallel to the returned list: which (global) chain, after which sweep
        for i in np.lexsort((s["chain"], s["sweep"])):  # the ring is filled through an atomic counter: order by (sweep, chain)
            k = int(s["k"][i])
            nodes = np.column_stack([s["z"][i, :k], s["v"][i, :k]])
            out.append((k, nodes[np.argsort(nodes[:, 0], kind="stable")], float(s["sd"][i]), float(s["logL"][i])))
            self.sample_chains.append(int(s["chain"][i]))
            self.sample_sweeps.append(int(s["sweep"][i]))
        self._refresh_stats()
        self._pull(range(self.nloc))
        return out

    def sync(self):
        """Bring every local chain's state and the counters back from the library."""
        self._refresh_stats()
        self._pull(range(self.nloc))
