"""Parallel-tempering driver (raytracerfortran_amd/tempering.py): host logic on CPU with a toy likelihood, the
multi-rank temperature exchange over gloo (world_size 2), and -- on the GPU box -- a short run on the reference's
test_1 problem through the batched HIP likelihood."""
import math
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from raytracerfortran_amd import tempering as T


class FlatEvaluator:
    """Constant residual sum: the sampler then draws from the prior."""
    def __call__(self, node_sets):
        return np.full(len(node_sets), 1.0)


class ToyEvaluator:
    """Residuals of a 1-parameter toy problem: pulls the mean node velocity towards 4000 m/s."""
    def __call__(self, node_sets):
        return np.array([((nodes[:, 1].mean() - 4000.0) / 50.0) ** 2 for nodes in node_sets])


def test_layering_matches_loglhood_rt_call_arguments():
    nodes = np.array([[0.0, 3100.0], [2500.0, 5800.0], [1000.0, 2400.0]])
    v, z = T.layering(nodes)
    assert v.tolist() == [3100.0, 2400.0, 5800.0] and z.tolist() == [1000.0, 2500.0]
    v, z = T.layering(np.array([[0.0, 4321.0]]))  # k = 1 (loglhood.f90:137-144)
    assert v.tolist() == [4321.0, 4321.0] and z.tolist() == [9999.9]


def test_bounds_follow_checkbounds():
    pr = T.Prior()
    ok = np.array([[0.0, 3000.0], [500.0, 4000.0], [1200.0, 5000.0]])
    assert T.inside_bounds(ok, pr)
    for bad in (np.array([[0.0, 1400.0]]), np.array([[0.0, 3000.0], [50.0, 4000.0]]),        # Vp < 1500; thickness < hmin
                np.array([[0.0, 3000.0], [500.0, 4000.0], [550.0, 5000.0]]),                  # thin inner layer
                np.array([[0.0, 3000.0], [10001.0, 4000.0]]), np.array([[0.0, 3000.0], [-5.0, 4000.0]])):
        assert not T.inside_bounds(bad, pr)


def test_flat_likelihood_samples_the_prior_on_k():
    """Birth/death bookkeeping: with a constant likelihood k follows the truncated Poisson prior
    (read_input.f90:78-81) when the minimum-thickness constraint is negligible."""
    pr = T.Prior(kmin=1, kmax=6, lam=2.0, hmin=1e-6, hmx=10000.0)
    s = T.Sampler(pr, FlatEvaluator(), ndata=20, nchains_total=8, nptchains1=8, seed=3)
    s.init_from_prior()
    samples = s.run(4000, burnin=500)
    ks = np.array([k for k, *_ in samples])
    w = np.array([math.exp(pr.log_pk(k)) for k in range(1, 7)])
    w /= w.sum()
    freq = np.array([(ks == k).mean() for k in range(1, 7)])
    assert np.abs(freq - w).max() < 0.03, (freq, w)
    assert s.stats["acc_bd"] > 0 and s.stats["acc_mh"] > 0 and s.stats["acc_sd"] > 0


def test_tempering_helps_and_levels_stay_a_permutation():
    pr = T.Prior(kmin=1, kmax=3)
    s = T.Sampler(pr, ToyEvaluator(), ndata=20, nchains_total=6, nptchains1=2, dTlog=1.6, seed=5)
    s.init_from_prior()
    samples = s.run(600, burnin=200)
    assert sorted(s.level.tolist()) == [0, 0, 1, 2, 3, 4]
    assert s.stats["acc_swap"] > 0
    vmean = np.mean([nodes[:, 1].mean() for _, nodes, _, _ in samples])
    assert abs(vmean - 4000.0) < 300.0


def _pt_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from raytracerfortran_amd import tempering as TT
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pr = TT.Prior(kmin=1, kmax=3)
    s = TT.Sampler(pr, ToyEvaluator(), ndata=20, nchains_total=6, nptchains1=2, dTlog=1.6, seed=11,
                   rank=rank, world=world, dist=dist, dist_device="cpu")
    s.init_from_prior()
    samples = s.run(300, burnin=100)
    q.put((rank, s.level.tolist(), s.stats["acc_swap"], s.stats["prop_swap"], len(samples),
           [float(c.logL) for c in s.chains]))
    dist.destroy_process_group()


def test_two_ranks_take_identical_swap_decisions_gloo():
    """Chains sharded over 2 ranks: after every all-gather both ranks hold the same temperature assignment, the
    number of T = 1 samples adds up, and swaps do happen across the rank boundary."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 2000
    procs = [ctx.Process(target=_pt_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, lev0, acc0, prop0, n0, _), (_, lev1, acc1, prop1, n1, _) = res
    assert lev0 == lev1 and sorted(lev0) == [0, 0, 1, 2, 3, 4]
    assert (acc0, prop0) == (acc1, prop1) and acc0 > 0
    assert n0 + n1 == 2 * 200  # two chains at T = 1 at every kept sweep, wherever they live
    assert lev0 != [0, 0, 1, 2, 3, 4]  # the ladder moved


@pytest.mark.gpu
def test_tempered_run_on_test_1(kat):
    """BASELINE config 3 in miniature: 8 tempered chains on the reference's test_1 data through the batched HIP
    likelihood.  The T = 1 chain must stay in the high-likelihood region of the true model (logL 55.1 at sd = 0.02)
    and its logL bookkeeping must equal an independent evaluation by the oracle."""
    from oracle import oracle as O
    m = kat["model"]
    off, dep, obs = kat["test_1_src_offset"], kat["test_1_src_depth"], kat["test_1_obs_RT"]
    ev = T.GpuEvaluator(off, dep, obs)
    s = T.Sampler(T.Prior(), ev, ndata=20, nchains_total=8, nptchains1=1, dTlog=1.4, seed=2)
    nodes = np.column_stack([[0.0] + m["depths"], m["vels"]])
    s.init_from_model(nodes, 0.02)
    assert abs(s.chains[0].logL - kat["test_1_map_logL_survey"]["value"]) < 1e-8
    samples = s.run(400, burnin=100)
    assert len(samples) == 300 and ev.launches > 400
    ll = np.array([x[3] for x in samples])
    assert ll.max() > 45.0 and np.median(ll) > 30.0
    for c in s.chains:  # bookkeeping vs the oracle
        v, z = T.layering(c.nodes)
        t = O.oracle_dff(v, z, off, dep)
        ref = O.oracle_loglhood_rt(obs, t, c.sd, stable=True)
        assert abs(c.logL - ref) <= 1e-9 * max(1.0, abs(ref))
    assert s.stats["acc_mh"] > 0 and s.stats["acc_swap"] > 0


@pytest.mark.gpu
def test_native_sweep_samples_the_prior_at_beta_zero(kat):
    """rtmi_pt_sweep (csrc/rtmi_pt.cpp) with beta = 0: the likelihood drops out of every acceptance ratio, so k must
    follow the truncated Poisson prior -- birth/death/bounds bookkeeping of the native sweep -- while every proposal
    still goes through the GPU likelihood."""
    import raytracerfortran_amd as rt
    from raytracerfortran_amd import raymod
    rt.set_sources(kat["test_1_src_offset"], kat["test_1_src_depth"], kat["test_1_obs_RT"])
    pr = T.Prior(kmin=1, kmax=6, lam=2.0, hmin=1e-6, hmx=10000.0)
    nat = raymod.NativeChains(16, 32, 20, pr, seed=4, stream_id=0)
    for i in range(16):
        nat.set_chain(i, np.array([[0.0, 4000.0]]), 0.02, 0.0)
    nat.evaluate()
    nat.sweep(300)
    ks = []
    for it in range(1500):
        nat.sweep(1)
        ks += [nat.get_chain(i)[0].shape[0] for i in range(16)]
    ks = np.array(ks)
    w = np.array([math.exp(pr.log_pk(k)) for k in range(1, 7)])
    w /= w.sum()
    freq = np.array([(ks == k).mean() for k in range(1, 7)])
    assert np.abs(freq - w).max() < 0.03, (freq, w)
    st = nat.stats()
    assert st["acc_bd"] > 0 and st["acc_mh"] > 0 and st["acc_sd"] > 0 and st["models"] > st["launches"] > 1500


@pytest.mark.gpu
def test_native_sampler_on_test_1(kat):
    """NativeSampler (sweeps inside librtmi.so, swaps here) on the reference's test_1 problem: logL bookkeeping of
    every chain equals the oracle, the T = 1 chain stays in the high-likelihood region, swaps happen."""
    from oracle import oracle as O
    m = kat["model"]
    off, dep, obs = kat["test_1_src_offset"], kat["test_1_src_depth"], kat["test_1_obs_RT"]
    ev = T.GpuEvaluator(off, dep, obs)
    s = T.NativeSampler(T.Prior(), ev, ndata=20, nchains_total=8, nptchains1=1, dTlog=1.4, seed=2)
    s.init_from_model(np.column_stack([[0.0] + m["depths"], m["vels"]]), 0.02)
    assert abs(s.chains[0].logL - kat["test_1_map_logL_survey"]["value"]) < 1e-8
    samples = s.run(600, burnin=100)
    s.sync()
    assert len(samples) == 500
    ll = np.array([x[3] for x in samples])
    assert ll.max() > 45.0 and np.median(ll) > 30.0
    for c in s.chains:
        v, z = T.layering(c.nodes)
        t = O.oracle_dff(v, z, off, dep)
        ref = O.oracle_loglhood_rt(obs, t, c.sd, stable=True)
        assert abs(c.logL - ref) <= 1e-9 * max(1.0, abs(ref))
        assert T.inside_bounds(c.nodes, s.prior)
    assert s.stats["acc_mh"] > 0 and s.stats["acc_swap"] > 0 and s.stats["acc_bd"] >= 0


def _native_pt_worker(rank, world, port, q, off, dep, obs, nodes):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import raytracerfortran_amd as rt
    from raytracerfortran_amd import tempering as TT
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rt.init(0)  # both ranks rehearse on the one GPU of the box; on a node each rank has its own
    ev = TT.GpuEvaluator(off, dep, obs)
    s = TT.NativeSampler(TT.Prior(), ev, ndata=len(obs), nchains_total=8, nptchains1=2, dTlog=1.3, seed=5,
                         rank=rank, world=world, dist=dist, dist_device="cpu")
    s.init_from_model(np.array(nodes), 0.02)
    samples = s.run(200, burnin=50)
    s.sync()
    q.put((rank, s.level.tolist(), s.stats["acc_swap"], len(samples), [float(c.logL) for c in s.chains],
           [c.nodes.tolist() for c in s.chains], [c.sd for c in s.chains]))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_chains_sharded_over_two_ranks_native(kat):
    """BASELINE config 4's structure in miniature: 8 tempered chains sharded over 2 ranks (here both on the one
    GPU of the box, all-gather over gloo; on a node one rank per GPU over RCCL), native sweeps, temperature swaps
    decided identically on both ranks, every chain's logL equal to the oracle's for its final state."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    m = kat["model"]
    off, dep, obs = kat["test_1_src_offset"], kat["test_1_src_depth"], kat["test_1_obs_RT"]
    nodes = np.column_stack([[0.0] + m["depths"], m["vels"]]).tolist()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 2000
    procs = [ctx.Process(target=_native_pt_worker, args=(r, 2, port, q, off, dep, obs, nodes)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, lev0, sw0, n0, ll0, nd0, sd0), (_, lev1, sw1, n1, ll1, nd1, sd1) = res
    assert lev0 == lev1 and sorted(lev0) == [0, 0, 1, 2, 3, 4, 5, 6] and sw0 == sw1 > 0
    assert n0 + n1 == 2 * 150
    for ll, nd, sd in zip(ll0 + ll1, nd0 + nd1, sd0 + sd1):
        v, z = T.layering(np.array(nd))
        ref = O.oracle_loglhood_rt(obs, O.oracle_dff(v, z, off, dep), sd, stable=True)
        assert abs(ll - ref) <= 1e-9 * max(1.0, abs(ref))
