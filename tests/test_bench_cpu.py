"""bench.py's line, checked on CPU: the `chain_parallel` block the driver's scaling run must carry (Next #3) and
the honesty labels of the roofline block (Next #7).  No GPU: the exchange rule runs through the C symbol rtmi_pt_swap_decide on the
host (what the device's pt_swap_kernel runs), the gather through gloo."""
import json
import os
import socket
import sys

import numpy as np

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, corrupt):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    from raytracerfortran_amd import raymod
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nloc, nct = 32, 32 * world
    level = np.repeat(np.arange(nct // 2), 2).astype(np.int32)            # two chains per temperature level
    beta = 1.15 ** -np.arange(float(nct // 2))
    rng = np.random.default_rng(100 + rank)                               # every rank owns the logL of ITS chains only
    for rnd in range(25):
        mine = torch.tensor(rng.normal(27000.0, 30.0, nloc))
        allv = torch.empty(nct, dtype=torch.float64)
        dist.all_gather_into_tensor(allv, mine)                           # the path's one collective: 8 B per chain
        level, acc, prop = raymod.swap_decide(allv.numpy(), level, beta, seed=7, round_=rnd)
    if corrupt and rank == 1:
        level = level.copy()
        level[[0, 1]] = level[[1, 0]] if level[0] != level[1] else (level[0] + 1) % (nct // 2)
    blk = bench.chain_parallel_block(level, rccl_ranks=world, world=world, elapsed_s=0.05, sweeps=25, models=1.2e4, nsrc=10_000,
                                     nloc=nloc, dist=dist, red_device="cpu", rccl_info={"rccl_version": 22100, "rccl_library": "x"},
                                     engine="test", exchange="gloo")
    q.put((rank, blk))
    dist.destroy_process_group()


def _run(corrupt):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, corrupt)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_chain_parallel_block_two_ranks_gloo():
    """Two ranks rehearse the sharded ladder: after 25 swap rounds decided by the library's rule from all-gathered logL both must
    hold the same level vector -- the block says so (hash all-reduced MIN/MAX) and carries what an N-GPU line has to show."""
    res = _run(corrupt=False)
    b0, b1 = res[0], res[1]
    assert b0["level_hash"] == b1["level_hash"] and b0["level_hash_equal_across_ranks"] and b1["level_hash_equal_across_ranks"]
    for key in ("rccl_ranks", "ranks", "chains", "chains_per_gpu", "sources", "sweeps", "ms_per_sweep", "models_per_s",
                "travel_times_per_s", "swap_every", "exchange", "engine", "rccl_version", "rccl_library"):
        assert key in b0, key
    assert b0["rccl_ranks"] == 2 and b0["chains"] == 64 and b0["chains_per_gpu"] == 32
    assert abs(b0["ms_per_sweep"] - 2.0) < 1e-9 and abs(b0["travel_times_per_s"] - 1.2e4 * 1e4 / 0.05) < 1e-3
    json.dumps(b0)


def test_chain_parallel_block_flags_ranks_that_disagree():
    res = _run(corrupt=True)
    assert res[0]["level_hash"] != res[1]["level_hash"]
    assert not res[0]["level_hash_equal_across_ranks"] and not res[1]["level_hash_equal_across_ranks"]


def test_roofline_block_labels_what_was_not_measured(tmp_path, monkeypatch):
    """roofline.kernel is whatever the library's dispatch reported; traffic and VALU counts come from a committed profile and say
    so, with that profile's build commit."""
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles" / "r99z"
    prof.mkdir(parents=True)
    name = "void rtmi::fused_kernel<double, false, true, false, false, 1, false>(rtmi::FusedArgsT<double>)"
    other = "void rtmi::fused_kernel<double, false, false, true, false, 1, true>(rtmi::FusedArgsT<double>)"   # the sampler's: same command
    json.dump({"build_commit": "abc1234", name: {"hbm_bytes_per_launch": {"total": 24.6e6},
                                                  "SQ_INSTS_VALU": {"mean_per_launch": 39.4e6}},
               other: {"hbm_bytes_per_launch": {"total": 2.8e6}, "SQ_INSTS_VALU": {"mean_per_launch": 1.0e6}}},
              open(prof / "pmc_summary.json", "w"))      # (the counters are looked up by the EXACT name: round 4 once quoted the sampler's)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    roof, valu = bench.c2_roofline(1_000_000, 0.1, name)
    assert roof["kernel"] == name and roof["bound"] == "hbm" and abs(roof["achieved"] - 240.0) < 1e-9
    assert abs(roof["frac"] - 240.0 / 8000.0) < 1e-12 and roof["traffic"] == 24.6e6
    assert "NOT MEASURED IN THIS RUN" in roof["traffic_source"] and "abc1234" in roof["traffic_source"] and "r99z" in roof["traffic_source"]
    assert "NOT MEASURED IN THIS RUN" in valu["source"] and "abc1234" in valu["source"]
    assert "useful_frac_of_peak" not in valu and "useful_term_instr_per_travel_time" not in valu   # no literals posing as measurements
    # other sizes: no borrowed counters at all
    roof2, valu2 = bench.c2_roofline(500_000, 0.05, name)
    assert roof2["traffic"] is None and "executed_frac_of_peak" not in valu2


def test_kernel_name_of_the_line_is_the_one_rocprof_saw():
    """The newest committed kernel trace of the bench command and the name the library reports for that launch agree (the name is
    composed by the launcher from the template arguments it dispatched on)."""
    import csv
    import glob
    import re
    traces = sorted(glob.glob(os.path.join(ROOT, "profiles", "r04*", "kernel_stats.csv")))
    if not traces:
        import pytest
        pytest.skip("no round-4 kernel trace committed yet")
    names = [r["Name"] for r in csv.DictReader(open(traces[-1]))]
    fused = [n for n in names if "fused_kernel<double" in n]
    assert fused, names
    src = open(os.path.join(ROOT, "raytracerfortran_amd", "csrc", "rtmi_kernels.hip")).read()
    fmt = re.search(r'note_kernel\("(void rtmi::fused_kernel<%s, %s, %s, %s, %s, %d, false>\(rtmi::FusedArgsT<%s>\))"', src)
    assert fmt, "the launcher no longer records the kernel it dispatches"
    want = fmt.group(1) % ("double", "false", "true", "false", "false", 1, "double")   # C2: travel times only, row tables
    assert want in fused, (want, fused)


def test_fence_of_the_chain_parallel_block(tmp_path):
    """bench.fenced: a block that raises costs the line its object, not the line; a block that never returns is abandoned by
    the watchdog, which lets rank 0 write the line and ends the process with exit code 0 (every rank carries the same fence)."""
    import subprocess
    import bench
    out, broken = bench.fenced(lambda: {"ok": 1}, 30.0, None)
    assert out == {"ok": 1} and not broken

    def boom():
        raise RuntimeError("rtmi_comm_create failed: unhandled system error")
    out, broken = bench.fenced(boom, 30.0, None)
    assert broken and "rtmi_comm_create failed" in out["error"] and out["error"].startswith("RuntimeError")
    # the hang: a child process whose block sleeps for ever
    code = ("import sys, time, json; sys.path.insert(0, %r); import bench\n"
            "bench.fenced(lambda: time.sleep(600), 1.0, lambda c: print(json.dumps({'value': 1.0, 'chain_parallel': c}), flush=True))\n"
            "print('not reached')\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and "not reached" not in p.stdout
    assert "abandoned" in json.loads(lines[0])["chain_parallel"]["error"]
