"""N > 1 path on CPU: world_size-2 gloo process group.  Sharding arithmetic, the episode plan against the oracle's shared
ep_index order, and the one collective (all-gather of finished-episode returns) with ragged per-rank counts."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
from rl_ptg_amd import dist as ptg_dist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank r finished r + 2 episodes (ragged): returns r*100 + i, lengths 10 + i
        r = np.array([rank * 100.0 + i for i in range(rank + 2)])
        l = np.array([10 + i for i in range(rank + 2)])
        ra, la = ptg_dist.all_gather_finished(r, l)
        # more than the inline capacity on one rank only (rank 1: 23 episodes, rank 0: 3): the second collective
        r2 = np.array([rank * 1000.0 + i * 0.5 for i in range(rank * 20 + 3)])
        l2 = np.array([100 + i for i in range(rank * 20 + 3)])
        ra2, la2 = ptg_dist.all_gather_finished(r2, l2)
        rz, lz = ptg_dist.all_gather_finished(np.zeros(0), np.zeros(0, np.int64))     # nobody finished anything
        assert len(rz) == 0 and len(lz) == 0
        # the episode-boundary decision is the same on every rank: no rank has one (both far from the end), one rank's batch is
        # de-synchronised (0 = unknown), one rank's episode ends inside the window
        assert ptg_dist.episode_boundary_in_window(4000, 20) is False
        assert ptg_dist.episode_boundary_in_window(0 if rank == 1 else 4000, 20) is True
        assert ptg_dist.episode_boundary_in_window(15 if rank == 0 else 4000, 20) is True
        assert ptg_dist.episode_boundary_in_window(21 if rank == 0 else 4000, 20) is False
        # reward-normalisation moments: rank r holds envs [r*5, r*5+3+2r) of a ragged split; merged = moments over all envs
        x = np.random.default_rng(42).normal(2.0, 3.0, (4, 8))[:, rank * 3:rank * 3 + 3 + 2 * rank]
        mom = np.stack([np.full(4, x.shape[1], float), x.mean(1), ((x - x.mean(1, keepdims=True)) ** 2).sum(1)], -1)
        merged = ptg_dist.all_merge_moments(torch.from_numpy(mom)).numpy()
        # the oracle stands in for the GPU engine: shard `rank` of a 6-env batch, episode order of the shared ep_index
        case = "synth_bs2_op2_term_penalty"
        tr, consts, tables, market = H.load_traj(case)
        n_total, n = 6, 3
        lo, hi = ptg_dist.shard_range(n_total, world, rank)
        first_ptr, stride = ptg_dist.episode_plan(n_total, world, rank)
        env = H.po.OracleVecEnv(consts, tables, market, n, ep_index0=first_ptr - n)   # constructor consumes n, reset the next n
        env.reset()
        ints, _ = env.state()
        q.put((rank, ra.tolist(), la.tolist(), ints[:, 11].tolist(), (lo, hi), (first_ptr, stride), merged.tolist(), ra2.tolist(), la2.tolist()))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    exp_r = [0.0, 1.0, 100.0, 101.0, 102.0]
    allx = np.random.default_rng(42).normal(2.0, 3.0, (4, 8))
    exp_r2 = [i * 0.5 for i in range(3)] + [1000.0 + i * 0.5 for i in range(23)]
    exp_l2 = [100 + i for i in range(3)] + [100 + i for i in range(23)]
    for rank, ra, la, act_d, rng, plan, merged, ra2, la2 in out:
        assert ra == exp_r and la == [10, 11, 10, 11, 12]
        assert ra2 == exp_r2 and la2 == exp_l2
        assert rng == (rank * 3, rank * 3 + 3) and plan == (6 + rank * 3, 6)
        merged = np.array(merged)                   # both ranks hold the moments of all 8 envs (3 on rank 0 + 5 on rank 1)
        assert merged[:, 0].tolist() == [8.0] * 4
        np.testing.assert_allclose(merged[:, 1], allx.mean(1), rtol=1e-14)
        np.testing.assert_allclose(merged[:, 2] / 8.0, allx.var(1), rtol=1e-13)
    assert out[0][6] == out[1][6]                   # identical floating-point result on every rank
    # the episode every shard env starts in == what 6 reference envs sharing ep_index get (constructor 0..5, reset 6..11)
    tr, consts, tables, market = H.load_traj("synth_bs2_op2_term_penalty")
    full = H.po.OracleVecEnv(consts, tables, market, 6)
    full.reset()
    ints, _ = full.state()
    assert out[0][3] + out[1][3] == ints[:, 11].tolist()
    assert ints[:, 11].tolist() == (market["eps_ind"][6:12] * consts["eps_len_d"]).astype(int).tolist()


def test_shard_helpers():
    assert ptg_dist.shard_range(524288, 8, 3) == (196608, 262144)
    assert ptg_dist.episode_plan(524288, 8, 3) == (524288 + 196608, 524288)
    with pytest.raises(ValueError):
        ptg_dist.shard_range(10, 4, 0)
    a = ptg_dist.mixed_scenario_assignment(12, 2, 1, 3)
    assert a.tolist() == [0, 1, 2, 0, 1, 2] and a.dtype == np.uint8
    r, l = ptg_dist.all_gather_finished([1.0, 2.0], [3, 4])        # no process group: identity
    assert r.tolist() == [1.0, 2.0] and l.tolist() == [3, 4]


def _worker8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_total = 524288                                          # BASELINE config 5: 524 288 envs over 8 ranks
        lo, hi = ptg_dist.shard_range(n_total, world, rank)
        first_ptr, stride = ptg_dist.episode_plan(n_total, world, rank)
        # every rank finishes all of its envs on the same step (a synchronised batch): 19 of them reported here, ragged by rank, so that
        # ranks below and above the inline capacity (15) mix in one call
        cnt = 10 + 2 * rank
        r = (lo + np.arange(cnt)) * 0.25
        l = np.full(cnt, 4603)
        ra, la = ptg_dist.all_gather_finished(r, l)
        flags = [ptg_dist.episode_boundary_in_window(12 if rank == 5 else 4000, 20), ptg_dist.episode_boundary_in_window(4000, 20)]
        x = np.random.default_rng(7).normal(0.0, 2.0, (3, 64))[:, rank * 8:(rank + 1) * 8]
        mom = np.stack([np.full(3, 8.0), x.mean(1), ((x - x.mean(1, keepdims=True)) ** 2).sum(1)], -1)
        merged = ptg_dist.all_merge_moments(torch.from_numpy(mom)).numpy()
        a = ptg_dist.mixed_scenario_assignment(n_total, world, rank, 3)
        q.put((rank, (lo, hi), (first_ptr, stride), ra.tolist(), la.tolist(), flags, merged.tolist(), int(a[0]), int(a[-1]), len(a)))
    finally:
        dist.destroy_process_group()


def test_world_size_8_gloo():
    """The rank count of BASELINE config 5 (8 x MI355X), rehearsed on CPU: shards tile the batch, the episode plans tile the shared ep_index
    order, one ragged all-gather with ranks on both sides of the inline capacity, the boundary decision and the moment merge agree on all ranks."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n_total = 524288
    exp_r, exp_l = [], []
    for rank in range(world):
        cnt = 10 + 2 * rank
        exp_r += ((rank * 65536 + np.arange(cnt)) * 0.25).tolist()
        exp_l += [4603] * cnt
    allx = np.random.default_rng(7).normal(0.0, 2.0, (3, 64))
    for rank, rng, plan, ra, la, flags, merged, a0, a1, alen in out:
        assert rng == (rank * 65536, (rank + 1) * 65536)                      # contiguous, equal shards
        assert plan == (n_total + rank * 65536, n_total)                      # env e of the job takes eps_ind[N + e + m N]
        assert ra == exp_r and la == exp_l                                    # rank order, every rank holds everything
        assert flags == [True, False]
        merged = np.array(merged)
        assert merged[:, 0].tolist() == [64.0] * 3
        np.testing.assert_allclose(merged[:, 1], allx.mean(1), rtol=1e-13)
        np.testing.assert_allclose(merged[:, 2] / 64.0, allx.var(1), rtol=1e-12)
        assert alen == 65536 and a0 == (rank * 65536) % 3 and a1 == ((rank + 1) * 65536 - 1) % 3      # scenario of GLOBAL env e = e mod 3
    assert all(o[6] == out[0][6] for o in out)                                # bit-identical merge on every rank
