"""BASELINE.json configs[4] -- N = 524 288 envs sharded over 8 GPUs, mixed BS1/2/3 -- rehearsed on ONE GPU (no 8-GPU node is
available to the build): rank r of 8 (65 536 envs, its env offset, scenario assignment and episode plan from rl_ptg_amd.dist) must
reproduce envs [r * 65 536, (r + 1) * 65 536) of a single 524 288-env engine bit for bit -- observations, rewards, done flags, RNG
streams, episode order and finished-episode lists.  Reference coupling preserved: the module-global ep_index
(/root/reference/env/ptg_gym_env.py:9,487-493)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rank_of_8_equals_its_slice_of_the_524288_env_batch():
    import torch
    from rl_ptg_amd import dist as ptg_dist
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import EnvSpec, synthetic_spec
    from rl_ptg_amd.synthetic import sticky_actions_device
    world, n = 8, 65536
    n_total = world * n
    specs = [synthetic_spec(scenario=sc, operation="OP2", eps_len_d=1, train_steps=200000)[0] for sc in (1, 2, 3)]
    spec = EnvSpec.merge_scenarios(specs)
    K, CH = 150, 10                                       # 1-day episodes: every env terminates at step 139 (one synchronised reset inside)
    dev = torch.device("cuda", 0)

    def make(n_envs, rank, ws):
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n_envs, device=0, out_dtype="float32", obs_layout="row")
        first_ptr, stride = ptg_dist.episode_plan(n_total, ws, rank)
        eng.set_global_env_offset(first_ptr - n_total)
        eng.set_market_assignment(ptg_dist.mixed_scenario_assignment(n_total, ws, rank, 3))
        eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
        eng.set_noise_rng(2025)
        eng.reset()
        return eng

    ranks = (0, 5, 7)
    full = make(n_total, 0, 1)
    shards = {r: make(n, r, world) for r in ranks}
    acts = sticky_actions_device(K, n_total, seed=9, device=dev, p_switch=0.2)
    n_done = 0
    for c in range(0, K, CH):
        of, rf, df = full.rollout(acts[c:c + CH])
        for r, eng in shards.items():
            lo = r * n
            o, rw, d = eng.rollout(acts[c:c + CH, lo:lo + n].contiguous())
            assert torch.equal(of[:, lo:lo + n], o), (c, r)
            assert torch.equal(rf[:, lo:lo + n], rw) and torch.equal(df[:, lo:lo + n], d), (c, r)
        n_done += int(df.sum())
        del of, rf, df
    full.sync()
    assert n_done == n_total                              # everybody finished exactly one episode
    fields = ["meth_state", "i", "j", "k", "hot_cold", "current_action", "act_ep_d", "ep_ptr", "noise_count", "market_set", "cum_rew", "T_cat"]
    fs = {f: full.get_state(f) for f in fields}
    rF, lF, iF = full.finished_episodes()
    assert len(rF) == n_total and set(lF.tolist()) == {139}
    order = np.argsort(iF)
    rF = rF[order]
    for r, eng in shards.items():
        lo = r * n
        for f in fields:
            assert np.array_equal(eng.get_state(f), fs[f][lo:lo + n]), (r, f)
        rr, ll, ii = eng.finished_episodes()
        assert len(rr) == n and np.array_equal(rr[np.argsort(ii)], rF[lo:lo + n]), r
        eng.close()
    # the three business scenarios are really mixed inside every shard (global env e -> scenario e % 3), and episodes differ between envs
    ms = fs["market_set"]
    assert np.array_equal(ms, np.arange(n_total) % 3)
    assert len(np.unique(fs["act_ep_d"])) > 1
    full.close()
