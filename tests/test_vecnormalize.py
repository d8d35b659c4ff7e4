"""Reward normalisation of the reference's VecNormalize(env, norm_obs=False) wrapper (src/rl_utils.py:453), §8(f) rank 2.
Checker: oracle/vecnormalize_oracle.py, the restated SB3 2.0.0a13 algorithm (un-vendored: parity vs SB3 itself is unpinned)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vecnormalize_oracle as vo          # noqa: E402
from rl_ptg_amd import dist as ptg_dist   # noqa: E402


def test_oracle_known_answers():
    """Hand-derived values of the published algorithm: one env, constant reward 1, gamma 0.5 -> returns 1, 1.5, 1.75;
    RunningMeanStd starts at (0, 1, 1e-4) and a single-element batch has variance 0."""
    n = vo.RewardNormalizer(1, gamma=0.5, epsilon=0.0, clip_reward=100.0)
    out = [float(n.step(np.array([1.0]), np.array([False]))[0]) for _ in range(3)]
    c0 = 1e-4
    m1 = 0.0 + 1.0 * 1 / (c0 + 1)
    v1 = (1.0 * c0 + 0.0 + 1.0 ** 2 * c0 * 1 / (c0 + 1)) / (c0 + 1)
    assert abs(float(n.ret_rms.count) - (3 + c0)) < 1e-12
    assert abs(out[0] - 1.0 / np.sqrt(v1)) < 1e-9 * out[0]
    d2 = 1.5 - m1
    v2 = (v1 * (c0 + 1) + 0.0 + d2 ** 2 * (c0 + 1) * 1 / (c0 + 2)) / (c0 + 2)
    assert abs(out[1] - 1.0 / np.sqrt(v2)) < 1e-9 * out[1]
    n.step(np.array([1.0]), np.array([True]))
    assert n.returns[0] == 0.0                              # returns[dones] = 0
    assert float(vo.RewardNormalizer(2, clip_reward=0.5).step(np.array([5.0, -5.0]), np.array([0, 0]))[0]) == 0.5


def test_merge_moments_equals_numpy_on_concatenated_shards():
    rng = np.random.default_rng(0)
    T = 6
    shards = [rng.normal(3, 2, (T, n)) for n in (7, 64, 1, 130)]
    parts = np.stack([np.stack([np.full(T, x.shape[1], float), x.mean(1), ((x - x.mean(1, keepdims=True)) ** 2).sum(1)], -1) for x in shards])
    m = ptg_dist.merge_moments(parts)
    allx = np.concatenate(shards, 1)
    assert m[:, 0].tolist() == [202.0] * T
    np.testing.assert_allclose(m[:, 1], allx.mean(1), rtol=1e-14)
    np.testing.assert_allclose(m[:, 2] / m[:, 0], allx.var(1), rtol=1e-13)


def _rollout_rewards(n, K, layout="feature", out_dtype="float32"):
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=1, train_steps=400000)
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=out_dtype, obs_layout=layout)
    eng.set_episode_plan(spec.eps_ind, n, n)
    eng.set_noise_rng(2)
    eng.reset()
    acts = np.random.default_rng(3).integers(0, 5, (K, n)).astype(np.int32)
    _, r, d = eng.rollout(acts)
    eng.sync()
    return eng, r, d


@pytest.mark.gpu
@pytest.mark.parametrize("out_dtype", ["float32", "float64"])
def test_device_reward_normalisation_vs_restated_sb3(out_dtype):
    """Rewards and done flags of a real rollout (1-day episodes: two terminations in 300 steps), n = 1000 (ragged last wave):
    normalised rewards, running statistics and per-env returns against the restated SB3 algorithm; one call over 300 steps ==
    calls over 120 + 1 + 179 steps (state carried across calls; T = 1 is the per-step path)."""
    n, K = 1000, 300
    eng, r, d = _rollout_rewards(n, K, out_dtype=out_dtype)
    rn, dn = r.cpu().numpy().astype(np.float64), d.cpu().numpy()
    assert int(dn.sum()) == 2 * n
    ora = vo.RewardNormalizer(n)
    exp = ora.rollout(rn, dn)
    eng.vn_init()
    got = eng.vn_normalize(r, d)
    st, ret = eng.vn_get()
    tol = 1e-6 if out_dtype == "float32" else 1e-12         # float32 outputs: one rounding of the float64 quotient
    np.testing.assert_allclose(got.cpu().numpy(), exp, rtol=tol, atol=1e-30)
    np.testing.assert_allclose([st["mean"], st["var"], st["count"]], [ora.ret_rms.mean, ora.ret_rms.var, ora.ret_rms.count], rtol=1e-11)
    np.testing.assert_allclose(ret, ora.returns, rtol=1e-12, atol=1e-300)
    assert np.abs(got.cpu().numpy()).max() <= 10.0
    # the same in three calls
    eng.vn_init()
    parts = [eng.vn_normalize(r[:120], d[:120]), eng.vn_normalize(r[120], d[120]).unsqueeze(0), eng.vn_normalize(r[121:], d[121:])]
    import torch
    st2, ret2 = eng.vn_get()
    # the running moments are merged as a prefix scan whose grouping depends on the call length: equal to rounding, not bitwise
    np.testing.assert_allclose(torch.cat(parts).cpu().numpy(), got.cpu().numpy(), rtol=tol, atol=1e-30)
    np.testing.assert_allclose([st2[k] for k in ("mean", "var", "count")], [st[k] for k in ("mean", "var", "count")], rtol=1e-12)
    assert np.array_equal(ret2, ret)
    # evaluation mode: frozen statistics, returns untouched
    frozen = eng.vn_normalize(r[:50], d[:50], training=False)
    st3, ret3 = eng.vn_get()
    assert st3 == st2 and np.array_equal(ret3, ret)
    np.testing.assert_allclose(frozen.cpu().numpy(), np.clip(rn[:50] / np.sqrt(st2["var"] + 1e-8), -10, 10), rtol=tol, atol=1e-30)
    # checkpoint round trip
    eng.vn_init()
    eng.vn_set(stats=st, returns=ret)
    assert eng.vn_get()[0] == st
    eng.close()


@pytest.mark.gpu
def test_sharded_reward_normalisation_equals_one_batch():
    """Two handles over the halves of a batch, their per-step moments merged (the multi-GPU path without the process group):
    same statistics and the same normalised rewards as one handle over all envs."""
    import ctypes as C
    import torch
    n, K = 768, 200
    full, r, d = _rollout_rewards(n, K)
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=1, train_steps=400000)
    halves = [HipEngine(spec.consts, spec.tables, spec.markets, n // 2, device=0, out_dtype="float32", obs_layout="feature") for _ in range(2)]
    full.vn_init()
    exp = full.vn_normalize(r, d)
    st_full, _ = full.vn_get()
    rs = [r[:, :n // 2].contiguous(), r[:, n // 2:].contiguous()]
    ds = [d[:, :n // 2].contiguous(), d[:, n // 2:].contiguous()]
    moms = []
    for h, rr, dd in zip(halves, rs, ds):
        h.vn_init()
        m = torch.empty((K, 3), dtype=torch.float64, device=rr.device)
        h._chk(h._L.ptg_vn_batch_moments(h._h, C.c_void_p(rr.data_ptr()), C.c_void_p(dd.data_ptr()), K, C.c_void_p(m.data_ptr()), h._stream()))
        moms.append(m)
    merged = ptg_dist.merge_moments(torch.stack(moms)).contiguous()
    outs = []
    for h, rr in zip(halves, rs):
        o = torch.empty_like(rr)
        h._chk(h._L.ptg_vn_apply(h._h, C.c_void_p(rr.data_ptr()), K, C.c_void_p(merged.data_ptr()), C.c_void_p(o.data_ptr()), 1, h._stream()))
        outs.append(o)
    got = torch.cat(outs, dim=1)
    np.testing.assert_allclose(got.cpu().numpy(), exp.cpu().numpy(), rtol=1e-6, atol=1e-30)
    for h in halves:
        st, _ = h.vn_get()
        np.testing.assert_allclose([st["mean"], st["var"], st["count"]], [st_full["mean"], st_full["var"], st_full["count"]], rtol=1e-11)
        h.close()
    full.close()


@pytest.mark.gpu
def test_vecenv_norm_reward_is_the_reference_wrapper():
    """PtGVecEnv(..., norm_reward=True).step() == VecNormalize(PtGVecEnv(...), norm_obs=False).step() as restated from SB3:
    normalised rewards, get_original_reward(), ret_rms, frozen statistics with training = False."""
    from rl_ptg_amd.prep import synthetic_spec
    from rl_ptg_amd.vec_env import PtGVecEnv
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=1, train_steps=200000)
    n, K = 48, 160                                        # 139-step episodes: one termination inside
    plain = PtGVecEnv(spec, n_envs=n, seed=5, noise="device")
    wrapped = PtGVecEnv(spec, n_envs=n, seed=5, noise="device", norm_reward=True)
    ora = vo.RewardNormalizer(n)
    plain.reset(); wrapped.reset()
    rng = np.random.default_rng(1)
    for t in range(K):
        a = rng.integers(0, 5, n)
        _, r0, d0, _ = plain.step(a)
        _, r1, d1, _ = wrapped.step(a)
        exp = ora.step(r0.astype(np.float64), d0)
        assert np.array_equal(d0, d1) and np.array_equal(wrapped.get_original_reward(), r0)
        np.testing.assert_allclose(r1, exp, rtol=2e-6, atol=1e-30)
    st = wrapped.ret_rms
    # the oracle above is fed the float32 rewards the VecEnv hands out, the device normaliser the float64 ones: 1e-7 apart
    np.testing.assert_allclose([st["mean"], st["var"], st["count"]], [ora.ret_rms.mean, ora.ret_rms.var, ora.ret_rms.count], rtol=1e-6)
    wrapped.training = False
    a = rng.integers(0, 5, n)
    _, r0, _, _ = plain.step(a)
    _, r1, _, _ = wrapped.step(a)
    np.testing.assert_allclose(r1, np.clip(r0.astype(np.float64) / np.sqrt(st["var"] + 1e-8), -10, 10), rtol=2e-6, atol=1e-30)
    assert wrapped.ret_rms == st
    plain.close(); wrapped.close()


@pytest.mark.gpu
def test_vecenv_norm_reward_reset_and_frozen_statistics():
    """(1) VecNormalize.reset() starts the discounted returns at zero again -- PtGVecEnv.reset() must too; (2) with
    training = False the statistics are frozen and the returns are NOT advanced, but returns[dones] = 0 still happens.  Checked
    against the restated SB3 algorithm (oracle/vecnormalize_oracle.py; unpinned against SB3 itself)."""
    from rl_ptg_amd.prep import synthetic_spec
    from rl_ptg_amd.vec_env import PtGVecEnv
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=1, train_steps=200000)
    n = 16
    env = PtGVecEnv(spec, n_envs=n, seed=9, noise="device", norm_reward=True)
    ora = vo.RewardNormalizer(n)
    env.reset()
    rng = np.random.default_rng(2)
    for t in range(20):
        a = rng.integers(0, 5, n)
        _, r, d, _ = env.step(a)
        ora.step(env.get_original_reward().astype(np.float64), d)
    _, ret = env.engine.vn_get()
    np.testing.assert_allclose(ret, ora.returns, rtol=1e-6, atol=1e-6)
    assert np.abs(ret).max() > 0
    env.reset()                                           # (1)
    _, ret = env.engine.vn_get()
    assert np.array_equal(ret, np.zeros(n))
    ora.returns[:] = 0.0
    # (2) frozen statistics across an episode end (139-step episodes): run up to the termination with training on, then freeze
    for t in range(130):
        a = rng.integers(0, 5, n)
        _, r, d, _ = env.step(a)
        ora.step(env.get_original_reward().astype(np.float64), d)
    env.training = False
    ora.training = False
    stats = env.ret_rms
    seen_done = False
    for t in range(15):
        a = rng.integers(0, 5, n)
        _, r, d, _ = env.step(a)
        exp = ora.step(env.get_original_reward().astype(np.float64), d)
        np.testing.assert_allclose(r, exp, rtol=2e-6, atol=1e-30)
        seen_done |= bool(d.any())
        _, ret = env.engine.vn_get()
        np.testing.assert_allclose(ret, ora.returns, rtol=1e-6, atol=1e-6)      # unchanged except zeroed where an episode ended
    assert seen_done and env.ret_rms == stats
    env.close()
