"""Direct parity of the HEADLINE kernels -- the fused rollout (k_rollout_pc) and the one-launch step (k_step_hot), float32 and
float64 -- against the reference's golden trajectories and against the CPU oracle (Next #5): no transitive hop
through another HIP kernel.  Integers bit-exact; floats: float64 rtol 1e-11, float32 2e-7 (north_star bar: 1e-5)."""
import os
import sys

import numpy as np
import pytest

import helpers as H

sys.path.insert(0, os.path.join(H.ROOT, "oracle"))
import sb3_flat_oracle as flat_oracle  # noqa: E402

pytestmark = pytest.mark.gpu
RTOL64, ATOL64 = 1e-11, 1e-13
RTOL32, ATOL32 = 2e-7, 1e-9
INT_FIELDS = ["meth_state", "i", "j", "hot_cold", "standby_tid", "startup_tid", "partial_tid", "full_tid", "k", "current_action"]


def _ints(eng):
    cols = [eng.get_state(f) for f in INT_FIELDS]
    actd = eng.get_state("act_ep_d")
    return np.stack(cols + [actd * 24, actd], axis=1)


@pytest.mark.parametrize("out_dtype,layout", [("float32", "row"), ("float32", "feature"), ("float32", "sb3_flat"),
                                              ("float64", "row"), ("float64", "feature")])
@pytest.mark.parametrize("case", H.TRAJ_CASES)
def test_rollout_vs_reference_golden(case, out_dtype, layout):
    """eng.rollout(all actions of the fixture) == the unmodified reference, step by step: observations (post-reset rows where an
    episode ended), rewards, done flags, final integer state, finished-episode returns / lengths."""
    tr, eng = H.make_engine(case, out_dtype, obs_layout=layout)
    rtol, atol = (RTOL64, ATOL64) if out_dtype == "float64" else (RTOL32, ATOL32)
    K, n = tr["actions"].shape
    mode = "mod" if tr["meta"]["consts"]["raw_modified"] else "raw"
    eng.reset()
    obs, rew, done = eng.rollout(tr["actions"])
    eng.sync()
    obs, rew, done = eng.rows(obs).cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
    assert np.array_equal(done, tr["done"])
    np.testing.assert_allclose(rew, tr["f64s"][:, :, 0], rtol=rtol, atol=atol)
    want = tr["obs"].copy()                                   # [K, n, F] canonical; finished envs show the post-reset observation
    for q, (t, e) in enumerate(tr["post_reset_at"].tolist()):
        want[t, e] = tr["post_reset_obs"][q]
    if layout == "sb3_flat":
        want = flat_oracle.flatten_rows(want.reshape(K * n, -1), mode).reshape(K, n, -1)
        assert obs.shape == want.shape
    np.testing.assert_allclose(obs, want, rtol=rtol, atol=atol)
    if layout == "sb3_flat":                                  # the one-hot block is exact
        c0 = sorted(k for k, _ in flat_oracle.reference_keys(mode))
        off = sum(dict(flat_oracle.reference_keys(mode))[k] for k in c0[:c0.index("METH_STATUS")])
        assert np.array_equal(obs[:, :, off:off + 6], want[:, :, off:off + 6])
    # final state: the reference's last step (envs that finished on the last step are checked through post_reset_int)
    last_done = tr["done"][K - 1].astype(bool)
    ints = _ints(eng)
    assert np.array_equal(ints[~last_done], tr["ints"][K - 1][~last_done])
    assert np.array_equal(eng.get_state("T_cat")[~last_done], tr["f64s"][K - 1, :, 2][~last_done])
    np.testing.assert_allclose(eng.get_state("cum_rew")[~last_done], tr["f64s"][K - 1, :, 1][~last_done], rtol=1e-11, atol=1e-9)
    assert np.array_equal(eng.get_state("noise_count"), tr["noise_len"])
    # Monitor statistics: return and length of every finished episode
    r, l, ids = eng.finished_episodes()
    ret = np.zeros(n)
    exp = []
    for t in range(K):
        ret += tr["f64s"][t, :, 0]
        for e in np.nonzero(tr["done"][t])[0]:
            exp.append((int(e), int(tr["ints"][t, e, 8]), ret[e]))
            ret[e] = 0.0
    got = sorted(zip(ids.tolist(), l.tolist(), r.tolist()))
    assert len(got) == len(exp)
    for (e1, l1, r1), (e2, l2, r2) in zip(got, sorted(exp)):
        assert e1 == e2 and l1 == l2 and abs(r1 - r2) <= 1e-9 * max(1.0, abs(r2))
    eng.close()


def _synthetic(n, scenario, operation, out_dtype, layout):
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=scenario, operation=operation, eps_len_d=32)
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=out_dtype, obs_layout=layout)
    eng.set_episode_plan(spec.eps_ind, n, n)
    return spec, eng


def _oracle(spec, n, tape):
    m = spec.markets[0]
    consts = dict(spec.consts, scenario=m["scenario"], rew_l_b=m["rew_l_b"], rew_u_b=m["rew_u_b"], r_0=m["r_0"])
    ora = H.po.OracleVecEnv(consts, spec.tables, dict(m, eps_ind=None), n)
    ora.set_noise_tape(tape)
    return ora


@pytest.mark.parametrize("out_dtype,layout", [("float32", "row"), ("float32", "feature"), ("float64", "row")])
def test_config2_hot_kernels_vs_oracle_n4096(out_dtype, layout):
    """BASELINE.json configs[1]: N = 4096, BS2/OP2.  BOTH hot kernels (k_step_hot: the first 60 steps one launch each, k_rollout_pc:
    the next 100 fused) against the CPU oracle on the same action and noise tapes."""
    n, K1, K2, L = 4096, 60, 100, 192
    spec, eng = _synthetic(n, 2, "OP2", out_dtype, layout)
    rtol, atol = (RTOL64, ATOL64) if out_dtype == "float64" else (RTOL32, ATOL32)
    eng.fill_noise_tape(seed=5, per_env_len=L)
    tape = eng.get_noise_tape(L)
    ora = _oracle(spec, n, tape)
    o_ref, _ = ora.reset()
    np.testing.assert_allclose(eng.rows(eng.reset()).cpu().numpy(), o_ref, rtol=rtol, atol=atol)
    rng = np.random.default_rng(8)
    acts = rng.integers(0, 5, (K1 + K2, n)).astype(np.int32)
    acts[20:] = np.where(rng.random((K1 + K2 - 20, n)) < 0.8, acts[19:-1], acts[20:])       # some held actions: partial / full load get reached
    refs = [ora.step(acts[t])[:3] for t in range(K1 + K2)]
    assert eng.rollout_launches(K2) == 1
    for t in range(K1):
        o, r, d = eng.step(acts[t])
        eng.sync()
        np.testing.assert_allclose(eng.rows(o).cpu().numpy(), refs[t][0], rtol=rtol, atol=atol, err_msg=f"step {t}")
        np.testing.assert_allclose(r.cpu().numpy(), refs[t][1], rtol=rtol, atol=max(atol, 1e-6 if out_dtype == "float32" else 0))
        assert np.array_equal(d.cpu().numpy(), refs[t][2])
    o, r, d = eng.rollout(acts[K1:])
    eng.sync()
    o, r = eng.rows(o).cpu().numpy(), r.cpu().numpy()
    for t in range(K2):
        np.testing.assert_allclose(o[t], refs[K1 + t][0], rtol=rtol, atol=atol, err_msg=f"fused step {t}")
        np.testing.assert_allclose(r[t], refs[K1 + t][1], rtol=rtol, atol=max(atol, 1e-6 if out_dtype == "float32" else 0))
    ints, f64s = ora.state()
    for col, name in enumerate(INT_FIELDS):
        assert np.array_equal(eng.get_state(name), ints[:, col]), name
    assert np.array_equal(eng.get_state("T_cat"), f64s[:, 2])
    assert set(np.unique(ints[:, 0])) == {0, 1, 2, 3, 4}                   # every METH_STATUS occurred
    eng.close(); ora.close()


@pytest.mark.parametrize("n,scenario,operation,noise,out_dtype", [
    (65536, 1, "OP1", "tape", "float32"), (65536, 1, "OP1", "tape", "float64"),
    (65536, 1, "OP1", "rng", "float32"),               # bench.py's exact instantiation (in-kernel counter RNG) at full size
    (262144, 3, "OP2", "tape", "float32"),             # BASELINE.json configs[3]: the CHP / EEG reward path (:291-293), 4 launches per step segment
    (262144, 3, "OP2", "rng", "float64")])
def test_full_size_slice_vs_oracle(n, scenario, operation, noise, out_dtype):
    """BASELINE.json configs[2] (N = 65 536, BS1/OP1) and configs[3] (N = 262 144, BS3/OP2) at FULL size.  The oracle follows a
    256-env slice that straddles two workgroups (envs are independent: the slice's action and noise tapes fed to a 256-env oracle) for
    240 fused steps from reset -- across the refresher's rolling phase, several launch segments and (262 144) all four env slices of
    a step segment.  noise = "rng": the kernels draw the state-change noise inline (what bench.py times); the oracle's tape then
    comes from ptg_fill_noise_tape on a 256-env twin at the slice's global env offset -- the same counter streams
    (include/ptg_env.h, ptg_set_noise_rng)."""
    import torch
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.synthetic import sticky_actions_device
    K, L, CH = 240, 256, 60
    lo, m = (n // 8) * 5 + 128, 256                          # 40 960 + 128 at 65 536 envs; inside the third env slice at 262 144
    spec, eng = _synthetic(n, scenario, operation, out_dtype, "row")
    rtol, atol = (RTOL64, ATOL64) if out_dtype == "float64" else (RTOL32, ATOL32)
    if noise == "tape":
        eng.fill_noise_tape(seed=31, per_env_len=L)
        tape = eng.get_noise_tape(L)[lo:lo + m]
    else:
        eng.set_noise_rng(seed=31)
        twin = HipEngine(spec.consts, spec.tables, spec.markets, m, device=0, out_dtype=out_dtype, obs_layout="row")
        twin.set_global_env_offset(lo)
        twin.fill_noise_tape(seed=31, per_env_len=L)
        tape = twin.get_noise_tape(L)
        twin.close()
    ora = _oracle(spec, m, tape)
    ora.reset()
    eng.reset()
    acts = sticky_actions_device(K, n, seed=3, device=torch.device("cuda", 0))
    a_host = acts[:, lo:lo + m].cpu().numpy()
    o_parts, r_parts, n_done = [], [], 0
    for t0 in range(0, K, CH):                               # 60 steps per call: 2.2 GB (float32) of observations at 262 144 envs
        o, r, d = eng.rollout(acts[t0:t0 + CH])
        eng.sync()
        o_parts.append(o[:, lo:lo + m].cpu().numpy()); r_parts.append(r[:, lo:lo + m].cpu().numpy())
        n_done += int(d.sum())
        del o, r, d
    o, r = np.concatenate(o_parts), np.concatenate(r_parts)
    assert n_done == 0
    for t in range(K):
        o_ref, r_ref, d_ref, _, _ = ora.step(a_host[t])
        np.testing.assert_allclose(o[t], o_ref, rtol=rtol, atol=atol, err_msg=f"step {t}")
        np.testing.assert_allclose(r[t], r_ref, rtol=rtol, atol=max(atol, 1e-6 if out_dtype == "float32" else 0))
    ints, f64s = ora.state()
    for col, name in enumerate(INT_FIELDS):
        assert np.array_equal(eng.get_state(name)[lo:lo + m], ints[:, col]), name
    np.testing.assert_allclose(eng.get_state("cum_rew")[lo:lo + m], f64s[:, 1], rtol=1e-11, atol=1e-9)
    assert len(np.unique(ints[:, 0])) >= 4                   # the slice went through (nearly) every METH_STATUS
    assert int(eng.get_state("noise_count")[lo:lo + m].max()) <= L
    eng.close(); ora.close()


@pytest.mark.parametrize("scenario,operation,out_dtype,layout", [(2, "OP2", "float32", "row"), (1, "OP1", "float64", "row"), (3, "OP2", "float32", "feature")])
def test_real_training_set_every_env_its_own_episode_vs_oracle(scenario, operation, out_dtype, layout):
    """The reference's REAL training set (tests/golden/market_real.npz: 36 552 hours) with its default episode length (37 days, 41 episodes)
    through `rl_ptg_amd.prep.Preprocessing`: 1 024 envs in DummyVecEnv order, so every env sits in its own draw of the shuffled eps_ind
    (`src/rl_utils.py:315-335`) and the market windows are per-lane gathers -- 120 single steps then 200 fused steps against the oracle."""
    from rl_ptg_amd.config import EnvConfig
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import EnvSpec, Preprocessing
    z = np.load(os.path.join(H.GOLD, "market_real.npz"))
    prices = {k: z[k] for k in z.files}
    cfg = EnvConfig(scenario=scenario, operation=operation)
    pre = Preprocessing(prices, H.load_tables(operation), cfg, seed_train=3654, train_steps=1500000, action_type="discrete")
    spec = EnvSpec.from_dict_input(pre.dict_env_kwargs("train"), "train")
    assert pre.n_eps == 41 and len(spec.eps_ind) == 2460 and spec.consts["eps_sim_steps"] == 5328      # SURVEY.md 8(c) known answers
    n, K1, K2 = 1024, 120, 200          # (constructors + first resets draw 2 n of the 2 460 eps_ind entries: the reference raises IndexError beyond)
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=out_dtype, obs_layout=layout)
    eng.set_episode_plan(spec.eps_ind, n, n)
    rng = np.random.default_rng(31 + scenario)
    tape = rng.normal(0.0, spec.consts["noise"], size=(n, 64))
    eng.set_noise_tape(tape)
    m = spec.markets[0]
    consts = dict(spec.consts, scenario=m["scenario"], rew_l_b=m["rew_l_b"], rew_u_b=m["rew_u_b"], r_0=m["r_0"])
    ora = H.po.OracleVecEnv(consts, spec.tables, dict(m, eps_ind=spec.eps_ind), n, ep_index0=0)
    ora.set_noise_tape(tape)
    rtol, atol = (RTOL64, ATOL64) if out_dtype == "float64" else (RTOL32, ATOL32)
    ratol = max(atol, 1e-6 if out_dtype == "float32" else 0)
    o_ref, _ = ora.reset()
    np.testing.assert_allclose(eng.rows(eng.reset()).cpu().numpy(), o_ref, rtol=rtol, atol=atol)
    assert len(np.unique(eng.get_state("act_ep_d"))) > 30          # the envs really are spread over the episodes
    hold = rng.integers(1, 10, n)
    cur = np.full(n, 2)

    def acts_at(t):
        nonlocal cur
        flip = t % hold == 0
        cur = np.where(flip & (t > 8), rng.integers(0, 5, n), cur)
        return cur.astype(np.int32)

    for t in range(K1):
        a = acts_at(t)
        o, r, d = eng.step(a)
        eng.sync()
        o_ref, r_ref, d_ref, _, _ = ora.step(a)
        np.testing.assert_allclose(eng.rows(o).cpu().numpy(), o_ref, rtol=rtol, atol=atol, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=rtol, atol=ratol, err_msg=f"reward step {t}")
    acts = np.stack([acts_at(K1 + t) for t in range(K2)])
    obs, rew, done = eng.rollout(acts)
    eng.sync()
    obs, rew = obs.cpu(), rew.cpu().numpy()
    for t in range(K2):
        o_ref, r_ref, d_ref, _, _ = ora.step(acts[t])
        np.testing.assert_allclose(eng.rows(obs[t]).numpy(), o_ref, rtol=rtol, atol=atol, err_msg=f"obs fused step {t}")
        np.testing.assert_allclose(rew[t], r_ref, rtol=rtol, atol=ratol, err_msg=f"reward fused step {t}")
    ints, f64s = ora.state()
    got = _ints(eng)
    assert np.array_equal(got[:, :10], ints[:, :10]) and np.array_equal(got[:, 11], ints[:, 11])
    assert np.array_equal(eng.get_state("T_cat"), f64s[:, 2])
    eng.close(); ora.close()
