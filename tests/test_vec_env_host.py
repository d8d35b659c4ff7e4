"""Host-side contract of PtGVecEnv.step (rl_ptg_amd/vec_env.py over ptg_step_host): what the caller may keep across steps,
the zero-copy and the staged route, lazy eval infos, checkpointing.  Reference semantics: DummyVecEnv returns copies of its
buffers every step (SB3 dummy_vec_env.py `_obs_from_buf`), SB3 algorithms keep `_last_obs` across exactly one step()."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _spec(**kw):
    from rl_ptg_amd.prep import synthetic_spec
    return synthetic_spec(scenario=2, operation="OP2", eps_len_d=8, **kw)[0]


@pytest.mark.parametrize("n,dtype", [(1, "float64"), (6, "float64"), (6, "float32")])
def test_small_batch_observations_are_copies(n, dtype):
    """with n_envs = 1 / float64 the returned observations aliased the reused staging buffer.  Every array handed out
    by a small batch is a fresh copy: keeping obs_t and stepping again leaves obs_t untouched (off-policy _last_obs)."""
    from rl_ptg_amd.vec_env import PtGVecEnv
    env = PtGVecEnv(_spec(), n, seed=3, out_dtype=dtype, noise="device")
    env.reset()
    rng = np.random.default_rng(0)
    kept = []
    for t in range(12):
        obs, rew, done, infos = env.step(rng.integers(0, 5, n))
        kept.append(({k: v.copy() for k, v in obs.items()}, obs, rew.copy(), rew, done.copy(), done))
    for snap, obs, r0, r, d0, d in kept:
        for k in obs:
            assert np.array_equal(snap[k], obs[k]), k
            assert obs[k].dtype == (np.int64 if k == "METH_STATUS" else np.dtype(dtype))
        assert np.array_equal(r0, r) and np.array_equal(d0, d)
    env.close()


def test_single_env_adapter_keeps_previous_observation():
    from rl_ptg_amd.vec_env import PTGEnv
    tr, kw = H.kwargs_from_fixture("real_bs2_op2_mod_disc_evalval")
    env = PTGEnv(kw, "eval")
    env.reset(seed=1)
    o1, *_ = env.step(2)
    snap = {k: np.array(v, copy=True) for k, v in o1.items()}
    for a in (2, 2, 3, 0, 1):
        env.step(a)
    for k in snap:
        assert np.array_equal(snap[k], np.asarray(o1[k])), k
    env.close()


def test_large_batch_ring_and_routes_agree():
    """A batch above the zero-copy / copy-out limits: observations are views of a ring of OBS_RING pinned blocks -- valid for the next
    OBS_RING - 1 steps -- and equal, step by step, to what the device-tensor route returns for the same actions."""
    import torch
    from rl_ptg_amd.vec_env import PtGVecEnv
    n = 4096
    spec = _spec()
    a_env = PtGVecEnv(spec, n, seed=5, out_dtype="float32", obs_layout="row", noise="device")
    b_env = PtGVecEnv(spec, n, seed=5, out_dtype="float32", obs_layout="row", noise="device")
    assert not a_env._copy_out and len(a_env._blk) == a_env.OBS_RING
    a_env.reset(); b_env.reset()
    rng = np.random.default_rng(1)
    hist = []
    for t in range(10):
        a = rng.integers(0, 5, n)
        obs, rew, done, infos = a_env.step(a)
        o, r, d = b_env.step_tensors(torch.from_numpy(a.astype(np.int32)).cuda())
        b_env.engine.sync()
        o = o.cpu().numpy()
        cols = a_env._cols
        for k, sl in cols.items():
            ref = np.rint(o[:, sl.start]).astype(np.int64) if k == "METH_STATUS" else o[:, sl]
            assert np.array_equal(obs[k], ref), (t, k)
        assert np.array_equal(rew, r.cpu().numpy()) and not done.any() and infos == [{} for _ in range(n)]
        hist.append((obs, {k: v.copy() for k, v in obs.items()}))
        for back in range(1, min(len(hist), a_env.OBS_RING)):       # still intact OBS_RING - 1 steps later
            old, snap = hist[-1 - back]
            assert all(np.array_equal(old[k], snap[k]) for k in snap)
    a_env.close(); b_env.close()


def test_feature_major_host_views():
    from rl_ptg_amd.vec_env import PtGVecEnv
    n = 2048
    spec = _spec()
    a_env = PtGVecEnv(spec, n, seed=5, out_dtype="float64", obs_layout="feature", noise="device")
    b_env = PtGVecEnv(spec, n, seed=5, out_dtype="float64", obs_layout="row", noise="device")
    a_env.reset(); b_env.reset()
    rng = np.random.default_rng(2)
    for t in range(6):
        a = rng.integers(0, 5, n)
        oa, ra, da, _ = a_env.step(a)
        ob, rb, db, _ = b_env.step(a)
        assert all(np.array_equal(oa[k], ob[k]) and oa[k].shape == ob[k].shape for k in oa) and np.array_equal(ra, rb)
    a_env.close(); b_env.close()


def test_lazy_eval_infos_match_eager_dicts():
    """Eval batches larger than EAGER_INFO_MAX hand out _InfoRow views: same keys, same values, dict protocol SB3 uses
    (get / in / keys / item assignment)."""
    from rl_ptg_amd.vec_env import INFO_KEYS, PtGVecEnv
    spec = _spec()
    n = 96
    lazy = PtGVecEnv(spec, n, train_or_eval="eval", seed=7, noise="device")
    eager = PtGVecEnv(spec, n, train_or_eval="eval", seed=7, noise="device")
    eager.EAGER_INFO_MAX = 1 << 30
    eager._setup_host()
    assert lazy._lazy_info and not eager._lazy_info
    lazy.reset(); eager.reset()
    rng = np.random.default_rng(3)
    for t in range(8):
        a = rng.integers(0, 5, n)
        _, _, _, il = lazy.step(a)
        _, _, _, ie = eager.step(a)
        for e in (0, 17, n - 1):
            assert list(il[e].keys()) == INFO_KEYS == list(ie[e].keys())
            assert all(il[e][k] == ie[e][k] for k in INFO_KEYS)
            assert il[e].get("episode") is None and "terminal_observation" not in il[e] and "reward [ct]" in il[e]
            assert il[e].get("step") == t
    lazy.close(); eager.close()


def test_vec_env_checkpoint_resume_numpy_noise():
    from rl_ptg_amd.vec_env import PtGVecEnv
    spec = _spec()
    n = 5
    rng = np.random.default_rng(4)
    acts = rng.integers(0, 5, (80, n))
    env = PtGVecEnv(spec, n, seed=11, noise="numpy", noise_tape_len=16)
    env.reset()
    for t in range(30):
        env.step(acts[t])
    sd = env.state_dict()
    ref = [env.step(acts[t]) for t in range(30, 80)]
    env.close()
    env2 = PtGVecEnv(spec, n, seed=999, noise="numpy", noise_tape_len=16)
    env2.load_state_dict(sd)
    for t in range(30, 80):
        o, r, d, _ = env2.step(acts[t])
        ro, rr, rd, _ = ref[t - 30]
        assert all(np.array_equal(o[k], ro[k]) for k in o) and np.array_equal(r, rr) and np.array_equal(d, rd), t
    env2.close()


def test_create_destroy_does_not_leak_device_memory():
    """50 handles created, used (reset, a hot rollout with its refresher stream, a host step, profiling events) and destroyed: the
    device's free memory returns to where it was (tables, staging buffers, streams and events are all released)."""
    import torch
    from rl_ptg_amd.engine import HipEngine
    spec = _spec()
    torch.cuda.synchronize()
    acts = np.random.default_rng(0).integers(0, 5, (8, 512)).astype(np.int32)

    def cycle():
        eng = HipEngine(spec.consts, spec.tables, spec.markets, 512, device=0, out_dtype="float32", obs_layout="row")
        eng.set_episode_plan(spec.eps_ind, 512, 512)
        eng.set_noise_rng(1)
        eng.reset()
        eng.profile(True)
        o, r, d = eng.rollout(acts)
        eng.step(acts[0])
        eng.sync()
        assert len(eng.profile_read()) == 2
        del o, r, d
        eng.close()
    for _ in range(3):
        cycle()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(50):
        cycle()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free0 - free1) < (32 << 20), (free0, free1)
