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


@pytest.mark.parametrize("n,layout,dtype", [(6, "row", "float64"), (4096, "row", "float32"), (4096, "feature", "float64"), (333, "row", "float32")])
def test_host_step_phases_and_status_section(n, layout, dtype):
    """ptg_step_host_begin / _tail / _end (include/ptg_env.h) through raw ctypes, as a C caller would drive them: after `tail` the rewards,
    done flags and the contiguous METH_STATUS bytes are in the block, after `end` the observations; the three phases produce what the
    one-call ptg_step_host produces on a twin handle (zero-copy route at n = 6, staged route above), and the status bytes equal the
    METH_STATUS column of the returned rows.  Calling the phases out of order is PTG_E_INVALID, not UB."""
    import ctypes as C
    import torch
    from rl_ptg_amd.engine import HipEngine
    spec = _spec()
    outs = []
    for phased in (True, False):
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=dtype, obs_layout=layout)
        eng.set_episode_plan(spec.eps_ind, n, n)
        eng.set_noise_rng(4)
        eng.reset()
        L, h = eng._L, eng._h
        o_rew, o_done, o_stat, total = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        assert L.ptg_host_layout_ex(h, C.byref(o_rew), C.byref(o_done), C.byref(o_stat), C.byref(total)) == 0
        t3 = C.c_size_t()
        assert L.ptg_host_layout(h, None, None, C.byref(t3)) == 0 and t3.value == total.value      # the old call reports the same block size
        assert o_rew.value % 16 == 0 and o_done.value % 16 == 0 and o_stat.value % 16 == 0 and total.value >= o_stat.value + n
        blk = torch.empty(total.value, dtype=torch.uint8, pin_memory=True)
        act = torch.empty(n, dtype=torch.int32, pin_memory=True)
        fin = torch.empty(n * eng.obs_dim * (8 if dtype == "float64" else 4), dtype=torch.uint8, pin_memory=True)
        nd = C.c_int(-1)
        if phased:
            assert L.ptg_step_host_tail(h, C.byref(nd)) == -1 and L.ptg_step_host_end(h) == -1          # nothing in flight: PTG_E_INVALID
        rng = np.random.default_rng(9)
        rows = []
        for t in range(25):
            act.numpy()[:] = rng.integers(0, 5, n)
            args = (h, C.c_void_p(act.data_ptr()), 0, C.c_void_p(blk.data_ptr()), C.c_void_p(fin.data_ptr()), None)
            if phased:
                assert L.ptg_step_host_begin(*args, eng._stream()) == 0
                assert L.ptg_step_host_begin(*args, eng._stream()) == -1                                # one host step at a time
                assert L.ptg_step_host_tail(h, C.byref(nd)) == 0 and nd.value == 0
                assert L.ptg_step_host_end(h) == 0
            else:
                assert L.ptg_step_host(*args, C.byref(nd), eng._stream()) == 0 and nd.value == 0
            b = blk.numpy()
            npdt = np.float64 if dtype == "float64" else np.float32
            flat = b[:n * eng.obs_dim * np.dtype(npdt).itemsize].view(npdt)
            mat = flat.reshape(eng.obs_dim, n).T if layout == "feature" else flat.reshape(n, eng.obs_dim)
            status = b[o_stat.value:o_stat.value + n]
            assert np.array_equal(status.astype(np.int64), np.rint(mat[:, eng.obs_dim - 9]).astype(np.int64)), t
            rows.append(b[:o_stat.value + n].copy())
        assert len({int(x) for r in rows for x in np.unique(r[o_stat.value:o_stat.value + n])}) >= 3    # several METH_STATUS values occurred
        outs.append(rows)
        eng.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_copy_obs_switch_and_layout_guard():
    """copy_obs=True on a large batch: fresh arrays and a NEW infos list every step (DummyVecEnv semantics);
    copy_obs=False: ring views and one persistent list; layouts whose columns are not the Dict observation's raise at once."""
    from rl_ptg_amd.vec_env import PtGVecEnv
    spec = _spec()
    n = 2048
    with pytest.raises(ValueError):
        PtGVecEnv(spec, n, obs_layout="split", noise="device")
    with pytest.raises(ValueError):
        PtGVecEnv(spec, n, obs_layout="sb3_flat", noise="device")
    rng = np.random.default_rng(2)
    acts = [rng.integers(0, 5, n) for _ in range(8)]
    res = {}
    for copy in (True, False):
        env = PtGVecEnv(spec, n, seed=1, out_dtype="float32", noise="device", copy_obs=copy)
        env.reset()
        kept = [env.step(a) for a in acts]
        res[copy] = [({k: v.copy() for k, v in o.items()}, r.copy()) for o, r, d, i in kept]
        if copy:
            assert len({id(i) for _, _, _, i in kept}) == len(kept)                                      # a new list object per step
            assert all(np.array_equal(kept[0][0][k], res[True][0][0][k]) for k in kept[0][0])            # step 0's arrays untouched 7 steps later
        else:
            assert len({id(i) for _, _, _, i in kept}) == 1
        env.close()
    # the last OBS_RING - 1 steps of the view route are still intact and equal the copies
    for t in range(len(acts) - 3, len(acts)):
        assert np.array_equal(res[True][t][1], res[False][t][1])
        for k in res[True][t][0]:
            assert np.array_equal(res[True][t][0][k], res[False][t][0][k]), (t, k)
