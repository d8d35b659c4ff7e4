"""hipGraph replay of the hot path (include/ptg_env.h, "hipGraph capture"): `ptg_step` / `ptg_rollout` captured once and replayed step after
step give what eager calls give, bit for bit -- the kernels take the step count from the device state -- and `ptg_note_replays` keeps the
host's count (which routes an episode's terminating step, env/ptg_gym_env.py:508-511, to the generic kernel) in step.  A captured step
carries its own terminating-step kernel and replays across episode ends; a captured rollout that runs over one is reported, not silently wrong."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(n, out_dtype="float32", layout="row", **kw):
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=4, **kw)
    engs = []
    for _ in range(2):
        e = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=out_dtype, obs_layout=layout)
        e.set_episode_plan(spec.eps_ind, n, n)
        e.set_noise_rng(seed=77)
        engs.append(e)
    return spec, engs[0], engs[1]


def _capture(fn):
    import torch
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    return g


@pytest.mark.parametrize("n,out_dtype,layout", [(777, "float32", "row"), (4096, "float64", "feature"), (513, "float32", "sb3_flat")])
def test_captured_step_replayed_many_times_equals_eager_steps(n, out_dtype, layout):
    import torch
    spec, A, B = _pair(n, out_dtype, layout)
    R = 60
    rng = np.random.default_rng(5)
    acts = torch.as_tensor(rng.integers(0, 5, (R + 6, n)).astype(np.int32), device="cuda")
    A.reset(); B.reset()
    act_buf = torch.zeros(n, dtype=torch.int32, device="cuda")
    obs, rew, done = B.alloc_obs(1)[0], torch.zeros(n, dtype=B.out_dtype, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
    g = _capture(lambda: B.step(act_buf, obs, rew, done, want_final=False))
    for t in range(R):
        act_buf.copy_(acts[t])
        g.replay()
        o_ref, r_ref, d_ref = A.step(acts[t], want_final=False)
        torch.cuda.synchronize()
        assert torch.equal(obs, o_ref) and torch.equal(rew, r_ref) and torch.equal(done, d_ref), f"replay {t}"
    B.note_replays(R - 1)                                    # (the capture call itself counted as one step)
    assert A.steps_to_episode_end() == B.steps_to_episode_end()
    for t in range(R, R + 6):                               # eager calls after the replays: both handles in step
        oa, ra, da = A.step(acts[t], want_final=False)
        ob, rb, db = B.step(acts[t], want_final=False)
        torch.cuda.synchronize()
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    A.sync(); B.sync()
    for f in ("meth_state", "i", "j", "k", "noise_count", "current_action", "cum_rew"):
        assert np.array_equal(A.get_state(f), B.get_state(f)), f
    A.close(); B.close()


def test_captured_rollout_replayed_equals_one_long_rollout():
    import torch
    n, T, R = 1000, 12, 5
    spec, A, B = _pair(n)
    rng = np.random.default_rng(6)
    acts = torch.as_tensor(rng.integers(0, 5, (T * R, n)).astype(np.int32), device="cuda")
    A.reset(); B.reset()
    o_ref, r_ref, d_ref = A.rollout(acts)
    act_buf = torch.zeros((T, n), dtype=torch.int32, device="cuda")
    obs, rew, done = B.alloc_obs(T), torch.zeros((T, n), dtype=B.out_dtype, device="cuda"), torch.zeros((T, n), dtype=torch.uint8, device="cuda")
    g = _capture(lambda: B.rollout(act_buf, obs, rew, done))
    for q in range(R):
        act_buf.copy_(acts[q * T:(q + 1) * T])
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(obs, o_ref[q * T:(q + 1) * T]) and torch.equal(rew, r_ref[q * T:(q + 1) * T]), f"replay {q}"
        assert torch.equal(done, d_ref[q * T:(q + 1) * T])
    B.note_replays((R - 1) * T)
    assert A.steps_to_episode_end() == B.steps_to_episode_end()
    A.sync(); B.sync()
    for f in ("meth_state", "i", "j", "k", "noise_count", "cum_rew"):
        assert np.array_equal(A.get_state(f), B.get_state(f)), f
    A.close(); B.close()


def test_captured_step_replays_across_episode_ends():
    """After ptg_set_replay_proof a captured ptg_step is enqueued as the hot kernel + the generic kernel behind it (each a no-op when the step is the other's): replayed
    over two and a half episodes it terminates, auto-resets over the episode plan and fills the finished-episode list exactly as eager
    calls do."""
    import torch
    n = 300
    spec, A, B = _pair(n, sim_step=3600)                    # 4-day episodes of hourly steps: 96 steps, the 91st call terminates
    B.set_replay_proof(True)
    to_end = spec.consts["eps_sim_steps"] - 5
    R = 2 * to_end + 40
    rng = np.random.default_rng(9)
    acts = torch.as_tensor(rng.integers(0, 5, (R + 4, n)).astype(np.int32), device="cuda")
    A.reset(); B.reset()
    act_buf = torch.zeros(n, dtype=torch.int32, device="cuda")
    obs, rew, done = B.alloc_obs(1)[0], torch.zeros(n, dtype=B.out_dtype, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
    fin = torch.zeros_like(obs)
    g = _capture(lambda: B.step(act_buf, obs, rew, done, final_obs=fin))
    n_done = 0
    for t in range(R):
        act_buf.copy_(acts[t])
        g.replay()
        o_ref, r_ref, d_ref = A.step(acts[t])
        torch.cuda.synchronize()
        assert torch.equal(done, d_ref), f"done flags, replay {t}"
        assert torch.equal(obs, o_ref) and torch.equal(rew, r_ref), f"replay {t}"
        if bool(d_ref.any()):
            n_done += 1
            assert bool(d_ref.all()) and torch.equal(fin, A.final_obs)      # terminal observations of the step that ended the episodes
    assert n_done == 2
    B.note_replays(R - 1)
    assert A.steps_to_episode_end() == B.steps_to_episode_end()
    ra, la, ea = A.finished_episodes()
    rb, lb, eb = B.finished_episodes()
    assert len(ra) == 2 * n and np.array_equal(np.sort(ea), np.sort(eb))
    oa, ob = np.lexsort((la, ea)), np.lexsort((lb, eb))
    assert np.array_equal(ra[oa], rb[ob]) and np.array_equal(la[oa], lb[ob])
    for t in range(R, R + 4):
        xa = A.step(acts[t], want_final=False)
        xb = B.step(acts[t], want_final=False)
        torch.cuda.synchronize()
        assert all(torch.equal(p, q) for p, q in zip(xa, xb))
    for f in ("meth_state", "i", "j", "k", "noise_count", "act_ep_d", "cum_rew"):
        assert np.array_equal(A.get_state(f), B.get_state(f)), f
    A.close(); B.close()


def test_captured_rollout_replayed_over_an_episode_end_is_reported():
    """A fused launch cannot cross an episode end (the host cuts eager rollouts there): a replay that does is flagged, not silently wrong."""
    import torch
    n, T = 256, 30
    spec, A, B = _pair(n, sim_step=3600)
    A.close()
    B.reset()
    act_buf = torch.full((T, n), 2, dtype=torch.int32, device="cuda")
    obs, rew, done = B.alloc_obs(T), torch.zeros((T, n), dtype=B.out_dtype, device="cuda"), torch.zeros((T, n), dtype=torch.uint8, device="cuda")
    g = _capture(lambda: B.rollout(act_buf, obs, rew, done))
    for _ in range(3):                                      # steps 0 .. 89: the hot kernels' share of the 96-step episode
        g.replay()
    B.sync()
    g.replay()                                              # steps 90 .. 119 would run over the terminating step (call 91)
    with pytest.raises(RuntimeError, match="terminating step"):
        B.sync()
    B.close()


def test_default_captured_step_over_the_terminating_step_is_reported():
    """Without ptg_set_replay_proof a captured step is the hot kernel alone: all the steps it may take are fine, the terminating one is flagged."""
    import torch
    n = 256
    spec, A, B = _pair(n, sim_step=3600)
    A.close()
    B.reset()
    to_end = B.steps_to_episode_end()
    assert to_end == spec.consts["eps_sim_steps"] - 5
    act_buf = torch.full((n,), 2, dtype=torch.int32, device="cuda")
    obs, rew, done = B.alloc_obs(1)[0], torch.zeros(n, dtype=B.out_dtype, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
    g = _capture(lambda: B.step(act_buf, obs, rew, done, want_final=False))
    for _ in range(to_end - 1):
        g.replay()
    B.sync()
    g.replay()
    with pytest.raises(RuntimeError, match="terminating step"):
        B.sync()
    B.close()
