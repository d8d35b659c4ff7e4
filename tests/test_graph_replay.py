"""hipGraph replay of the hot path (include/ptg_env.h, "hipGraph capture"): `ptg_step` / `ptg_rollout` captured once and replayed step after
step give what eager calls give, bit for bit -- the kernels take the step count from the device state -- and `ptg_note_replays` keeps the
host's count (which routes an episode's terminating step, env/ptg_gym_env.py:508-511, to the generic kernel) in step.  A replay that runs
over the terminating step is reported, not silently wrong."""
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


def test_replay_over_the_terminating_step_is_reported():
    import torch
    n = 256
    spec, A, B = _pair(n, sim_step=3600)                    # 4-day episodes of hourly steps: 96 steps, the 91st call terminates
    A.close()
    B.reset()
    to_end = B.steps_to_episode_end()
    assert to_end == spec.consts["eps_sim_steps"] - 5
    act_buf = torch.full((n,), 2, dtype=torch.int32, device="cuda")
    obs, rew, done = B.alloc_obs(1)[0], torch.zeros(n, dtype=B.out_dtype, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
    g = _capture(lambda: B.step(act_buf, obs, rew, done, want_final=False))
    with pytest.raises(RuntimeError, match="run over the terminating step"):
        B.note_replays(to_end)                              # more than the hot kernels may run
    for _ in range(to_end - 1):                             # all the steps a hot kernel may take ...
        g.replay()
    B.sync()                                                # ... are fine
    g.replay()                                              # the terminating step through a hot kernel
    with pytest.raises(RuntimeError, match="terminating step"):
        B.sync()
    B.close()
