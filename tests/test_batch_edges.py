"""Batch sizes at every wave / workgroup boundary: the hot kernels (shadow lanes past N, partial row-tile flushes) write exactly
their own output rows -- canary words on both sides of every output buffer stay untouched -- and agree bit for bit with the
generic kernels on the same actions and noise stream.  (The reference has no batch dimension; its DummyVecEnv runs 6 envs,
src/rl_utils.py:448-453 -- these sizes are the edge cases of THIS path's own geometry.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 6, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 777]
T = 37


def _guarded(torch, shape, dtype, dev, guard=512):
    """A tensor of `shape` carved out of a larger canary-filled allocation: returns (view, whole, guard elements)."""
    numel = int(np.prod(shape))
    whole = torch.full((numel + 2 * guard,), 113, dtype=dtype, device=dev)
    return whole[guard:guard + numel].view(shape), whole, guard


def _canaries_intact(whole, guard):
    w = whole.cpu().numpy()
    return bool((w[:guard] == 113).all() and (w[-guard:] == 113).all())


def _run(spec, n, layout, out_dtype, acts, route):
    import torch
    from rl_ptg_amd.engine import HipEngine
    env = {"PTG_NO_HOT_KERNELS": "1"} if route == "generic" else {}
    os.environ.update(env)
    try:
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=out_dtype, obs_layout=layout)
        eng.set_episode_plan(spec.eps_ind, n, n)
        eng.set_noise_rng(11)
        eng.reset()
        dev = torch.device("cuda", 0)
        F = eng.obs_dim
        oshape = (T, F, n) if eng.feature_major else (T, n, F)
        obs, obs_w, g = _guarded(torch, oshape, eng.out_dtype, dev)
        rew, rew_w, _ = _guarded(torch, (T, n), eng.out_dtype, dev)
        done, done_w, _ = _guarded(torch, (T, n), torch.uint8, dev)
        if route == "steps":                                  # one launch per step (k_step_hot), each into its own guarded row
            for t in range(T):
                eng.step(acts[t], obs[t], rew[t], done[t], want_final=False)
        else:
            eng.rollout(acts, obs, rew, done)
        eng.sync()
        assert _canaries_intact(obs_w, g) and _canaries_intact(rew_w, g) and _canaries_intact(done_w, g), (route, n, layout, out_dtype)
        out = (obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), done.cpu().numpy().copy(),
               {f: eng.get_state(f) for f in ("meth_state", "i", "j", "k", "current_action", "noise_count", "cum_rew")})
        eng.close()
        return out
    finally:
        for k in env:
            os.environ.pop(k, None)


@pytest.mark.parametrize("layout,out_dtype", [("row", "float32"), ("feature", "float32"), ("sb3_flat", "float32"), ("split", "float32"),
                                              ("row", "float64"), ("feature", "float64"), ("sb3_flat", "float64"), ("split", "float64")])
def test_batch_sizes_at_wave_and_workgroup_boundaries(layout, out_dtype):
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=8)
    rng = np.random.default_rng(77)
    for n in SIZES:
        acts = rng.integers(0, 5, (T, n)).astype(np.int32)
        hot = _run(spec, n, layout, out_dtype, acts, "rollout")
        stp = _run(spec, n, layout, out_dtype, acts, "steps")
        if layout == "split":                                 # the generic kernels do not write the split layout: hot rollout == hot steps
            ref = stp
        else:
            ref = _run(spec, n, layout, out_dtype, acts, "generic")
        for got, name in ((hot, "rollout"), (stp, "steps")):
            for a, b, what in zip(got[:3], ref[:3], ("obs", "rew", "done")):
                assert np.array_equal(a, b), (name, what, n, layout, out_dtype)
            for f, v in ref[3].items():
                assert np.array_equal(got[3][f], v), (name, f, n, layout, out_dtype)


@pytest.mark.parametrize("n", [65537, 100000])
def test_batches_wider_than_one_launch(n):
    """More than 65 536 envs: the rollout is cut into equal slices (100 000 -> 2 x 50 176 with shadow lanes in the last workgroup);
    same canaries, same bit-for-bit agreement with the generic kernels."""
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=3, operation="OP2", eps_len_d=8)
    acts = np.random.default_rng(78).integers(0, 5, (T, n)).astype(np.int32)
    hot = _run(spec, n, "row", "float32", acts, "rollout")
    ref = _run(spec, n, "row", "float32", acts, "generic")
    for a, b, what in zip(hot[:3], ref[:3], ("obs", "rew", "done")):
        assert np.array_equal(a, b), (what, n)
    for f, v in ref[3].items():
        assert np.array_equal(hot[3][f], v), (f, n)


def test_rollout_captured_into_a_graph_and_replayed_once():
    """ptg_rollout inside a stream capture (torch.cuda.graph): nothing runs at capture time, one replay produces what the eager call
    produces on a twin engine, and the host-side step count stays in step."""
    import torch
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=8)
    n, Tg = 4096, 24
    dev = torch.device("cuda", 0)
    acts = torch.from_numpy(np.random.default_rng(5).integers(0, 5, (2 * Tg, n)).astype(np.int32)).to(dev)
    outs = []
    for mode in ("eager", "graph"):
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
        eng.set_episode_plan(spec.eps_ind, n, n)
        eng.set_noise_rng(9)
        eng.reset()
        obs = torch.zeros((Tg, n, eng.obs_dim), device=dev); rew = torch.zeros((Tg, n), device=dev)
        done = torch.zeros((Tg, n), dtype=torch.uint8, device=dev)
        if mode == "eager":
            eng.rollout(acts[:Tg], obs, rew, done)
        else:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    eng.rollout(acts[:Tg], obs, rew, done)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            assert float(obs.abs().sum()) == 0.0               # captured, not run
            g.replay()
        eng.sync()
        first = (obs.cpu().numpy().copy(), rew.cpu().numpy().copy())
        eng.rollout(acts[Tg:], obs, rew, done)                  # eager continuation: the handle's step count matches the device state
        eng.sync()
        outs.append(first + (obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), eng.get_state("k"), eng.get_state("cum_rew")))
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("warm,Tg", [(5, 20), (0, 150)])
def test_captured_rollout_keeps_the_table_refresher(warm, Tg):
    """A rollout replayed as a hipGraph right after a synchronised reset keeps the table refresher: the pass at the head
    of a launch is part of the rollout kernel, so a captured launch has it.  65 536 envs, the bench's workload; (5, 20) is the driver's
    window, (0, 150) a long launch from reset.  Results: ONE replay == the eager call, bit for bit.  Time: the SECOND replay of the
    graph (the first also pays the runtime's one-off graph set-up) against the eager call at the same place -- the second rollout after
    the reset --, both timed by stream events recorded while a filler keeps the stream busy (no host launch latency inside either
    interval).  Allowed: 10 % + 8 us for the 20-step window -- hipGraphLaunch itself puts 5-8 us between the event and the kernel
    (tools/graph_probe.py: 48.8 vs 41.9 us for the same 20 steps, 246 vs 250 us for 150; without the refresher the 20 steps take 54 us) --
    and 15 % for the long launch: the rolling passes of a long eager launch run on a forked stream, and that branch is NOT captured,
    because the runtime replays the two branches of such a graph serially (measured 420 vs 253 us; rl_ptg_amd/csrc/ptg_env.hip,
    launch_refresher), so the captured launch runs with its head pass only (+ 8 % in the A/B runs of profiles/r03_refresh_ab.txt).
    (The second replay re-runs the captured step counts on the advanced state: fine for a stopwatch, and why the values are compared
    after the first one.)"""
    import torch
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    from rl_ptg_amd.synthetic import sticky_actions_device
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
    n = 65536
    dev = torch.device("cuda", 0)
    acts = sticky_actions_device(warm + 2 * Tg, n, seed=77, device=dev, p_switch=1.0 / 12.0)
    filler = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    res = {}
    for mode in ("eager", "graph"):
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
        eng.set_episode_plan(spec.eps_ind, n, n)
        eng.set_noise_rng(20250614)
        eng.reset()
        obs = torch.zeros((Tg, n, eng.obs_dim), device=dev); rew = torch.zeros((Tg, n), device=dev)
        done = torch.zeros((Tg, n), dtype=torch.uint8, device=dev)
        if warm:
            eng.rollout(acts[:warm], obs[:warm], rew[:warm], done[:warm])
        eng.sync()
        obs.zero_()
        torch.cuda.synchronize()
        assert eng.rollout_launches(Tg) == 1
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = None
        if mode == "graph":
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    eng.rollout(acts[warm:warm + Tg], obs, rew, done)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            assert float(obs.abs().sum()) == 0.0               # captured, not run
            g.replay()
        else:
            eng.rollout(acts[warm:warm + Tg], obs, rew, done)
        eng.sync()
        torch.cuda.synchronize()
        first = (obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), eng.get_state("i"), eng.get_state("cum_rew"))
        for _ in range(3):
            filler.zero_()                                      # ~0.6 ms of work ahead of the timed interval (longer than any host launch path; 3 GB through the caches)
        ev0.record()
        if g is not None:
            g.replay()
        else:
            eng.rollout(acts[warm + Tg:], obs, rew, done)
        ev1.record()
        torch.cuda.synchronize()
        res[mode] = (ev0.elapsed_time(ev1) * 1e3,) + first
        eng.close()
    t_e, t_g = res["eager"][0], res["graph"][0]
    for a, b in zip(res["eager"][1:], res["graph"][1:]):
        assert np.array_equal(a, b)
    assert t_g <= (1.10 * t_e + 8.0 if Tg <= 20 else 1.15 * t_e), (t_e, t_g)


def test_two_handles_on_two_streams_at_once():
    """Two independent handles driven from two streams at the same time (rollouts and per-step launches interleaved): each gets what
    it gets alone -- no state shared between handles (error words, refresher stream, profiling lists are per handle)."""
    import torch
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    dev = torch.device("cuda", 0)
    specs = [synthetic_spec(scenario=1, operation="OP1", eps_len_d=8)[0], synthetic_spec(scenario=3, operation="OP2", eps_len_d=8)[0]]
    n, Tn = 8192, 40
    rng = np.random.default_rng(11)
    acts = [torch.from_numpy(rng.integers(0, 5, (Tn, n)).astype(np.int32)).to(dev) for _ in range(2)]

    def make(i):
        eng = HipEngine(specs[i].consts, specs[i].tables, specs[i].markets, n, device=0, out_dtype=("float32", "float64")[i], obs_layout="row")
        eng.set_episode_plan(specs[i].eps_ind, n, n)
        eng.set_noise_rng(20 + i)
        eng.reset()
        return eng

    def drive(eng, a):                                       # 16 fused steps, 8 single steps, 16 fused steps
        o1, r1, _ = eng.rollout(a[:16])
        for t in range(16, 24):
            eng.step(a[t], want_final=False)
        o2, r2, _ = eng.rollout(a[24:])
        return o1, r1, o2, r2

    alone = []
    for i in range(2):
        eng = make(i)
        res = drive(eng, acts[i])
        eng.sync()
        alone.append([x.cpu().numpy().copy() for x in res] + [eng.get_state("cum_rew")])
        eng.close()
    engs = [make(0), make(1)]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    torch.cuda.synchronize()
    res = [None, None]
    for i in (0, 1):
        with torch.cuda.stream(streams[i]):
            res[i] = drive(engs[i], acts[i])
    torch.cuda.synchronize()
    for i in (0, 1):
        engs[i].sync()
        got = [x.cpu().numpy() for x in res[i]] + [engs[i].get_state("cum_rew")]
        for a, b in zip(got, alone[i]):
            assert np.array_equal(a, b), i
        engs[i].close()


def test_api_misuse_returns_error_codes():
    """Call-order and argument mistakes come back as PTG_E_INVALID (-1) with a message, never as a crash or a silent success:
    stepping before the first reset, null buffers, an action kind that contradicts cfg.action_type, zero steps, an unknown state
    field, asking for a tape that was never set; and the handle keeps working afterwards."""
    import ctypes as C
    import torch
    from rl_ptg_amd import _lib
    from rl_ptg_amd.engine import HipEngine, PtgError
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=8)
    L = _lib.lib()
    n = 300
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
    eng.set_episode_plan(spec.eps_ind, n, n)
    h = eng._h
    dev = torch.device("cuda", 0)
    a_i = torch.zeros(n, dtype=torch.int32, device=dev); a_f = torch.zeros(n, dtype=torch.float32, device=dev)
    obs = torch.zeros((4, n, eng.obs_dim), device=dev); rew = torch.zeros((4, n), device=dev); done = torch.zeros((4, n), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    E = -1
    assert L.ptg_step(h, p(a_i), _lib.ACT_I32, p(obs), p(rew), p(done), None, None, None) == E          # not reset yet
    assert b"reset" in L.ptg_last_error(h)
    assert L.ptg_rollout(h, p(a_i), _lib.ACT_I32, 1, p(obs), p(rew), p(done), None) == E
    eng.reset()
    assert L.ptg_step(h, None, _lib.ACT_I32, p(obs), p(rew), p(done), None, None, None) == E             # null actions
    assert L.ptg_step(h, p(a_i), _lib.ACT_I32, None, p(rew), p(done), None, None, None) == E             # null observations
    assert L.ptg_step(h, p(a_f), _lib.ACT_F32, p(obs), p(rew), p(done), None, None, None) == E           # float actions, discrete env
    assert L.ptg_step(h, p(a_i), 7, p(obs), p(rew), p(done), None, None, None) == E                      # no such action kind
    assert L.ptg_rollout(h, p(a_i), _lib.ACT_I32, 0, p(obs), p(rew), p(done), None) == E                 # zero steps
    assert L.ptg_rollout_launches(h, 0) < 0
    buf = (C.c_double * n)()
    assert L.ptg_get_state(h, 77, buf) == E and L.ptg_set_state(h, 77, buf) == E                         # unknown field
    assert L.ptg_get_noise_tape(h, buf) == E                                                             # no tape set
    assert L.ptg_set_noise_tape(h, None, 8) == E
    assert L.ptg_finished_episodes(h, None, None, None, 4, None) == E
    assert L.ptg_step(None, p(a_i), _lib.ACT_I32, p(obs), p(rew), p(done), None, None, None) == E        # null handle
    with pytest.raises(PtgError):
        eng._chk(L.ptg_get_state(h, 78, buf))                                                          # the Python binding raises on a code
    # ... and the handle is still good
    o, r, d = eng.rollout(np.full((4, n), 2, np.int32), obs, rew, done)
    eng.sync()
    assert np.isfinite(o.cpu().numpy()).all() and int(eng.get_state("k")[0]) == 4
    eng.close()


@pytest.mark.parametrize("out_dtype", ["float32", "float64"])
def test_feature_major_plane_pitch(out_dtype):
    """ptg_set_feature_pitch (HipEngine obs_pitch): feature planes `pitch` elements apart -- the [F, N] view of [F, pitch] storage holds
    what the back-to-back layout holds, bit for bit: reset rows, per-step launches (hot and generic kernel, incl. a terminating step with
    its terminal observations) and fused rollouts; the padding between the planes is never written."""
    import torch
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=1, train_steps=200000)      # 139-step episodes: a termination inside
    n, Tn = 1000, 160
    dev = torch.device("cuda", 0)
    acts = torch.from_numpy(np.random.default_rng(3).integers(0, 5, (Tn, n)).astype(np.int32)).to(dev)
    outs = {}
    for pitch in (None, n + 24):
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=out_dtype, obs_layout="feature", obs_pitch=pitch)
        assert eng.pitch == (n if pitch is None else pitch)
        eng.set_episode_plan(spec.eps_ind, n, n)
        eng.set_noise_rng(8)
        o0 = eng.reset().clone()
        store = torch.full((Tn, eng.obs_dim, eng.pitch), -7.0, dtype=eng.out_dtype, device=dev)
        obs = store[:, :, :n]
        o_s, r_s, d_s = [], [], []
        for t in range(20):                                              # hot per-step launches into rows of the pitched rollout buffer
            o, r, d = eng.step(acts[t], obs=obs[t])
            o_s.append(o.clone()); r_s.append(r.clone())
        o_r, r_r, d_r = eng.rollout(acts[20:], obs[20:])                  # fused, across the terminating step at 139 (generic kernel + reset rows)
        eng.sync()
        assert int(d_r.sum()) == n
        if eng.pitch != n:
            assert bool((store[:, :, n:] == -7.0).all())                  # the padding stays untouched
            with pytest.raises(ValueError):
                eng.rollout(acts[:4], torch.empty((4, eng.obs_dim, n), dtype=eng.out_dtype, device=dev))      # a back-to-back buffer no longer fits
        outs[pitch] = (o0.cpu().numpy(), torch.stack(o_s).cpu().numpy(), torch.stack(r_s).cpu().numpy(), o_r.cpu().numpy(), r_r.cpu().numpy(),
                       eng.final_obs.cpu().numpy().copy(), eng.get_state("i"), eng.get_state("cum_rew"))
        eng.close()
    for a, b in zip(outs[None], outs[n + 24]):
        assert np.array_equal(a, b)
    # "auto": a pitch only where the plane stride would be a multiple of 64 KiB
    e1 = HipEngine(spec.consts, spec.tables, spec.markets, 16384, device=0, out_dtype="float64", obs_layout="feature", obs_pitch="auto")
    e2 = HipEngine(spec.consts, spec.tables, spec.markets, 1000, device=0, out_dtype="float64", obs_layout="feature", obs_pitch="auto")
    assert e1.pitch == 16384 + 128 and e2.pitch == 1000
    e1.close(); e2.close()
    with pytest.raises(ValueError):
        HipEngine(spec.consts, spec.tables, spec.markets, 64, device=0, obs_layout="row", obs_pitch=80)
