"""GPU parity: the HIP path (through the C ABI) against the golden vectors of the unmodified reference and
against the CPU oracle.  Integers bit-exact; floats within the tolerances written below (north_star: 1e-5
relative; measured margins are far tighter)."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

# float64 outputs: identical operand order -> at most a few ulp (pow() vs multiply in the efficiency polynomial)
RTOL64, ATOL64 = 1e-11, 1e-13
# float32 outputs: one rounding of the float64 result
RTOL32, ATOL32 = 2e-7, 1e-9

INT_FIELDS = ["meth_state", "i", "j", "hot_cold", "standby_tid", "startup_tid", "partial_tid", "full_tid", "k", "current_action"]


def _ints(eng):
    cols = [eng.get_state(f) for f in INT_FIELDS]
    actd = eng.get_state("act_ep_d")
    return np.stack(cols + [actd * 24, actd], axis=1)


@pytest.mark.parametrize("out_dtype,layout", [("float64", "row"), ("float32", "row"), ("float32", "feature"), ("float64", "feature")])
@pytest.mark.parametrize("case", H.TRAJ_CASES)
def test_trajectory_vs_reference_golden(case, out_dtype, layout):
    tr, eng = H.make_engine(case, out_dtype, obs_layout=layout)
    rtol, atol = (RTOL64, ATOL64) if out_dtype == "float64" else (RTOL32, ATOL32)
    K, n = tr["actions"].shape
    obs0 = eng.rows(eng.reset()).cpu().numpy()
    np.testing.assert_allclose(obs0, tr["reset_obs"], rtol=rtol, atol=atol)
    assert np.array_equal(_ints(eng), tr["reset_int"])
    post_at = [tuple(x) for x in tr["post_reset_at"].tolist()]
    n_post = 0
    check_every = 1 if K <= 1000 else 7
    ret_acc = np.zeros(n)
    fin_expect, fin_got = [], []
    for t in range(K):
        obs, rew, done = eng.step(tr["actions"][t])
        eng.sync()
        obs, rew, done = eng.rows(obs).cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        assert np.array_equal(done, tr["done"][t]), f"done differs at step {t}"
        np.testing.assert_allclose(rew, tr["f64s"][t, :, 0], rtol=rtol, atol=atol, err_msg=f"reward, step {t}")
        ret_acc += tr["f64s"][t, :, 0]
        final = eng.rows(eng.final_obs).cpu().numpy()
        any_done = bool(done.any())
        if any_done or t % check_every == 0 or t == K - 1:
            ints = _ints(eng)
            T = eng.get_state("T_cat")
            cum = eng.get_state("cum_rew")
        for e in range(n):
            if done[e]:
                assert post_at[n_post] == (t, e)
                np.testing.assert_allclose(final[e], tr["obs"][t, e], rtol=rtol, atol=atol)
                np.testing.assert_allclose(obs[e], tr["post_reset_obs"][n_post], rtol=rtol, atol=atol)
                assert np.array_equal(ints[e], tr["post_reset_int"][n_post])
                fin_expect.append((e, ret_acc[e], int(tr["ints"][t, e, 8])))
                ret_acc[e] = 0.0
                n_post += 1
            else:
                np.testing.assert_allclose(obs[e], tr["obs"][t, e], rtol=rtol, atol=atol, err_msg=f"obs, step {t} env {e}")
                if any_done or t % check_every == 0 or t == K - 1:
                    assert np.array_equal(ints[e], tr["ints"][t, e]), f"int state, step {t} env {e}: {ints[e]} vs {tr['ints'][t, e]}"
                    assert T[e] == tr["f64s"][t, e, 2]
                    np.testing.assert_allclose(cum[e], tr["f64s"][t, e, 1], rtol=1e-11, atol=1e-9)
        if any_done:
            r, l, ids = eng.finished_episodes()
            fin_got.extend(zip(ids.tolist(), l.tolist(), r.tolist()))
        if "infos" in tr:
            info = eng.info.cpu().numpy()
            np.testing.assert_allclose(info, tr["infos"][t], rtol=RTOL64, atol=ATOL64, err_msg=f"info rows, step {t}")
    assert n_post == len(post_at)
    assert np.array_equal(eng.get_state("noise_count"), tr["noise_len"])
    r, l, ids = eng.finished_episodes()
    assert len(r) == 0
    assert len(fin_got) == len(fin_expect)
    got = sorted(fin_got)
    exp = sorted((e, ln, rr) for e, rr, ln in fin_expect)
    for (e1, l1, r1), (e2, l2, r2) in zip(got, exp):
        assert e1 == e2 and l1 == l2
        assert abs(r1 - r2) <= 1e-9 * max(1.0, abs(r2))
    eng.close()


@pytest.mark.parametrize("op,case", [("OP1", "synth_bs1_op1_mod_disc_train"), ("OP2", "synth_bs2_op2_mod_disc_train")])
def test_device_built_get_index_lut(op, case):
    """k_build_argmin == the reference's _get_index for every distinct catalyst temperature."""
    u = H.load_npz(f"{H.GOLD}/units_{op}.npz")
    tr, eng = H.make_engine(case)
    T, lut = eng.debug_get_index_lut()
    assert np.array_equal(T, u["T"])
    assert u["dests"].tolist() == ["cooldown", "standby_up", "standby_down", "startup_cold", "startup_hot", "op1_start_p"]
    assert np.array_equal(lut, u["get_index"])
    eng.close()


@pytest.mark.parametrize("case", ["synth_bs2_op2_mod_disc_train", "synth_bs1_op1_s60_toggle"])
def test_device_built_window_records_match_numpy_average(case):
    """k_build_records == slicing + np.average (NumPy pairwise order), incl. table-end and splice windows."""
    tr, eng = H.make_engine(case)
    _, consts, tables, _ = H.load_traj(case)
    S = consts["sim_step"] // consts["time_step_op"]
    rng = np.random.default_rng(1)
    for tid, key in enumerate(H.po.TABLE_KEYS):
        tab = tables[key]
        n = len(tab)
        starts = sorted(set([0, 1, n - S - 1, n - S, n - S + 1, n - 1, n] + rng.integers(0, n, 6).tolist()))
        for r in starts:
            if r < 0:
                continue
            if r == n:
                win = np.ones((S, 7)) * tab[-1]
            elif r + S <= n:
                win = tab[r:r + S]
            elif tid <= 1:
                win = np.concatenate((tab[r:], tables["op1_start_p"][:r + S - n]), axis=0)
            else:
                win = np.concatenate((tab[r:], np.ones((r + S - n, 7)) * tab[-1]), axis=0)
            rec = eng.debug_window_record(tid, r)
            assert rec[0] == win[-1, 1]
            for c in range(5):
                assert rec[1 + c] == np.average(win[:, 2 + c]), (key, r, c)
    eng.close()


@pytest.mark.parametrize("out_dtype,layout", [("float64", "row"), ("float32", "row"), ("float32", "feature")])
def test_rollout_equals_steps(out_dtype, layout):
    case = "synth_bs2_op2_term_penalty"
    tr, e1 = H.make_engine(case, out_dtype, obs_layout=layout)
    _, e2 = H.make_engine(case, out_dtype, obs_layout=layout)
    K = 400
    e1.reset(); e2.reset()
    obs_r, rew_r, done_r = e2.rollout(tr["actions"][:K])
    e2.sync()
    for t in range(K):
        o, r, d = e1.step(tr["actions"][t])
        e1.sync()
        assert np.array_equal(o.cpu().numpy(), obs_r[t].cpu().numpy())
        assert np.array_equal(r.cpu().numpy(), rew_r[t].cpu().numpy())
        assert np.array_equal(d.cpu().numpy(), done_r[t].cpu().numpy())
    for f in INT_FIELDS + ["act_ep_d", "ep_ptr", "noise_count", "n_state_changes", "T_cat", "cum_rew"]:
        assert np.array_equal(e1.get_state(f), e2.get_state(f)), f
    e1.close(); e2.close()


def test_invalid_discrete_action_is_an_error_not_ub():
    from rl_ptg_amd.engine import PtgError
    tr, eng = H.make_engine("synth_bs2_op2_mod_disc_train")
    eng.reset()
    bad = np.array([7, 0], dtype=np.int32)
    eng.step(bad)
    with pytest.raises(PtgError) as ei:
        eng.sync()
    assert ei.value.code == -3
    eng.step(np.array([-1, 4], dtype=np.int32))      # python-style negative index is legal: actions[-1] = full_load
    eng.sync()
    assert eng.get_state("current_action").tolist()[0] == 4
    eng.close()


def _synthetic_engine(n, scenario=1, operation="OP1", out_dtype="float64", layout="row"):
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=scenario, operation=operation, eps_len_d=32)
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=out_dtype, obs_layout=layout)
    eng.set_episode_plan(spec.eps_ind, n, n)
    return spec, eng


def _oracle_for(spec, n, tape):
    m = spec.markets[0]
    consts = dict(spec.consts, scenario=m["scenario"], rew_l_b=m["rew_l_b"], rew_u_b=m["rew_u_b"], r_0=m["r_0"])
    ora = H.po.OracleVecEnv(consts, spec.tables, dict(m, eps_ind=None), n)
    ora.set_noise_tape(tape)
    return ora


@pytest.mark.parametrize("scenario,operation", [(1, "OP1"), (2, "OP2"), (3, "OP2")])
def test_device_rng_tape_vs_oracle_and_inline_mode(scenario, operation):
    """N = 4096 (BASELINE.json configs[1] size), uniform-random actions (43 % state changes), every business scenario: the HIP
    path with a device-generated noise tape equals the oracle fed the same tape, and the in-kernel RNG mode equals the tape
    mode bit for bit."""
    n, K, L = 4096, 120, 128
    spec, e_tape = _synthetic_engine(n, scenario, operation)
    _, e_rng = _synthetic_engine(n, scenario, operation)
    e_tape.fill_noise_tape(seed=77, per_env_len=L)
    e_rng.set_noise_rng(seed=77)
    tape = e_tape.get_noise_tape(L)
    assert abs(tape.mean()) < 0.05 and abs(tape.std() - 10.0) < 0.05          # N(0, noise = 10)
    ora = _oracle_for(spec, n, tape)
    o_ref, _ = ora.reset()
    np.testing.assert_allclose(e_tape.reset().cpu().numpy(), o_ref, rtol=RTOL64, atol=ATOL64)
    e_rng.reset()
    rng = np.random.default_rng(3)
    for t in range(K):
        a = rng.integers(0, 5, n).astype(np.int32)
        o1, r1, d1 = e_tape.step(a)
        o2, r2, d2 = e_rng.step(a)
        e_tape.sync(); e_rng.sync()
        o_ref, r_ref, d_ref, _, _ = ora.step(a)
        np.testing.assert_allclose(o1.cpu().numpy(), o_ref, rtol=RTOL64, atol=ATOL64)
        np.testing.assert_allclose(r1.cpu().numpy(), r_ref, rtol=RTOL64, atol=ATOL64)
        assert np.array_equal(d1.cpu().numpy(), d_ref)
        assert np.array_equal(o1.cpu().numpy(), o2.cpu().numpy()) and np.array_equal(r1.cpu().numpy(), r2.cpu().numpy())
    ints, f64s = ora.state()
    for col, name in [(0, "meth_state"), (1, "i"), (2, "j"), (3, "hot_cold"), (4, "standby_tid"), (5, "startup_tid"),
                      (6, "partial_tid"), (7, "full_tid"), (8, "k"), (9, "current_action")]:
        assert np.array_equal(e_tape.get_state(name), ints[:, col]), name
        assert np.array_equal(e_rng.get_state(name), ints[:, col]), name
    assert np.array_equal(e_tape.get_state("T_cat"), f64s[:, 2])
    assert e_tape.get_state("noise_count").max() <= L
    e_tape.close(); e_rng.close(); ora.close()


def test_full_size_properties_n65536():
    """BASELINE.json size (N = 65 536, BS1/OP1): size-independent properties instead of an oracle run.
    (1) permutation equivariance: env e of a batch stepped with permuted actions/noise streams equals env perm[e];
    (2) fused rollout == per-step launches; (3) float32 fast path == float64 reference-order path within 2e-7;
    (4) replicas: envs given identical actions and noise stay identical (checksum of checksums)."""
    n, K = 65536, 48
    spec, a64 = _synthetic_engine(n, out_dtype="float64", layout="row")
    _, a32 = _synthetic_engine(n, out_dtype="float32", layout="feature")
    _, b32 = _synthetic_engine(n, out_dtype="float32", layout="feature")
    rng = np.random.default_rng(11)
    L = 64
    tape = rng.normal(0, 10, (n, L))
    acts = rng.integers(0, 5, (K, n)).astype(np.int32)
    acts[:, : n // 2] = acts[:, :1]                      # first half of the batch: replicas of env 0
    tape[: n // 2] = tape[0]
    perm = rng.permutation(n)
    a64.set_noise_tape(tape); a32.set_noise_tape(tape); b32.set_noise_tape(tape[perm])
    a64.reset(); a32.reset(); b32.reset()
    ro, rr, rd = b32.rollout(acts[:, perm])
    b32.sync()
    for t in range(K):
        o64, r64, d64 = a64.step(acts[t])
        o32, r32, d32 = a32.step(acts[t])
        a64.sync(); a32.sync()
        o64n, o32n = o64.cpu().numpy(), a32.rows(o32).cpu().numpy()
        np.testing.assert_allclose(o32n, o64n, rtol=RTOL32, atol=ATOL32)
        np.testing.assert_allclose(r32.cpu().numpy(), r64.cpu().numpy(), rtol=RTOL32, atol=1e-6)
        assert np.array_equal(b32.rows(ro[t]).cpu().numpy(), o32n[perm])           # (1) + (2)
        assert np.array_equal(rr[t].cpu().numpy(), r32.cpu().numpy()[perm])
        assert np.array_equal(o32n[: n // 2], np.broadcast_to(o32n[0], (n // 2, o32n.shape[1])))   # (4)
    for f in INT_FIELDS:
        v = a32.get_state(f)
        assert np.array_equal(v, a64.get_state(f)) and np.array_equal(b32.get_state(f), v[perm]), f
    a64.close(); a32.close(); b32.close()


def test_sharded_engines_equal_one_batch():
    """Two shards (rank 0 / 1 of world_size 2, here on one GPU) == one engine over all envs: episode plan over the shared
    eps_ind order, RNG streams keyed by the global env index, mixed business scenarios per env (BASELINE.json config 5)."""
    from rl_ptg_amd import dist as ptg_dist
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import EnvSpec, synthetic_spec
    specs = [synthetic_spec(scenario=sc, operation="OP2", eps_len_d=2, train_steps=40000)[0] for sc in (1, 2, 3)]
    spec = EnvSpec.merge_scenarios(specs)
    n_total, world = 512, 2
    n = n_total // world
    K = 300                                               # 2-day episodes: 283 steps -> every env terminates once

    def make(n_envs, rank, ws):
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n_envs, device=0, out_dtype="float32", obs_layout="feature")
        first_ptr, stride = ptg_dist.episode_plan(n_total, ws, rank)
        eng.set_global_env_offset(first_ptr - n_total)
        eng.set_market_assignment(ptg_dist.mixed_scenario_assignment(n_total, ws, rank, 3))
        eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
        eng.set_noise_rng(99)
        eng.reset()
        return eng

    full, s0, s1 = make(n_total, 0, 1), make(n, 0, 2), make(n, 1, 2)
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 5, (K, n_total)).astype(np.int32)
    of, rf, df = full.rollout(acts)
    o0, r0, d0 = s0.rollout(acts[:, :n])
    o1, r1, d1 = s1.rollout(acts[:, n:])
    full.sync(); s0.sync(); s1.sync()
    assert int(df.sum()) == n_total                      # each env finished exactly one episode
    assert np.array_equal(of[:, :, :n].cpu().numpy(), o0.cpu().numpy()) and np.array_equal(of[:, :, n:].cpu().numpy(), o1.cpu().numpy())
    assert np.array_equal(rf[:, :n].cpu().numpy(), r0.cpu().numpy()) and np.array_equal(rf[:, n:].cpu().numpy(), r1.cpu().numpy())
    assert np.array_equal(df[:, n:].cpu().numpy(), d1.cpu().numpy())
    for f in INT_FIELDS + ["act_ep_d", "market_set"]:
        assert np.array_equal(full.get_state(f), np.concatenate([s0.get_state(f), s1.get_state(f)])), f
    # episodic returns: the per-shard lists, gathered, are the batch's list
    rF, lF, iF = full.finished_episodes()
    rA, lA, iA = s0.finished_episodes()
    rB, lB, iB = s1.finished_episodes()
    got = sorted(zip(np.concatenate([iA, iB + n]).tolist(), np.concatenate([rA, rB]).tolist()))
    assert got == sorted(zip(iF.tolist(), rF.tolist())) and len(got) == n_total
    assert set(np.concatenate([lA, lB]).tolist()) == {283}
    # scenario mix actually differs: BS3 envs (gas = EUA = 0, CHP revenue) earn differently from BS1 / BS2 envs
    ms = full.get_state("market_set")
    assert len({round(float(np.mean(rF[np.argsort(iF)][ms == q])), 3) for q in range(3)}) == 3
    full.close(); s0.close(); s1.close()


def test_full_size_properties_n262144_bs3():
    """BASELINE.json configs[3]: N = 262 144, BS3/OP2 (CHP / EEG reward path).  Fused rollout == per-step launches (bit-equal),
    float32 hot path == float64 reference-order path within one float32 rounding, CHP revenue present."""
    n, K = 262144, 10
    spec, a64 = _synthetic_engine(n, 3, "OP2", out_dtype="float64", layout="feature")
    _, a32 = _synthetic_engine(n, 3, "OP2", out_dtype="float32", layout="feature")
    _, b32 = _synthetic_engine(n, 3, "OP2", out_dtype="float32", layout="feature")
    for e in (a64, a32, b32):
        e.set_noise_rng(7)
        e.reset()
    rng = np.random.default_rng(2)
    acts = rng.integers(0, 5, (K, n)).astype(np.int32)
    acts[:4] = 2                                          # everybody starts up first
    ro, rr, rd = b32.rollout(acts)
    b32.sync()
    for t in range(K):
        o64, r64, _ = a64.step(acts[t])
        o32, r32, _ = a32.step(acts[t])
        a64.sync(); a32.sync()
        assert np.array_equal(ro[t].cpu().numpy(), o32.cpu().numpy()) and np.array_equal(rr[t].cpu().numpy(), r32.cpu().numpy())
        np.testing.assert_allclose(o32.cpu().numpy(), o64.cpu().numpy(), rtol=RTOL32, atol=ATOL32)
        np.testing.assert_allclose(r32.cpu().numpy(), r64.cpu().numpy(), rtol=RTOL32, atol=1e-6)
    for f in INT_FIELDS:
        assert np.array_equal(a32.get_state(f), a64.get_state(f)) and np.array_equal(b32.get_state(f), a32.get_state(f)), f
    assert int(rd.sum()) == 0
    a64.close(); a32.close(); b32.close()


def test_hot_and_generic_kernels_agree_across_episode_boundaries():
    """ptg_rollout / ptg_step route non-terminating steps of a synchronised batch to the hot kernels and the terminating step
    to the generic kernel.  1-day episodes (139 steps), 1000 steps = 7 terminations: identical to the all-generic route,
    for the fused and the per-step path, float32 row- and feature-major."""
    import os
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=1, train_steps=200000, state_change_penalty=0.25)
    n, K = 128, 1000                                      # 7 x 128 finished episodes fit the library's list (>= 1024 entries)
    rng = np.random.default_rng(9)
    acts = rng.integers(0, 5, (K, n)).astype(np.int32)
    results = {}
    for route in ("hot", "generic"):
        for layout in ("feature", "row"):
            if route == "generic":
                os.environ["PTG_NO_HOT_KERNELS"] = "1"
            try:
                eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout=layout)
                eng.set_episode_plan(spec.eps_ind, n, n)
                eng.set_noise_rng(3)
                eng.reset()
                o, r, d = eng.rollout(acts[:600])
                eng.sync()
                outs = [eng.rows(o).cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()]
                so, sr, sd = [], [], []
                for t in range(600, K):
                    oo, rr, dd = eng.step(acts[t])
                    eng.sync()
                    so.append(eng.rows(oo).cpu().numpy().copy()); sr.append(rr.cpu().numpy().copy()); sd.append(dd.cpu().numpy().copy())
                fin = eng.finished_episodes()
                state = {f: eng.get_state(f) for f in INT_FIELDS + ["act_ep_d", "ep_ptr", "noise_count", "n_state_changes", "cum_rew"]}
                results[(route, layout)] = (outs, np.array(so), np.array(sr), np.array(sd), fin, state)
                eng.close()
            finally:
                os.environ.pop("PTG_NO_HOT_KERNELS", None)
    ref = results[("generic", "row")]
    assert int(ref[0][2].sum()) + int(ref[3].sum()) == 7 * n            # every env terminated seven times
    for key, res in results.items():
        for a, b in zip(res[0], ref[0]):
            assert np.array_equal(a, b), key
        assert np.array_equal(res[1], ref[1]) and np.array_equal(res[2], ref[2]) and np.array_equal(res[3], ref[3]), key
        assert sorted(zip(res[4][2].tolist(), res[4][0].tolist())) == sorted(zip(ref[4][2].tolist(), ref[4][0].tolist())), key
        for f, v in res[5].items():
            assert np.array_equal(v, ref[5][f]), (key, f)


@pytest.mark.parametrize("sim_step,action_type,raw_modified,out_dtype", [
    (60, "discrete", "mod", "float32"), (60, "continuous", "raw", "float32"), (300, "discrete", "raw", "float64"),
    (1200, "continuous", "mod", "float32"), (120, "discrete", "mod", "float32")])
def test_differential_vs_oracle_other_step_sizes(sim_step, action_type, raw_modified, out_dtype):
    """Differential test against the CPU oracle for step sizes the golden fixtures do not cover (30, 60, 150, 600 rows per
    step), toggling partial / full load with short holds so that every rung of both ladders is visited."""
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=4, sim_step=sim_step, raw_modified=raw_modified,
                             action_type=action_type, train_steps=60 * 8 * (4 * 86400 // sim_step))   # eps_ind long enough for 2 x 2048 draws
    n, K = 2048, 260
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=out_dtype, obs_layout="feature")
    eng.set_episode_plan(spec.eps_ind, n, n)
    eng.fill_noise_tape(seed=sim_step, per_env_len=64)
    tape = eng.get_noise_tape(64)
    m = spec.markets[0]
    consts = dict(spec.consts, scenario=m["scenario"], rew_l_b=m["rew_l_b"], rew_u_b=m["rew_u_b"], r_0=m["r_0"])
    ora = H.po.OracleVecEnv(consts, spec.tables, dict(m, eps_ind=None if spec.eps_ind is None else spec.eps_ind), n, ep_index0=0)
    ora.set_noise_tape(tape)
    rtol, atol = (RTOL64, ATOL64) if out_dtype == "float64" else (RTOL32, ATOL32)
    # the oracle's construction consumed eps_ind[0:n], its reset takes eps_ind[n + e] -- the plan given to the engine
    o_ref, _ = ora.reset()
    np.testing.assert_allclose(eng.rows(eng.reset()).cpu().numpy(), o_ref, rtol=rtol, atol=atol)
    rng = np.random.default_rng(sim_step)
    warm = max(3, int(3600 * 2.2 / sim_step))                       # ~2.2 h of startup reaches production from cold
    hold = rng.integers(1, 12, n)
    cur = np.full(n, 3)
    for t in range(K):
        if t < warm:
            a = np.full(n, 2)
        else:
            flip = (t - warm) % hold == 0
            cur = np.where(flip, 7 - cur, cur)
            a = cur.copy()
            detour = rng.random(n) < 0.01
            a[detour] = rng.integers(0, 3, int(detour.sum()))
        if action_type == "continuous":
            acts = (-1 + 0.4 * (a + 0.5) + rng.uniform(-0.19, 0.19, n)).astype(np.float32)
        else:
            acts = a.astype(np.int32)
        o, r, d = eng.step(acts)
        eng.sync()
        o_ref, r_ref, d_ref, _, _ = ora.step(acts)
        np.testing.assert_allclose(eng.rows(o).cpu().numpy(), o_ref, rtol=rtol, atol=atol, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=rtol, atol=max(atol, 1e-6 if out_dtype == "float32" else 0), err_msg=f"reward step {t}")
        assert np.array_equal(d.cpu().numpy(), d_ref)
    ints, f64s = ora.state()
    for col, name in [(0, "meth_state"), (1, "i"), (2, "j"), (3, "hot_cold"), (4, "standby_tid"), (5, "startup_tid"),
                      (6, "partial_tid"), (7, "full_tid"), (8, "k"), (9, "current_action"), (11, "act_ep_d")]:
        assert np.array_equal(eng.get_state(name), ints[:, col]), name
    assert np.array_equal(eng.get_state("T_cat"), f64s[:, 2])
    # ladder coverage: every partial-load and full-load table was in use somewhere in the batch
    assert len(set(eng.get_state("partial_tid").tolist())) >= 3 and len(set(eng.get_state("full_tid").tolist())) >= 3
    eng.close(); ora.close()


@pytest.mark.gpu
@pytest.mark.parametrize("action_type", ["discrete", "continuous"])
def test_fused_rollout_slices_segments_and_kept_actions(action_type):
    """The fused rollout runs as env slices x step segments with the launch's actions staged in LDS as one-byte codes.
    n = 1000 envs in 256-env slices (ragged last slice and workgroup), 700 steps (two segments of the 512-step stage), action
    rows that hold "keep the previous action" codes (continuous: values >= 1 and NaN, :351-355): bit-identical to the
    all-generic route and to per-step launches; the launch count reported by the library matches the geometry."""
    import os
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=8, train_steps=400000, action_type=action_type)
    n, K = 1000, 700
    rng = np.random.default_rng(21)
    if action_type == "discrete":
        acts = rng.integers(-5, 5, (K, n)).astype(np.int64)          # python-style negative indices included (:347)
    else:
        acts = rng.uniform(-1.2, 1.3, (K, n)).astype(np.float32)
        acts[rng.random((K, n)) < 0.05] = np.nan
    res = {}
    for route in ("generic", "sliced", "steps"):
        env = {"generic": {"PTG_NO_HOT_KERNELS": "1"}, "sliced": {"PTG_PC_CHUNK": "256"}, "steps": {}}[route]
        os.environ.update(env)
        try:
            eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="feature")
            eng.set_episode_plan(spec.eps_ind, n, n)
            eng.set_noise_rng(5)
            eng.reset()
            if route == "sliced":
                assert eng.rollout_launches(K) == 2 * 4                  # ceil(700 / 512) segments x ceil(1000 / 256) slices
            if route == "steps":
                oo, rr = [], []
                for t in range(K):
                    o, r, d = eng.step(acts[t])
                    eng.sync()
                    oo.append(o.cpu().numpy().copy()); rr.append(r.cpu().numpy().copy())
                out = (np.array(oo), np.array(rr))
            else:
                o, r, d = eng.rollout(acts)
                eng.sync()
                out = (o.cpu().numpy(), r.cpu().numpy())
            res[route] = (out, {f: eng.get_state(f) for f in INT_FIELDS + ["noise_count", "cum_rew", "current_action"]})
            eng.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    for route in ("sliced", "steps"):
        assert np.array_equal(res[route][0][0], res["generic"][0][0]), route
        assert np.array_equal(res[route][0][1], res["generic"][0][1]), route
        for f, v in res["generic"][1].items():
            assert np.array_equal(res[route][1][f], v), (route, f)


@pytest.mark.gpu
def test_fused_rollout_flags_invalid_discrete_action():
    """An out-of-range discrete action inside a fused rollout raises PTG_E_ACTION at the next sync (IndexError in the
    reference, :347) and leaves the env on its previous action."""
    from rl_ptg_amd.engine import HipEngine, PtgError
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=8)
    n, K = 300, 20
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="feature")
    eng.set_episode_plan(spec.eps_ind, n, n)
    eng.set_noise_rng(1)
    eng.reset()
    acts = np.full((K, n), 2, np.int32)
    acts[7, 123] = 9
    eng.rollout(acts)
    with pytest.raises(PtgError):
        eng.sync()
    assert int(eng.get_state("current_action")[123]) == 2
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["row", "feature"])
@pytest.mark.parametrize("variant", ["default", "no_lds_lut", "block128", "no_refresh", "refresh_always"])
def test_hot_kernels_raw_features_and_launch_variants(layout, variant):
    """'raw' observations (26 columns: the row-major tile has an even stride) and the launch variants of the fused rollout
    (lookup table left in global memory, 64-env workgroups) against the generic kernels, bit for bit."""
    import os
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=3, operation="OP2", eps_len_d=8, raw_modified="raw", train_steps=400000)
    n, K = 640, 120
    acts = np.random.default_rng(17).integers(0, 5, (K, n)).astype(np.int32)
    env = {"default": {}, "no_lds_lut": {"PTG_NO_LDS_LUT": "1"}, "block128": {"PTG_BLOCK": "128"}, "no_refresh": {"PTG_NO_REFRESH": "1"},
           "refresh_always": {"PTG_REFRESH_ALWAYS": "1"}}[variant]
    out = {}
    for route in ("generic", "hot"):
        e2 = dict(env) if route == "hot" else {"PTG_NO_HOT_KERNELS": "1"}
        os.environ.update(e2)
        try:
            eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout=layout)
            assert eng.obs_dim == 26
            eng.set_episode_plan(spec.eps_ind, n, n)
            eng.set_noise_rng(11)
            eng.reset()
            o, r, d = eng.rollout(acts[:100])
            rest = [eng.step(acts[t]) for t in range(100, K)]
            eng.sync()
            out[route] = (o.cpu().numpy(), r.cpu().numpy(), np.array([x[0].cpu().numpy() for x in rest]),
                          {f: eng.get_state(f) for f in INT_FIELDS + ["cum_rew", "noise_count"]})
            eng.close()
        finally:
            for k in e2:
                os.environ.pop(k, None)
    assert np.array_equal(out["hot"][0], out["generic"][0]) and np.array_equal(out["hot"][1], out["generic"][1])
    assert np.array_equal(out["hot"][2], out["generic"][2])
    for f, v in out["generic"][3].items():
        assert np.array_equal(out["hot"][3][f], v), f


@pytest.mark.gpu
@pytest.mark.parametrize("raw_modified,out_dtype", [("mod", "float32"), ("raw", "float32"), ("mod", "float64")])
def test_sb3_flat_layout_equals_flattened_dict_observation(raw_modified, out_dtype):
    """obs_layout="sb3_flat": rows are what SB3's CombinedExtractor feeds a MultiInputPolicy (sub-spaces in sorted key order,
    Discrete(6) METH_STATUS one-hot).  stable-baselines3 (2.x, un-vendored, not installed here) is restated in
    oracle/sb3_flat_oracle.py (test infrastructure, pinned by hand-built known answers); the native layout must equal that
    restatement applied to the row-major output, bit for bit, and so must the product's torch helper sb3_flat_features -- hot kernels (step + fused rollout, ragged last wave), generic kernels (float64, terminating steps with reset rows and
    terminal observations) alike."""
    import torch
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    from rl_ptg_amd.vec_env import sb3_flat_features
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=1, raw_modified=raw_modified, train_steps=200000)
    n, K = 200, 300                                       # 1-day episodes: 139 steps -> two terminations inside
    acts = np.random.default_rng(4).integers(0, 5, (K, n)).astype(np.int32)
    res = {}
    for layout in ("row", "sb3_flat"):
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=out_dtype, obs_layout=layout)
        eng.set_episode_plan(spec.eps_ind, n, n)
        eng.set_noise_rng(8)
        o0 = eng.reset().clone()
        o, r, d = eng.rollout(acts[:250])
        so, fo = [], []
        for t in range(250, K):
            oo, rr, dd = eng.step(acts[t], want_final=True)
            eng.sync()
            so.append(oo.clone())
            if bool(dd.any()):
                fo.append((dd.bool().clone(), eng.final_obs.clone()))
        eng.sync()
        res[layout] = (eng.obs_dim, o0, o, r.cpu().numpy(), d.cpu().numpy(), torch.stack(so), fo)
        eng.close()
    F = res["row"][0]
    assert res["sb3_flat"][0] == F + 5 == (40 if raw_modified == "mod" else 31)
    import os, sys
    sys.path.insert(0, os.path.join(H.ROOT, "oracle"))
    import sb3_flat_oracle

    def flat(x):                                          # the independent checker; float64 engines keep float64 rows (values are exact either way)
        xs = x.cpu().numpy()
        y = sb3_flat_oracle.flatten_rows(xs.reshape(-1, xs.shape[-1]).astype(np.float64), raw_modified).reshape(xs.shape[:-1] + (-1,))
        assert np.array_equal(y, sb3_flat_features(x, raw_modified=raw_modified).cpu().numpy().astype(np.float32))     # product helper == oracle
        return torch.from_numpy(y.astype(xs.dtype)).to(x.device) if xs.dtype == np.float32 else sb3_flat_features(x, raw_modified=raw_modified)
    assert torch.equal(res["sb3_flat"][1], flat(res["row"][1]))
    assert torch.equal(res["sb3_flat"][2], flat(res["row"][2]))
    assert torch.equal(res["sb3_flat"][5], flat(res["row"][5]))
    assert np.array_equal(res["sb3_flat"][3], res["row"][3]) and np.array_equal(res["sb3_flat"][4], res["row"][4])
    assert int(res["row"][4].sum()) == n                  # every env terminated once inside the rollout ...
    assert len(res["row"][6]) == 1 and len(res["sb3_flat"][6]) == 1      # ... and once more in the per-step part (step 278)
    (dr, fr), (df, ff) = res["row"][6][0], res["sb3_flat"][6][0]
    assert bool(dr.all()) and torch.equal(dr, df)
    assert torch.equal(ff, flat(fr))                      # terminal observations in the flat layout


@pytest.mark.gpu
def test_rollout_info_stream_equals_reference_infos():
    """§8(f) rank 4: the 24 `_get_info` fields of every step recorded on the device ([T][N][24]) == the info dicts of the
    unmodified reference (golden fixture), and the `stats` table Postprocessing.test_performance builds from them
    (src/rl_utils.py:528-565: key order of stats_names, Meth_Action as its index, zero rows at terminated steps)."""
    from rl_ptg_amd.vec_env import stats_table, INFO_KEYS
    from rl_ptg_amd.config import STATS_NAMES
    case = next(c for c in H.TRAJ_CASES if "infos" in H.load_traj(c)[0])
    tr, eng = H.make_engine(case, "float64", obs_layout="row")
    eng.reset()
    obs, rew, done, info = eng.rollout_info(tr["actions"])
    eng.sync()
    K, n = tr["actions"].shape
    assert tuple(info.shape) == (K, n, 24) and len(INFO_KEYS) == len(STATS_NAMES) == 24
    np.testing.assert_allclose(info.cpu().numpy(), tr["infos"], rtol=RTOL64, atol=ATOL64)
    assert np.array_equal(done.cpu().numpy(), tr["done"])
    np.testing.assert_allclose(rew.cpu().numpy(), tr["f64s"][:, :, 0], rtol=RTOL64, atol=ATOL64)
    tab = stats_table(info[:, 0], done[:, 0])
    assert list(tab) == list(STATS_NAMES)
    exp = tr["infos"][:, 0].copy()
    exp[tr["done"][:, 0].astype(bool)] = 0.0
    for m, nme in enumerate(STATS_NAMES):
        np.testing.assert_allclose(tab[nme], exp[:, m], rtol=RTOL64, atol=ATOL64)
    eng.close()


@pytest.mark.gpu
def test_full_episode_rollout_hot_equals_generic_on_device():
    """One whole 32-day episode and beyond (4 700 steps, termination + reset of the synchronised batch at step 4 602) at
    N = 16 384 in ONE ptg_rollout call: the hot route (9 + 1 launches of the fused kernel around one generic terminating step) and
    the all-generic route write bit-identical observations, rewards and done flags (compared on the device, 2 x 10.8 GB) and
    leave identical state, returns and finished-episode lists."""
    import os
    import torch
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    from rl_ptg_amd.synthetic import sticky_actions_device
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
    n, K = 16384, 4700
    acts = sticky_actions_device(K, n, seed=77, device=torch.device("cuda", 0))
    out = {}
    for route in ("hot", "generic"):
        if route == "generic":
            os.environ["PTG_NO_HOT_KERNELS"] = "1"
        try:
            eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="feature")
            eng.set_episode_plan(spec.eps_ind, n, n)
            eng.set_noise_rng(123)
            eng.reset()
            if route == "hot":
                assert eng.rollout_launches(K) == 9 + 1 + 1           # ceil(4602 / 512) hot launches (64-env workgroups), the terminating step, 97 more steps
            o, r, d = eng.rollout(acts)
            eng.sync()
            fin = eng.finished_episodes()
            st = {f: eng.get_state(f) for f in INT_FIELDS + ["cum_rew", "noise_count", "act_ep_d", "ep_ptr"]}
            out[route] = (o, r, d, fin, st)
            eng.close()
        finally:
            os.environ.pop("PTG_NO_HOT_KERNELS", None)
    h, g = out["hot"], out["generic"]
    assert torch.equal(h[0], g[0]) and torch.equal(h[1], g[1]) and torch.equal(h[2], g[2])
    assert int(h[2].sum()) == n and int(h[2][4602].sum()) == n       # everybody terminates at k = eps_sim_steps - 6
    assert sorted(zip(h[3][2].tolist(), h[3][0].tolist())) == sorted(zip(g[3][2].tolist(), g[3][0].tolist())) and len(h[3][0]) == n
    for f, v in g[4].items():
        assert np.array_equal(h[4][f], v), f


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_fused_rollout_every_short_length():
    """Rollout lengths 1..9 (the producer / consumer loops' prologue, peeled iteration, paired body and tail in every
    combination; every wave must pass the same number of hand-off barriers) == per-step launches, bit for bit."""
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=8)
    n = 300
    rng = np.random.default_rng(31)
    a = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
    b = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
    for e in (a, b):
        e.set_episode_plan(spec.eps_ind, n, n)
        e.set_noise_rng(6)
        e.reset()
    for T in list(range(1, 10)) + [17, 2, 1]:
        acts = rng.integers(0, 5, (T, n)).astype(np.int32)
        o, r, d = a.rollout(acts)
        a.sync()
        for t in range(T):
            so, sr, sd = b.step(acts[t])
            b.sync()
            assert np.array_equal(o[t].cpu().numpy(), so.cpu().numpy()) and np.array_equal(r[t].cpu().numpy(), sr.cpu().numpy()), (T, t)
    for f in INT_FIELDS + ["cum_rew", "noise_count"]:
        assert np.array_equal(a.get_state(f), b.get_state(f)), f
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["step_hot", "rollout_hot", "generic"])
def test_price_index_past_the_series_is_an_error_not_an_out_of_bounds_read(route):
    """An episode offset that pushes the 13-hour price window past the end of the series (reference: IndexError at
    :446-447) is flagged as PTG_E_RANGE at the next sync by every kernel family; the kernels clamp the index instead of
    reading out of bounds, and the other envs are unaffected."""
    import os
    from rl_ptg_amd.engine import HipEngine, PtgError
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
    n = 130
    env = {"PTG_NO_HOT_KERNELS": "1"} if route == "generic" else {}
    os.environ.update(env)
    try:
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="feature")
        eng.set_episode_plan(spec.eps_ind, n, n)
        eng.set_noise_rng(1)
        eng.reset()
        d = eng.get_state("act_ep_d")
        d[77] = 10_000                                       # day 10 000 of a 38-day trace
        eng.set_state("act_ep_d", d)
        acts = np.full((8, n), 2, np.int32)
        if route == "rollout_hot":
            eng.rollout(acts)
        else:
            eng.step(acts[0])
        with pytest.raises(PtgError) as ei:
            eng.sync()
        assert ei.value.code == -4
        eng.step(acts[1])                                    # the flag was cleared by the sync that reported it ...
        with pytest.raises(PtgError):
            eng.sync()                                       # ... and the env is still out of range on the next step
        eng.close()
    finally:
        for k in env:
            os.environ.pop(k, None)


@pytest.mark.gpu
@pytest.mark.parametrize("noise", ["rng", "tape"])
def test_checkpoint_resume_continues_bit_identically(noise):
    """state_dict() after 120 steps -> a fresh engine -> load_state_dict(): the next 100 steps (fused, hot path: equal step
    counts mark the batch as synchronised again) and the reward normaliser continue exactly where the first engine would."""
    import torch
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=3, operation="OP2", eps_len_d=2, train_steps=400000, state_change_penalty=0.1)
    n = 320
    rng = np.random.default_rng(8)
    acts = rng.integers(0, 5, (220, n)).astype(np.int32)

    def make():
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
        eng.set_episode_plan(spec.eps_ind, n, n)
        if noise == "rng":
            eng.set_noise_rng(77)
        else:
            eng.fill_noise_tape(seed=77, per_env_len=64)
        eng.reset()
        eng.vn_init()
        return eng

    a = make()
    _, r, d = a.rollout(acts[:120])
    a.vn_normalize(r, d)
    sd = a.state_dict()
    oa, ra, da = a.rollout(acts[120:])
    na = a.vn_normalize(ra, da)
    b = make()
    b.load_state_dict(sd)
    assert b.rollout_launches(100) == 1                    # the hot fused kernel, not 100 generic launches
    ob, rb, db = b.rollout(acts[120:])
    nb = b.vn_normalize(rb, db)
    a.sync(); b.sync()
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(na, nb)
    for f in INT_FIELDS + ["cum_rew", "noise_count", "n_state_changes", "ep_ptr", "act_ep_d"]:
        assert np.array_equal(a.get_state(f), b.get_state(f)), f
    assert a.vn_get()[0] == b.vn_get()[0]
    a.close(); b.close()
