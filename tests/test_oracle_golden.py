"""Pins the CPU oracle (oracle/ptg_oracle.c) to golden vectors produced by the unmodified reference
(tests/golden/make_golden.py).  Everything is compared BIT-EXACTLY: the oracle keeps the reference's operand
order, libm calls and NumPy's pairwise summation."""
import numpy as np
import pytest

import helpers as H
from helpers import po


@pytest.mark.parametrize("case", H.TRAJ_CASES)
def test_trajectory_bit_exact(case):
    tr, env = H.make_oracle(case)
    K = tr["actions"].shape[0]
    obs0, info0 = env.reset()
    assert np.array_equal(obs0, tr["reset_obs"])
    assert np.array_equal(info0, tr["reset_info"])
    ints0, _ = env.state()
    assert np.array_equal(ints0, tr["reset_int"])
    n_post = 0
    post_at = [tuple(x) for x in tr["post_reset_at"].tolist()]
    for t in range(K):
        obs, rew, done, final, info = env.step(tr["actions"][t])
        li, lf = env.last()
        assert np.array_equal(li, tr["ints"][t]), f"int state differs at step {t}"
        assert np.array_equal(lf, tr["f64s"][t]), f"float state differs at step {t}"
        assert np.array_equal(done, tr["done"][t])
        assert np.array_equal(rew, tr["f64s"][t, :, 0])
        for e in range(env.n):
            if done[e]:
                assert post_at[n_post] == (t, e)
                assert np.array_equal(final[e], tr["obs"][t, e])
                assert np.array_equal(obs[e], tr["post_reset_obs"][n_post])
                si, _ = env.state()
                assert np.array_equal(si[e], tr["post_reset_int"][n_post])
                n_post += 1
            else:
                assert np.array_equal(obs[e], tr["obs"][t, e]), f"obs differs at step {t} env {e}"
        if "infos" in tr:
            assert np.array_equal(info, tr["infos"][t])
        for e in range(env.n):
            assert env.noise_count(e) == tr["n_noise"][t, e]
    assert n_post == len(post_at)
    assert env.ep_index == int(tr["ep_index_end"])


@pytest.mark.parametrize("op", ["OP1", "OP2"])
def test_get_index_all_temperatures(op):
    u = H.load_npz(f"{H.GOLD}/units_{op}.npz")
    case = "synth_bs2_op2_mod_disc_train" if op == "OP2" else "synth_bs1_op1_mod_disc_train"
    _, env = H.make_oracle(case)
    tids = {k: i for i, k in enumerate(po.TABLE_KEYS)}
    for d, name in enumerate(u["dests"].tolist()):
        got = np.array([env.get_index(tids[name], T) for T in u["T"]])
        assert np.array_equal(got, u["get_index"][d]), name


def test_pairwise_mean_matches_numpy_average():
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 30, 127, 128, 129, 300, 301, 1000, 4097):
        a = rng.normal(size=(n, 7)) * 10.0 ** rng.integers(-6, 6, size=(n, 1))
        for c in range(7):
            assert po.pairwise_mean(a[:, c]) == np.average(a[:, c])


def test_continuous_decode_edges():
    # thresholds are float64 -1 + ival*0.4; the action arrives as float32 (env/ptg_gym_env.py:147-155, :351-355)
    _, env = H.make_oracle("real_bs2_op2_raw_cont_test")
    thr = np.ones(6)
    for ival in range(6):
        thr[ival] = -1 + ival * ((1 - (-1)) / 5)
    for a in [-1.0, 1.0, np.nan, -1.5, 1.5, -0.6, -0.2, 0.2, 0.6, 0.0, 0.99999994, np.inf, -np.inf, 0.3, -0.9]:
        a32 = np.float32(a)
        expect = 2                                   # previous action kept when no threshold fires
        chk = thr > np.array([a32], dtype=np.float32)
        for ival in range(6):
            if chk[ival]:
                expect = [0, 1, 2, 3, 4][ival - 1]
                break
        assert env.decode_continuous(a32, 2) == expect, a
    assert env.decode_continuous(np.float32(-0.6), 0) == 0      # f32(-0.6) < -0.6  -> standby
    assert env.decode_continuous(np.float32(0.2), 0) == 3       # f32(0.2) > 0.20000000000000018 -> partial_load
    assert env.decode_continuous(np.float32(-1.5), 0) == 4      # below -1 wraps to actions[-1] = full_load


def test_known_answers_from_survey():
    """600 steps, actions default_rng(0).integers(0,5,600), reset(seed=3654), real BS2/OP2 train kwargs."""
    tr, consts, tables, market = H.load_traj("real_bs2_op2_mod_disc_train")
    assert market["eps_ind"][:8].astype(int).tolist() == [33, 3, 38, 0, 8, 23, 9, 5]
    prep = H.load_prep("real_bs2_OP2")["meta"]
    assert abs(prep["r_level"] - 128.86368073) < 1e-8
    assert prep["rew_l_b"] == -685.2843133116897 and prep["rew_u_b"] == 1372.5540663690715
    assert consts["max_h2_volumeflow"] == 0.001087082395 and consts["eps_sim_steps"] == 5328
    # ep_index: 3 constructions + 3 resets; env 0's first episode is eps_ind[3] = 0
    assert tr["reset_int"][0, 10] == 0 * 37 * 24 and tr["reset_int"][1, 10] == 8 * 37 * 24
    assert tr["reset_int"][0, 1] == 41218                      # argmin |cooldown.T - 16|
