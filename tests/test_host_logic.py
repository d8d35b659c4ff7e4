"""Host-side logic that needs no GPU: spaces, observation column map, SB3 feature flattening, EnvSpec validation."""
import numpy as np
import pytest
import torch

from rl_ptg_amd.prep import EnvSpec, synthetic_spec
from rl_ptg_amd.spaces import make_spaces, obs_columns
from rl_ptg_amd.vec_env import sb3_flat_features


def test_spaces_match_reference_declaration():
    for rm, n_keys, F in (("mod", 11, 35), ("raw", 12, 26)):
        obs_space, act_space = make_spaces(rm, "discrete")
        assert len(obs_space.spaces) == n_keys and list(obs_space.spaces) == sorted(obs_space.spaces)
        assert obs_space["METH_STATUS"].n == 6 and obs_space["T_CAT"].shape == (1,) and obs_space["T_CAT"].dtype == np.float64
        assert act_space.n == 5
        cols, width = obs_columns(rm)
        assert width == F and sorted(cols) == sorted(obs_space.spaces)
        assert sum(sl.stop - sl.start for sl in cols.values()) == F
    _, box = make_spaces("mod", "continuous")
    assert box.shape == (1,) and box.dtype == np.float32 and float(box.low[0]) == -1.0 and float(box.high[0]) == 1.0
    with pytest.raises(AssertionError):
        make_spaces("both", "discrete")
    with pytest.raises(AssertionError):
        make_spaces("mod", "hybrid")


def test_sb3_flat_features_order_and_one_hot():
    rng = np.random.default_rng(0)
    n = 7
    obs = rng.random((n, 35)).astype(np.float32)
    obs[:, 26] = rng.integers(0, 5, n)
    flat = sb3_flat_features(torch.from_numpy(obs), "mod").numpy()
    assert flat.shape == (n, 40)
    cols, _ = obs_columns("mod")
    # sorted keys: CH4_syn, Elec_Heating, H2O_DE, H2_in, H2_res, METH_STATUS(6), Part_Full(13), Pot_Reward(13), T_CAT, cos, sin
    assert np.array_equal(flat[:, 0], obs[:, cols["CH4_syn_MolarFlow"].start])
    onehot = flat[:, 5:11]
    assert np.array_equal(onehot.argmax(1), obs[:, 26].astype(int)) and np.all(onehot.sum(1) == 1)
    assert np.array_equal(flat[:, 11:24], obs[:, cols["Part_Full"]]) and np.array_equal(flat[:, 24:37], obs[:, cols["Pot_Reward"]])
    assert np.array_equal(flat[:, 38], obs[:, cols["Temp_hour_enc_cos"].start]) and np.array_equal(flat[:, 39], obs[:, cols["Temp_hour_enc_sin"].start])
    fm = sb3_flat_features(torch.from_numpy(np.ascontiguousarray(obs.T)), "mod", feature_major=True).numpy()
    assert np.array_equal(fm, flat)
    raw = sb3_flat_features(torch.zeros((3, 26)), "raw")
    assert raw.shape == (3, 31)


def test_env_spec_validation_and_merge():
    spec, pre = synthetic_spec(scenario=2, operation="OP2", eps_len_d=32)
    assert spec.consts["raw_modified"] == 1 and spec.consts["action_type"] == 0 and spec.consts["eps_sim_steps"] == 4608
    assert spec.eps_ind is not None and len(spec.eps_ind) == 3250 and not spec.eps_ind.any()
    assert len(spec.markets[0]["el"]) == 38 * 24 and len(spec.markets[0]["gas"]) == 38
    kw = pre.dict_env_kwargs("val")
    assert kw["eps_ind"] is None and kw["state_change_penalty"] == 0.0            # validation envs: offset 0, no penalty (:381-385)
    bad = dict(pre.dict_env_kwargs("train"), raw_modified="both")
    with pytest.raises(AssertionError):
        EnvSpec.from_dict_input(bad)
    bad = dict(pre.dict_env_kwargs("train"), ptg_standby=3)
    with pytest.raises(ValueError):
        EnvSpec.from_dict_input(bad)
    with pytest.raises(ValueError):
        pre.dict_env_kwargs("holdout")
    s1, _ = synthetic_spec(scenario=1, operation="OP2", eps_len_d=32)
    s3, _ = synthetic_spec(scenario=3, operation="OP2", eps_len_d=32)
    merged = EnvSpec.merge_scenarios([s1, spec, s3])
    assert [m["scenario"] for m in merged.markets] == [1, 2, 3]
    assert np.all(merged.markets[1]["gas"] == 15.0) and not merged.markets[2]["gas"].any() and not merged.markets[2]["eua"].any()


def test_sticky_action_tape_holds_actions_between_switches():
    """Synthetic workload generator: actions are uniform over {0..4} and change only at switch times drawn
    with probability p_switch per step; p_switch = 1 is the i.i.d. tape."""
    import torch
    from rl_ptg_amd.synthetic import sticky_actions_device
    a = sticky_actions_device(3000, 257, seed=3, device=torch.device("cpu"), p_switch=1.0 / 12.0).numpy()
    assert a.shape == (3000, 257) and a.dtype == np.int32 and a.min() == 0 and a.max() == 4
    changes = (a[1:] != a[:-1]).mean()
    assert abs(changes - (1.0 / 12.0) * 0.8) < 0.004          # a switch redraws the same action one time in five
    assert np.allclose(np.bincount(a.ravel(), minlength=5) / a.size, 0.2, atol=0.02)
    b = sticky_actions_device(3000, 257, seed=3, device=torch.device("cpu"), p_switch=1.0 / 12.0).numpy()
    assert np.array_equal(a, b)                                # reproducible
    iid = sticky_actions_device(400, 64, seed=1, device=torch.device("cpu"), p_switch=1.0).numpy()
    assert abs((iid[1:] != iid[:-1]).mean() - 0.8) < 0.02
    assert sticky_actions_device(0, 5, seed=1, device=torch.device("cpu")).shape == (0, 5)


def test_sb3_flat_oracle_known_answers():
    """oracle/sb3_flat_oracle.py against hand-built rows: sub-spaces in sorted-key order, METH_STATUS as a one-hot of 6 (SB3
    preprocess_obs + CombinedExtractor, restated; unpinned against SB3 itself)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import sb3_flat_oracle as fo
    # 'mod', canonical order: Pot_Reward[13] Part_Full[13] METH_STATUS T_CAT H2_in CH4_syn H2_res H2O_DE Elec_Heating sin cos
    row = np.concatenate([100 + np.arange(13), 200 + np.arange(13), [4], [0.31, 0.32, 0.33, 0.34, 0.35, 0.36, 0.37, 0.38]]).astype(np.float64)
    flat = fo.flatten_rows(row[None, :], "mod")[0]
    # sorted keys: CH4_syn, Elec_Heating, H2O_DE, H2_in, H2_res, METH_STATUS(6), Part_Full(13), Pot_Reward(13), T_CAT, cos, sin
    want = np.concatenate([[0.33, 0.36, 0.35, 0.32, 0.34], [0, 0, 0, 0, 1, 0], 200 + np.arange(13), 100 + np.arange(13), [0.31, 0.38, 0.37]]).astype(np.float32)
    assert flat.dtype == np.float32 and flat.shape == (40,) and np.array_equal(flat, want)
    # 'raw': Elec_Price[13] Gas_Price[2] EUA_Price[2] METH_STATUS ...
    row = np.concatenate([100 + np.arange(13), [21, 22], [31, 32], [0], [0.31, 0.32, 0.33, 0.34, 0.35, 0.36, 0.37, 0.38]]).astype(np.float64)
    flat = fo.flatten_rows(row[None, :], "raw")[0]
    # sorted: CH4_syn, EUA_Price(2), Elec_Heating, Elec_Price(13), Gas_Price(2), H2O_DE, H2_in, H2_res, METH_STATUS(6), T_CAT, cos, sin
    want = np.concatenate([[0.33], [31, 32], [0.36], 100 + np.arange(13), [21, 22], [0.35, 0.32, 0.34], [1, 0, 0, 0, 0, 0], [0.31, 0.38, 0.37]]).astype(np.float32)
    assert flat.shape == (31,) and np.array_equal(flat, want)
    assert sorted(k for k, _ in fo.reference_keys("mod"))[5] == "METH_STATUS"


def test_split_layout_columns_agree_with_the_flat_oracle():
    """rl_ptg_amd.policy_split's column tables against oracle/sb3_flat_oracle.py: the 14 env columns of a split row land where the
    flattened observation has them, and the market windows where the oracle puts Pot_Reward / Part_Full (Elec / Gas / EUA)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import sb3_flat_oracle as fo
    from rl_ptg_amd.policy_split import FLAT_MOD, FLAT_RAW, SPLIT_ENV, env_columns, flat_rows_from_split
    for rm, table in (("mod", FLAT_MOD), ("raw", FLAT_RAW)):
        keys = dict(fo.reference_keys(rm))
        off, c = {}, 0
        for k in sorted(keys):
            off[k] = c
            c += 6 if k == "METH_STATUS" else keys[k]
        assert off == table
        assert env_columns(rm) == [off["METH_STATUS"] + j for j in range(6)] + [off[k] for k in SPLIT_ENV]
    # a hand-built split row + series -> the flat row the oracle builds from the canonical row
    fa, fb = np.arange(100, 160, dtype=np.float32), np.arange(200, 260, dtype=np.float32)
    canon = np.concatenate([fa[7:20], fb[7:20], [3], [0.31, 0.32, 0.33, 0.34, 0.35, 0.36, 0.37, 0.38]]).astype(np.float64)
    split = np.array([[0, 0, 0, 1, 0, 0, 0.31, 0.32, 0.33, 0.34, 0.35, 0.36, 0.37, 0.38, 7, 0]], dtype=np.float32)
    got = flat_rows_from_split(split, {"featA": fa[None], "featB": fb[None], "gas_n": np.zeros((1, 4), np.float32), "eua_n": np.zeros((1, 4), np.float32)}, "mod")
    assert np.array_equal(got, fo.flatten_rows(canon[None], "mod"))
