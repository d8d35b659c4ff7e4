"""TEST INFRASTRUCTURE -- independent NumPy restatement of what Stable-Baselines3 does to the reference's Dict observation before the
policy network sees it (the checker for the PTG_OBS_SB3_FLAT row layout; nothing under rl_ptg_amd/ imports this file).

Reference consumer: the reference trains every agent with policy "MultiInputPolicy" (src/rl_config_agent.py:126-149), whose features
extractor is SB3's CombinedExtractor.  stable-baselines3 is a dependency that is NOT vendored in /root/reference and not installable
here (requirements.txt pins stable-baselines3==2.0.0a13), so its published algorithm is restated:
  * common/preprocessing.py::preprocess_obs  -- Box: obs.float();  Discrete(n): one_hot(obs.long(), n).float();  Dict: per key;
  * common/torch_layers.py::CombinedExtractor -- for key, subspace in observation_space.spaces.items(): nn.Flatten() each,
    torch.cat(encoded, dim=1);  gymnasium.spaces.Dict orders its sub-spaces by sorted key (spaces/dict.py: `sorted(spaces.items())`
    for a plain dict argument, which is what the reference passes, env/ptg_gym_env.py:166-204).
Parity of this restatement against SB3 itself is UNPINNED (no SB3 here); it is pinned only by the hand-built known answers in
tests/test_host_logic.py::test_sb3_flat_oracle_known_answers.
"""
import numpy as np

N_STATUS = 6        # METH_STATUS = Discrete(6) (env/ptg_gym_env.py:170,190)


def reference_keys(raw_modified, price_ahead=13):
    """(key, width) in the INSERTION order of PTGEnv._get_obs (env/ptg_gym_env.py:219-249) = the canonical column order."""
    P = price_ahead
    market = [("Elec_Price", P), ("Gas_Price", 2), ("EUA_Price", 2)] if raw_modified == "raw" else [("Pot_Reward", P), ("Part_Full", P)]
    tail = ["METH_STATUS", "T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow", "Elec_Heating",
            "Temp_hour_enc_sin", "Temp_hour_enc_cos"]
    return market + [(k, 1) for k in tail]


def flatten_rows(rows, raw_modified, price_ahead=13):
    """rows [N, F] in canonical column order -> [N, F + 5] as CombinedExtractor concatenates them, float32 like SB3's tensors."""
    rows = np.asarray(rows)
    out, col = {}, 0
    for key, w in reference_keys(raw_modified, price_ahead):
        x = rows[:, col:col + w]
        if key == "METH_STATUS":
            idx = np.rint(x[:, 0]).astype(np.int64)
            x = (idx[:, None] == np.arange(N_STATUS)[None, :])
        out[key] = x.astype(np.float32)
        col += w
    assert col == rows.shape[1]
    return np.concatenate([out[k] for k in sorted(out)], axis=1)
