"""Drive the UNMODIFIED reference (/root/reference) to produce golden vectors.  ORACLE HARNESS ONLY.

This module is test infrastructure.  It runs only in the build container (the reference never
travels to the GPU box) and is imported only by tests/golden/make_golden.py.  It

  * puts a stand-in `gymnasium` / name-only `stable_baselines3` (./stubs) and /root/reference on
    sys.path, so `env/ptg_gym_env.py` and `src/rl_utils.py` import as they are;
  * builds a scratch working directory holding a copy of `config/` (with the requested edits to
    config_env.yaml) and a `data/` tree (symlinks to the reference's read-only data, or synthetic
    market CSVs), because the reference opens its YAML relative to CWD
    (src/rl_config_env.py:15, src/rl_opt.py:37) and its CSVs relative to TrainConfig.path
    (src/rl_utils.py:30,54);
  * builds the env kwargs with the reference's own load_data / Preprocessing / dict_env_kwargs
    (src/rl_utils.py:70-144,146-405);
  * steps reference PTGEnv objects with given action tapes, records the normal draws consumed at
    state changes (env/ptg_gym_env.py:585,599,621) and dumps per-step integer / float state.

The multi-env runner restates SB3 2.0.0a13 `make_vec_env` + `DummyVecEnv` + `Monitor` ordering from
memory (un-vendored, requirements.txt:5): construct envs 0..n-1, `reset(seed=seed+e)` in env order,
then per vector step call env.step in env order and reset a finished env immediately.
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile

import numpy as np
import yaml

REF = os.environ.get("PTG_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

TABLE_KEYS = ["startup_cold", "startup_hot", "cooldown", "standby_down", "standby_up",
              "op1_start_p", "op2_start_f", "op3_p_f", "op4_p_f_p_5", "op5_p_f_p_10",
              "op6_p_f_p_15", "op7_p_f_p_22", "op8_f_p", "op9_f_p_f_5", "op10_f_p_f_10",
              "op11_f_p_f_15", "op12_f_p_f_20"]
ACTIONS = ["standby", "cooldown", "startup", "partial_load", "full_load"]
OBS_ORDER = {
    "mod": ["Pot_Reward", "Part_Full", "METH_STATUS", "T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow",
            "H2_res_MolarFlow", "H2O_DE_MassFlow", "Elec_Heating", "Temp_hour_enc_sin", "Temp_hour_enc_cos"],
    "raw": ["Elec_Price", "Gas_Price", "EUA_Price", "METH_STATUS", "T_CAT", "H2_in_MolarFlow",
            "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow", "Elec_Heating",
            "Temp_hour_enc_sin", "Temp_hour_enc_cos"],
}
INFO_KEYS = ["step", "el_price_act", "gas_price_act", "eua_price_act", "Meth_State", "Meth_Action",
             "Meth_Hot_Cold", "Meth_T_cat", "Meth_H2_flow", "Meth_CH4_flow", "Meth_H2O_flow",
             "Meth_el_heating", "ch4_revenues [ct/h]", "steam_revenues [ct/h]", "o2_revenues [ct/h]",
             "eua_revenues [ct/h]", "chp_revenues [ct/h]", "elec_costs_heating [ct/h]",
             "elec_costs_electrolyzer [ct/h]", "water_costs [ct/h]", "reward [ct]", "cum_reward",
             "Pot_Reward", "Part_Full"]

_imported = {}


def _import_reference():
    if _imported:
        return _imported
    for p in (os.path.join(HERE, "stubs"), REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    import env.ptg_gym_env as ptg_mod          # noqa: E402  (the reference, unmodified)
    import src.rl_utils as rl_utils            # noqa: E402
    import src.rl_config_env as rl_config_env  # noqa: E402
    _imported.update(ptg=ptg_mod, utils=rl_utils, cfg=rl_config_env)
    return _imported


class _Cfg:
    pass


@contextlib.contextmanager
def _quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def _write_market_csv(path, times, col, values):
    with open(path, "w") as f:
        f.write(f"Time;{col}\n")
        for t, v in zip(times, values):
            f.write(f"{t};{float(v)!r}\n")


def make_workdir(env_overrides=None, synthetic_market=None):
    """Scratch dir with config/ (edited) and data/.  synthetic_market: dict split -> (el, gas, eua)
    with el in Euro/MWh hourly, gas in Euro/MWh daily, eua in Euro/t daily (CSV units)."""
    wd = tempfile.mkdtemp(prefix="ptg_ref_")
    shutil.copytree(os.path.join(REF, "config"), os.path.join(wd, "config"))
    with open(os.path.join(wd, "config", "config_env.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg.update(env_overrides or {})
    with open(os.path.join(wd, "config", "config_env.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    os.mkdir(os.path.join(wd, "data"))
    for op in ("OP1", "OP2"):
        os.symlink(os.path.join(REF, "data", op), os.path.join(wd, "data", op))
    if synthetic_market is None:
        os.symlink(os.path.join(REF, "data", "spot_market_data"), os.path.join(wd, "data", "spot_market_data"))
    else:
        import pandas as pd
        md = os.path.join(wd, "data", "spot_market_data")
        os.mkdir(md)
        for split, (el, gas, eua) in synthetic_market.items():
            th = pd.date_range("2020-07-01", periods=len(el), freq="h").strftime("%d-%m-%Y %H:%M")
            td = pd.date_range("2020-07-01", periods=len(gas), freq="D").strftime("%d-%m-%Y %H:%M")
            _write_market_csv(os.path.join(md, f"data-day-ahead-el-{split}.csv"), th, "Day-Ahead-price [Euro/MWh]", el)
            _write_market_csv(os.path.join(md, f"data-day-ahead-gas-{split}.csv"), td, "THE_DA_Gas [Euro/MWh]", gas)
            _write_market_csv(os.path.join(md, f"data-day-ahead-eua-{split}.csv"), td, "EUA_CO2 [Euro/t]", eua)
    return wd


class RefSetup:
    """Reference preprocessing products for one configuration (one scratch working dir)."""

    def __init__(self, env_overrides=None, synthetic_market=None, action_type="discrete",
                 seed_train=3654, seed_test=605, train_steps=1500000, parallel="Singleprocessing"):
        m = _import_reference()
        self.mods = m
        self.wd = make_workdir(env_overrides, synthetic_market)
        self._cwd = os.getcwd()
        os.chdir(self.wd)
        try:
            with _quiet():
                self.EnvConfig = m["cfg"].EnvConfiguration()
                tc = _Cfg()
                tc.path = self.wd
                tc.seed_train, tc.seed_test = seed_train, seed_test
                tc.train_steps = train_steps
                tc.parallel = parallel
                self.TrainConfig = tc
                ac = _Cfg()
                ac.rl_alg_hyp = {"action_type": action_type}
                self.AgentConfig = ac
                self.price, self.op = m["utils"].load_data(self.EnvConfig, tc)
                self.pre = m["utils"].Preprocessing(self.price, self.op, ac, self.EnvConfig, tc)
        finally:
            os.chdir(self._cwd)

    def kwargs(self, split="train"):
        return self.pre.dict_env_kwargs(split)

    def close(self):
        shutil.rmtree(self.wd, ignore_errors=True)


class _RecRNG:
    """Wraps the env's Generator and logs the values `normal` returned (one per state change)."""

    def __init__(self, gen, log):
        self._g, self._log = gen, log

    def normal(self, loc, scale, size=None):
        v = self._g.normal(loc, scale, size=size)
        self._log.append(float(np.asarray(v).reshape(-1)[0]))
        return v

    def __getattr__(self, name):
        return getattr(self._g, name)


def _tid(env, arr):
    for i, k in enumerate(TABLE_KEYS):
        if arr is getattr(env, k):
            return i
    raise RuntimeError("unknown table object")


def flat_obs(obs, raw_modified):
    out = []
    for k in OBS_ORDER[raw_modified]:
        out.extend(np.asarray(obs[k], dtype=np.float64).reshape(-1).tolist())
    return out


def int_state(env):
    return [int(env.Meth_State), int(env.i), int(env.j), int(env.hot_cold), _tid(env, env.standby),
            _tid(env, env.startup), _tid(env, env.partial), _tid(env, env.full), int(env.k),
            ACTIONS.index(env.current_action), int(env.act_ep_h), int(env.act_ep_d)]


INT_COLS = ["meth_state", "i", "j", "hot_cold", "standby_tid", "startup_tid", "partial_tid", "full_tid",
            "k", "current_action", "act_ep_h", "act_ep_d"]
F64_COLS = ["reward", "cum_rew", "T_cat", "H2", "CH4", "H2_res", "H2O", "el_heating"]


def f64_state(env, reward):
    return [float(reward), float(env.cum_rew), float(env.Meth_T_cat), float(env.Meth_H2_flow),
            float(env.Meth_CH4_flow), float(env.Meth_H2_res_flow), float(env.Meth_H2O_flow),
            float(env.Meth_el_heating)]


def info_row(info):
    row = []
    for k in INFO_KEYS:
        v = info[k]
        row.append(float(ACTIONS.index(v)) if k == "Meth_Action" else float(v))
    return row


def run_vector(kwargs, actions, seed, train_or_eval="train", ep_index0=0):
    """Emulate make_vec_env + DummyVecEnv over n reference envs sharing the module-global ep_index.

    actions: array [K, n] (int for discrete, float32 for continuous).
    Returns dict of arrays; per-step arrays are [K, n, ...] and describe the state AFTER the env's
    step() (and BEFORE the auto-reset); reset_* arrays describe the state after each reset.
    """
    m = _import_reference()
    ptg = m["ptg"]
    actions = np.asarray(actions)
    K, n = actions.shape
    rm = kwargs["raw_modified"]
    ptg.ep_index = ep_index0
    logs = [[] for _ in range(n)]
    envs = []
    for e in range(n):
        envs.append(ptg.PTGEnv(kwargs, train_or_eval))
    reset_obs, reset_int, reset_info = [], [], []
    for e, env in enumerate(envs):
        o, inf = env.reset(seed=seed + e)
        env._np_random = _RecRNG(env._np_random, logs[e])
        reset_obs.append(flat_obs(o, rm))
        reset_int.append(int_state(env))
        reset_info.append(info_row(inf))
    F = len(reset_obs[0])
    ints = np.zeros((K, n, len(INT_COLS)), np.int64)
    f64s = np.zeros((K, n, len(F64_COLS)), np.float64)
    obs = np.zeros((K, n, F), np.float64)
    done = np.zeros((K, n), np.uint8)
    infos = np.zeros((K, n, len(INFO_KEYS)), np.float64) if train_or_eval == "eval" else None
    n_noise = np.zeros((K, n), np.int64)          # draws consumed so far (after this step)
    post_reset_obs, post_reset_int, post_reset_at = [], [], []
    for t in range(K):
        for e, env in enumerate(envs):
            a = actions[t, e]
            if kwargs["action_type"] == "discrete":
                a = int(a)
            else:
                a = np.array([a], dtype=np.float32)
            o, r, term, trunc, inf = env.step(a)
            ints[t, e] = int_state(env)
            f64s[t, e] = f64_state(env, r)
            obs[t, e] = flat_obs(o, rm)
            done[t, e] = 1 if term else 0
            n_noise[t, e] = len(logs[e])
            if infos is not None:
                infos[t, e] = info_row(inf)
            if term:
                o2, _ = env.reset()
                post_reset_obs.append(flat_obs(o2, rm))
                post_reset_int.append(int_state(env))
                post_reset_at.append((t, e))
    L = max(1, max(len(l) for l in logs))
    noise = np.zeros((n, L), np.float64)
    for e, l in enumerate(logs):
        noise[e, :len(l)] = l
    out = dict(actions=actions, ints=ints, f64s=f64s, obs=obs, done=done, noise=noise,
               noise_len=np.array([len(l) for l in logs], np.int64), n_noise=n_noise,
               reset_obs=np.array(reset_obs), reset_int=np.array(reset_int, np.int64),
               reset_info=np.array(reset_info),
               post_reset_obs=np.array(post_reset_obs).reshape(-1, F),
               post_reset_int=np.array(post_reset_int, np.int64).reshape(-1, len(INT_COLS)),
               post_reset_at=np.array(post_reset_at, np.int64).reshape(-1, 2),
               ep_index_end=np.int64(ptg.ep_index))
    if infos is not None:
        out["infos"] = infos
    return out
