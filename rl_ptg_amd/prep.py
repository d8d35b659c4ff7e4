"""Host-side input preparation: what the reference computes once before training and hands to the env.

Restates (vectorised, same float64 operand order, checked bit-for-bit against tests/golden/prep_*.npz):
  * scenario price overrides and episode-length checks of load_data        src/rl_utils.py:119-142
  * calculate_optimum: potential reward / load identifier / T-OPT          src/rl_opt.py:26-152
  * define_episodes + rand_eps_ind                                          src/rl_utils.py:283-335
  * dict_env_kwargs: the env's input contract                               src/rl_utils.py:337-405
The 13-wide `e_r_b` and 2-wide `g_e` tensors of preprocessing_array (:243-281) are NOT materialised: the kernels
read the 1-D series (e_r_b[c, i, t] == series_c[t + i]).  `EnvSpec.from_dict_input` also accepts a reference-made
kwargs dict (with e_r_b / g_e) and folds it back, so the reference's own Preprocessing output drops in unchanged.
"""
import math

import numpy as np

from .config import KWARG_KEYS, EnvConfig
from .engine import TABLE_KEYS

SPLITS = ("train", "val", "test")


def apply_scenario_overrides(prices, cfg):
    """prices: {"el_train": ..., "gas_train": ..., "eua_train": ..., *_val, *_test} (ct/kWh, ct/kWh, Euro/t).
    Returns a new dict with the business-scenario overrides of src/rl_utils.py:119-126 applied."""
    out = {k: np.asarray(v, dtype=np.float64) for k, v in prices.items()}
    if cfg.scenario in (2, 3):
        gas_price = cfg.ch4_price_fix if cfg.scenario == 2 else 0
        for s in SPLITS:
            out[f"gas_{s}"] = np.full(len(out[f"gas_{s}"]), gas_price, dtype=np.float64)
        if cfg.scenario == 3:
            for s in SPLITS:
                out[f"eua_{s}"] = np.zeros(len(out[f"eua_{s}"]))
    return out


def check_episode_length(n_gas_train_days, eps_len_d):
    """src/rl_utils.py:133-142 -> train_len_d"""
    min_train_len = 6
    train_len_d = n_gas_train_days - min_train_len
    assert train_len_d > 0, f"The training set size must be greater than {min_train_len} days"
    if train_len_d % eps_len_d != 0:
        divisors = [i for i in range(1, train_len_d + 1) if train_len_d % i == 0]
        assert False, (f"The training set size {train_len_d} must be divisible by the episode length "
                       f"eps_len_d : {eps_len_d}; Possible divisors are: {divisors}")
    assert train_len_d >= eps_len_d
    return train_len_d


def _pem_efficiency(load_fraction, cfg):
    """LHV efficiency of the PEM electrolyzer at a load fraction (regression polynomial of env/ptg_gym_env.py:311-317,
    src/rl_opt.py:84-90; Python float arithmetic incl. its `**`, so the value is bit-identical to the reference's)."""
    x = load_fraction
    if x < cfg.min_load_electrolyzer:
        return 0.02
    return (0.598 - 0.325 * x ** 2 + 0.218 * x ** 3 +
            0.01 * x ** (-1) - 1.68 * 10 ** (-3) * x ** (-2) +
            2.51 * 10 ** (-5) * x ** (-3))


def _steady_state_terms(cfg, level):
    """Price-independent factors of the hourly margin at one steady-state load level (1 = partial, 2 = full load), each a
    Python float built with the operand order of src/rl_opt.py:57-96 so that products with the price arrays round alike."""
    ms = cfg.meth_stats_load
    mol = cfg.convert_mol_to_Nm3
    b_s3 = 1 if cfg.scenario == 3 else 0
    n_ch4, n_h2res, n_h2 = ms["Meth_CH4_flow"][level], ms["Meth_H2_res_flow"][level], ms["Meth_H2_flow"][level]
    m_h2o, p_heat = ms["Meth_H2O_flow"][level], ms["Meth_el_heating"][level]
    q_ch4 = n_ch4 * mol * cfg.H_u_CH4 * 1000                       # kW_th of methane
    q_h2res = n_h2res * mol * cfg.H_u_H2 * 1000                    # kW_th of residual hydrogen
    p_chp = q_ch4 * cfg.eta_CHP * b_s3                             # kW_el of the CHP plant (scenario 3)
    q_chp = q_ch4 * (1 - cfg.eta_CHP) * b_s3
    q_steam = m_h2o * (cfg.dt_water * cfg.cp_water + cfg.h_H2O_evap) / 3600
    v_h2 = n_h2 * mol                                              # Nm3/s of hydrogen
    eta = _pem_efficiency(v_h2 / cfg.max_h2_volumeflow, cfg)
    return dict(
        gas_coef=q_ch4 + q_h2res,                                  # x gas price              -> SNG revenues
        chp=p_chp * cfg.eeg_el_price,                              # EEG tender revenues
        steam=(q_steam + q_chp) * cfg.heat_price,
        oxygen=1 / 2 * v_h2 * 3600 * cfg.o2_price,
        eua_coef=n_ch4 * cfg.Molar_mass_CO2 / 1000 / 1000 * 3600,  # x EUA price x 100       -> EUA revenues
        heat_coef=p_heat / 1000,                                   # x electricity price      -> heating costs
        elz_coef=v_h2 * cfg.H_u_H2 * 1000 / eta,                   # x electricity price      -> electrolysis costs
        water=(m_h2o + n_h2 * cfg.Molar_mass_H2O / 1000 * 3600) / cfg.rho_water * cfg.water_price)


def calculate_optimum(el, gas, eua, cfg):
    """Potential reward (ct/h) and load identifier per hour, ignoring plant dynamics (restates src/rl_opt.py:26-152).

    Returns dict(pot_rew, part_full, cum_rew).  The reference loops over hours and the two load levels in Python; here the
    price-independent factors are scalars (`_steady_state_terms`) and the hour loop is float64 array arithmetic in the same
    association order, which reproduces the reference's numbers bit for bit (tests/test_prep.py)."""
    el = np.asarray(el, dtype=np.float64)
    gas = np.asarray(gas, dtype=np.float64)
    eua = np.asarray(eua, dtype=np.float64)
    hour = np.arange(len(el))
    day = np.minimum(hour // 24, len(gas) - 1)                    # the last partial day reuses the last daily price (:51-52)
    gas_h, eua_h = gas[day], eua[day]
    margin = []
    for level in (1, 2):
        k = _steady_state_terms(cfg, level)
        revenue_sng = k["gas_coef"] * gas_h
        revenue_eua = k["eua_coef"] * eua_h * 100
        cost_el = k["heat_coef"] * el + k["elz_coef"] * el
        margin.append(revenue_sng + k["chp"] + k["steam"] + revenue_eua + k["oxygen"] - cost_el - k["water"])
    partial, full = margin
    full_wins = full > partial                                     # Python's max() keeps the first maximum (:102-103)
    best = np.where(full_wins, full, partial)
    part_full = np.where(best > 0, full_wins.astype(np.float64), -1.0)          # -1: no profitable operation (:112-135)
    cum_rew = np.cumsum(np.where(best > 0, best, 0.0))             # sequential adds like the loop (:125,139)
    return dict(pot_rew=best, part_full=part_full, cum_rew=cum_rew)


def rand_eps_ind(seed_train, n_eps, num_loops, train_len_d, eps_len_d, overhead_factor=10):
    """Order in which training envs visit the n_eps sub-periods of the training set: `overhead_factor * loops` independent
    random permutations of 0..n_eps-1, concatenated (restates src/rl_utils.py:315-335; the reference seeds NumPy's global
    legacy generator and shuffles a row per round -- RandomState(seed).shuffle on an equally long array draws the same
    permutations).  A training set that is one single episode gives all-zero indices."""
    if train_len_d == eps_len_d:
        return np.zeros(n_eps * int(num_loops) * overhead_factor)
    rounds = (1 if num_loops < 1 else int(num_loops)) * overhead_factor
    legacy = np.random.RandomState(seed_train)
    out = np.empty((rounds, n_eps), dtype=int)
    for r in range(rounds):
        perm = np.arange(n_eps)
        legacy.shuffle(perm)
        out[r] = perm
    return out.reshape(-1)


class Preprocessing:
    """The products of the reference's Preprocessing class (src/rl_utils.py:146-405) for one configuration.

    prices: dict with el_/gas_/eua_ x train/val/test series in the loader's units (ct/kWh, ct/kWh, Euro/t), BEFORE the
    scenario overrides.  tables: dict of the 17 process tables (rl_ptg_amd.tables.load_op_tables)."""

    def __init__(self, prices, tables, cfg: EnvConfig, seed_train=3654, train_steps=1500000, action_type="discrete",
                 parallel="Singleprocessing"):
        self.cfg = cfg
        self.tables = tables
        self.action_type = action_type
        self.parallel = parallel
        self.prices = apply_scenario_overrides(prices, cfg)
        cfg.train_len_d = check_episode_length(len(self.prices["gas_train"]), cfg.eps_len_d)
        self.opt = {s: calculate_optimum(self.prices[f"el_{s}"], self.prices[f"gas_{s}"], self.prices[f"eua_{s}"], cfg)
                    for s in SPLITS}
        r0 = cfg.r_0_values
        self.r_level = calculate_optimum(r0["el_price"], r0["gas_price"], r0["eua_price"], cfg)["pot_rew"]
        P = cfg.price_ahead
        self.t_opt = {s: float(self.opt[s]["cum_rew"][-P]) for s in SPLITS}      # src/rl_opt.py:147
        # define_episodes (:283-312)
        val_len_d = len(self.prices["gas_val"]) - 1
        test_len_d = len(self.prices["gas_test"]) - 1
        self.n_eps = int(cfg.train_len_d / cfg.eps_len_d)
        self.eps_len = 24 * 3600 * cfg.eps_len_d
        self.eps_sim_steps = dict(train=int(self.eps_len / cfg.sim_step), val=int(24 * 3600 * val_len_d / cfg.sim_step),
                                  test=int(24 * 3600 * test_len_d / cfg.sim_step))
        self.num_loops = train_steps / (self.eps_sim_steps["train"] * self.n_eps)
        self.eps_ind = rand_eps_ind(seed_train, self.n_eps, self.num_loops, cfg.train_len_d, cfg.eps_len_d)
        self.n_eps_loops = self.n_eps * int(self.num_loops)
        pr = self.opt["train"]["pot_rew"]
        self.rew_l_b = float(np.min(pr[:len(pr) - P]))                             # np.min(e_r_b_train[1, 0, :]) (:378)
        self.rew_u_b = float(np.max(pr[:len(pr) - P]))

    def dict_env_kwargs(self, split="train", materialize=False):
        """Env kwargs in the reference's layout (src/rl_utils.py:337-405).  Price data as 1-D series
        (`el_series`, `pot_rew_series`, `part_full_series`, `gas_series`, `eua_series`); with materialize=True the
        reference's `e_r_b` / `g_e` tensors are built as well (needed only to feed the reference's own PTGEnv)."""
        if split not in SPLITS:
            raise ValueError(f'Invalid type: {split}. Must be "train", "val", or "test".')
        cfg = self.cfg
        kw = {f"ptg_{k}": cfg.ptg_state_space[k] for k in ["standby", "cooldown", "startup", "partial_load", "full_load"]}
        kw.update({k: getattr(cfg, k) for k in KWARG_KEYS})
        kw.update(parallel=self.parallel, n_eps_loops=self.n_eps_loops, reward_level=self.r_level, action_type=self.action_type)
        kw.update({k: self.tables[k] for k in TABLE_KEYS})
        kw.update(eps_ind=self.eps_ind if split == "train" else None,
                  state_change_penalty=cfg.state_change_penalty if split == "train" else 0.0,
                  eps_sim_steps=self.eps_sim_steps[split], rew_l_b=self.rew_l_b, rew_u_b=self.rew_u_b,
                  el_series=self.prices[f"el_{split}"], pot_rew_series=self.opt[split]["pot_rew"],
                  part_full_series=self.opt[split]["part_full"], gas_series=self.prices[f"gas_{split}"],
                  eua_series=self.prices[f"eua_{split}"])
        if materialize:
            P = cfg.price_ahead
            T = len(kw["el_series"]) - P
            e_r_b = np.zeros((3, P, T))
            for i in range(P):
                e_r_b[0, i, :] = kw["el_series"][i:i + T]
                e_r_b[1, i, :] = kw["pot_rew_series"][i:i + T]
                e_r_b[2, i, :] = kw["part_full_series"][i:i + T]
            D = len(kw["gas_series"]) - 1
            g_e = np.zeros((2, 2, D))
            g_e[0, 0], g_e[0, 1] = kw["gas_series"][:-1], kw["gas_series"][1:]
            g_e[1, 0], g_e[1, 1] = kw["eua_series"][:-1], kw["eua_series"][1:]
            kw.update(e_r_b=e_r_b, g_e=g_e)
        return kw


def _fold(a):
    """a[i, t] == s[t + i]  ->  s"""
    a = np.asarray(a, dtype=np.float64)
    return np.concatenate([a[0, :], a[1:, -1]])


class EnvSpec:
    """Everything HipEngine needs, extracted from an env kwargs dict (reference-made or from Preprocessing above)."""

    def __init__(self, consts, tables, markets, eps_ind):
        self.consts, self.tables, self.markets, self.eps_ind = consts, tables, markets, eps_ind

    @classmethod
    def from_dict_input(cls, kw, train_or_eval="train"):
        assert train_or_eval in ["train", "eval"], 'train_or_eval must be either "train" or "eval".'
        consts = {}
        for k in KWARG_KEYS:
            if k in ("scenario", "raw_modified"):
                continue
            consts[k] = kw[k]
        if kw["raw_modified"] not in ("raw", "mod"):
            raise AssertionError(f"state design raw_modified {kw['raw_modified']} must match 'raw' or 'mod'!")
        if kw["action_type"] not in ("discrete", "continuous"):
            raise AssertionError(f"invalid action type ({kw['action_type']}) - must match ['discrete', 'continuous']!")
        for name in ("standby", "cooldown", "startup", "partial_load", "full_load"):
            if kw.get(f"ptg_{name}", ["standby", "cooldown", "startup", "partial_load", "full_load"].index(name)) != \
                    ["standby", "cooldown", "startup", "partial_load", "full_load"].index(name):
                raise ValueError("ptg_state_space must map standby..full_load to 0..4 (the reference indexes a list with it)")
        consts.update(raw_modified={"raw": 0, "mod": 1}[kw["raw_modified"]],
                      action_type={"discrete": 0, "continuous": 1}[kw["action_type"]],
                      train_or_eval={"train": 0, "eval": 1}[train_or_eval],
                      eps_sim_steps=int(kw["eps_sim_steps"]), state_change_penalty=float(kw["state_change_penalty"]),
                      t_cat_initial=16.0)
        tables = {k: np.ascontiguousarray(kw[k], dtype=np.float64) for k in TABLE_KEYS}
        if "el_series" in kw:
            ser = {k: np.asarray(kw[f"{k}_series"], dtype=np.float64) for k in ("el", "pot_rew", "part_full", "gas", "eua")}
        else:
            e_r_b, g_e = kw["e_r_b"], kw["g_e"]
            ser = dict(el=_fold(e_r_b[0]), pot_rew=_fold(e_r_b[1]), part_full=_fold(e_r_b[2]), gas=_fold(g_e[0]), eua=_fold(g_e[1]))
        market = dict(ser, scenario=int(kw["scenario"]), rew_l_b=float(kw["rew_l_b"]), rew_u_b=float(kw["rew_u_b"]),
                      r_0=float(np.asarray(kw["reward_level"]).reshape(-1)[0]))
        eps_ind = kw.get("eps_ind")
        eps_ind = None if eps_ind is None else np.asarray(eps_ind, dtype=np.float64)
        return cls(consts, tables, [market], eps_ind)

    @classmethod
    def merge_scenarios(cls, specs):
        """Several single-scenario specs over the same trace -> one spec with one market set per scenario
        (BASELINE.json config 5: envs of mixed business scenarios in one batch)."""
        base = specs[0]
        return cls(base.consts, base.tables, [s.markets[0] for s in specs], base.eps_ind)


def synthetic_spec(scenario=2, operation="OP2", eps_len_d=32, raw_modified="mod", action_type="discrete",
                   train_or_eval="train", sim_step=600, seed_train=3654, train_steps=1500000, state_change_penalty=0.0):
    """The BASELINE.json workload: synthetic 38-day trace (32-day episodes), real process tables."""
    from .synthetic import synthetic_market
    from .tables import load_op_tables
    cfg = EnvConfig(scenario=scenario, operation=operation, eps_len_d=eps_len_d, raw_modified=raw_modified,
                    sim_step=sim_step, state_change_penalty=state_change_penalty)
    prices = {}
    for split, (days, seed) in dict(train=(38, 20250614), val=(9, 20250615), test=(9, 20250616)).items():
        el, gas, eua = synthetic_market(days, seed)
        prices.update({f"el_{split}": el, f"gas_{split}": gas, f"eua_{split}": eua})
    pre = Preprocessing(prices, load_op_tables(operation), cfg, seed_train=seed_train, train_steps=train_steps,
                        action_type=action_type)
    return EnvSpec.from_dict_input(pre.dict_env_kwargs("train"), train_or_eval), pre
