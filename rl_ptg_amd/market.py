"""Market and process data import: the reference's file loaders for the env's input contract.

Restates, without pandas, what src/rl_utils.py does to the files under data/:
  * import_market_data  (:21-43)   "Time;<series>" CSVs, ';' separated, Time as %d-%m-%Y %H:%M; electricity and gas prices arrive in
                                    Euro/MWh and are divided by 10 (ct/kWh), EUA prices in Euro/t stay as they are;
  * import_data         (:46-67)   the 17 process tables (7 named columns -> [rows, 7]);
  * load_data           (:70-144)  all splits, the size warning, business-scenario overrides, reward-level values, train_len_d checks.
Numbers are parsed with Python's float() (correctly rounded); the reference's pandas parser gives the same doubles for these files
(tests/test_prep.py::test_market_csv_import_matches_reference_loader, bit-exact against tests/golden/market_real.npz).
"""
import csv
import os
import warnings
from datetime import datetime

import numpy as np

from .engine import TABLE_KEYS
from .prep import apply_scenario_overrides, check_episode_length
from .tables import CSV_NAMES

MARKET_COLUMNS = {"el": ("Day-Ahead-price [Euro/MWh]", 10.0), "gas": ("THE_DA_Gas [Euro/MWh]", 10.0), "eua": ("EUA_CO2 [Euro/t]", 1.0)}
MARKET_FILES = {(k, s): f"data/spot_market_data/data-day-ahead-{k}-{s}.csv" for k in ("el", "gas", "eua") for s in ("train", "val", "test")}
OP_COLUMNS = ["Time [s]", "T_cat [gradC]", "n_h2 [mol/s]", "n_ch4 [mol/s]", "n_h2_res [mol/s]", "m_DE [kg/h]", "Pel [W]"]


def _read_columns(file_path, wanted):
    with open(file_path, newline="") as f:
        rd = csv.reader(f, delimiter=";")
        header = [h.strip() for h in next(rd)]
        missing = [c for c in wanted if c not in header]
        if missing:
            raise KeyError(missing[0])                      # pandas: KeyError on df[<column>]
        idx = [header.index(c) for c in wanted]
        rows = [r for r in rd if r and any(x.strip() for x in r)]
    return header, rows, idx


def import_market_data(csvfile: str, type: str, path: str):
    """Same signature and result as the reference's import_market_data (src/rl_utils.py:21-43): float64 array of the series in
    `path + "/" + csvfile`; type 'el' | 'gas' | 'eua'."""
    if type not in MARKET_COLUMNS:
        assert False, "Invalid market data type. Must be one of ['el', 'gas', 'eua']!"
    col, div = MARKET_COLUMNS[type]
    header, rows, idx = _read_columns(path + "/" + csvfile, ["Time", col])
    out = np.empty(len(rows), dtype=np.float64)
    for q, r in enumerate(rows):
        datetime.strptime(r[idx[0]].strip(), "%d-%m-%Y %H:%M")      # the reference parses (and so validates) the Time column
        out[q] = float(r[idx[1]])
    return out / div if div != 1.0 else out


def import_data(csvfile: str, path: str):
    """src/rl_utils.py:46-67: one process table -> float64 [rows, 7] = t, T_cat, n_h2, n_ch4, n_h2_res, m_h2o, P_el."""
    header, rows, idx = _read_columns(path + "/" + csvfile, OP_COLUMNS)
    out = np.empty((len(rows), 7), dtype=np.float64)
    for q, r in enumerate(rows):
        for c in range(7):
            out[q, c] = float(r[idx[c]])
    return out


def load_data(cfg, path, market_files=None, op_dir=None):
    """The reference's load_data (src/rl_utils.py:70-144) for an rl_ptg_amd.config.EnvConfig: returns (dict_price_data, dict_op_data)
    with the reference's keys -- el_price_/gas_price_/eua_price_ x train/val/test (+ *_reward_level) and the 17 table names -- and sets
    cfg.train_len_d.  `path` = the RL_PtG project directory; market_files {(kind, split): relative path} and op_dir default to the
    reference's layout (config/config_env.yaml:37-73: data/spot_market_data/..., data/<operation>/...)."""
    files = dict(MARKET_FILES)
    files.update(market_files or {})
    price = {f"{k}_price_{s}": import_market_data(files[(k, s)].lstrip("/"), k, path) for k in ("el", "gas", "eua") for s in ("train", "val", "test")}
    for s in ("train", "val", "test"):                      # :94-105
        el_h = len(price[f"el_price_{s}"])
        sizes = {el_h // 24, len(price[f"gas_price_{s}"]), len(price[f"eua_price_{s}"])}
        if len(sizes) > 1:
            warnings.warn(f"Market data size does not match for {s}: electricity ({el_h}h = {el_h // 24}d), gas ({len(price[f'gas_price_{s}'])}d), "
                          f"and EUA ({len(price[f'eua_price_{s}'])}d)! -> Check size!", UserWarning)
    op_dir = os.path.join("data", cfg.operation) if op_dir is None else op_dir
    ops = {k: import_data(os.path.join(op_dir, CSV_NAMES[k]), path) for k in TABLE_KEYS}
    over = apply_scenario_overrides({f"{k}_{s}": price[f"{k}_price_{s}"] for k in ("el", "gas", "eua") for s in ("train", "val", "test")}, cfg)   # :119-126
    for k in ("el", "gas", "eua"):
        for s in ("train", "val", "test"):
            price[f"{k}_price_{s}"] = over[f"{k}_{s}"]
    price.update({f"{k}_reward_level": cfg.r_0_values[k] for k in ("el_price", "gas_price", "eua_price")})       # :129-130
    cfg.train_len_d = check_episode_length(len(price["gas_price_train"]), cfg.eps_len_d)                         # :133-142
    return price, ops


def prices_for_preprocessing(dict_price_data):
    """load_data's dict -> the {el_train, gas_train, ...} form rl_ptg_amd.prep.Preprocessing takes."""
    return {f"{k}_{s}": np.asarray(dict_price_data[f"{k}_price_{s}"], dtype=np.float64) for k in ("el", "gas", "eua") for s in ("train", "val", "test")}
