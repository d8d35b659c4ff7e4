"""The 17 methanation process tables per load level (reference: data/OP1, data/OP2; src/rl_utils.py:46-67,108-113).

Packaged as rl_ptg_amd/data/tables_<OP>.npz (float64 [rows, 7]: t, T_cat, n_h2, n_ch4, n_h2_res, m_h2o, P_el, exactly
the values the reference's pandas loader yields).  `import_op_tables_from_csv` reads a reference-style data directory.
"""
import os

import numpy as np

from .engine import TABLE_KEYS

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
CSV_NAMES = {
    "startup_cold": "data-meth_startup_cold.csv", "startup_hot": "data-meth_startup_hot.csv",
    "cooldown": "data-meth_cooldown.csv", "standby_down": "data-meth_standby_down.csv",
    "standby_up": "data-meth_standby_up.csv", "op1_start_p": "data-meth_op1_start_p.csv",
    "op2_start_f": "data-meth_op2_start_f.csv", "op3_p_f": "data-meth_op3_p_f.csv",
    "op4_p_f_p_5": "data-meth_op4_p_f_p_5.csv", "op5_p_f_p_10": "data-meth_op5_p_f_p_10.csv",
    "op6_p_f_p_15": "data-meth_op6_p_f_p_15.csv", "op7_p_f_p_22": "data-meth_op7_p_f_p_20.csv",
    "op8_f_p": "data-meth_op8_f_p.csv", "op9_f_p_f_5": "data-meth_op9_f_p_f_5.csv",
    "op10_f_p_f_10": "data-meth_op10_f_p_f_10.csv", "op11_f_p_f_15": "data-meth_op11_f_p_f_15.csv",
    "op12_f_p_f_20": "data-meth_op12_f_p_f_20.csv",
}

_cache = {}


def load_op_tables(operation):
    if operation not in _cache:
        z = np.load(os.path.join(DATA, f"tables_{operation}.npz"), allow_pickle=False)
        _cache[operation] = {k: np.ascontiguousarray(z[k], dtype=np.float64) for k in TABLE_KEYS}
    return _cache[operation]


def import_op_tables_from_csv(directory):
    """`directory` holds the 17 ';'-separated files of one load level (header + 7 columns)."""
    out = {}
    for k in TABLE_KEYS:
        a = np.loadtxt(os.path.join(directory, CSV_NAMES[k]), delimiter=";", skiprows=1, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != 7:
            raise ValueError(f"{CSV_NAMES[k]}: expected 7 columns")
        out[k] = np.ascontiguousarray(a)
    return out
