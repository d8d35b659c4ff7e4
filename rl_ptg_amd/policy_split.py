"""Consumer side of the "split" observation layout (include/ptg_env.h, PTG_OBS_SPLIT): a policy's first Linear layer evaluated
from the 16-column env rows plus a per-hour projection table, instead of from the 40-column flattened observation.

The reference's policies are SB3 "MultiInputPolicy" MLPs (src/rl_config_agent.py:126-149): CombinedExtractor flattens the Dict
observation into x (40 columns for 'mod', 31 for 'raw', sorted-key order), the first layer computes  y = W x + b.  26 ('raw': 17)
of those columns are the env's 13-hour (2-day) windows of the market feature series -- a function of the env's hour (day) index
alone.  So
    y = W_env x_env + G_hour[hour index] (+ G_day[day index]) + b,     G_hour[h] = sum_q W[:, col(q)] * series[h + q]
with G computed once per weight update for every hour of the series (a [n_hours, 13] x [13, H] product per series) -- and the env
kernel writes 64 instead of 160 bytes of observation per env-step.  torch only; nothing here is required by the env itself.
"""
import numpy as np

# flat (sorted-key) column layout of SB3's CombinedExtractor output, price_ahead = 13
FLAT_MOD = {"CH4_syn_MolarFlow": 0, "Elec_Heating": 1, "H2O_DE_MassFlow": 2, "H2_in_MolarFlow": 3, "H2_res_MolarFlow": 4, "METH_STATUS": 5,
            "Part_Full": 11, "Pot_Reward": 24, "T_CAT": 37, "Temp_hour_enc_cos": 38, "Temp_hour_enc_sin": 39}
FLAT_RAW = {"CH4_syn_MolarFlow": 0, "EUA_Price": 1, "Elec_Heating": 3, "Elec_Price": 4, "Gas_Price": 17, "H2O_DE_MassFlow": 19,
            "H2_in_MolarFlow": 20, "H2_res_MolarFlow": 21, "METH_STATUS": 22, "T_CAT": 28, "Temp_hour_enc_cos": 29, "Temp_hour_enc_sin": 30}
# split row: 0-5 METH_STATUS one-hot, then these, then 14 = hour index, 15 = day index
SPLIT_ENV = ["T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow", "Elec_Heating", "Temp_hour_enc_sin",
             "Temp_hour_enc_cos"]


def env_columns(raw_modified):
    """flat column of each of the 14 env columns of a split row"""
    f = FLAT_MOD if raw_modified == "mod" else FLAT_RAW
    return [f["METH_STATUS"] + j for j in range(6)] + [f[k] for k in SPLIT_ENV]


def flat_rows_from_split(rows, series, raw_modified):
    """Rebuild the SB3_FLAT rows from split rows [..., 16] and the feature series (HipEngine.market_feature_series()): the exact inverse of
    what the SPLIT layout leaves out (used by the tests; a policy never needs it).  NumPy or torch input; returns the same kind."""
    import torch
    is_np = isinstance(rows, np.ndarray)
    r = torch.from_numpy(rows) if is_np else rows
    mod = raw_modified == "mod"
    out = torch.zeros(r.shape[:-1] + (40 if mod else 31,), dtype=r.dtype, device=r.device)
    out[..., env_columns(raw_modified)] = r[..., :14]
    hi, di = r[..., 14].long(), r[..., 15].long()
    ar = torch.arange(13, device=r.device)
    fa = torch.as_tensor(series["featA"], device=r.device).reshape(-1)
    if mod:
        fb = torch.as_tensor(series["featB"], device=r.device).reshape(-1)
        out[..., FLAT_MOD["Pot_Reward"]:FLAT_MOD["Pot_Reward"] + 13] = fa[hi[..., None] + ar]
        out[..., FLAT_MOD["Part_Full"]:FLAT_MOD["Part_Full"] + 13] = fb[hi[..., None] + ar]
    else:
        g = torch.as_tensor(series["gas_n"], device=r.device).reshape(-1)
        u = torch.as_tensor(series["eua_n"], device=r.device).reshape(-1)
        out[..., FLAT_RAW["Elec_Price"]:FLAT_RAW["Elec_Price"] + 13] = fa[hi[..., None] + ar]
        a2 = torch.arange(2, device=r.device)
        out[..., FLAT_RAW["Gas_Price"]:FLAT_RAW["Gas_Price"] + 2] = g[di[..., None] + a2]
        out[..., FLAT_RAW["EUA_Price"]:FLAT_RAW["EUA_Price"] + 2] = u[di[..., None] + a2]
    return out.numpy() if is_np else out


class FirstLayerSplit:
    """y = W x + b of a first layer with weight W [H, 40 | 31] (flat column order), evaluated from split rows.

    prepare(W, b) builds the projection tables (call after every weight update); __call__(rows) -> [..., H]."""

    def __init__(self, series, raw_modified="mod", device=None):
        import torch
        self.mod = raw_modified == "mod"
        self.raw_modified = raw_modified
        dev = device
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a).reshape(-1), device=dev)
        self.fa = t(series["featA"])
        self.fb = t(series["featB"]) if self.mod else None
        self.g = None if self.mod else t(series["gas_n"])
        self.u = None if self.mod else t(series["eua_n"])
        self.env_cols = torch.as_tensor(env_columns(raw_modified), device=dev)

    @staticmethod
    def _windows(s, width):
        return s.unfold(0, width, 1)                      # [len - width + 1, width]: row h = s[h : h + width]

    def prepare(self, W, b=None):
        f = FLAT_MOD if self.mod else FLAT_RAW
        self.W_env = W[:, self.env_cols].t().contiguous()                  # [14, H]
        self.b = b
        if self.mod:
            self.G_hour = (self._windows(self.fa, 13) @ W[:, f["Pot_Reward"]:f["Pot_Reward"] + 13].t()
                           + self._windows(self.fb, 13) @ W[:, f["Part_Full"]:f["Part_Full"] + 13].t())
            self.G_day = None
        else:
            self.G_hour = self._windows(self.fa, 13) @ W[:, f["Elec_Price"]:f["Elec_Price"] + 13].t()
            self.G_day = (self._windows(self.g, 2) @ W[:, f["Gas_Price"]:f["Gas_Price"] + 2].t()
                          + self._windows(self.u, 2) @ W[:, f["EUA_Price"]:f["EUA_Price"] + 2].t())
        return self

    def __call__(self, rows):
        y = rows[..., :14] @ self.W_env + self.G_hour[rows[..., 14].long()]
        if self.G_day is not None:
            y = y + self.G_day[rows[..., 15].long()]
        return y if self.b is None else y + self.b
