"""Observation / action space objects.  Uses Gymnasium's classes when the package is installed (what SB3 expects);
otherwise a minimal stand-alone equivalent with the attributes this package and SB3-style callers read."""
import numpy as np

try:                                     # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box, Dict, Discrete  # noqa: F401
    HAVE_GYMNASIUM = True
except Exception:                        # gymnasium is not installed in the build image
    HAVE_GYMNASIUM = False

    class Space:
        def __init__(self, shape, dtype):
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                shape = np.shape(low)
            super().__init__(shape, dtype)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Discrete(Space):
        def __init__(self, n):
            super().__init__((), np.int64)
            self.n = int(n)

        def contains(self, x):
            return 0 <= int(x) < self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class Dict(Space):
        def __init__(self, spaces):
            self.spaces = dict(sorted(spaces.items()))        # Gymnasium orders Dict keys
            self.shape, self.dtype = None, None

        def keys(self):
            return self.spaces.keys()

        def items(self):
            return self.spaces.items()

        def __getitem__(self, k):
            return self.spaces[k]

        def __iter__(self):
            return iter(self.spaces)

        def __len__(self):
            return len(self.spaces)

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k!r}: {v!r}" for k, v in self.spaces.items()) + ")"


def make_spaces(raw_modified, action_type, price_ahead=13):
    """observation_space / action_space exactly as the reference declares them (env/ptg_gym_env.py:140-204)."""
    b_norm, b_enc = [0, 1], [-1, 1]
    one = dict(low=b_norm[0], high=b_norm[1], shape=(1,), dtype=np.float64)
    common = {
        "METH_STATUS": Discrete(6),
        "T_CAT": Box(**one), "H2_in_MolarFlow": Box(**one), "CH4_syn_MolarFlow": Box(**one), "H2_res_MolarFlow": Box(**one),
        "H2O_DE_MassFlow": Box(**one), "Elec_Heating": Box(**one),
        "Temp_hour_enc_sin": Box(low=b_enc[0], high=b_enc[1], shape=(1,), dtype=np.float64),
        "Temp_hour_enc_cos": Box(low=b_enc[0], high=b_enc[1], shape=(1,), dtype=np.float64),
    }
    P = price_ahead
    if raw_modified == "raw":
        market = {"Elec_Price": Box(low=b_norm[0] * np.ones((P,)), high=b_norm[1] * np.ones((P,)), dtype=np.float64),
                  "Gas_Price": Box(low=b_norm[0] * np.ones((2,)), high=b_norm[1] * np.ones((2,)), dtype=np.float64),
                  "EUA_Price": Box(low=b_norm[0] * np.ones((2,)), high=b_norm[1] * np.ones((2,)), dtype=np.float64)}
    elif raw_modified == "mod":
        market = {"Pot_Reward": Box(low=b_norm[0] * np.ones((P,)), high=b_norm[1] * np.ones((P,)), dtype=np.float64),
                  "Part_Full": Box(low=b_enc[0] * np.ones((P,)), high=b_enc[1] * np.ones((P,)), dtype=np.float64)}
    else:
        assert False, f"ptg_gym_env.py error: state design raw_modified {raw_modified} must match 'raw' or 'mod'!"
    obs_space = Dict({**market, **common})
    if action_type == "discrete":
        act_space = Discrete(5)
    elif action_type == "continuous":
        act_space = Box(low=-1, high=1, shape=(1,), dtype=np.float32)
    else:
        assert False, f"ptg_gym_env.py error: invalid action type ({action_type}) - must match ['discrete', 'continuous']!"
    return obs_space, act_space


# column slices of the flat observation matrix (reference dict insertion order, env/ptg_gym_env.py:219-249)
def obs_columns(raw_modified, price_ahead=13):
    P = price_ahead
    cols = {}
    if raw_modified == "raw":
        cols["Elec_Price"] = slice(0, P); cols["Gas_Price"] = slice(P, P + 2); cols["EUA_Price"] = slice(P + 2, P + 4)
        o = P + 4
    else:
        cols["Pot_Reward"] = slice(0, P); cols["Part_Full"] = slice(P, 2 * P)
        o = 2 * P
    names = ["METH_STATUS", "T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow",
             "Elec_Heating", "Temp_hour_enc_sin", "Temp_hour_enc_cos"]
    for q, nme in enumerate(names):
        cols[nme] = slice(o + q, o + q + 1)
    return cols, o + 9
