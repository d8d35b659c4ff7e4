"""ctypes binding of the CPU oracle (oracle/ptg_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as
the checker / reported CPU baseline.  Nothing under rl_ptg_amd/ imports it.

Inputs mirror what the reference env receives (src/rl_utils.py:337-405 `dict_env_kwargs`):
`split_reference_kwargs` turns such a kwargs dict into (consts, tables, market) with the 13-wide `e_r_b`
and 2-wide `g_e` tensors folded back into the 1-D series they were built from (:254-281).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libptg_oracle.so")

TABLE_KEYS = ["startup_cold", "startup_hot", "cooldown", "standby_down", "standby_up",
              "op1_start_p", "op2_start_f", "op3_p_f", "op4_p_f_p_5", "op5_p_f_p_10",
              "op6_p_f_p_15", "op7_p_f_p_22", "op8_f_p", "op9_f_p_f_5", "op10_f_p_f_10",
              "op11_f_p_f_15", "op12_f_p_f_20"]
INT_COLS = ["meth_state", "i", "j", "hot_cold", "standby_tid", "startup_tid", "partial_tid", "full_tid",
            "k", "current_action", "act_ep_h", "act_ep_d"]
F64_COLS = ["reward", "cum_rew", "T_cat", "H2", "CH4", "H2_res", "H2O", "el_heating"]

_D = ["noise"]
_I1 = ["eps_len_d", "sim_step", "time_step_op", "price_ahead", "scenario"]
_D2 = ["convert_mol_to_Nm3", "H_u_CH4", "H_u_H2", "dt_water", "cp_water", "rho_water", "Molar_mass_CO2",
       "Molar_mass_H2O", "h_H2O_evap", "eeg_el_price", "heat_price", "o2_price", "water_price",
       "min_load_electrolyzer", "max_h2_volumeflow", "eta_CHP",
       "t_cat_standby", "t_cat_startup_cold", "t_cat_startup_hot"]
_I2 = ["time1_start_p_f", "time2_start_f_p", "time_p_f", "time_f_p", "time1_p_f_p", "time2_p_f_p",
       "time23_p_f_p", "time3_p_f_p", "time34_p_f_p", "time4_p_f_p", "time45_p_f_p", "time5_p_f_p",
       "time1_f_p_f", "time2_f_p_f", "time23_f_p_f", "time3_f_p_f", "time34_f_p_f", "time4_f_p_f",
       "time45_f_p_f", "time5_f_p_f", "i_fully_developed", "j_fully_developed"]
_D3 = ["el_l_b", "el_u_b", "gas_l_b", "gas_u_b", "eua_l_b", "eua_u_b", "T_l_b", "T_u_b", "h2_l_b", "h2_u_b",
       "ch4_l_b", "ch4_u_b", "h2_res_l_b", "h2_res_u_b", "h2o_l_b", "h2o_u_b", "heat_l_b", "heat_u_b",
       "rew_l_b", "rew_u_b"]
_I3 = ["raw_modified", "action_type", "train_or_eval", "eps_sim_steps"]
_D4 = ["state_change_penalty", "r_0"]
CONST_KEYS = _D + _I1 + _D2 + _I2 + _D3 + _I3 + _D4


class _Config(C.Structure):
    _fields_ = ([(k, C.c_double) for k in _D] + [(k, C.c_int32) for k in _I1] +
                [(k, C.c_double) for k in _D2] + [(k, C.c_int32) for k in _I2] +
                [(k, C.c_double) for k in _D3] + [(k, C.c_int32) for k in _I3] +
                [(k, C.c_double) for k in _D4])


class _Tables(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_double) * 17), ("rows", C.c_int32 * 17)]


class _Market(C.Structure):
    _fields_ = [("n_hours", C.c_int32), ("el", C.POINTER(C.c_double)), ("pot_rew", C.POINTER(C.c_double)),
                ("part_full", C.POINTER(C.c_double)), ("n_days", C.c_int32), ("gas", C.POINTER(C.c_double)),
                ("eua", C.POINTER(C.c_double)), ("n_eps_ind", C.c_int32), ("eps_ind", C.POINTER(C.c_double))]


def build(force=False):
    """gcc-compile the oracle in place (idempotent)."""
    src = os.path.join(HERE, "ptg_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "libptg_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        dp, i64p, u8p = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
        L.ptgo_create.argtypes = [C.POINTER(_Config), C.POINTER(_Tables), C.POINTER(_Market), C.c_int, C.c_int64,
                                  C.POINTER(C.c_void_p)]
        L.ptgo_destroy.argtypes = [C.c_void_p]
        L.ptgo_destroy.restype = None
        L.ptgo_obs_dim.argtypes = [C.c_void_p]
        L.ptgo_set_noise_tape.argtypes = [C.c_void_p, dp, C.c_int]
        L.ptgo_reset.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.ptgo_step.argtypes = [C.c_void_p, C.c_void_p, dp, dp, u8p, dp, dp]
        L.ptgo_step_mt.argtypes = [C.c_void_p, C.c_void_p, dp, dp, u8p, dp, dp, C.c_int]
        L.ptgo_get_last.argtypes = [C.c_void_p, i64p, dp]
        L.ptgo_get_state.argtypes = [C.c_void_p, i64p, dp]
        L.ptgo_ep_index.argtypes = [C.c_void_p]
        L.ptgo_ep_index.restype = C.c_int64
        L.ptgo_noise_count.argtypes = [C.c_void_p, C.c_int]
        L.ptgo_noise_count.restype = C.c_int64
        L.ptgo_get_index.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.ptgo_get_index.restype = C.c_int32
        L.ptgo_decode_continuous.argtypes = [C.c_void_p, C.c_float, C.c_int32]
        L.ptgo_decode_continuous.restype = C.c_int32
        L.ptgo_pairwise_mean.argtypes = [dp, C.c_int64, C.c_int64]
        L.ptgo_pairwise_mean.restype = C.c_double
        L.ptgo_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def split_reference_kwargs(kw, train_or_eval="train"):
    """(consts, tables, market) from a reference-style kwargs dict (src/rl_utils.py:337-405)."""
    consts = {}
    for k in CONST_KEYS:
        if k == "raw_modified":
            consts[k] = {"raw": 0, "mod": 1}[kw[k]]
        elif k == "action_type":
            consts[k] = {"discrete": 0, "continuous": 1}[kw[k]]
        elif k == "train_or_eval":
            consts[k] = {"train": 0, "eval": 1}[train_or_eval]
        elif k == "r_0":
            consts[k] = float(np.asarray(kw["reward_level"]).reshape(-1)[0])
        else:
            consts[k] = kw[k]
    tables = {k: np.ascontiguousarray(kw[k], dtype=np.float64) for k in TABLE_KEYS}
    e_r_b, g_e = np.asarray(kw["e_r_b"]), np.asarray(kw["g_e"])

    def series(a):     # a[i, t] == s[t + i]  ->  s
        return np.concatenate([a[0, :], a[1:, -1]]).astype(np.float64)
    market = dict(el=series(e_r_b[0]), pot_rew=series(e_r_b[1]), part_full=series(e_r_b[2]),
                  gas=series(g_e[0]), eua=series(g_e[1]),
                  eps_ind=None if kw.get("eps_ind") is None else np.asarray(kw["eps_ind"], dtype=np.float64))
    return consts, tables, market


class OracleVecEnv:
    """N scalar oracle envs behind the DummyVecEnv stepping order (env order, immediate reset when done)."""

    def __init__(self, consts, tables, market, n_envs, ep_index0=0):
        L = lib()
        cfg = _Config()
        for k in CONST_KEYS:
            setattr(cfg, k, consts[k])
        self._keep = []
        tb = _Tables()
        for t, k in enumerate(TABLE_KEYS):
            a = np.ascontiguousarray(tables[k], dtype=np.float64)
            assert a.ndim == 2 and a.shape[1] == 7
            self._keep.append(a)
            tb.data[t] = _dptr(a)
            tb.rows[t] = a.shape[0]
        mk = _Market()
        arrs = {k: np.ascontiguousarray(market[k], dtype=np.float64) for k in ("el", "pot_rew", "part_full", "gas", "eua")}
        assert len(arrs["el"]) == len(arrs["pot_rew"]) == len(arrs["part_full"]) and len(arrs["gas"]) == len(arrs["eua"])
        self._keep.extend(arrs.values())
        mk.n_hours, mk.n_days = len(arrs["el"]), len(arrs["gas"])
        mk.el, mk.pot_rew, mk.part_full = _dptr(arrs["el"]), _dptr(arrs["pot_rew"]), _dptr(arrs["part_full"])
        mk.gas, mk.eua = _dptr(arrs["gas"]), _dptr(arrs["eua"])
        ei = market.get("eps_ind")
        if ei is None:
            mk.n_eps_ind, mk.eps_ind = 0, None
        else:
            ei = np.ascontiguousarray(ei, dtype=np.float64)
            self._keep.append(ei)
            mk.n_eps_ind, mk.eps_ind = len(ei), _dptr(ei)
        h = C.c_void_p()
        rc = L.ptgo_create(C.byref(cfg), C.byref(tb), C.byref(mk), int(n_envs), int(ep_index0), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"ptgo_create failed ({rc}): {L.ptgo_last_error().decode()}")
        self._h, self._L = h, L
        self.n = int(n_envs)
        self.obs_dim = L.ptgo_obs_dim(h)
        self.action_type = consts["action_type"]
        self.eval = bool(consts["train_or_eval"])

    def close(self):
        if getattr(self, "_h", None):
            self._L.ptgo_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(f"oracle error {rc}: {self._L.ptgo_last_error().decode()}")

    def set_noise_tape(self, tape):
        if tape is None:
            self._chk(self._L.ptgo_set_noise_tape(self._h, None, 0))
            return
        tape = np.ascontiguousarray(tape, dtype=np.float64)
        assert tape.shape[0] == self.n
        self._chk(self._L.ptgo_set_noise_tape(self._h, _dptr(tape), tape.shape[1]))

    def reset(self, e=-1):
        obs = np.zeros((self.n, self.obs_dim))
        info = np.zeros((self.n, 24))
        self._chk(self._L.ptgo_reset(self._h, e, _dptr(obs), _dptr(info)))
        return obs, info

    def _actions(self, actions):
        if self.action_type == 0:
            a = np.ascontiguousarray(actions, dtype=np.int32).reshape(-1)
        else:
            a = np.ascontiguousarray(actions, dtype=np.float32).reshape(-1)
        assert a.shape[0] == self.n
        return a

    def step(self, actions, n_threads=0):
        a = self._actions(actions)
        obs = np.zeros((self.n, self.obs_dim))
        rew = np.zeros(self.n)
        done = np.zeros(self.n, np.uint8)
        final = np.zeros((self.n, self.obs_dim))
        info = np.zeros((self.n, 24)) if self.eval else None
        u8 = done.ctypes.data_as(C.POINTER(C.c_uint8))
        if n_threads and n_threads > 1:
            self._chk(self._L.ptgo_step_mt(self._h, a.ctypes.data, _dptr(obs), _dptr(rew), u8, _dptr(final),
                                           _dptr(info), int(n_threads)))
        else:
            self._chk(self._L.ptgo_step(self._h, a.ctypes.data, _dptr(obs), _dptr(rew), u8, _dptr(final), _dptr(info)))
        return obs, rew, done, final, info

    def step_reuse(self, actions, n_threads=0):
        """step() into buffers allocated once (throughput timing: no per-step allocation / zero-fill on the Python side)."""
        b = self.__dict__.get("_bufs")
        if b is None:
            b = self._bufs = (np.zeros((self.n, self.obs_dim)), np.zeros(self.n), np.zeros(self.n, np.uint8), np.zeros((self.n, self.obs_dim)))
        obs, rew, done, final = b
        a = self._actions(actions)
        u8 = done.ctypes.data_as(C.POINTER(C.c_uint8))
        if n_threads and n_threads > 1:
            self._chk(self._L.ptgo_step_mt(self._h, a.ctypes.data, _dptr(obs), _dptr(rew), u8, _dptr(final), None, int(n_threads)))
        else:
            self._chk(self._L.ptgo_step(self._h, a.ctypes.data, _dptr(obs), _dptr(rew), u8, _dptr(final), None))
        return obs, rew, done, final, None

    def last(self):
        ints = np.zeros((self.n, 12), np.int64)
        f = np.zeros((self.n, 8))
        self._chk(self._L.ptgo_get_last(self._h, ints.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(f)))
        return ints, f

    def state(self):
        ints = np.zeros((self.n, 12), np.int64)
        f = np.zeros((self.n, 8))
        self._chk(self._L.ptgo_get_state(self._h, ints.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(f)))
        return ints, f

    @property
    def ep_index(self):
        return int(self._L.ptgo_ep_index(self._h))

    def noise_count(self, e):
        return int(self._L.ptgo_noise_count(self._h, e))

    def get_index(self, table_id, t_cat):
        return int(self._L.ptgo_get_index(self._h, int(table_id), float(t_cat)))

    def decode_continuous(self, a, previous):
        return int(self._L.ptgo_decode_continuous(self._h, float(np.float32(a)), int(previous)))


def pairwise_mean(col):
    """np.average of a 1-D (possibly strided) float64 array, via the oracle's pairwise summation."""
    col = np.asarray(col, dtype=np.float64)
    assert col.ndim == 1 and col.strides[0] % 8 == 0
    return float(lib().ptgo_pairwise_mean(col.ctypes.data_as(C.POINTER(C.c_double)), col.shape[0], col.strides[0] // 8))
