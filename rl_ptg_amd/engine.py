"""HipEngine: the batched PtG env state on one MI355X, driven through the C ABI (include/ptg_env.h).

PyTorch is plumbing here: it owns the observation / reward / done / action buffers (ROCm tensors whose
data_ptr() is handed to the library) and the HIP stream; all env arithmetic runs in the hand-written
kernels of csrc/ptg_env.hip.  There is no CPU path: constructing an engine without the extension or
without a GPU raises.

`consts` uses the key names of the reference's env kwargs (src/rl_utils.py:345-365) with the two strings
already mapped to ints: raw_modified {0 raw, 1 mod}, action_type {0 discrete, 1 continuous},
train_or_eval {0 train, 1 eval}.  `markets` is a list (one per business scenario in the batch) of dicts
with el, pot_rew, part_full, gas, eua (1-D float64), scenario, rew_l_b, rew_u_b, r_0.
"""
import ctypes as C

import numpy as np

from . import _lib

TABLE_KEYS = ["startup_cold", "startup_hot", "cooldown", "standby_down", "standby_up",
              "op1_start_p", "op2_start_f", "op3_p_f", "op4_p_f_p_5", "op5_p_f_p_10",
              "op6_p_f_p_15", "op7_p_f_p_22", "op8_f_p", "op9_f_p_f_5", "op10_f_p_f_10",
              "op11_f_p_f_15", "op12_f_p_f_20"]
ACTIONS = ["standby", "cooldown", "startup", "partial_load", "full_load"]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class PtgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libptg_env error {code}: {msg}")
        self.code = code


class HipEngine:
    def __init__(self, consts, tables, markets, n_envs, device=0, out_dtype="float32", obs_layout="row", obs_pitch=None):
        """obs_pitch (obs_layout="feature" only): elements between two feature planes of the observation buffers -- None: n_envs (planes
        back to back, `[F, N]` contiguous); an int >= n_envs; or "auto": n_envs + 1 KiB worth of elements for float64 outputs whose plane
        stride would be a multiple of 64 KiB (power-of-two batches: planes 2^19 bytes apart run at 0.59 of the HBM peak, 0.70 with the
        pitch; float32 planes do not need it and run slower with one -- profiles/r03_fm_pitch.txt; include/ptg_env.h ptg_set_feature_pitch).  With a pitch the observation tensors are `[F, N]` VIEWS of `[F, pitch]` storage."""
        import torch
        self._torch = torch
        self._L = _lib.lib()                      # raises if the extension is missing
        if not torch.cuda.is_available():
            raise RuntimeError("HipEngine needs a ROCm GPU (torch.cuda.is_available() is False); rl_ptg_amd has no CPU path")
        self.n = int(n_envs)
        self._fin_buf = None
        self.device = torch.device("cuda", int(device))
        self.out_dtype = {"float32": torch.float32, "float64": torch.float64}[out_dtype]
        cfg = _lib.PtgConfig()
        c = dict(consts)
        c.setdefault("t_cat_initial", 16.0)
        c["out_dtype"] = _lib.OUT_F64 if out_dtype == "float64" else _lib.OUT_F32
        c["obs_layout"] = {"row": _lib.OBS_ROW_MAJOR, "feature": _lib.OBS_FEATURE_MAJOR, "sb3_flat": _lib.OBS_SB3_FLAT,
                           "split": _lib.OBS_SPLIT}[obs_layout]
        self.feature_major = obs_layout == "feature"
        for k in _lib.CONFIG_KEYS:
            setattr(cfg, k, c[k])
        self.consts = c
        self._keep = []
        tb = _lib.PtgTables()
        for t, k in enumerate(TABLE_KEYS):
            a = np.ascontiguousarray(tables[k], dtype=np.float64)
            if a.ndim != 2 or a.shape[1] != 7:
                raise ValueError(f"table {k}: expected [rows, 7], got {a.shape}")
            self._keep.append(a)
            tb.data_host[t] = _dp(a)
            tb.rows[t] = a.shape[0]
        if isinstance(markets, dict):
            markets = [markets]
        mk = (_lib.PtgMarket * len(markets))()
        for s, m in enumerate(markets):
            arrs = {k: np.ascontiguousarray(m[k], dtype=np.float64) for k in ("el", "pot_rew", "part_full", "gas", "eua")}
            if not (len(arrs["el"]) == len(arrs["pot_rew"]) == len(arrs["part_full"])) or len(arrs["gas"]) != len(arrs["eua"]):
                raise ValueError("market series lengths differ")
            self._keep.extend(arrs.values())
            mk[s].n_hours, mk[s].n_days = len(arrs["el"]), len(arrs["gas"])
            mk[s].el_host, mk[s].pot_rew_host, mk[s].part_full_host = _dp(arrs["el"]), _dp(arrs["pot_rew"]), _dp(arrs["part_full"])
            mk[s].gas_host, mk[s].eua_host = _dp(arrs["gas"]), _dp(arrs["eua"])
            mk[s].scenario = int(m["scenario"])
            mk[s].rew_l_b, mk[s].rew_u_b, mk[s].r_0 = float(m["rew_l_b"]), float(m["rew_u_b"]), float(m["r_0"])
        self.n_sets = len(markets)
        h = C.c_void_p()
        rc = self._L.ptg_create(C.byref(cfg), C.byref(tb), mk, len(markets), self.n, int(device), C.byref(h))
        if rc != 0:
            raise PtgError(rc, self._L.ptg_last_error(None).decode())
        self._h = h
        self.obs_dim = self._L.ptg_obs_dim(h)
        osz = 8 if out_dtype == "float64" else 4
        if obs_pitch == "auto":
            import os
            pad = int(os.environ.get("PTG_FM_PAD_BYTES", "1024"))         # (experiments: tools/r03_pitch2.sh -> profiles/r03_fm_pitch.txt)
            obs_pitch = self.n + pad // osz if (self.feature_major and osz == 8 and pad and (self.n * osz) % 65536 == 0) else None
        self.pitch = self.n if obs_pitch is None else int(obs_pitch)
        if self.pitch != self.n:
            if not self.feature_major:
                raise ValueError("obs_pitch applies to obs_layout='feature' only")
            self._chk(self._L.ptg_set_feature_pitch(h, self.pitch))
        self.action_type = int(c["action_type"])
        self.eval_mode = bool(c["train_or_eval"])
        with torch.cuda.device(self.device):
            self.obs = self.alloc_obs(zero=True)
            self.final_obs = self.alloc_obs(zero=True)
            self.rew = torch.zeros(self.n, dtype=self.out_dtype, device=self.device)
            self.done = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
            self.info = torch.zeros((self.n, _lib.N_INFO), dtype=torch.float64, device=self.device) if self.eval_mode else None

    # ------------------------------------------------------------------ plumbing
    def alloc_obs(self, T=None, zero=False):
        """An observation buffer in this engine's layout: [N, F] row-major, [F, N] feature-major (a view of [F, pitch] storage when the
        engine has a pitch); with T: [T, ...] for a rollout."""
        torch = self._torch
        make = torch.zeros if zero else torch.empty
        lead = () if T is None else (int(T),)
        if not self.feature_major:
            return make(lead + (self.n, self.obs_dim), dtype=self.out_dtype, device=self.device)
        t = make(lead + (self.obs_dim, self.pitch), dtype=self.out_dtype, device=self.device)
        return t[..., :self.n] if self.pitch != self.n else t

    def _check_obs(self, obs):
        """a caller-supplied feature-major buffer must have the engine's plane pitch (and a contiguous env axis)"""
        if self.feature_major:
            ok = obs.stride(-1) == 1 and obs.stride(-2) == self.pitch and (obs.dim() == 2 or obs.stride(0) == self.obs_dim * self.pitch)
        else:
            ok = obs.is_contiguous()
        if not ok:
            raise ValueError(f"observation buffer strides {tuple(obs.stride())} do not match the engine's layout (pitch {self.pitch}); use alloc_obs()")

    def close(self):
        if getattr(self, "_h", None):
            self._L.ptg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise PtgError(rc, self._L.ptg_last_error(self._h).decode())

    def rows(self, obs):
        """[N, F] (or [T, N, F]) view of an observation buffer in either layout (a transpose view for feature-major)."""
        return obs.transpose(-1, -2) if self.feature_major else obs

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _action_kind(self, t):
        torch = self._torch
        if t.dtype == torch.int32:
            return _lib.ACT_I32
        if t.dtype == torch.int64:
            return _lib.ACT_I64
        if t.dtype == torch.float32:
            return _lib.ACT_F32
        raise TypeError(f"actions must be int32 / int64 / float32, got {t.dtype}")

    def as_device_actions(self, actions):
        """numpy / torch actions -> contiguous device tensor of a dtype the kernels read."""
        torch = self._torch
        if not torch.is_tensor(actions):
            a = np.asarray(actions)
            a = a.astype(np.float32) if self.action_type == 1 else a.astype(np.int32)
            actions = torch.from_numpy(np.ascontiguousarray(a))
        if actions.device != self.device:
            actions = actions.to(self.device, non_blocking=True)
        if self.action_type == 1 and actions.dtype != torch.float32:
            actions = actions.float()
        return actions.contiguous()

    # ------------------------------------------------------------------ configuration
    def set_market_assignment(self, set_of_env):
        a = np.ascontiguousarray(set_of_env, dtype=np.uint8)
        assert a.shape == (self.n,)
        self._chk(self._L.ptg_set_market_assignment(self._h, a.ctypes.data_as(C.POINTER(C.c_uint8))))

    def set_episode_plan(self, eps_ind, first_ptr, stride):
        if eps_ind is None or len(eps_ind) == 0:
            self._chk(self._L.ptg_set_episode_plan(self._h, None, 0, 0, 0))
            return
        a = np.ascontiguousarray(eps_ind, dtype=np.float64)
        self._chk(self._L.ptg_set_episode_plan(self._h, _dp(a), len(a), int(first_ptr), int(stride)))

    def set_noise_tape(self, tape):
        if tape is None:
            self._chk(self._L.ptg_set_noise_tape(self._h, None, 0))
            self._noise_cfg = {"mode": "none"}
            return
        a = np.ascontiguousarray(tape, dtype=np.float64)
        assert a.ndim == 2 and a.shape[0] == self.n
        self._chk(self._L.ptg_set_noise_tape(self._h, _dp(a), a.shape[1]))
        self._noise_cfg = {"mode": "tape", "per_env_len": int(a.shape[1])}

    def set_noise_rng(self, seed):
        """Draw the state-change noise inside the kernels from the counter-based generator (no tape)."""
        self._chk(self._L.ptg_set_noise_rng(self._h, int(seed) & (2 ** 64 - 1)))
        self._noise_cfg = {"mode": "rng", "seed": int(seed) & (2 ** 64 - 1)}

    def set_global_env_offset(self, offset):
        self._chk(self._L.ptg_set_global_env_offset(self._h, int(offset)))
        self._env_offset = int(offset)

    def fill_noise_tape(self, seed, per_env_len):
        self._chk(self._L.ptg_fill_noise_tape(self._h, int(seed) & (2 ** 64 - 1), int(per_env_len), self._stream()))
        self.tape_len = int(per_env_len)
        self._noise_cfg = {"mode": "tape", "per_env_len": int(per_env_len)}

    def get_noise_tape(self, per_env_len):
        out = np.zeros((self.n, per_env_len))
        self._chk(self._L.ptg_get_noise_tape(self._h, _dp(out)))
        return out

    # ------------------------------------------------------------------ hot path
    def reset(self, mask=None):
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert m.shape == (self.n,)
        with self._torch.cuda.device(self.device):
            self._chk(self._L.ptg_reset(self._h, None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8)),
                                        C.c_void_p(self.obs.data_ptr()), self._stream()))
        return self.obs

    def step(self, actions, obs=None, rew=None, done=None, final_obs=None, want_final=True):
        """Enqueue one vector step on the current stream; returns (obs, rew, done) device tensors (no sync)."""
        a = self.as_device_actions(actions)
        assert a.numel() == self.n
        if obs is not None:
            self._check_obs(obs)
        if final_obs is not None:
            self._check_obs(final_obs)
        obs = self.obs if obs is None else obs
        rew = self.rew if rew is None else rew
        done = self.done if done is None else done
        fo = (self.final_obs if final_obs is None else final_obs) if want_final else None
        with self._torch.cuda.device(self.device):
            self._chk(self._L.ptg_step(self._h, C.c_void_p(a.data_ptr()), self._action_kind(a), C.c_void_p(obs.data_ptr()),
                                       C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
                                       C.c_void_p(fo.data_ptr()) if fo is not None else None,
                                       C.c_void_p(self.info.data_ptr()) if self.info is not None else None, self._stream()))
        return obs, rew, done

    def rollout(self, actions, obs=None, rew=None, done=None):
        """T fused steps in one launch: actions [T, N] -> obs [T, N, F] ([T, F, N] feature-major), rew [T, N], done [T, N]."""
        torch = self._torch
        a = self.as_device_actions(actions)
        assert a.dim() == 2 and a.shape[1] == self.n
        T = a.shape[0]
        with torch.cuda.device(self.device):
            if obs is None:
                obs = self.alloc_obs(T)
            else:
                self._check_obs(obs)
            if rew is None:
                rew = torch.empty((T, self.n), dtype=self.out_dtype, device=self.device)
            if done is None:
                done = torch.empty((T, self.n), dtype=torch.uint8, device=self.device)
            self._chk(self._L.ptg_rollout(self._h, C.c_void_p(a.data_ptr()), self._action_kind(a), T, C.c_void_p(obs.data_ptr()),
                                          C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()), self._stream()))
        return obs, rew, done

    def rollout_info(self, actions):
        """T steps with the 24 `_get_info` fields of every step recorded on the device: returns (obs, rew, done, info) with
        info [T, N, 24] float64 in the key order of rl_ptg_amd.vec_env.INFO_KEYS (what Postprocessing.test_performance gathers
        from per-step info dicts, src/rl_utils.py:528-565)."""
        torch = self._torch
        a = self.as_device_actions(actions)
        assert a.dim() == 2 and a.shape[1] == self.n
        T = a.shape[0]
        with torch.cuda.device(self.device):
            obs = self.alloc_obs(T)
            rew = torch.empty((T, self.n), dtype=self.out_dtype, device=self.device)
            done = torch.empty((T, self.n), dtype=torch.uint8, device=self.device)
            info = torch.empty((T, self.n, 24), dtype=torch.float64, device=self.device)
            self._chk(self._L.ptg_rollout_info(self._h, C.c_void_p(a.data_ptr()), self._action_kind(a), T, C.c_void_p(obs.data_ptr()),
                                               C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()), C.c_void_p(info.data_ptr()), self._stream()))
        return obs, rew, done, info

    def rollout_launches(self, n_steps):
        """Kernel launches a rollout of n_steps would issue from the envs' current position (per-launch timing)."""
        k = self._L.ptg_rollout_launches(self._h, int(n_steps))
        if k < 0:
            self._chk(k)
        return k

    def profile(self, enable=True):
        """Start / stop collecting the device time of every hot-kernel launch (kernel-attached HIP events)."""
        self._chk(self._L.ptg_profile(self._h, 1 if enable else 0))

    def profile_read(self, cap=65536):
        """Durations [us] of the launches recorded since profile(True), in launch order; waits for them and clears the list."""
        out = np.zeros(cap)
        cnt = C.c_int(0)
        self._chk(self._L.ptg_profile_read(self._h, _dp(out), cap, C.byref(cnt)))
        return out[:cnt.value].copy()

    def profile_read_ex(self, cap=65536):
        """profile_read with each launch's helper kernel (the table refresher beside it) accounted for: (us, helper_us, span_us),
        span = first start to last end of the launch and its helper (include/ptg_env.h, ptg_profile_read_ex)."""
        us, hp, sp = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        cnt = C.c_int(0)
        self._chk(self._L.ptg_profile_read_ex(self._h, _dp(us), _dp(hp), _dp(sp), cap, C.byref(cnt)))
        n = cnt.value
        return us[:n].copy(), hp[:n].copy(), sp[:n].copy()

    def finished_dropped(self):
        """Finished episodes that were never handed out (ring overflow / a query's cap) since the engine was created."""
        d = C.c_uint64(0)
        self._chk(self._L.ptg_finished_dropped(self._h, C.byref(d)))
        return int(d.value)

    def steps_to_episode_end(self):
        """Vector steps from now up to and including the one on which a synchronised batch's episodes end; 0 = unknown to the
        host (de-synchronised batch).  Pure host bookkeeping: no device call."""
        s = C.c_int(0)
        self._chk(self._L.ptg_steps_to_episode_end(self._h, C.byref(s)))
        return int(s.value)

    def note_replays(self, n_steps):
        """Tell the handle that captured hot launches were REPLAYED for `n_steps` vector steps in all (include/ptg_env.h, "hipGraph
        capture"): step() / rollout() captured into a graph can be replayed because the kernels read the step count from the device state;
        the host-side count that routes an episode's terminating step only sees eager calls and the capture itself."""
        self._chk(self._L.ptg_note_replays(self._h, int(n_steps)))

    def set_replay_proof(self, enable=True):
        """A step() captured into a graph AFTER this call replays across episode ends by itself (hot kernel + generic kernel, one of them a
        no-op per step: +1.5-2 us); without it a captured step must stop before the episode's terminating step (include/ptg_env.h)."""
        self._chk(self._L.ptg_set_replay_proof(self._h, 1 if enable else 0))

    def sync(self):
        self._chk(self._L.ptg_sync(self._h, self._stream()))

    # ------------------------------------------------------------------ state access
    def get_state(self, name):
        f = _lib.STATE_FIELDS[name]
        out = np.zeros(self.n, np.float64 if f >= 32 else np.int32)
        self._chk(self._L.ptg_get_state(self._h, f, C.c_void_p(out.ctypes.data)))
        return out

    def set_state(self, name, values):
        f = _lib.STATE_FIELDS[name]
        a = np.ascontiguousarray(values, dtype=np.float64 if f >= 32 else np.int32)
        assert a.shape == (self.n,)
        self._chk(self._L.ptg_set_state(self._h, f, C.c_void_p(a.ctypes.data)))

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        """Everything a resumed run needs beyond the constructor arguments and the episode plan: every per-env state field,
        the noise source, the global env offset and -- if started -- the reward normaliser with its hyper-parameters (NumPy arrays /
        plain numbers; synchronises).  Not included: the list of finished episodes not yet collected (ptg_finished_episodes)."""
        sd = {"fields": {k: self.get_state(k) for k in _lib.STATE_FIELDS}, "n": self.n,
              "env_offset": self.__dict__.get("_env_offset", 0)}       # keys the in-kernel RNG streams (global env index)
        sd["noise"] = dict(self.__dict__.get("_noise_cfg", {"mode": "none"}))
        if sd["noise"].get("mode") == "tape":
            sd["noise"]["tape"] = self.get_noise_tape(sd["noise"]["per_env_len"])
        try:
            st, ret = self.vn_get()
            sd["vn"] = {"stats": st, "returns": ret, "hyper": dict(self.__dict__.get("_vn_hyper", {}))}
        except PtgError:
            pass
        return sd

    def load_state_dict(self, sd):
        """Inverse of state_dict() on an engine built with the same arguments (reset() first, then the episode plan)."""
        assert sd["n"] == self.n
        if "env_offset" in sd:
            self.set_global_env_offset(sd["env_offset"])
        nz = sd.get("noise", {})
        if nz.get("mode") == "rng":
            self.set_noise_rng(nz["seed"])
        elif nz.get("mode") == "tape":
            self.set_noise_tape(nz["tape"])
        for k, v in sd["fields"].items():
            if k != "k":
                self.set_state(k, v)
        self.set_state("k", sd["fields"]["k"])              # last: equal step counts mark the batch as synchronised again
        if "vn" in sd:
            hyper = sd["vn"].get("hyper") or {}
            if hyper and hyper != self.__dict__.get("_vn_hyper"):
                self.vn_init(**hyper)                           # the checkpoint's gamma / epsilon / clip_reward
            else:
                try:
                    self.vn_get()
                except PtgError:
                    self.vn_init()
            self.vn_set(stats=sd["vn"]["stats"], returns=sd["vn"]["returns"])

    def finished_episodes(self, cap=None):
        cap = max(2 * self.n, 1024) if cap is None else int(cap)          # the library's ring holds max(2 n, 1024) entries
        buf = self._fin_buf
        if buf is None or buf[0].shape[0] < cap:                          # receive buffers kept across calls (2 MB at 65 536 envs)
            buf = self._fin_buf = (np.zeros(cap), np.zeros(cap, np.int32), np.zeros(cap, np.int32))
        r, l, ids = buf
        cnt = C.c_int(0)
        self._chk(self._L.ptg_finished_episodes(self._h, _dp(r), l.ctypes.data_as(C.POINTER(C.c_int32)),
                                                ids.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(cnt)))
        n = cnt.value
        return r[:n].copy(), l[:n].copy(), ids[:n].copy()

    # ------------------------------------------------------------------ VecNormalize(norm_obs=False) on the device
    def vn_init(self, gamma=0.99, epsilon=1e-8, clip_reward=10.0):
        """Start reward normalisation as the reference wraps its envs (src/rl_utils.py:453, SB3 defaults)."""
        self._chk(self._L.ptg_vn_init(self._h, float(gamma), float(epsilon), float(clip_reward)))
        self._vn_hyper = {"gamma": float(gamma), "epsilon": float(epsilon), "clip_reward": float(clip_reward)}

    def vn_normalize(self, rew, done, training=True, out=None, group=None):
        """Normalise a [T, N] (or [N]) reward tensor in place of VecNormalize.step_wait: advances the discounted returns,
        updates the running moments step by step (training=True) and returns the clipped, scaled rewards.  With an
        initialised torch.distributed process group the per-step moments of all ranks' envs are merged first (one
        all-gather per call), so every rank holds the statistics of the whole job."""
        torch = self._torch
        from . import dist as ptg_dist
        r2 = rew if rew.dim() == 2 else rew.unsqueeze(0)
        d2 = done if done.dim() == 2 else done.unsqueeze(0)
        T = r2.shape[0]
        assert r2.shape == (T, self.n) and d2.shape == (T, self.n) and r2.is_contiguous() and d2.is_contiguous()
        res = torch.empty_like(r2) if out is None else (out if out.dim() == 2 else out.unsqueeze(0))
        with torch.cuda.device(self.device):
            mom = None
            if training:
                mom = torch.empty((T, 3), dtype=torch.float64, device=self.device)
                self._chk(self._L.ptg_vn_batch_moments(self._h, C.c_void_p(r2.data_ptr()), C.c_void_p(d2.data_ptr()), T,
                                                       C.c_void_p(mom.data_ptr()), self._stream()))
                mom = ptg_dist.all_merge_moments(mom, group=group)
            self._chk(self._L.ptg_vn_apply(self._h, C.c_void_p(r2.data_ptr()), T, C.c_void_p(mom.data_ptr()) if mom is not None else None,
                                           C.c_void_p(res.data_ptr()), 1 if training else 0, self._stream()))
        return res if rew.dim() == 2 else res[0]

    def vn_get(self):
        st, ret = np.zeros(3), np.zeros(self.n)
        self._chk(self._L.ptg_vn_get(self._h, _dp(st), _dp(ret)))
        return dict(mean=st[0], var=st[1], count=st[2]), ret

    def vn_set(self, stats=None, returns=None):
        st = None if stats is None else np.array([stats["mean"], stats["var"], stats["count"]], dtype=np.float64)
        rt = None if returns is None else np.ascontiguousarray(returns, dtype=np.float64)
        self._chk(self._L.ptg_vn_set(self._h, None if st is None else _dp(st), None if rt is None else _dp(rt)))

    def market_feature_series(self):
        """The pre-normalised float32 feature series the kernels read, each [n_sets, length]: dict(featA, featB (hourly), gas_n, eua_n
        (daily)).  Columns 14 / 15 of a "split" observation row index the flattened arrays."""
        out = {}
        for which, name in enumerate(("featA", "featB", "gas_n", "eua_n")):
            cnt = C.c_int(0)
            self._chk(self._L.ptg_market_feature_series(self._h, which, None, 0, C.byref(cnt)))
            a = np.zeros(cnt.value, np.float32)
            self._chk(self._L.ptg_market_feature_series(self._h, which, a.ctypes.data_as(C.POINTER(C.c_float)), cnt.value, C.byref(cnt)))
            out[name] = a.reshape(self.n_sets, -1)
        return out

    def debug_get_index_lut(self):
        nT = C.c_int(0)
        self._chk(self._L.ptg_debug_get_index_lut(self._h, None, None, C.byref(nT)))
        T = np.zeros(nT.value)
        lut = np.zeros((6, nT.value), np.int32)
        self._chk(self._L.ptg_debug_get_index_lut(self._h, _dp(T), lut.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nT)))
        return T, lut

    def debug_window_record(self, table_id, start_row):
        out = np.zeros(7)
        self._chk(self._L.ptg_debug_window_record(self._h, int(table_id), int(start_row), _dp(out)))
        return out
