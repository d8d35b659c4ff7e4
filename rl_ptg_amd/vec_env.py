"""PtGVecEnv: the batched PtG environment behind the Stable-Baselines3 VecEnv surface.

Replaces, for the reference's training loop, the stack  make_vec_env -> Monitor -> DummyVecEnv  around N reference
`PTGEnv` objects (src/rl_utils.py:448-453, :480-500); it stays wrappable by `VecNormalize(env, norm_obs=False)` and
usable by `EvalCallback`.  Constructor input is the reference's own env kwargs dict (`Preprocessing.dict_env_kwargs`,
src/rl_utils.py:337-405) -- the reference's or rl_ptg_amd.prep's.

    env = PtGVecEnv(dict_input, n_envs=6, train_or_eval="train", seed=3654)
    obs = env.reset()                                  # dict of np.ndarray [N, ...] in the declared float64 spaces
    obs, rewards, dones, infos = env.step(actions)     # rewards float32 [N], dones bool [N], infos list of N dicts

Semantics kept from the reference stack
  * envs are stepped "in env order" and a finished env is reset at once: infos[e]["terminal_observation"] holds the last
    observation, the returned row is the post-reset observation, infos[e]["episode"] = {"r", "l", "t"} (Monitor),
    infos[e]["TimeLimit.truncated"] = False; `truncated` never occurs (env/ptg_gym_env.py:478-481);
  * training envs take their episodes from the shared `eps_ind` sequence in the order N reference envs would
    (constructor + reset consumption, env/ptg_gym_env.py:59-62,490-493);
  * `train_or_eval="eval"`: infos carry the reference's 24-key dict (env/ptg_gym_env.py:251-278);
  * `seed(s)`: env e draws its state-change noise from numpy Generator(PCG64(SeedSequence(s + e))) -- the stream Gymnasium
    gives the reference env -- when noise="numpy" (bit parity, host-generated tape, meant for small N); noise="device"
    uses the in-kernel counter RNG (statistically equivalent; any N).

Device-side consumers (a policy living on the same GPU) can skip the NumPy hand-off: `step_tensors(actions)` returns the
ROCm tensors (obs matrix [N, F] or feature-major [F, N], rewards, dones) without synchronising.
"""
import time

import numpy as np

from .engine import ACTIONS, HipEngine
from .prep import EnvSpec
from .spaces import make_spaces, obs_columns

INFO_KEYS = ["step", "el_price_act", "gas_price_act", "eua_price_act", "Meth_State", "Meth_Action", "Meth_Hot_Cold",
             "Meth_T_cat", "Meth_H2_flow", "Meth_CH4_flow", "Meth_H2O_flow", "Meth_el_heating", "ch4_revenues [ct/h]",
             "steam_revenues [ct/h]", "o2_revenues [ct/h]", "eua_revenues [ct/h]", "chp_revenues [ct/h]",
             "elec_costs_heating [ct/h]", "elec_costs_electrolyzer [ct/h]", "water_costs [ct/h]", "reward [ct]",
             "cum_reward", "Pot_Reward", "Part_Full"]

try:                                     # pragma: no cover - SB3 is not installed in the build image
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase
except Exception:
    _VecEnvBase = object


_INFO_INT = frozenset(("step", "Meth_State", "Meth_Hot_Cold"))
_INFO_POS = {k: q for q, k in enumerate(INFO_KEYS)}


class _InfoRow(dict):
    """info dict of one env of a LARGE eval batch: the 24 reference keys are read on demand from the step's info matrix (building
    65 536 x 24-entry dicts per step is the O(N) Python loop SURVEY.md section 7 warns about); keys stored explicitly (episode,
    terminal_observation, ...) behave as in a plain dict.  Reads reflect the latest step of the owning PtGVecEnv."""
    __slots__ = ("_own", "_e")

    def __init__(self, owner, e):
        super().__init__()
        self._own, self._e = owner, e

    def __missing__(self, k):
        q = _INFO_POS[k]
        v = self._own._info_cur[self._e, q]
        return int(v) if k in _INFO_INT else (ACTIONS[int(v)] if k == "Meth_Action" else float(v))

    def __contains__(self, k):
        return dict.__contains__(self, k) or (k in _INFO_POS and self._own._info_cur is not None)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def keys(self):
        return list(INFO_KEYS) + [k for k in dict.keys(self)]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(INFO_KEYS) + dict.__len__(self)

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class PtGVecEnv(_VecEnvBase):
    metadata = {"render_modes": ["None"]}
    OBS_RING = 4            # large batches: step() hands out views of a ring of pinned blocks (see _setup_host)
    EAGER_INFO_MAX = 64     # eval batches up to this size get plain 24-key info dicts every step, larger ones _InfoRow views

    def __init__(self, dict_input, n_envs, train_or_eval="train", seed=None, device=0, out_dtype="float64", obs_layout="row",
                 noise="numpy", noise_tape_len=256, world_size=1, rank=0, render_mode="None", engine_cls=HipEngine,
                 norm_reward=False, gamma=0.99, epsilon=1e-8, clip_reward=10.0, copy_obs=None):
        if obs_layout not in ("row", "feature"):
            # the Dict observation is carved out of the canonical [N, F] matrix; SB3's flattened rows ("sb3_flat") and the split rows
            # ("split") have other columns -- they belong to the tensor API (HipEngine.step / rollout, step_tensors)
            raise ValueError(f"PtGVecEnv: obs_layout must be 'row' or 'feature', got {obs_layout!r} (use HipEngine for 'sb3_flat' / 'split' rows)")
        spec = dict_input if isinstance(dict_input, EnvSpec) else EnvSpec.from_dict_input(dict_input, train_or_eval)
        want = {"train": 0, "eval": 1}[train_or_eval]
        if int(spec.consts.get("train_or_eval", want)) != want:      # a prepared EnvSpec used for the other mode: the argument wins
            spec = EnvSpec(dict(spec.consts, train_or_eval=want), spec.tables, spec.markets, spec.eps_ind)
        self.spec = spec
        self.train_or_eval = train_or_eval
        self.raw_modified = "mod" if spec.consts["raw_modified"] else "raw"
        self.action_type = "continuous" if spec.consts["action_type"] else "discrete"
        self.observation_space, self.action_space = make_spaces(self.raw_modified, self.action_type, spec.consts["price_ahead"])
        self.num_envs = int(n_envs)
        self.render_mode = render_mode
        self.world_size, self.rank = int(world_size), int(rank)
        self.n_total = self.num_envs * self.world_size
        self.env_offset = self.num_envs * self.rank
        if _VecEnvBase is not object:    # pragma: no cover
            super().__init__(self.num_envs, self.observation_space, self.action_space)
        self.engine = engine_cls(spec.consts, spec.tables, spec.markets, self.num_envs, device=device, out_dtype=out_dtype,
                                 obs_layout=obs_layout)
        self.engine.set_global_env_offset(self.env_offset)
        # DummyVecEnv order: n_total constructions consume eps_ind[0:n_total]; the first vector reset takes eps_ind[n_total + e]
        self.engine.set_episode_plan(spec.eps_ind, first_ptr=self.n_total + self.env_offset, stride=self.n_total)
        self._cols, self._F = obs_columns(self.raw_modified, spec.consts["price_ahead"])
        keys = list(self.observation_space.spaces) if hasattr(self.observation_space, "spaces") else list(self._cols)
        self._key_slices = [(k, self._cols[k]) for k in keys]
        self.noise_mode = noise
        self.noise_sigma = float(spec.consts["noise"])
        self._tape_len = int(noise_tape_len)
        self._gens = None
        self._steps_since_refill = 0
        self._tape = None
        self._seed = seed
        self._apply_seed(seed)
        self._actions = None
        self.reset_infos = [{} for _ in range(self.num_envs)]
        self._t0 = time.time()
        self._ep_start = np.full(self.num_envs, self._t0)
        self._needs_reset = True
        # norm_reward=True: step() returns rewards normalised as the reference's VecNormalize(env, norm_obs=False) wrapper does
        # (src/rl_utils.py:453), computed on the device (HipEngine.vn_normalize); `training` = False freezes the statistics
        self.norm_reward = bool(norm_reward)
        self.training = True
        self._old_reward = None
        # copy_obs: True = step() returns fresh arrays and a new infos list every step, like DummyVecEnv (_save_obs copies);
        # False = views of a ring of OBS_RING pinned blocks (valid until step() has been called OBS_RING - 1 more times) and ONE
        # persistent infos list whose entries are replaced in place; None = copies up to 64 KiB per step, views above (copying
        # 9-18 MB and building a 65 536-entry list per step costs more than the step).  INTEGRATION.md, "Lifetime of what step() returns".
        self._copy_obs_arg = copy_obs
        if self.norm_reward:
            self.engine.vn_init(gamma=gamma, epsilon=epsilon, clip_reward=clip_reward)
        self._setup_host()

    def get_original_reward(self):
        """Unnormalised rewards of the last step (VecNormalize.get_original_reward)."""
        return None if self._old_reward is None else self._old_reward.copy()

    @property
    def ret_rms(self):
        """Running moments of the discounted returns: dict(mean, var, count), as VecNormalize.ret_rms holds them."""
        return self.engine.vn_get()[0] if self.norm_reward else None

    # ------------------------------------------------------------------ seeding / noise
    def _apply_seed(self, seed):
        if self.noise_mode == "device":
            self.engine.set_noise_rng(0 if seed is None else int(seed))
            return
        if self.noise_mode != "numpy":
            raise ValueError("noise must be 'numpy' or 'device'")
        # Gymnasium seeding (gymnasium.utils.seeding.np_random): Generator(PCG64(SeedSequence(seed))); None -> OS entropy
        self._gens = [np.random.Generator(np.random.PCG64(np.random.SeedSequence(None if seed is None else int(seed) + self.env_offset + e)))
                      for e in range(self.num_envs)]
        self._tape = np.stack([g.normal(0, self.noise_sigma, size=self._tape_len) for g in self._gens])
        self.engine.set_noise_tape(self._tape)
        self._steps_since_refill = 0

    def _refill_tape(self):
        """At most one draw per env and step is consumed: after tape_len steps move the unread draws to the front and top up."""
        used = self.engine.get_state("noise_count")
        L = self._tape_len
        for e, g in enumerate(self._gens):
            u = int(min(used[e], L))
            if u:
                self._tape[e, :L - u] = self._tape[e, u:]
                self._tape[e, L - u:] = g.normal(0, self.noise_sigma, size=u)
        self.engine.set_noise_tape(self._tape)
        self._steps_since_refill = 0

    def seed(self, seed=None):
        self._seed = seed
        self._apply_seed(seed)
        return [None if seed is None else seed + self.env_offset + e for e in range(self.num_envs)]

    # ------------------------------------------------------------------ host buffers
    def _setup_host(self):
        """Pinned host blocks for ptg_step_host (include/ptg_env.h): actions in, [obs | rewards | dones] out, laid out by the
        library.  Small batches (<= 64 KiB per step) get ONE block -- the kernels write it in place over PCIe -- and every step()
        hands out fresh copies, like DummyVecEnv.  Large batches rotate through a ring of OBS_RING blocks and hand out VIEWS: an
        observation stays valid until step() has been called OBS_RING - 1 more times (SB3 keeps `_last_obs` across exactly one
        step()); copying 9-18 MB per step would cost more than the step."""
        import ctypes as C
        import torch
        eng = self.engine
        o_rew, o_done, o_stat, total = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        eng._chk(eng._L.ptg_host_layout_ex(eng._h, C.byref(o_rew), C.byref(o_done), C.byref(o_stat), C.byref(total)))
        self._off_rew, self._off_done, self._off_status, self._blk_bytes = o_rew.value, o_done.value, o_stat.value, total.value
        n, F = self.num_envs, eng.obs_dim
        self._odt = np.float64 if eng.out_dtype == torch.float64 else np.float32
        self._copy_out = total.value <= (64 << 10) if self._copy_obs_arg is None else bool(self._copy_obs_arg)
        k = 1 if self._copy_out else self.OBS_RING

        def pinned(nbytes):
            try:
                t = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, pin_memory=True)
            except RuntimeError:                           # pinning refused: the library stages through device buffers
                t = torch.empty(max(int(nbytes), 16), dtype=torch.uint8)
            self._keep.append(t)
            return t.numpy()
        self._keep = []
        self._blk = [pinned(total.value) for _ in range(k)]
        self._blk_ptr = [C.c_void_p(b.ctypes.data) for b in self._blk]
        self._views = [self._carve(b) for b in self._blk]
        self._slot = 0
        adt = np.float32 if self.action_type == "continuous" else np.int32
        self._act_buf = pinned(n * 4).view(adt)[:n]
        self._act_ptr = C.c_void_p(self._act_buf.ctypes.data)
        self._act_kind = 1 if self.action_type == "continuous" else 0      # PTG_ACT_F32 / PTG_ACT_I32
        self._final = pinned(n * F * np.dtype(self._odt).itemsize).view(self._odt)
        self._final_ptr = C.c_void_p(self._final.ctypes.data)
        self._final_mat = self._final.reshape(F, n).T if eng.feature_major else self._final.reshape(n, F)
        self._info_host, self._info_ptr = None, None
        self._lazy_info = eng.eval_mode and n > self.EAGER_INFO_MAX
        if eng.eval_mode:
            # large eval batches: TWO info blocks used in turn -- the rows of the latest step are read in place through _InfoRow (no
            # 12.6 MB copy per step at 65 536 envs) while the next step's rows land in the other block
            self._info_blocks = [pinned(n * 24 * 8).view(np.float64)[:n * 24].reshape(n, 24) for _ in range(2 if self._lazy_info else 1)]
            self._info_ptrs = [C.c_void_p(b.ctypes.data) for b in self._info_blocks]
            self._info_slot = 0
            self._info_host, self._info_ptr = self._info_blocks[0], self._info_ptrs[0]
        self._n_done = C.c_int(0)
        self._n_done_ref = C.byref(self._n_done)
        self._in_flight = None                                  # slot of the host step begun by step_async
        # infos: N persistent dicts, replaced only for envs whose episode ended (and put back empty on the next step)
        self._infos = [{} for _ in range(n)]
        self._dirty = []
        if self._lazy_info:
            self._infos = [_InfoRow(self, e) for e in range(n)]
        self._info_cur = None

    def _carve(self, blk):
        """(obs matrix [N, F] view, rewards view, dones view) of one host block"""
        n, F = self.num_envs, self.engine.obs_dim
        nb = n * F * np.dtype(self._odt).itemsize
        flat = blk[:nb].view(self._odt)
        mat = flat.reshape(F, n).T if self.engine.feature_major else flat.reshape(n, F)
        rew = blk[self._off_rew:self._off_rew + n * np.dtype(self._odt).itemsize].view(self._odt)
        done = blk[self._off_done:self._off_done + n]
        status = blk[self._off_status:self._off_status + n]    # METH_STATUS of every row, contiguous bytes (ptg_host_layout_ex)
        return mat, rew, done, status

    # ------------------------------------------------------------------ observations
    def _obs_dict(self, mat, copy=True, status=None):
        """[N, F] matrix (reference dict order) -> dict keyed like observation_space.  copy=False: column views of `mat`
        (METH_STATUS is always a new int64 array; `status` = that array, if the caller has already made it from the library's
        contiguous status bytes).  Element type = the engine's out_dtype (float64 by default, as declared)."""
        out = {}
        for k, sl in self._key_slices:
            if k == "METH_STATUS":
                out[k] = status if status is not None else np.rint(mat[:, sl.start]).astype(np.int64)
            else:
                out[k] = np.array(mat[:, sl], copy=True) if copy else mat[:, sl]
        return out

    def _obs_row_dict(self, row):
        d = self._obs_dict(row[None, :])
        return {k: (v[0] if k != "METH_STATUS" else int(v[0])) for k, v in d.items()}

    @staticmethod
    def _info_dict(row):
        d = {}
        for q, k in enumerate(INFO_KEYS):
            v = row[q]
            if k in _INFO_INT:
                d[k] = int(v)
            elif k == "Meth_Action":
                d[k] = ACTIONS[int(v)]
            else:
                d[k] = float(v)
        return d

    def _to_host(self, name, t):
        """Enqueue a device -> host copy into a reusable pinned buffer (pageable if pinning is refused); no synchronisation."""
        import torch
        bufs = self.__dict__.setdefault("_hbufs", {})
        b = bufs.get(name)
        if b is None or b.shape != t.shape or b.dtype != t.dtype:
            try:
                b = torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)
            except RuntimeError:
                b = torch.empty(tuple(t.shape), dtype=t.dtype)
            bufs[name] = b
        b.copy_(t, non_blocking=True)
        return b

    # ------------------------------------------------------------------ VecEnv API
    def reset(self):
        obs = self.engine.rows(self.engine.reset()).cpu().numpy()
        self.engine.sync()
        self._ep_start[:] = time.time()
        self._needs_reset = False
        # DummyVecEnv.reset keeps what each env's reset() returned as info -- the reference returns _get_info() there (:504).  Built
        # for batches of the reference's size; a 65 536-env batch gets empty dicts (nothing in SB3's loops reads them; 24 keys x N dicts)
        self.reset_infos = self.reset_info_rows() if self.num_envs <= self.RESET_INFO_MAX_ENVS else [{} for _ in range(self.num_envs)]
        if self.norm_reward:                                  # VecNormalize.reset(): self.returns = np.zeros(self.num_envs)
            self.engine.vn_set(returns=np.zeros(self.num_envs))
        return self._obs_dict(obs)

    RESET_INFO_MAX_ENVS = 256

    def reset_info_rows(self, indices=None):
        """_get_info() of freshly reset envs (env/ptg_gym_env.py:251-278 with the state of :105-138): zero revenue terms, the flows of the
        single cooldown row the env starts in, the prices of its episode's first hour / day."""
        eng, spec = self.engine, self.spec
        st = {f: eng.get_state(f) for f in ("act_ep_d", "i", "meth_state", "current_action", "hot_cold", "T_cat", "market_set")}
        cooldown = spec.tables["cooldown"]
        rows = []
        for e in (range(self.num_envs) if indices is None else indices):
            m = spec.markets[int(st["market_set"][e])]
            d = int(st["act_ep_d"][e])
            row = cooldown[int(st["i"][e])]
            rows.append({"step": 0, "el_price_act": float(m["el"][d * 24]), "gas_price_act": float(m["gas"][d]), "eua_price_act": float(m["eua"][d]),
                         "Meth_State": int(st["meth_state"][e]), "Meth_Action": ACTIONS[int(st["current_action"][e])],
                         "Meth_Hot_Cold": int(st["hot_cold"][e]), "Meth_T_cat": float(st["T_cat"][e]),
                         "Meth_H2_flow": float(row[2]), "Meth_CH4_flow": float(row[3]), "Meth_H2O_flow": float(row[5]), "Meth_el_heating": float(row[6]),
                         "ch4_revenues [ct/h]": 0.0, "steam_revenues [ct/h]": 0.0, "o2_revenues [ct/h]": 0.0, "eua_revenues [ct/h]": 0.0,
                         "chp_revenues [ct/h]": 0.0, "elec_costs_heating [ct/h]": -0.0, "elec_costs_electrolyzer [ct/h]": -0.0,
                         "water_costs [ct/h]": -0.0, "reward [ct]": 0.0, "cum_reward": 0, "Pot_Reward": float(m["pot_rew"][d * 24]),
                         "Part_Full": float(m["part_full"][d * 24])})
        return rows

    def step_async(self, actions):
        a = np.asarray(actions)
        if self.action_type == "continuous":
            a = a.reshape(self.num_envs, -1)[:, 0]
        else:
            a = a.reshape(self.num_envs)
        np.copyto(self._act_buf, a, casting="unsafe")         # into the pinned block the kernel (or the H2D copy) reads
        self._actions = self._act_buf
        if self.norm_reward:
            return                                             # the device path enqueues in step_wait (_step_wait_device)
        if self._needs_reset:
            raise RuntimeError("PtGVecEnv: call reset() before step()")
        # enqueue the whole step now (ptg_step_host_begin): actions in, kernel(s), outputs back; step_wait collects it.  The stream is
        # torch's current stream at THIS call (a caller inside `with torch.cuda.stream(s)` gets its step on s)
        eng = self.engine
        slot = self._slot
        self._slot = (slot + 1) % len(self._blk)
        if self._lazy_info:
            self._info_slot ^= 1
            self._info_host, self._info_ptr = self._info_blocks[self._info_slot], self._info_ptrs[self._info_slot]
        rc = eng._L.ptg_step_host_begin(eng._h, self._act_ptr, self._act_kind, self._blk_ptr[slot], self._final_ptr, self._info_ptr, eng._stream())
        if rc:
            eng._chk(rc)
        self._in_flight = slot

    def _finish_infos(self, dones, n_done, final_mat):
        """Monitor / DummyVecEnv conventions for the envs whose episode ended; everything else keeps its persistent entry."""
        infos = self._infos
        if self._dirty:                                       # last step's finished envs: back to a plain entry
            for e in self._dirty:
                infos[e] = _InfoRow(self, e) if self._lazy_info else {}
            self._dirty = []
        if self.engine.eval_mode:
            if self._lazy_info:
                self._info_cur = self._info_host              # rows are read through _InfoRow on demand, in place (the next step fills the other block)
            else:
                info = self._info_host
                for e in range(self.num_envs):
                    infos[e] = self._info_dict(info[e])
        if n_done:
            r, l, ids = self.engine.finished_episodes()
            now = time.time()
            ep = {int(i): (float(rr), int(ll)) for rr, ll, i in zip(r, l, ids)}
            for e in np.nonzero(dones)[0]:
                d = dict(infos[e]) if self.engine.eval_mode else {}
                d["terminal_observation"] = self._obs_row_dict(final_mat[e])
                d["TimeLimit.truncated"] = False
                rr, ll = ep.get(int(e), (float("nan"), 0))
                d["episode"] = {"r": round(rr, 6), "l": ll, "t": round(now - self._t0, 6)}
                infos[e] = d
                self._ep_start[e] = now
                self._dirty.append(int(e))
        return infos

    def step_wait(self):
        if self._needs_reset:
            raise RuntimeError("PtGVecEnv: call reset() before step()")
        if self.norm_reward:
            return self._step_wait_device()
        eng = self.engine
        slot = self._in_flight
        if slot is None:
            raise RuntimeError("PtGVecEnv: step_wait() without step_async()")
        self._in_flight = None
        mat, rew, done, status = self._views[slot]
        if self._copy_out:                                    # small batch: nothing to overlap, one call (raises on an invalid action: reference IndexError)
            rc = eng._L.ptg_step_host_finish(eng._h, self._n_done_ref)
            if rc:
                eng._chk(rc)
            rews, dones, stat64 = rew.astype(np.float32), done.astype(bool), status.astype(np.int64)
        else:
            # phase 2 (ptg_step_host_tail): rewards, done flags and the METH_STATUS bytes are on the host; the observations of a large
            # batch are still crossing PCIe while the arrays below are made
            rc = eng._L.ptg_step_host_tail(eng._h, self._n_done_ref)
            if rc:
                eng._L.ptg_step_host_end(eng._h)
                eng._chk(rc)
            rews = rew.astype(np.float32)                     # always a new array
            dones = done.astype(bool)
            stat64 = status.astype(np.int64)
            # phase 3 (ptg_step_host_end): observations in; raises on an invalid action (reference: IndexError)
            rc = eng._L.ptg_step_host_end(eng._h)
            if rc:
                eng._chk(rc)
        self._old_reward = rews
        obs = self._obs_dict(mat, copy=self._copy_out, status=stat64)
        infos = self._finish_infos(dones, self._n_done.value, self._final_mat)
        if self._copy_out:
            infos = list(infos)                               # a new list object every step (entries shared), like DummyVecEnv's deepcopy'd buf_infos
        if self.noise_mode == "numpy":
            self._steps_since_refill += 1
            if self._steps_since_refill >= self._tape_len:
                self._refill_tape()
        return obs, rews, dones, infos

    def _step_wait_device(self):
        """norm_reward=True: the step and the VecNormalize kernels run on device tensors, then one synchronisation for the copies."""
        eng = self.engine
        obs_t, rew_t, done_t = eng.step(self._actions)
        raw_rew_t = rew_t
        rew_t = eng.vn_normalize(rew_t, done_t, training=self.training)      # same stream: moments, running statistics, clip(r / sqrt(var + eps))
        h_obs, h_rew, h_done = self._to_host("obs", eng.rows(obs_t)), self._to_host("rew", rew_t), self._to_host("done", done_t)
        h_raw = self._to_host("raw_rew", raw_rew_t)
        h_info = self._to_host("info", eng.info) if eng.info is not None else None
        eng.sync()                                            # raises on an invalid action (reference: IndexError)
        obs = h_obs.numpy()
        rews = h_rew.numpy().astype(np.float32)
        dones = h_done.numpy().astype(bool)
        self._old_reward = h_raw.numpy().astype(np.float32)
        if h_info is not None:
            self._info_host[...] = h_info.numpy()
        final = None
        n_done = int(dones.sum())
        if n_done:
            h_final = self._to_host("final", eng.rows(eng.final_obs))      # rare: only steps on which an episode ends
            eng.sync()
            final = h_final.numpy()
            if not self.training:                             # VecNormalize: self.returns[dones] = 0 also with frozen statistics
                _, ret = eng.vn_get()
                ret[dones] = 0.0
                eng.vn_set(returns=ret)
        infos = self._finish_infos(dones, n_done, final)
        if self.noise_mode == "numpy":
            self._steps_since_refill += 1
            if self._steps_since_refill >= self._tape_len:
                self._refill_tape()
        return self._obs_dict(obs), rews, dones, infos          # copies: the staging buffers are reused by the next step

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        """HipEngine.state_dict() plus the host side of the noise streams (noise="numpy": the per-env PCG64 generator states, the
        tape and the refill counter), so that a resumed env continues with the draws the reference env would see."""
        sd = {"engine": self.engine.state_dict(), "seed": self._seed, "noise_mode": self.noise_mode,
              "steps_since_refill": self._steps_since_refill, "needs_reset": self._needs_reset}
        if self._gens is not None:
            sd["generators"] = [g.bit_generator.state for g in self._gens]
            sd["tape"] = self._tape.copy()
        return sd

    def load_state_dict(self, sd):
        assert sd["noise_mode"] == self.noise_mode
        if self._needs_reset:
            self.reset()
        self.engine.load_state_dict(sd["engine"])
        self._seed = sd.get("seed")
        if "generators" in sd:
            for g, st in zip(self._gens, sd["generators"]):
                g.bit_generator.state = st
            self._tape = sd["tape"].copy()
        self._steps_since_refill = int(sd["steps_since_refill"])
        self._needs_reset = bool(sd["needs_reset"])

    def step_tensors(self, actions):
        """Device path: enqueue one step, return (obs, rewards, dones) ROCm tensors without synchronising."""
        return self.engine.step(actions, want_final=False)

    def close(self):
        self.engine.close()

    def get_attr(self, attr_name, indices=None):
        n = len(self._indices(indices))
        return [getattr(self, attr_name)] * n

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        raise NotImplementedError(f"PtGVecEnv holds no per-env Python objects (env_method {method_name!r})")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode=None):
        return None

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)


class PTGEnv:
    """Single-env Gymnasium-style adapter (the reference's env/ptg_gym_env.py:23 surface) over an N = 1 batch.

    `reset(seed=None, options=None) -> (obs, info)`, `step(action) -> (obs, reward, terminated, truncated, info)`.
    One reference env in a fresh process: its constructor takes eps_ind[0], every reset the next entry.  The kernels reset a
    finished env themselves, so the reset() that follows a terminated step hands out that already-prepared episode.
    """
    metadata = {"render_modes": ["None"]}

    def __init__(self, dict_input, train_or_eval="train", render_mode="None", device=0, noise="numpy"):
        assert train_or_eval in ["train", "eval"], 'ptg_gym_env.py error: train_or_eval must be either "train" or "eval".'
        spec = EnvSpec.from_dict_input(dict_input, "eval")      # info rows are always available to the adapter
        self._vec = PtGVecEnv(spec, 1, train_or_eval="eval", device=device, noise=noise)
        self._vec.engine.set_episode_plan(spec.eps_ind, first_ptr=1, stride=1)
        self.train_or_eval = train_or_eval
        self.render_mode = render_mode
        self.observation_space, self.action_space = self._vec.observation_space, self._vec.action_space
        self._pending = None

    def reset(self, seed=None, options=None):
        if seed is not None:
            self._vec.seed(seed)
        if self._pending is not None and seed is None:
            obs, self._pending = self._pending, None
        else:
            self._pending = None
            obs = {k: (v[0] if k != "METH_STATUS" else int(v[0])) for k, v in self._vec.reset().items()}
        return obs, self._reset_info()

    def _reset_info(self):
        return self._vec.reset_info_rows([0])[0]

    def step(self, action):
        a = np.asarray(action).reshape(-1)[:1]
        obs, rew, done, infos = self._vec.step(a)
        info = infos[0]
        terminated = bool(done[0])
        row = {k: (v[0] if k != "METH_STATUS" else int(v[0])) for k, v in obs.items()}
        if terminated:
            self._pending = row
            row = info.pop("terminal_observation")
            info.pop("episode", None)
            info.pop("TimeLimit.truncated", None)
        if self.train_or_eval == "train":
            info = {}
        return row, float(rew[0]), terminated, False, info

    def close(self):
        self._vec.close()


def sb3_flat_features(obs, raw_modified="mod", price_ahead=13, feature_major=False):
    """Observation matrix (torch tensor [N, F], or [F, N] with feature_major=True) -> the flat feature tensor SB3's
    `CombinedExtractor` builds from the dict observation: sub-spaces in sorted key order, every Box flattened, the
    `Discrete(6)` METH_STATUS one-hot encoded (40 columns for 'mod', 31 for 'raw').  Runs where `obs` lives, so a policy
    on the same GPU consumes the env's output without a host round trip."""
    import torch
    if feature_major:
        obs = obs.transpose(-1, -2)
    cols, _ = obs_columns(raw_modified, price_ahead)
    parts = []
    for key in sorted(cols):
        x = obs[..., cols[key]]
        if key == "METH_STATUS":
            x = torch.nn.functional.one_hot(x[..., 0].round().long(), num_classes=6).to(obs.dtype)
        parts.append(x)
    return torch.cat(parts, dim=-1)


def stats_table(info, done, stats_names=None):
    """The `stats` array of the reference's Postprocessing.test_performance (src/rl_utils.py:528-565) from a recorded info
    stream: info [T, 24] of ONE env (HipEngine.rollout_info(...)[3][:, e]), done [T]; rows of terminated steps stay zero, as
    there.  Returns {stats_name: array [T]} in the column order of EnvConfig.stats_names (src/rl_config_env.py:44-49)."""
    from .config import STATS_NAMES
    names = list(STATS_NAMES if stats_names is None else stats_names)
    a = np.asarray(info.cpu() if hasattr(info, "cpu") else info, dtype=np.float64).copy()
    d = np.asarray(done.cpu() if hasattr(done, "cpu") else done).astype(bool)
    assert a.ndim == 2 and a.shape[1] == len(INFO_KEYS) == len(names) and d.shape == (a.shape[0],)
    a[d] = 0.0
    return {nme: a[:, m] for m, nme in enumerate(names)}
