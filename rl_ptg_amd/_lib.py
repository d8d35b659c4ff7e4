"""ctypes binding of libptg_env.so (include/ptg_env.h).  There is no CPU fallback: if the HIP library is
missing or fails to load, importing code gets a loud error."""
import ctypes as C
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc", "ptg_env.hip")
HDR = os.path.join(ROOT, "include", "ptg_env.h")
LIB_PATH = os.environ.get("PTG_LIB_PATH") or os.path.join(PKG, "lib", "libptg_env.so")      # PTG_LIB_PATH: an experiment build of the same source
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall"]
N_PARTS = 9          # translation units of the parallel build: -DPTG_PART=0..8 (ptg_env.hip, "PTG_PART")

N_TABLES, N_INFO = 17, 24
ACT_I32, ACT_F32, ACT_I64 = 0, 1, 2
OUT_F32, OUT_F64 = 0, 1
OBS_ROW_MAJOR, OBS_FEATURE_MAJOR, OBS_SB3_FLAT, OBS_SPLIT = 0, 1, 2, 3

_D1 = ["noise"]
_I1 = ["eps_len_d", "sim_step", "time_step_op", "price_ahead"]
_D2 = ["convert_mol_to_Nm3", "H_u_CH4", "H_u_H2", "dt_water", "cp_water", "rho_water", "Molar_mass_CO2",
       "Molar_mass_H2O", "h_H2O_evap", "eeg_el_price", "heat_price", "o2_price", "water_price",
       "min_load_electrolyzer", "max_h2_volumeflow", "eta_CHP",
       "t_cat_standby", "t_cat_startup_cold", "t_cat_startup_hot"]
_I2 = ["time1_start_p_f", "time2_start_f_p", "time_p_f", "time_f_p", "time1_p_f_p", "time2_p_f_p",
       "time23_p_f_p", "time3_p_f_p", "time34_p_f_p", "time4_p_f_p", "time45_p_f_p", "time5_p_f_p",
       "time1_f_p_f", "time2_f_p_f", "time23_f_p_f", "time3_f_p_f", "time34_f_p_f", "time4_f_p_f",
       "time45_f_p_f", "time5_f_p_f", "i_fully_developed", "j_fully_developed"]
_D3 = ["el_l_b", "el_u_b", "gas_l_b", "gas_u_b", "eua_l_b", "eua_u_b", "T_l_b", "T_u_b", "h2_l_b", "h2_u_b",
       "ch4_l_b", "ch4_u_b", "h2_res_l_b", "h2_res_u_b", "h2o_l_b", "h2o_u_b", "heat_l_b", "heat_u_b"]
_I3 = ["raw_modified", "action_type", "train_or_eval", "eps_sim_steps"]
_D4 = ["state_change_penalty", "t_cat_initial"]
_I4 = ["out_dtype", "obs_layout"]
CONFIG_KEYS = _D1 + _I1 + _D2 + _I2 + _D3 + _I3 + _D4 + _I4


class PtgConfig(C.Structure):
    _fields_ = ([(k, C.c_double) for k in _D1] + [(k, C.c_int32) for k in _I1] + [(k, C.c_double) for k in _D2] +
                [(k, C.c_int32) for k in _I2] + [(k, C.c_double) for k in _D3] + [(k, C.c_int32) for k in _I3] +
                [(k, C.c_double) for k in _D4] + [(k, C.c_int32) for k in _I4])


class PtgTables(C.Structure):
    _fields_ = [("data_host", C.POINTER(C.c_double) * N_TABLES), ("rows", C.c_int32 * N_TABLES)]


class PtgMarket(C.Structure):
    _fields_ = [("n_hours", C.c_int32), ("el_host", C.POINTER(C.c_double)), ("pot_rew_host", C.POINTER(C.c_double)),
                ("part_full_host", C.POINTER(C.c_double)), ("n_days", C.c_int32), ("gas_host", C.POINTER(C.c_double)),
                ("eua_host", C.POINTER(C.c_double)), ("scenario", C.c_int32), ("reserved", C.c_int32),
                ("rew_l_b", C.c_double), ("rew_u_b", C.c_double), ("r_0", C.c_double)]


# state fields of ptg_get_state / ptg_set_state
STATE_FIELDS = {"meth_state": 0, "i": 1, "j": 2, "k": 3, "hot_cold": 4, "standby_tid": 5, "startup_tid": 6,
                "partial_tid": 7, "full_tid": 8, "current_action": 9, "act_ep_d": 10, "ep_ptr": 11,
                "noise_count": 12, "n_state_changes": 13, "market_set": 14, "T_cat": 32, "cum_rew": 33}

EXPORTS = ["ptg_abi_version", "ptg_create", "ptg_destroy", "ptg_num_envs", "ptg_obs_dim", "ptg_last_error",
           "ptg_set_market_assignment", "ptg_set_episode_plan", "ptg_set_noise_tape", "ptg_set_noise_rng", "ptg_set_global_env_offset", "ptg_set_feature_pitch", "ptg_fill_noise_tape",
           "ptg_get_noise_tape", "ptg_reset", "ptg_step", "ptg_rollout", "ptg_rollout_info", "ptg_rollout_launches", "ptg_step_host", "ptg_host_layout", "ptg_host_layout_ex", "ptg_step_host_begin", "ptg_step_host_tail", "ptg_step_host_end", "ptg_step_host_finish", "ptg_host_buffers_changed", "ptg_profile", "ptg_profile_read", "ptg_profile_read_ex", "ptg_finished_dropped", "ptg_steps_to_episode_end", "ptg_note_replays", "ptg_set_replay_proof", "ptg_sync", "ptg_get_state", "ptg_set_state",
           "ptg_finished_episodes", "ptg_vn_init", "ptg_vn_batch_moments", "ptg_vn_apply", "ptg_vn_get", "ptg_vn_set",
           "ptg_market_feature_series", "ptg_debug_get_index_lut", "ptg_debug_window_record"]


def build(force=False, verbose=False):
    """hipcc cross-compiles the extension for gfx950 in-tree (no GPU needed to build)."""
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    newest = max(os.path.getmtime(SRC), os.path.getmtime(HDR))
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = int(os.environ.get("PTG_BUILD_JOBS", "0")) or min(N_PARTS, os.cpu_count() or 1)
    if jobs <= 1:                                             # one translation unit: the whole file in one hipcc call
        cmd = [hipcc] + HIPCC_FLAGS + ["-shared", "-o", LIB_PATH, SRC]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return LIB_PATH
    # the same source compiled N_PARTS times, each time with a different share of the hot-kernel instantiations, `jobs` at a time
    obj_dir = os.path.join(os.path.dirname(LIB_PATH), "obj")
    os.makedirs(obj_dir, exist_ok=True)
    objs = [os.path.join(obj_dir, f"ptg_env_part{k}.o") for k in range(N_PARTS)]
    cmds = [[hipcc] + HIPCC_FLAGS + ["-Wno-unused-function", f"-DPTG_PART={k}", "-c", "-o", objs[k], SRC] for k in range(N_PARTS)]
    running, todo, failed = [], list(enumerate(cmds)), []
    while todo or running:
        while todo and len(running) < jobs:
            k, cmd = todo.pop(0)
            if verbose:
                print(" ".join(cmd))
            running.append((k, subprocess.Popen(cmd)))
        k, proc = running.pop(0)
        if proc.wait() != 0:
            failed.append(k)
    if failed:
        raise RuntimeError(f"hipcc failed on part(s) {failed} of {SRC}")
    link = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(rl_ptg_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    L.ptg_abi_version.restype = C.c_int
    import re
    want = int(re.search(r"#define PTG_ABI_VERSION (\d+)", open(HDR).read()).group(1))
    if L.ptg_abi_version() != want:      # e.g. a stale experiment build behind PTG_LIB_PATH
        raise RuntimeError(f"{LIB_PATH} has ABI version {L.ptg_abi_version()}, include/ptg_env.h declares {want}: rebuild it "
                           "(python -c 'import __graft_entry__ as g; g.build()')")
    vp, dp, u8p, i32p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    L.ptg_abi_version.restype = C.c_int
    L.ptg_create.argtypes = [C.POINTER(PtgConfig), C.POINTER(PtgTables), C.POINTER(PtgMarket), C.c_int, C.c_int, C.c_int,
                             C.POINTER(vp)]
    L.ptg_destroy.argtypes = [vp]
    L.ptg_destroy.restype = None
    L.ptg_num_envs.argtypes = [vp]
    L.ptg_obs_dim.argtypes = [vp]
    L.ptg_last_error.argtypes = [vp]
    L.ptg_last_error.restype = C.c_char_p
    L.ptg_set_market_assignment.argtypes = [vp, u8p]
    L.ptg_set_episode_plan.argtypes = [vp, dp, C.c_int, C.c_int64, C.c_int64]
    L.ptg_set_noise_tape.argtypes = [vp, dp, C.c_int]
    L.ptg_fill_noise_tape.argtypes = [vp, C.c_uint64, C.c_int, vp]
    L.ptg_set_noise_rng.argtypes = [vp, C.c_uint64]
    L.ptg_set_global_env_offset.argtypes = [vp, C.c_int64]
    L.ptg_set_feature_pitch.argtypes = [vp, C.c_int64]
    L.ptg_get_noise_tape.argtypes = [vp, dp]
    L.ptg_reset.argtypes = [vp, u8p, vp, vp]
    L.ptg_step.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.ptg_rollout.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.ptg_rollout_info.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.ptg_rollout_launches.argtypes = [vp, C.c_int]
    L.ptg_step_host.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.POINTER(C.c_int), vp]
    L.ptg_host_layout.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ptg_host_layout_ex.argtypes = [vp] + [C.POINTER(C.c_size_t)] * 4
    L.ptg_step_host_begin.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
    L.ptg_step_host_tail.argtypes = [vp, C.POINTER(C.c_int)]
    L.ptg_step_host_end.argtypes = [vp]
    L.ptg_step_host_finish.argtypes = [vp, C.POINTER(C.c_int)]
    L.ptg_host_buffers_changed.argtypes = [vp]
    L.ptg_profile.argtypes = [vp, C.c_int]
    L.ptg_profile_read.argtypes = [vp, dp, C.c_int, C.POINTER(C.c_int)]
    L.ptg_profile_read_ex.argtypes = [vp, dp, dp, dp, C.c_int, C.POINTER(C.c_int)]
    L.ptg_finished_dropped.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ptg_steps_to_episode_end.argtypes = [vp, C.POINTER(C.c_int)]
    L.ptg_note_replays.argtypes = [vp, C.c_int]
    L.ptg_set_replay_proof.argtypes = [vp, C.c_int]
    L.ptg_sync.argtypes = [vp, vp]
    L.ptg_get_state.argtypes = [vp, C.c_int, vp]
    L.ptg_set_state.argtypes = [vp, C.c_int, vp]
    L.ptg_finished_episodes.argtypes = [vp, dp, i32p, i32p, C.c_int, C.POINTER(C.c_int)]
    L.ptg_vn_init.argtypes = [vp, C.c_double, C.c_double, C.c_double]
    L.ptg_vn_batch_moments.argtypes = [vp, vp, vp, C.c_int, vp, vp]
    L.ptg_vn_apply.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp]
    L.ptg_vn_get.argtypes = [vp, dp, dp]
    L.ptg_vn_set.argtypes = [vp, dp, dp]
    L.ptg_market_feature_series.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    L.ptg_debug_get_index_lut.argtypes = [vp, dp, i32p, C.POINTER(C.c_int)]
    L.ptg_debug_window_record.argtypes = [vp, C.c_int, C.c_int, dp]
    for name in EXPORTS:
        getattr(L, name)
    _lib = L
    return L
