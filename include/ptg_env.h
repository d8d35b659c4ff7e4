/*
 * ptg_env.h -- C ABI of the MI355X-native batched Power-to-Gas environment (libptg_env.so).
 *
 * Drop-in boundary for the reference's hot path: N independent copies of
 *     PTGEnv.step / PTGEnv.reset            /root/reference/env/ptg_gym_env.py:336-481, :483-506
 * as the reference vectorises them with SB3's DummyVecEnv (src/rl_utils.py:448-453, :484): envs are stepped
 * in env order and a finished env is reset at once.  The Python side (rl_ptg_amd/vec_env.py) binds these
 * entry points with ctypes and presents the VecEnv / gym.Env surface; INTEGRATION.md shows the reference-side
 * binding.  Plain C types only: no torch / numpy types cross this boundary.
 *
 * Pointer conventions
 *   *_host : host memory, read (or written) synchronously during the call.
 *   *_dev  : device memory on the handle's GPU (e.g. torch.Tensor.data_ptr() of a ROCm tensor), accessed
 *            asynchronously on the hipStream_t passed as `stream` (NULL = the default stream).
 * Ownership: the caller owns every buffer it passes; the library owns its device-resident state, tables and
 * price series (copies made in ptg_create) until ptg_destroy.
 * Errors: every function returns 0 on success or a negative PTG_E_* code; ptg_last_error() gives the text.
 * No C++ exception crosses the boundary.  A handle is not thread-safe; work is stream-ordered.
 */
#ifndef PTG_ENV_H
#define PTG_ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTG_ABI_VERSION 5   /* 2: + ptg_rollout_launches, ptg_rollout_info, ptg_vn_*, PTG_OBS_SB3_FLAT;  3: + ptg_profile*, ptg_step_host, ptg_host_layout, PTG_OBS_SPLIT, ptg_market_feature_series;
                             * 4: + ptg_profile_read_ex, ptg_finished_dropped, ptg_host_buffers_changed, ptg_steps_to_episode_end,
                             *    ptg_host_layout_ex (status section), ptg_step_host_begin / _tail / _end / _finish, ptg_set_feature_pitch;
                             * 5: + ptg_note_replays, ptg_set_replay_proof (the hot kernels read the step count from the device state: captured launches can be replayed) */
#define PTG_N_TABLES 17
#define PTG_N_COLS 7
#define PTG_N_INFO 24
#define PTG_MAX_MARKET_SETS 4

enum {
    PTG_OK = 0,
    PTG_E_INVALID = -1,        /* bad argument / call order */
    PTG_E_HIP = -2,            /* a HIP runtime call failed (no device, out of memory, ...) */
    PTG_E_ACTION = -3,         /* a discrete action outside [-5, 4] reached a kernel (reference: IndexError, :347) */
    PTG_E_RANGE = -4           /* a price index left the series (reference: IndexError, :446-447) */
};

/* table ids: order of op_data_files, src/rl_utils.py:108-113 */
enum {
    PTG_T_STARTUP_COLD = 0, PTG_T_STARTUP_HOT, PTG_T_COOLDOWN, PTG_T_STANDBY_DOWN, PTG_T_STANDBY_UP,
    PTG_T_OP1_START_P, PTG_T_OP2_START_F, PTG_T_OP3_P_F, PTG_T_OP4_P_F_P_5, PTG_T_OP5_P_F_P_10,
    PTG_T_OP6_P_F_P_15, PTG_T_OP7_P_F_P_22, PTG_T_OP8_F_P, PTG_T_OP9_F_P_F_5, PTG_T_OP10_F_P_F_10,
    PTG_T_OP11_F_P_F_15, PTG_T_OP12_F_P_F_20
};

enum { PTG_ACT_I32 = 0, PTG_ACT_F32 = 1, PTG_ACT_I64 = 2 };   /* element type of the action buffer */
enum { PTG_OUT_F32 = 0, PTG_OUT_F64 = 1 };                    /* element type of obs / reward buffers */
/* Observation matrix layout.  ROW_MAJOR [N][F] is what DummyVecEnv hands to SB3 (one row per env).  FEATURE_MAJOR [F][N]
 * is the struct-of-arrays form the kernels write with fully coalesced stores (a wave writes 64 consecutive envs of one
 * feature); its transpose view is the same [N][F] matrix, e.g. torch: obs.t(). Rollouts: [T][N][F] resp. [T][F][N].
 * SB3_FLAT [N][F + 5] is the row SB3's CombinedExtractor builds from the Dict observation (the reference's policies are
 * "MultiInputPolicy"): sub-spaces concatenated in sorted key order, the Discrete(6) METH_STATUS one-hot encoded -- 40
 * columns for 'mod', 31 for 'raw' at price_ahead 13; ptg_obs_dim() reports the width.  Replaces: obs_as_tensor +
 * preprocess_obs + CombinedExtractor.forward on the caller's side.
 * SPLIT [N][16] (float32): the part of the SB3_FLAT row that depends on the env's own state, plus WHERE its market features are:
 *   columns 0-5 METH_STATUS one-hot, 6 T_CAT, 7 H2_in, 8 CH4_syn, 9 H2_res, 10 H2O_DE, 11 Elec_Heating, 12 sin, 13 cos,
 *   14 hour index, 15 day index -- the start of the env's 13-hour (2-day) windows in the normalised feature series of
 *   ptg_market_feature_series (index = market_set * series_length + hour or day; exact in float32 up to 2^24).
 * The 26 ('raw': 17) market features of a row are a function of that index alone, so a policy's first layer can be evaluated as
 *   W_env . row[0:14] + G[hour index] (+ G_day[day index]),  G = the market columns of W applied to every window of the series
 * (rl_ptg_amd/policy_split.py): 73 instead of 169 bytes per env-step leave the env kernel, and the first layer multiplies 14
 * instead of 40 inputs.  Replaces: the market sub-spaces of the Dict observation going through CombinedExtractor into the first
 * Linear of the reference's MultiInputPolicy (src/rl_config_agent.py:126-149). */
enum { PTG_OBS_ROW_MAJOR = 0, PTG_OBS_FEATURE_MAJOR = 1, PTG_OBS_SB3_FLAT = 2, PTG_OBS_SPLIT = 3 };

/* Constants of the env: the flat kwargs of Preprocessing.dict_env_kwargs (src/rl_utils.py:345-365), same names.
 * Replaces: the attribute set PTGEnv.__init__ copies from dict_input (env/ptg_gym_env.py:40). */
typedef struct ptg_config {
    double noise;                       /* config_env.yaml:28; informational for host tapes, sigma of ptg_fill_noise_tape */
    int32_t eps_len_d, sim_step, time_step_op, price_ahead;
    double convert_mol_to_Nm3, H_u_CH4, H_u_H2, dt_water, cp_water, rho_water, Molar_mass_CO2,
           Molar_mass_H2O, h_H2O_evap, eeg_el_price, heat_price, o2_price, water_price,
           min_load_electrolyzer, max_h2_volumeflow, eta_CHP;
    double t_cat_standby, t_cat_startup_cold, t_cat_startup_hot;
    int32_t time1_start_p_f, time2_start_f_p, time_p_f, time_f_p, time1_p_f_p, time2_p_f_p, time23_p_f_p,
            time3_p_f_p, time34_p_f_p, time4_p_f_p, time45_p_f_p, time5_p_f_p, time1_f_p_f, time2_f_p_f,
            time23_f_p_f, time3_f_p_f, time34_f_p_f, time4_f_p_f, time45_f_p_f, time5_f_p_f,
            i_fully_developed, j_fully_developed;
    double el_l_b, el_u_b, gas_l_b, gas_u_b, eua_l_b, eua_u_b, T_l_b, T_u_b, h2_l_b, h2_u_b, ch4_l_b,
           ch4_u_b, h2_res_l_b, h2_res_u_b, h2o_l_b, h2o_u_b, heat_l_b, heat_u_b;
    int32_t raw_modified;               /* 0 = "raw" (26 features), 1 = "mod" (35 features); :165,185 */
    int32_t action_type;                /* 0 = "discrete", 1 = "continuous"; :144-158 */
    int32_t train_or_eval;              /* 0 = "train", 1 = "eval" (info rows available); :471-474 */
    int32_t eps_sim_steps;              /* :508-511 */
    double state_change_penalty;        /* :332 */
    double t_cat_initial;               /* 16 in the reference (:117) */
    int32_t out_dtype;                  /* PTG_OUT_F32 | PTG_OUT_F64 */
    int32_t obs_layout;                 /* PTG_OBS_ROW_MAJOR | PTG_OBS_FEATURE_MAJOR | PTG_OBS_SB3_FLAT | PTG_OBS_SPLIT */
} ptg_config;

/* The 17 process tables (src/rl_utils.py:46-67): row-major [rows][7] = t, T_cat, n_h2, n_ch4, n_h2_res, m_h2o, P_el */
typedef struct ptg_tables {
    const double* data_host[PTG_N_TABLES];
    int32_t rows[PTG_N_TABLES];
} ptg_tables;

/* One business scenario's market view as 1-D series.  Replaces the materialised tensors
 *   e_r_b[c, i, t] == series_c[t + i]  (src/rl_utils.py:250-263)   g_e[c, i, d] == series_c[d + i]  (:266-281)
 * plus the scenario-dependent scalars (b_s3 env/ptg_gym_env.py:76-77; rew_l_b/u_b src/rl_utils.py:378-379;
 * r_0 = reward_level[0] env/ptg_gym_env.py:125).  All sets of one handle share n_hours / n_days. */
typedef struct ptg_market {
    int32_t n_hours;                    /* >= last hour index used + price_ahead */
    const double* el_host;              /* ct/kWh */
    const double* pot_rew_host;         /* ct/h   (calculate_optimum, src/rl_opt.py:26-152, column 20) */
    const double* part_full_host;       /* -1/0/1 (column 23) */
    int32_t n_days;
    const double* gas_host;             /* ct/kWh, scenario override applied (src/rl_utils.py:119-126) */
    const double* eua_host;             /* Euro/t */
    int32_t scenario;                   /* 1, 2 or 3 */
    int32_t reserved;
    double rew_l_b, rew_u_b, r_0;
} ptg_market;

typedef struct ptg_env ptg_env;

/* ---- life cycle ---------------------------------------------------------------------------------------- */
/* Builds the device-resident tables for step_size = sim_step / time_step_op:
 *   window records (row 12 of SURVEY §8a: T of the last row + NumPy-pairwise means of the 5 flow columns for every
 *   possible window start, incl. the table-end / startup->partial splice cases of _perform_sim_step :525-557) and
 *   the _get_index lookup (:514-523) for every distinct catalyst temperature x 6 destination tables.
 * Replaces PTGEnv.__init__ (:28-79) for n_envs envs.  Envs must be reset before the first step. */
int ptg_create(const ptg_config* cfg, const ptg_tables* tables, const ptg_market* sets, int n_sets,
               int n_envs, int device_id, ptg_env** out);
void ptg_destroy(ptg_env* env);
int ptg_abi_version(void);
int ptg_num_envs(const ptg_env* env);
int ptg_obs_dim(const ptg_env* env);                 /* 35 ('mod') / 26 ('raw') for price_ahead = 13 */
const char* ptg_last_error(const ptg_env* env);      /* env may be NULL: error of the last failed ptg_create */

/* ---- configuration of the batch ------------------------------------------------------------------------ */
/* market set (business scenario) of every env; default 0 */
int ptg_set_market_assignment(ptg_env* env, const uint8_t* set_of_env_host);
/* Training episodes (env/ptg_gym_env.py:59-62, :490-493): eps_ind as the reference holds it.  Env e takes
 * eps_ind[(first_ptr + e + m*stride) mod n] at its m-th reset from now on: with first_ptr = N_total + shard_offset and
 * stride = N_total this is the order in which N_total reference envs sharing the module-global ep_index reset under
 * DummyVecEnv when they terminate together (exclusive prefix sum over done flags).  n = 0: validation/test env. */
int ptg_set_episode_plan(ptg_env* env, const double* eps_ind_host, int n, int64_t first_ptr, int64_t stride);
/* Normal draws consumed at state changes (:584-585,598-599,620-621): the c-th draw of env e is tape[e*len + c mod len].
 * Host tape (e.g. numpy Generator.normal(0, noise) per env for bit parity with the reference), env-major [n_envs][per_env_len] as
 * written here; the library keeps it draw-major on the device (ptg_set / get_noise_tape transpose, a temporary device buffer of the
 * tape's size while they run) ... */
int ptg_set_noise_tape(ptg_env* env, const double* tape_host, int per_env_len);
/* ... or the device's counter-based generator: the c-th draw of the env with GLOBAL index g is
 *   noise(seed, g, c) = cfg.noise * BoxMuller(u1, u2),  (u1, u2) from three rounds of the 32-bit integer finaliser "lowbias32"
 *   keyed by (seed, g, c); Box-Muller in float32 with the hardware log2 / sqrt / cos instructions.
 * ptg_set_noise_rng draws it inside the step kernels (no tape, unbounded); ptg_fill_noise_tape writes the first per_env_len
 * draws of the same streams to the tape (so both modes give identical trajectories while the tape does not wrap).
 * Both reset the per-env draw counters.  Statistically equivalent to, not bit-equal with, NumPy's Generator.normal. */
int ptg_set_noise_rng(ptg_env* env, uint64_t seed);
int ptg_fill_noise_tape(ptg_env* env, uint64_t seed, int per_env_len, void* stream);
/* FEATURE_MAJOR outputs only: elements between two feature planes of the caller's observation buffers (obs_dev, final_obs_dev, the
 * observation section of a host block; rollouts: [T][F][pitch]), n_envs <= pitch <= n_envs + 2^20; default n_envs (planes back to back).
 * Element (t, q, e) lives at ((t * F + q) * pitch + e).  Why: with float64 outputs and a power-of-two batch the planes are 2^19 bytes
 * apart and the 35 stores of a wave differ only above bit 19 -- measured 0.59 of the HBM peak; a pitch of n_envs + 128 elements
 * (1 KiB) brings 0.70 (profiles/r03_fm_pitch.txt: pads of 256 B .. 130 KiB; 4 KiB and 64 KiB multiples do not help) for a consumer
 * that can read a pitched matrix (torch: storage [F, pitch], view [:, :n_envs]).  float32 planes (2^18 bytes apart) do not need it.
 * No reference counterpart (the reference has no batch dimension). */
int ptg_set_feature_pitch(ptg_env* env, int64_t pitch_elems);
/* global index of this handle's env 0 (multi-GPU shards): keys the RNG streams; default 0 */
int ptg_set_global_env_offset(ptg_env* env, int64_t offset);
int ptg_get_noise_tape(ptg_env* env, double* tape_host);          /* [n_envs][per_env_len] */

/* ---- the hot path --------------------------------------------------------------------------------------- */
/* reset (:483-506) of all envs (mask_host NULL) or of those with mask != 0; obs rows of reset envs are written. */
int ptg_reset(ptg_env* env, const uint8_t* mask_host, void* obs_dev, void* stream);
/* One vector step (:336-481) + DummyVecEnv auto-reset.
 *   actions_dev  [N]     int32 / float32 / int64 per action_kind (PTG_ACT_*)
 *   obs_dev      [N][F]  out_dtype ([F][N] when cfg.obs_layout is FEATURE_MAJOR; final_obs_dev alike);
 *                        row of a finished env = observation after its reset
 *   rew_dev      [N]     out_dtype
 *   done_dev     [N]     uint8
 *   final_obs_dev[N][F]  (nullable) rows of finished envs = terminal observation
 *   info_dev     [N][24] (nullable, float64) _get_info (:251-278) in key order, Meth_Action as its index */
int ptg_step(ptg_env* env, const void* actions_dev, int action_kind, void* obs_dev, void* rew_dev,
             uint8_t* done_dev, void* final_obs_dev, double* info_dev, void* stream);
/* T vector steps fused, state held in registers: actions [T][N] -> obs [T][N][F], rew [T][N], done [T][N].
 * Same results as T calls of ptg_step.  One kernel launch covers up to 65 536 envs and as many steps as fit its LDS action
 * stage (a few hundred); wider batches / longer rollouts are issued as consecutive launches on `stream`. */
int ptg_rollout(ptg_env* env, const void* actions_dev, int action_kind, int n_steps, void* obs_dev, void* rew_dev,
                uint8_t* done_dev, void* stream);
/* ptg_rollout that also records the 24 _get_info fields of every step: info_dev [T][N][24] float64 (key order of
 * env/ptg_gym_env.py:251-278, Meth_Action as its index).  Replaces: the per-step info dicts Postprocessing.test_performance
 * collects into its stats array (src/rl_utils.py:528-565).  float64 outputs: the fused rollout kernel writes the rows (it evaluates
 * the reference-order reward terms anyway); float32 outputs: the generic step kernel, T launches. */
int ptg_rollout_info(ptg_env* env, const void* actions_dev, int action_kind, int n_steps, void* obs_dev, void* rew_dev,
                     uint8_t* done_dev, double* info_dev, void* stream);
/* One vector step with HOST buffers in and out -- the call behind VecEnv.step_wait.  Replaces DummyVecEnv.step_wait's loop
 * `for env_idx: obs, rew, terminated, truncated, info = envs[env_idx].step(actions[env_idx])` + `_save_obs` (SB3 dummy_vec_env.py,
 * as the reference builds it in src/rl_utils.py:448-453).
 *   actions_host  [N] of action_kind
 *   out_host      one block: observations [N][F] (layout per cfg.obs_layout) at offset 0, rewards [N] at off_rew, done flags
 *                 [N] uint8 at off_done (ptg_host_layout gives the offsets and the total size; all 16-byte aligned)
 *   final_obs_host [N][F] (nullable): rows of the envs whose episode ended = terminal observation; written only when *n_done > 0
 *   info_host     [N][24] float64 (nullable; needs cfg.train_or_eval = 1)
 *   n_done        number of envs whose episode ended on this step
 * Synchronises `stream` before it returns and reports kernel-flagged errors (PTG_E_ACTION / PTG_E_RANGE) like ptg_sync.
 * When the blocks are pinned, device-mapped host memory (hipHostMalloc, torch pin_memory) and small (<= 256 KiB in all) the
 * kernels read and write them in place -- no copies; otherwise the library stages through device buffers with one copy each way. */
int ptg_host_layout(const ptg_env* env, size_t* off_rew, size_t* off_done, size_t* total);
int ptg_step_host(ptg_env* env, const void* actions_host, int action_kind, void* out_host, void* final_obs_host, double* info_host,
                  int* n_done, void* stream);
/* The out_host block has a fourth section since ABI 4 (ptg_host_layout's `total` includes it): "status" [N] uint8 at off_status = the
 * METH_STATUS of every env's returned observation row, contiguous -- what a NumPy caller turns into the int64 METH_STATUS vector of
 * the Dict observation (:219-249) without gathering one column out of N rows. */
int ptg_host_layout_ex(const ptg_env* env, size_t* off_rew, size_t* off_done, size_t* off_status, size_t* total);
/* ptg_step_host in three phases, for a caller with work of its own to overlap (VecEnv.step_async / step_wait):
 *   ptg_step_host_begin  enqueues everything on `stream` and returns: actions in, the step kernel(s), the outputs back -- as two copies
 *                        for staged batches, [rewards | done flags | status] (+ info rows) first, the observations behind them;
 *   ptg_step_host_tail   waits until rewards, done flags, status (and info rows) are in out_host and counts the finished envs -- the
 *                        observations of a large batch are still crossing PCIe while the caller works on the small part;
 *   ptg_step_host_end    waits for the observations, reports kernel-flagged errors (PTG_E_ACTION / PTG_E_RANGE), and fetches the terminal
 *                        observations when episodes ended.  One host step at a time per handle; ptg_step_host == begin + tail + end. */
int ptg_step_host_begin(ptg_env* env, const void* actions_host, int action_kind, void* out_host, void* final_obs_host, double* info_host,
                        void* stream);
int ptg_step_host_tail(ptg_env* env, int* n_done);
int ptg_step_host_end(ptg_env* env);
int ptg_step_host_finish(ptg_env* env, int* n_done);      /* tail + end in one call */
/* ptg_step_host remembers, per buffer ADDRESS (the last 8), whether the buffer is device-mapped pinned memory.  A buffer must stay
 * allocated / registered for as long as it is passed to ptg_step_host; a caller that frees one and later passes memory of another
 * kind at the same address calls this first (forgets the classifications). */
int ptg_host_buffers_changed(ptg_env* env);
/* hipGraph capture.  ptg_step / ptg_rollout enqueue kernels only (no synchronisation, no host round trip), so they can be captured on
 * `stream` -- e.g. together with the policy's forward pass, whose ~10 launches per step otherwise bound a device-resident collect loop
 * (profiles/r03_policy_loop.txt) -- and the captured launches can be REPLAYED: the hot kernels read the common step count from the device
 * state.
 *   * A captured ptg_step, by default, is the hot kernel alone: replay it at most ptg_steps_to_episode_end() - 1 times and make the
 *     episode's terminating step an eager call (a replay that reaches that step raises PTG_E_INVALID at the next synchronising call).
 *     After ptg_set_replay_proof(env, 1) a captured ptg_step is enqueued as the hot kernel, which does nothing when it finds the batch on
 *     the terminating step, plus the generic kernel behind it, which does nothing otherwise -- a replay takes the right one by itself,
 *     across episode ends, auto-reset (episode plan) and finished-episode list included, for the price of one empty launch per step
 *     (+1.5-2 us).  final_obs_dev of the captured call receives the terminal observations.
 *   * A captured ptg_rollout must not be replayed across an episode end (a fused launch cannot terminate; eager calls are cut there by
 *     the host): ptg_steps_to_episode_end() says how far it may go; a replay that runs over raises PTG_E_INVALID at the next
 *     synchronising call.
 *   * ptg_note_replays(env, n): after replaying captured launches that together advanced the batch by n vector steps BEYOND the first
 *     replay (the capture call counts as executed once, like an eager call), so that eager calls, ptg_steps_to_episode_end,
 *     ptg_rollout_launches and ptg_finished_episodes stay in step; the count wraps at the episode length.
 * Buffers are the graph's (fixed addresses); kernel-flagged errors surface at the next ptg_sync / ptg_step_host / ptg_finished_episodes.
 * No reference counterpart. */
int ptg_note_replays(ptg_env* env, int n_steps);
int ptg_set_replay_proof(ptg_env* env, int enable);
/* Number of kernel launches ptg_rollout(env, ..., n_steps, ...) would issue from the envs' current position (for
 * per-launch timing); negative PTG_E_* on a bad argument. */
int ptg_rollout_launches(ptg_env* env, int n_steps);
/* Per-launch device time of the hot kernels (bench.py's roofline figure; no reference counterpart -- the reference times
 * env.step with time.perf_counter at best).  ptg_profile(env, 1) starts a collection: every k_step_hot / k_rollout_pc launch
 * from then on carries a (start, stop) HIP event pair stamped at the kernel's own begin and end (hipExtLaunchKernelGGL), i.e.
 * what `rocprofv3 --kernel-trace` reports for the dispatch, without host launch latency in the interval.  ptg_profile(env, 0)
 * stops it.  ptg_profile_read waits for the recorded launches, returns their durations in microseconds in launch order
 * (count = min(launches, cap)) and clears the collection.  Launches being captured into a hipGraph must not be profiled. */
int ptg_profile(ptg_env* env, int enable);
int ptg_profile_read(ptg_env* env, double* us_host, int cap, int* count);
/* ptg_profile_read with the launches' helper kernel accounted for.  A rollout launch may run a table refresher beside it (k_refresh on
 * a stream forked from the caller's: the rolling passes of a long launch right after a synchronised reset; the pass at the head of a
 * launch is part of the rollout kernel itself).  Per recorded launch: us_host = the kernel's own duration, helper_us_host (nullable)
 * = its helper's duration or 0, span_us_host (nullable) = the length of the UNION of the two intervals, first start to last end --
 * the figure bench.py's roofline uses. */
int ptg_profile_read_ex(ptg_env* env, double* us_host, double* helper_us_host, double* span_us_host, int cap, int* count);
/* hipStreamSynchronize(stream) + report an error a kernel flagged (PTG_E_ACTION / PTG_E_RANGE). */
int ptg_sync(ptg_env* env, void* stream);

/* ---- state access (parity tests, checkpointing) -------------------------------------------------------- */
enum {
    PTG_F_METH_STATE = 0, PTG_F_I, PTG_F_J, PTG_F_K, PTG_F_HOT_COLD, PTG_F_STANDBY_TID, PTG_F_STARTUP_TID,
    PTG_F_PARTIAL_TID, PTG_F_FULL_TID, PTG_F_CURRENT_ACTION, PTG_F_ACT_EP_D, PTG_F_EP_PTR, PTG_F_NOISE_COUNT,
    PTG_F_N_STATE_CHANGES, PTG_F_MARKET_SET,       /* int32[N] */
    PTG_F_T_CAT = 32, PTG_F_CUM_REW                /* float64[N] */
};
int ptg_get_state(ptg_env* env, int field, void* out_host);
int ptg_set_state(ptg_env* env, int field, const void* in_host);

/* Episodes finished since the last call (Monitor's info["episode"]: r = sum of returned rewards, l = steps),
 * compacted on the device with a wave ballot prefix.  Returns up to cap entries and clears the list.  The list is a ring
 * of max(2 * n_envs, 1024) entries: when more episodes finish between two calls the oldest are dropped.  Synchronises the
 * device only if a launch that can finish episodes (a generic / terminating step) ran since the last call. */
int ptg_finished_episodes(ptg_env* env, double* returns_host, int32_t* lengths_host, int32_t* env_ids_host,
                          int cap, int* count);
/* When can the next episode end?  A batch whose envs share one step count (reset together, stepped together -- every batch until a
 * partial reset or ptg_set_state de-synchronises it) ends its episodes on ONE known vector step (:508-511): *steps = the number of
 * vector steps from now up to and including that one (>= 1).  *steps = 0: not known to the host (an episode may end on any step).
 * A sharded job uses it to issue the episodic-return all-gather only in windows that contain an episode boundary. */
int ptg_steps_to_episode_end(ptg_env* env, int* steps);
/* Finished episodes that were never handed out since ptg_create: overwritten in the ring before a query came, or cut off by a
 * query's `cap`.  0 for every caller that queries at least once per 2 * n_envs finished episodes with cap >= that. */
int ptg_finished_dropped(ptg_env* env, uint64_t* dropped_total);

/* ---- VecNormalize(env, norm_obs=False) reward normalisation on the device ------------------------------------------
 * Replaces: stable_baselines3.common.vec_env.VecNormalize.step_wait / _update_reward / normalize_reward and
 * RunningMeanStd.update (SB3 2.0.0a13, the reference's pin; wrapped around the env in src/rl_utils.py:453), over a
 * [T][N] reward matrix as ptg_rollout (T >= 1) or ptg_step (T = 1) writes it:
 *   returns = returns * gamma + reward;  running moments of `returns` updated with the step's batch mean / variance;
 *   reward_out = clip(reward / sqrt(var + epsilon), +-clip_reward);  returns[done] = 0.
 * Two phases so that a job sharded over GPUs normalises with the moments of ALL envs: ptg_vn_batch_moments advances this
 * handle's returns and yields per-step (count, mean, M2) of its envs; the caller merges the shards' moments (Chan's formula,
 * rl_ptg_amd.dist.merge_moments -- one all-gather per rollout) and hands the merged [T][3] array to ptg_vn_apply, which
 * updates the running statistics step by step and writes the normalised rewards.  moments_dev NULL = single-GPU: the
 * handle's own moments are used. */
int ptg_vn_init(ptg_env* env, double gamma, double epsilon, double clip_reward);      /* SB3 defaults: 0.99, 1e-8, 10.0 */
int ptg_vn_batch_moments(ptg_env* env, const void* rew_dev, const uint8_t* done_dev, int n_steps, double* moments_dev, void* stream);
int ptg_vn_apply(ptg_env* env, const void* rew_dev, int n_steps, const double* moments_dev, void* rew_out_dev, int training,
                 void* stream);
/* running statistics {mean, var, count} and the per-env discounted returns (either pointer may be NULL) */
int ptg_vn_get(ptg_env* env, double* stats3_host, double* returns_host);
int ptg_vn_set(ptg_env* env, const double* stats3_host, const double* returns_host);

/* The pre-normalised float32 market feature series the kernels read, as [n_sets][series length]: which = 0 Pot_Reward ('raw':
 * Elec_Price) hourly, 1 Part_Full hourly ('mod' only), 2 Gas_Price daily, 3 EUA_Price daily.  out_host NULL: only *count.
 * For the SPLIT layout's consumer (column 14 / 15 of a row index these arrays). */
int ptg_market_feature_series(ptg_env* env, int which, float* out_host, int cap, int* count);

/* diagnostics for tests: the device-built lookup products */
int ptg_debug_get_index_lut(ptg_env* env, double* T_values_host, int32_t* lut_host /*[6][nT]*/, int* n_T);
int ptg_debug_window_record(ptg_env* env, int table_id, int start_row, double* out7_host /*T_last, 5 means, key*/);

#ifdef __cplusplus
}
#endif
#endif /* PTG_ENV_H */
