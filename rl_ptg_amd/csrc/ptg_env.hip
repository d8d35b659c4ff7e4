// ptg_env.hip -- MI355X (gfx950) implementation of include/ptg_env.h: HIP kernels + C ABI.
//
// What replaces what (reference = /root/reference/env/ptg_gym_env.py):
//   k_build_records   _perform_sim_step's window (:525-557) + the five np.average reductions and the last-row
//                     catalyst temperature (:452-458), evaluated ONCE for every possible window start of every
//                     table, with NumPy's pairwise summation order -> one 64-byte record per start row.
//   k_build_argmin    _get_index (:514-523) for every distinct catalyst temperature x 6 destination tables.
//   k_step            step() (:336-481) + DummyVecEnv auto-reset (reset :483-506) over a struct-of-arrays of N envs:
//                     one lane per env, integer state machine (:339-440), one record gather, reward (:280-334),
//                     normalisation (:206-217), observation row (:219-249), info row (:251-278).
//   k_reset           reset() (:483-506).
//   k_step_hot / k_rollout_pc   the float32 hot path of the same step (one launch per step / T fused steps).
//   k_fill_noise      the normal(0, noise) draws of :585/:599/:621 as a counter-based device RNG (noise_draw).
//   k_vn_*            VecNormalize(norm_obs=False) reward normalisation over the [T][N] rewards.
// Built with -ffp-contract=off: the float64 expressions keep the reference's operand order.
#include "../../include/ptg_env.h"

#include <hip/hip_runtime.h>
#include <chrono>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

namespace {

constexpr int NT = PTG_N_TABLES;
constexpr int NC = PTG_N_COLS;
constexpr int N_DEST = 6;   // destination tables of _get_index: cooldown, standby_up, standby_down, startup_cold, startup_hot, op1_start_p
constexpr int DEST_TID[N_DEST] = {PTG_T_COOLDOWN, PTG_T_STANDBY_UP, PTG_T_STANDBY_DOWN, PTG_T_STARTUP_COLD,
                                  PTG_T_STARTUP_HOT, PTG_T_OP1_START_P};

// ------------------------------------------------------------------------------------------------ device data
struct alignas(64) Rec {      // one window start: 64 B = half an L2 line, never straddles a line
    double T;                 // catalyst temperature of the window's last row (:452)
    double m[5];              // np.average of n_h2, n_ch4, n_h2_res, m_h2o, P_el over the window (:454-458)
    int tkey;                 // index of T in the sorted distinct-temperature list
    int pad;
    double spare;
};
static_assert(sizeof(Rec) == 64, "Rec must be 64 bytes");

// Strength-reduced record of the float32 fast path.  The reward (:280-334) is linear in the three prices once the
// window is fixed: rew = base + ch4*(b_s3*k_chp + k_eua*eua) + c_gas*gas - c_el*el, and the electrolyzer efficiency
// polynomial (:311-317) depends on the window only -- so k_build_fast evaluates it once per window start.
struct alignas(64) RecFast {
    double base, ch4, c_gas, c_el;   // all pre-multiplied by sim_step/3600 except ch4 (raw mean methane flow)
    float feat[6];                   // normalised T_cat, H2, CH4, H2_res, H2O, el_heating (:212-217)
    int tkey;
    int pad;
};
static_assert(sizeof(RecFast) == 64, "RecFast must be 64 bytes");

// Per-env state, three arrays of naturally aligned structs (16-B / 16-B / 8-B lanes -> dwordx4 / dwordx2 accesses)
struct alignas(16) StA { int i, j, k; unsigned flags; };   // flags: [0:3) meth_state [3] hot_cold [4] standby=up [5] startup=hot
                                                            // [6:9) part_op [9:12) full_op [12:15) current_action [15:17) market set [17:32) T key
struct alignas(16) StB { double cum; int act_d; int nctr; };   // cum_rew (:330), act_ep_d (:61,492), noise draws consumed so far
struct alignas(8) StC { int nchg; int epp; };                  // state changes this episode (tracked when the penalty is on), pointer into
                                                               // eps_ind -- touched on penalised state changes / resets only

struct Regs {
    StA a; StB b; StC c;
    bool c_loaded, c_dirty;
};

struct DevParams {
    int fm_pitch;                // feature-major outputs: elements between two feature planes (>= N; ptg_set_feature_pitch)
    int N, S, sim_step, eps_sim_steps, PA, F, mod, eps_len_d;
    int E, ep_stride;                      // eps_ind length (0 = eval env), pointer stride (mod E)
    int noise_inline, track_changes;       // draw noise from the counter RNG in the kernel; maintain StC.nchg (penalty != 0)
    unsigned long long noise_seed;
    long long env_offset;                  // global index of env 0 of this shard (keys the RNG streams)
    double noise_sigma;
    int key_cold_max, key_hot_min, key_standby_max, key_init, i_reset, nT, tape_len;
    int n_hours, n_days, hstride, dstride;
    int t1_start_p_f, t2_start_f_p, t_p_f, t_f_p, t1_p_f_p, t2_p_f_p, t3_p_f_p, t34_p_f_p, t4_p_f_p, t45_p_f_p,
        t5_p_f_p, t1_f_p_f, t2_f_p_f, t23_f_p_f, t3_f_p_f, t34_f_p_f, t4_f_p_f, t45_f_p_f, t5_f_p_f, i_full, j_full;
    // reward / normalisation constants (:280-334, :206-217)
    double c_mol, Hu_ch4, Hu_h2, dt_cp_evap, heat_price, o2_price, eeg, eta_chp, one_m_eta_chp, M_co2, M_h2o,
           rho, water_price, min_load, max_h2, c_m2, c_m3, sim_step_d;
    double T_lo, T_rng, h2_lo, h2_rng, ch4_lo, ch4_rng, h2r_lo, h2r_rng, h2o_lo, h2o_rng, heat_lo, heat_rng;
    double reset_flow[5], T_init;
    double k_chp, k_eua;                   // fast path: per-unit-CH4 CHP revenue and EUA revenue factors (x sim_step/3600)
    // tables
    const Rec* rec;
    const RecFast* recf;
    const int2* tabmeta;                   // [17] {rows, record base}
    const int* argidx;                     // [6][nT]
    const double* Tvals;                   // [nT]
    const double* tape;                    // [tape_len][N] (draw-major, see k_fill_noise)
    const int* eps_ind;                    // [E]
    const double2* sincos;                 // [eps_sim_steps + 1]
    const float2* sincos32;
    // market, [set][...] with strides hstride / dstride
    const double *el, *featA, *featB, *gas, *eua, *gas_n, *eua_n;
    const float *featA32, *featB32, *gas_n32, *eua_n32;
    const double *pot_raw, *pf_raw;        // un-normalised pot_rew / part_full for info rows
    const double2* setc;                   // [sets] {b_s3, r_0 * state_change_penalty}
    // state
    StA* st_a; StB* st_b; StC* st_c;
    // finished-episode list
    double* fin_ret; int* fin_len; int* fin_env; int* fin_count; int fin_cap;
    const int* cmap; int q_stat;          // SB3_FLAT layout: canonical column -> flat column; canonical index of METH_STATUS (else cmap = null)
    int split;                            // SPLIT layout (16 columns: status one-hot, 8 env features, hour / day series index; q_stat set too)
    int* err;                             // [2] in pinned HOST memory: {invalid action seen, price index out of range}; kernels store 1 (plain
                                          // stores of a constant need no atomic), the host reads it after a stream synchronise -- no copy
    int* term_flag;                       // device word: "the hot step kernel of this (captured) step found the batch on the terminating step
                                          // and skipped it" -- written by k_step_hot, read by the k_step enqueued behind it
};

__device__ __forceinline__ int part_tid(int p) { return p == 0 ? PTG_T_OP1_START_P : 7 + p; }          // 5, 8..12
__device__ __forceinline__ int full_tid(int q) { return q == 0 ? PTG_T_OP2_START_F : (q == 1 ? PTG_T_OP3_P_F : 11 + q); }  // 6,7,13..16

// ------------------------------------------------------------------------------------------------ table builders
struct WinSrc {               // the S-row window _perform_sim_step would hand to step(), read in place
    const double* tab;        // this table's rows
    const double* nxt;        // op1_start_p rows (startup tables splice into it, :547-550)
    int n, r, S;
    bool splice, all_last;
    __device__ double at(int q, int col) const {
        if (all_last) return tab[(size_t)(n - 1) * NC + col];
        int v = r + q;
        if (v < n) return tab[(size_t)v * NC + col];
        return splice ? nxt[(size_t)(v - n) * NC + col] : tab[(size_t)(n - 1) * NC + col];
    }
};

// NumPy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src): blocks of <=128 with 8 accumulators.
__device__ double pairwise_sum(const WinSrc& w, int col, int off, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int q = 0; q < n; q++) res += w.at(off + q, col);
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int q = 0; q < 8; q++) r[q] = w.at(off + q, col);
        int q8 = 8;
        for (; q8 < n - (n % 8); q8 += 8)
            for (int q = 0; q < 8; q++) r[q] += w.at(off + q8 + q, col);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; q8 < n; q8++) res += w.at(off + q8, col);
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(w, col, off, n2) + pairwise_sum(w, col, off + n2, n - n2);
    }
}

__global__ void k_build_records(const double* __restrict__ tab, const int* __restrict__ rowkey, int n,
                                const double* __restrict__ nxt, const int* __restrict__ nxtkey, int splice, int S,
                                Rec* __restrict__ out)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    WinSrc w{tab, nxt, n, r, S, splice != 0, r == n};
    Rec rec;
    for (int c = 0; c < 5; c++) {
        double s = 0.0 + pairwise_sum(w, 2 + c, 0, S);
        rec.m[c] = s / (double)S;
    }
    rec.T = w.at(S - 1, 1);
    int last = r + S - 1;
    rec.tkey = (r == n || last < n) ? rowkey[min(last, n - 1)] : (splice ? nxtkey[last - n] : rowkey[n - 1]);
    rec.pad = 0;
    rec.spare = 0.0;
    out[r] = rec;
}

// first index of min |T_r - Tq|  (ndarray.argmin keeps the first minimum).  One wave per temperature key: lanes stride over the
// table rows (each keeps its first strict minimum), then a butterfly picks the smallest distance, ties to the smaller row.
__global__ void __launch_bounds__(256)
k_build_argmin(const double* __restrict__ tab, int n, const double* __restrict__ Tvals, int nT, int* __restrict__ out)
{
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nT) return;                                    // whole wave
    const double t = Tvals[q];
    int best = 0x7FFFFFFF;
    double bd = INFINITY;
    for (int r = lane; r < n; r += 64) {
        const double d = fabs(tab[(size_t)r * NC + 1] - t);
        if (d < bd) { bd = d; best = r; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double od = __shfl_xor(bd, off, 64);
        const int ob = __shfl_xor(best, off, 64);
        if (od < bd || (od == bd && ob < best)) { bd = od; best = ob; }
    }
    if (lane == 0) out[q] = best;
}

// ------------------------------------------------------------------------------------------------ noise tape RNG
// c-th normal(0, sigma) draw of global env eg: a counter-based generator built from three rounds of the 32-bit
// integer finaliser "lowbias32" (x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16) keyed by
// (seed, eg, c), then Box-Muller in float32 with the hardware's log2 / sqrt / cos instructions (~30 VALU instead of ~150 for Philox4x32-10 +
// float64 Box-Muller; the draw only jitters a table row index by ~10 rows).  The same function fills tapes
// (k_fill_noise) and draws in-kernel, so both modes agree bit for bit.
__device__ __forceinline__ unsigned lowbias32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ double noise_draw(unsigned long long seed, long long eg, unsigned c, double sigma)
{
    const unsigned k0 = lowbias32((unsigned)eg ^ (unsigned)seed);
    const unsigned k1 = lowbias32(k0 + c * 0x9E3779B9u + (unsigned)(seed >> 32) + (unsigned)(eg >> 32) * 0x85EBCA6Bu);
    const unsigned k2 = lowbias32(k1 ^ 0xC2B2AE35u);
    const float u1 = ((float)(k1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(k2 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    // the hardware's own log2 / sqrt / cos-of-revolutions (v_log_f32, v_sqrt_f32, v_cos_f32: one instruction each; u1 is never a denormal
    // and u2 lies in (0, 1), so the library versions' range handling -- ~45 of the 75 instructions of a draw -- bought nothing):
    // -2 ln u1 = -2 ln 2 * log2 u1
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    return sigma * (double)(r * __builtin_amdgcn_cosf(u2));
}

// The device tape is draw-major, [L][N]: lane e of a wave reads draw (noise_count[e] % L) of env e, and envs that consumed the same
// number of draws -- most of a synchronised batch -- then share cache lines (env-major rows of L doubles put every lane on a line
// of its own, 8 KiB apart at L = 1024).  The C ABI keeps the env-major [N][L] host layout (ptg_set / get_noise_tape transpose).
__global__ void k_fill_noise(double* __restrict__ tape, int N, int L, unsigned long long seed, long long env_offset, double sigma)
{
    long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)N * L) return;
    tape[g] = noise_draw(seed, g % N + env_offset, (unsigned)(g / N), sigma);
}

// in [rows][cols] -> out [cols][rows], 32 x 32 tiles through LDS (block 32 x 8)
__global__ void k_transpose(const double* __restrict__ in, double* __restrict__ out, int rows, int cols)
{
    __shared__ double tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[threadIdx.x][j];
    }
}

// ------------------------------------------------------------------------------------------------ the env step
__device__ __forceinline__ void load_regs(const DevParams& P, int e, Regs& R)
{
    R.a = P.st_a[e]; R.b = P.st_b[e];
    R.c_loaded = false; R.c_dirty = false;
}
__device__ __forceinline__ void need_c(const DevParams& P, int e, Regs& R)
{
    if (!R.c_loaded) { R.c = P.st_c[e]; R.c_loaded = true; }
}
__device__ __forceinline__ void store_regs(const DevParams& P, int e, const Regs& R)
{
    P.st_a[e] = R.a; P.st_b[e] = R.b;
    if (R.c_dirty) P.st_c[e] = R.c;
}

// :351-355: prob_thre[ival] = -1 + ival*0.4 in float64 (:151-155); the first threshold > action picks actions[ival-1]
// (python index -1 = full_load); no threshold fires (a >= 1, NaN) -> the previous action stays
__device__ __forceinline__ int decode_continuous(float af, int previous)
{
    const double a = (double)af;
    const double ival = (1.0 - (-1.0)) / 5;
    int out = previous;
    bool hit = false;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const double thr = -1.0 + q * ival;
        const bool fire = (!hit) & (thr > a);
        out = fire ? ((q == 0) ? 4 : q - 1) : out;
        hit = hit | fire;
    }
    return out;
}

// :346-357 -> action id 0..4, or -1 for an invalid discrete action
__device__ __forceinline__ int decode_action(const void* actions, int kind, size_t idx, int previous)
{
    if (kind == PTG_ACT_F32) return decode_continuous(((const float*)actions)[idx], previous);
    long long a = (kind == PTG_ACT_I64) ? ((const long long*)actions)[idx] : (long long)((const int*)actions)[idx];
    if (a < -5 || a > 4) return -1;
    return (int)(a < 0 ? a + 5 : a);      // python list indexing: actions[-1] is full_load
}

// Where the small lookup tables live: global memory (L2-resident) or LDS copies staged by the workgroup.
struct LutGlobal {            // _get_index lookup [6][nT] int32
    const int* p; int nT;
    __device__ __forceinline__ int get(int dest, int tkey) const { return p[dest * nT + tkey]; }
};
// Integer state machine of step() (:339-440) and _perform_sim_step (:525-557).  Returns the record index.
// The two memory lookups a state change needs (_get_index entry, noise draw) are issued BEFORE the branchy part so
// that they overlap; the branches then hold integer arithmetic only.
template <class LUT>
__device__ __forceinline__ int step_ints(const DevParams& P, const int2* tabmeta, const LUT& lut, Regs& R, int act, int e, bool& changed)
{
    unsigned f = R.a.flags;
    int s = f & 7, hot = (f >> 3) & 1, sb = (f >> 4) & 1, su = (f >> 5) & 1, pp = (f >> 6) & 7, fq = (f >> 9) & 7;
    const unsigned mset = (f >> 15) & 3;
    const int tkey = (int)(f >> 17);
    // :339-342 hot/cold hysteresis on the previous catalyst temperature
    if (tkey <= P.key_cold_max) hot = 0;
    else if (tkey >= P.key_hot_min) hot = 1;
    const int prev = s;
    const int S = P.S;
    int i = R.a.i, j = R.a.j, table;
    const int time_op = i + j * S;
    // :368-440 dispatch
    int kind;   // 0 continue, 1 _standby, 2 _cooldown, 3 _startup, 4 _partial, 5 _full
    if (act == 0) kind = (s == 0) ? 0 : 1;
    else if (act == 1) kind = (s == 1) ? 0 : 2;
    else if (act == 2) kind = (s <= 1) ? 3 : 0;
    else if (act == 3) kind = (s == 4) ? 4 : 0;
    else kind = (s == 3) ? 5 : 0;
    // lookups for the transition, hoisted: destination table of _get_index and the noise draw
    const bool sb_new = (tkey <= P.key_standby_max);
    const bool noisy = (kind >= 1 && kind <= 3);
    const int dest = (kind == 1) ? (sb_new ? 1 : 2) : (kind == 3) ? (hot ? 4 : 3) : (kind == 4) ? 5 : 0;
    const int idx = lut.get(dest, tkey);
    double z = 0.0;
    if (noisy) {
        if (P.tape_len > 0) z = P.tape[(size_t)((unsigned)R.b.nctr % (unsigned)P.tape_len) * P.N + e];
        else if (P.noise_inline) z = noise_draw(P.noise_seed, P.env_offset + e, (unsigned)R.b.nctr, P.noise_sigma);
        R.b.nctr += 1;
    }

    if (kind == 0) {                                       // _cont (:559-570)
        table = (s == 0) ? (sb ? PTG_T_STANDBY_UP : PTG_T_STANDBY_DOWN)
              : (s == 1) ? PTG_T_COOLDOWN
              : (s == 2) ? (su ? PTG_T_STARTUP_HOT : PTG_T_STARTUP_COLD)
              : (s == 3) ? part_tid(pp) : full_tid(fq);
        j += 1;
    } else if (noisy) {                                    // _standby / _cooldown / _startup (:572-625)
        if (kind == 1) { s = 0; sb = sb_new; table = sb ? PTG_T_STANDBY_UP : PTG_T_STANDBY_DOWN; }
        else if (kind == 2) { s = 1; table = PTG_T_COOLDOWN; }
        else { s = 2; pp = 0; fq = 0; su = hot; table = su ? PTG_T_STARTUP_HOT : PTG_T_STARTUP_COLD; }
        double x = (double)idx + z;                        // int(max(idx + normal, 0)) (:584-585)
        if (0 > x) x = 0;
        i = (int)x;
        j = 1;
    } else if (kind == 4) {                                // _partial (:627-691)
        s = 3;
        if (fq == 0) {
            if (time_op < P.t2_start_f_p) { pp = 0; i = idx; j = 1; }
            else { pp = 5; i = 0; j = 1; }
        } else if (fq == 1) {
            if (time_op < P.t1_p_f_p) { pp = 5; i = P.i_full; j = P.j_full; }
            else if (P.t1_p_f_p < time_op && time_op < P.t2_p_f_p) { pp = 1; j += 1; }
            else if (P.t2_p_f_p < time_op && time_op < P.t_p_f) { pp = 1; i = P.t2_p_f_p; j = 1; }
            else if (P.t_p_f < time_op && time_op < P.t34_p_f_p) { pp = 2; i = P.t3_p_f_p; j = 1; }
            else if (P.t34_p_f_p < time_op && time_op < P.t45_p_f_p) { pp = 3; i = P.t4_p_f_p; j = 1; }
            else if (P.t45_p_f_p < time_op && time_op < P.t5_p_f_p) { pp = 4; i = P.t5_p_f_p; j = 1; }
            else { pp = 5; i = 0; j = 1; }
        } else { pp = 5; i = 0; j = 1; }
        table = part_tid(pp);
    } else {                                               // _full (:693-757)
        s = 4;
        if (pp == 0) {
            fq = (time_op < P.t1_start_p_f) ? 0 : 1; i = 0; j = 1;
        } else if (pp == 5) {
            if (time_op < P.t1_f_p_f) { fq = 1; i = P.i_full; j = P.j_full; }
            else if (P.t1_f_p_f < time_op && time_op < P.t_f_p) { fq = 2; j += 1; }
            else if (P.t_f_p < time_op && time_op < P.t23_f_p_f) { fq = 2; i = P.t2_f_p_f; j = 1; }
            else if (P.t23_f_p_f < time_op && time_op < P.t34_f_p_f) { fq = 3; i = P.t3_f_p_f; j = 1; }
            else if (P.t34_f_p_f < time_op && time_op < P.t45_f_p_f) { fq = 4; i = P.t4_f_p_f; j = 1; }
            else if (P.t45_f_p_f < time_op && time_op < P.t5_f_p_f) { fq = 5; i = P.t5_f_p_f; j = 1; }
            else { fq = 1; i = 0; j = 1; }
        } else { fq = 1; i = 0; j = 1; }
        table = full_tid(fq);
    }
    // _perform_sim_step (:525-557) against the virtual table [rows | splice-or-last-row padding]
    const int2 tm = tabmeta[table];
    const int n = tm.x;
    const int start = i + (j - 1) * S;
    int r;
    if (start + S < n) {
        r = start;
    } else {
        const int over = start + S - n;
        if (table <= PTG_T_STARTUP_HOT) {                   // change_operation: startup -> partial load
            s = 3;
            if (over < S) { r = start; i = over; j = 0; }
            else r = n;
        } else {
            r = min(start, n);
        }
    }
    changed = (prev != s);
    R.a.i = i; R.a.j = j;
    R.a.flags = (unsigned)s | (hot << 3) | (sb << 4) | (su << 5) | (pp << 6) | (fq << 9) | ((unsigned)act << 12) |
                (mset << 15) | ((unsigned)tkey << 17);
    return tm.y + r;
}

// Observation matrix addressing: ROW_MAJOR row e = base + e*F (feature stride 1); FEATURE_MAJOR = base + e (stride N):
// lanes of a wave are consecutive envs, so every feature store is one contiguous 256-B (f32) / 512-B (f64) segment.
template <typename OUT, bool FM>
struct ObsRow {
    OUT* p; size_t stride; const int* cmap; int q_stat; bool split;
    __device__ __forceinline__ ObsRow(OUT* base, const DevParams& P, int e)
        : p(base ? (FM ? base + e : base + (size_t)e * P.F) : nullptr), stride(FM ? (size_t)P.fm_pitch : 1), cmap(FM ? nullptr : P.cmap), q_stat(P.q_stat),
          split(!FM && P.split) {}
    // q = canonical column (the reference's observation order); SB3_FLAT rows hold the columns in sorted-key order with
    // METH_STATUS one-hot over 6 classes
    __device__ __forceinline__ void put(int q, OUT v) const
    {
        if (!FM && split) {                     // the market columns are not stored: the consumer reads them through the series indices
            if (q < q_stat) return;
            if (q == q_stat) {
#pragma unroll
                for (int j = 0; j < 6; j++) p[j] = (OUT)(((int)v == j) ? 1 : 0);
            } else p[5 + (q - q_stat)] = v;     // canonical q_stat + 1 .. + 8 -> columns 6 .. 13
        } else if (!FM && cmap) {
            const int c = cmap[q];
            if (q == q_stat) {
#pragma unroll
                for (int j = 0; j < 6; j++) p[c + j] = (OUT)(((int)v == j) ? 1 : 0);
            } else p[c] = v;
        } else if (FM) __builtin_nontemporal_store(v, p + (size_t)q * stride);     // coalesced plane segments, never re-read: keep them out of L2
        else p[(size_t)q * stride] = v;
    }
    // SPLIT layout: where this env's 13-hour / 2-day windows start in the (market-set-major) feature series
    __device__ __forceinline__ void put_idx(unsigned hour_idx, unsigned day_idx) const
    {
        if (!FM && split) { p[14] = (OUT)hour_idx; p[15] = (OUT)day_idx; }
    }
    __device__ __forceinline__ void copy_row_from(const ObsRow& src, int F) const     // same layout on both sides
    {
        for (int q = 0; q < F; q++) p[(size_t)q * stride] = src.p[(size_t)q * src.stride];
    }
    __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
};

// The 2*PA (+4) market features of an observation row.  load() issues every read before the first store so that the
// loads pipeline (PAC = 13 compile-time: fully unrolled into registers; PAC = 0: generic price_ahead, rolled loop).
template <typename OUT, bool FAST, bool FM, int PAC>
struct PriceFeatures {
    typedef typename std::conditional<FAST, float, double>::type V;
    V a[PAC > 0 ? PAC : 1], b[PAC > 0 ? PAC : 1], d[4];
    size_t hb, db;
    __device__ __forceinline__ void load(const DevParams& P, unsigned mset, int H, int D)
    {
        hb = (size_t)mset * P.hstride + H; db = (size_t)mset * P.dstride + D;
        if (PAC > 0) {
            const V* pa = FAST ? (const V*)P.featA32 : (const V*)P.featA;
            const V* pb = FAST ? (const V*)P.featB32 : (const V*)P.featB;
#pragma unroll
            for (int q = 0; q < PAC; q++) a[q] = pa[hb + q];
            if (P.mod) {
#pragma unroll
                for (int q = 0; q < PAC; q++) b[q] = pb[hb + q];
            } else {
                const V* pg = FAST ? (const V*)P.gas_n32 : (const V*)P.gas_n;
                const V* pe = FAST ? (const V*)P.eua_n32 : (const V*)P.eua_n;
                d[0] = pg[db]; d[1] = pg[db + 1]; d[2] = pe[db]; d[3] = pe[db + 1];
            }
        }
    }
    __device__ __forceinline__ void store(const DevParams& P, const ObsRow<OUT, FM>& row) const
    {
        if (PAC > 0) {
#pragma unroll
            for (int q = 0; q < PAC; q++) row.put(q, (OUT)a[q]);
            if (P.mod) {          // Pot_Reward (normalised), Part_Full (:238-239)
#pragma unroll
                for (int q = 0; q < PAC; q++) row.put(PAC + q, (OUT)b[q]);
            } else {              // Elec_Price, Gas_Price, EUA_Price (:223-225)
#pragma unroll
                for (int q = 0; q < 4; q++) row.put(PAC + q, (OUT)d[q]);
            }
        } else {
            const int PA = P.PA;
            const V* pa = FAST ? (const V*)P.featA32 : (const V*)P.featA;
            const V* pb = FAST ? (const V*)P.featB32 : (const V*)P.featB;
            for (int q = 0; q < PA; q++) row.put(q, (OUT)pa[hb + q]);
            if (P.mod) {
                for (int q = 0; q < PA; q++) row.put(PA + q, (OUT)pb[hb + q]);
            } else {
                const V* pg = FAST ? (const V*)P.gas_n32 : (const V*)P.gas_n;
                const V* pe = FAST ? (const V*)P.eua_n32 : (const V*)P.eua_n;
                row.put(PA, (OUT)pg[db]); row.put(PA + 1, (OUT)pg[db + 1]);
                row.put(PA + 2, (OUT)pe[db]); row.put(PA + 3, (OUT)pe[db + 1]);
            }
        }
    }
};

// reset() (:483-506): takes the env's next episode, _initialize_op_rew (:105-138), writes the observation row
template <typename OUT, bool FAST, bool FM, int PAC>
__device__ __forceinline__ void reset_env(const DevParams& P, Regs& R, int e, const ObsRow<OUT, FM>& row)
{
    if (P.E > 0) {
        need_c(P, e, R);
        R.b.act_d = P.eps_ind[R.c.epp] * P.eps_len_d;
        R.c.epp += P.ep_stride;
        if (R.c.epp >= P.E) R.c.epp %= P.E;
        R.c_dirty = true;
    } else {
        R.b.act_d = 0;
    }
    const unsigned keep = R.a.flags & ((7u << 12) | (3u << 15));     // current_action survives reset(); market set is fixed
    R.a.flags = 1u | keep | ((unsigned)P.key_init << 17);            // cooldown, cold, standby_down, startup_cold, op1, op2
    R.a.i = P.i_reset; R.a.j = 0; R.a.k = 0; R.b.cum = 0.0;
    if (P.track_changes) { need_c(P, e, R); R.c.nchg = 0; R.c_dirty = true; }
    if (row) {
        const unsigned mset = (R.a.flags >> 15) & 3;
        PriceFeatures<OUT, FAST, FM, PAC> pf;
        pf.load(P, mset, R.b.act_d * 24, R.b.act_d);
        pf.store(P, row);
        row.put_idx(mset * (unsigned)P.hstride + (unsigned)(R.b.act_d * 24), mset * (unsigned)P.dstride + (unsigned)R.b.act_d);
        const int o = P.mod ? 2 * P.PA : P.PA + 4;
        row.put(o + 0, (OUT)1.0);
        row.put(o + 1, (OUT)((P.T_init - P.T_lo) / P.T_rng));
        row.put(o + 2, (OUT)((P.reset_flow[0] - P.h2_lo) / P.h2_rng));
        row.put(o + 3, (OUT)((P.reset_flow[1] - P.ch4_lo) / P.ch4_rng));
        row.put(o + 4, (OUT)((P.reset_flow[2] - P.h2r_lo) / P.h2r_rng));
        row.put(o + 5, (OUT)((P.reset_flow[3] - P.h2o_lo) / P.h2o_rng));
        row.put(o + 6, (OUT)((P.reset_flow[4] - P.heat_lo) / P.heat_rng));
        row.put(o + 7, (OUT)0.0);     // sin(0)
        row.put(o + 8, (OUT)1.0);     // cos(0)
    }
}

// Reward of one step (:280-334) from the window means and the three prices, operand order of the reference.  C = DevParams or RewC
// (same field names): the generic kernels and the float64 hot kernels share this one expression tree, hence identical bits.
struct RewTerms { double ch4_rev, steam_rev, o2_rev, eua_rev, chp_rev, cost_heat, cost_elz, cost_water, rew; };
template <class C>
__device__ __forceinline__ RewTerms reward_ref(const C& P, double H2, double CH4, double H2r, double H2O, double heat, double el, double gas,
                                               double eua, double b_s3)
{
    RewTerms t;
    const double ch4_vol = CH4 * P.c_mol;
    const double h2r_vol = H2r * P.c_mol;
    const double Q_ch4 = ch4_vol * P.Hu_ch4 * 1000;
    const double Q_h2r = h2r_vol * P.Hu_h2 * 1000;
    t.ch4_rev = (Q_ch4 + Q_h2r) * gas;
    const double power_chp = Q_ch4 * P.eta_chp * b_s3;
    const double Q_chp = Q_ch4 * P.one_m_eta_chp * b_s3;
    t.chp_rev = power_chp * P.eeg;
    const double Q_steam = H2O * P.dt_cp_evap / 3600;
    t.steam_rev = (Q_steam + Q_chp) * P.heat_price;
    const double h2_vol = H2 * P.c_mol;
    const double o2_vol = 0.5 * h2_vol * 3600;
    t.o2_rev = o2_vol * P.o2_price;
    const double co2 = CH4 * P.M_co2 / 1000;
    t.eua_rev = co2 / 1000 * 3600 * eua * 100;
    t.cost_heat = heat / 1000 * el;
    const double load = h2_vol / P.max_h2;
    double eta;
    if (load < P.min_load) {
        eta = 0.02;
    } else {
        const double l2 = load * load, inv = 1.0 / load;
        eta = 0.598 - 0.325 * l2 + 0.218 * (l2 * load) + 0.01 * inv - P.c_m2 * (inv * inv) + P.c_m3 * (inv * inv * inv);
    }
    t.cost_elz = h2_vol * P.Hu_h2 * 1000 / eta * el;
    const double cost_el = t.cost_heat + t.cost_elz;
    const double water_elz = H2 * P.M_h2o / 1000 * 3600;
    t.cost_water = (H2O + water_elz) / P.rho * P.water_price;
    t.rew = (t.ch4_rev + t.chp_rev + t.steam_rev + t.eua_rev + t.o2_rev - cost_el - t.cost_water) * P.sim_step_d / 3600;
    return t;
}

// One env step.  Returns terminated.
template <typename OUT, bool FAST, bool INFO, bool FM, int PAC, class LUT>
__device__ __forceinline__ bool env_step(const DevParams& P, const int2* tabmeta, const LUT& lut, Regs& R, int e, int act,
                                         const ObsRow<OUT, FM>& row, OUT* rew_out, double* info_row)
{
    // :442-450 clock and price columns at time (k+1)*dt -- independent of the state machine, so these loads go first
    const unsigned mset = (R.a.flags >> 15) & 3;
    const int k1 = R.a.k + 1;
    const int secs = k1 * P.sim_step;
    int H = R.b.act_d * 24 + secs / 3600, D = R.b.act_d + secs / 86400;
    if (H + P.PA > P.n_hours || D + 2 > P.n_days || H < 0 || D < 0) {
        P.err[1] = 1;
        H = max(0, min(H, P.n_hours - P.PA)); D = max(0, min(D, P.n_days - 2));
    }
    const double el = P.el[(size_t)mset * P.hstride + H];
    const double gas = P.gas[(size_t)mset * P.dstride + D];
    const double eua = P.eua[(size_t)mset * P.dstride + D];
    const double2 setc = P.setc[mset];
    const int kk = k1 <= P.eps_sim_steps ? k1 : P.eps_sim_steps;
    const int o = P.mod ? 2 * P.PA : P.PA + 4;
    PriceFeatures<OUT, FAST, FM, PAC> pf;
    if (row) pf.load(P, mset, H, D);
    bool changed;
    const int ridx = step_ints(P, tabmeta, lut, R, act, e, changed);
    const int s = R.a.flags & 7;
    double rew;
    if (FAST) {
        const RecFast rec = P.recf[ridx];
        R.a.flags = (R.a.flags & 0x1FFFFu) | ((unsigned)rec.tkey << 17);   // Meth_T_cat = op[-1, 1] (:452)
        rew = rec.base + rec.ch4 * (setc.x * P.k_chp + P.k_eua * eua) + rec.c_gas * gas - rec.c_el * el;
        R.b.cum += rew;
        if (changed) { rew -= setc.y; if (P.track_changes) { need_c(P, e, R); R.c.nchg += 1; R.c_dirty = true; } }
        *rew_out = (OUT)rew;
        if (row) {
            const float2 sc = P.sincos32[kk];
            pf.store(P, row);
            row.put_idx(mset * (unsigned)P.hstride + (unsigned)H, mset * (unsigned)P.dstride + (unsigned)D);
            row.put(o + 0, (OUT)s);
            for (int q = 0; q < 6; q++) row.put(o + 1 + q, (OUT)rec.feat[q]);
            row.put(o + 7, (OUT)sc.x);
            row.put(o + 8, (OUT)sc.y);
        }
    } else {
        const Rec rec = P.rec[ridx];
        R.a.flags = (R.a.flags & 0x1FFFFu) | ((unsigned)rec.tkey << 17);
        const double2 sc = P.sincos[kk];
        const double H2 = rec.m[0], CH4 = rec.m[1], H2r = rec.m[2], H2O = rec.m[3], heat = rec.m[4];
        const RewTerms rt = reward_ref(P, H2, CH4, H2r, H2O, heat, el, gas, eua, setc.x);
        const double ch4_rev = rt.ch4_rev, steam_rev = rt.steam_rev, o2_rev = rt.o2_rev, eua_rev = rt.eua_rev, chp_rev = rt.chp_rev,
                     cost_heat = rt.cost_heat, cost_elz = rt.cost_elz, cost_water = rt.cost_water;
        rew = rt.rew;
        R.b.cum += rew;
        if (changed) { rew -= setc.y; if (P.track_changes) { need_c(P, e, R); R.c.nchg += 1; R.c_dirty = true; } }
        *rew_out = (OUT)rew;
        // :206-217 + :219-249 observation row
        if (row) {
            pf.store(P, row);
            row.put_idx(mset * (unsigned)P.hstride + (unsigned)H, mset * (unsigned)P.dstride + (unsigned)D);
            row.put(o + 0, (OUT)s);
            row.put(o + 1, (OUT)((rec.T - P.T_lo) / P.T_rng));
            row.put(o + 2, (OUT)((H2 - P.h2_lo) / P.h2_rng));
            row.put(o + 3, (OUT)((CH4 - P.ch4_lo) / P.ch4_rng));
            row.put(o + 4, (OUT)((H2r - P.h2r_lo) / P.h2r_rng));
            row.put(o + 5, (OUT)((H2O - P.h2o_lo) / P.h2o_rng));
            row.put(o + 6, (OUT)((heat - P.heat_lo) / P.heat_rng));
            row.put(o + 7, (OUT)sc.x);
            row.put(o + 8, (OUT)sc.y);
        }
        if (INFO && info_row) {   // :251-278
            const size_t hb = (size_t)mset * P.hstride + H;
            info_row[0] = (double)R.a.k; info_row[1] = el; info_row[2] = gas; info_row[3] = eua;
            info_row[4] = (double)s; info_row[5] = (double)act; info_row[6] = (double)((R.a.flags >> 3) & 1);
            info_row[7] = rec.T; info_row[8] = H2; info_row[9] = CH4; info_row[10] = H2O; info_row[11] = heat;
            info_row[12] = ch4_rev; info_row[13] = steam_rev; info_row[14] = o2_rev; info_row[15] = eua_rev;
            info_row[16] = chp_rev; info_row[17] = -cost_heat; info_row[18] = -cost_elz; info_row[19] = -cost_water;
            info_row[20] = rew; info_row[21] = R.b.cum;
            info_row[22] = P.pot_raw[hb]; info_row[23] = P.pf_raw[hb];
        }
    }
    const bool term = (R.a.k == P.eps_sim_steps - 6);   // :508-511, tested before k += 1
    R.a.k = k1;
    return term;
}

// Finished episodes are compacted with a wave ballot: one atomic per wave, rank = popcount of lower done lanes.
__device__ __forceinline__ void push_finished(const DevParams& P, bool done, int e, double ret, int len)
{
    const unsigned long long m = __ballot(done);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;                                      // the running count is unsigned: it may wrap after 2^32 episodes, a slot never goes negative
    if (lane == leader) base = atomicAdd((unsigned*)P.fin_count, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    if (done) {
        const unsigned slot = (base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))) % (unsigned)P.fin_cap;
        P.fin_ret[slot] = ret; P.fin_len[slot] = len; P.fin_env[slot] = e;
    }
}

template <typename OUT, bool FAST, bool INFO, bool FM, int PAC>
__global__ void __launch_bounds__(256)
k_step(const DevParams P, const void* __restrict__ actions, int action_kind, OUT* __restrict__ obs, OUT* __restrict__ rew,
       uint8_t* __restrict__ done, OUT* __restrict__ final_obs, double* __restrict__ info, int only_at_k)
{
    // only_at_k >= 0: the second kernel of a CAPTURED hot step (ptg_step while `stream` is being captured): the launch acts only when the
    // hot kernel in front of it found the synchronised batch on the episode's terminating step and skipped it; it is a no-op otherwise
    // (the hot kernel's flag, not the step count: after an ordinary step the count may have just ARRIVED at only_at_k)
    if (only_at_k >= 0 && *P.term_flag == 0) return;
    __shared__ int2 s_tm[NT];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e < P.N;
    bool term = false;
    Regs R;
    double ret = 0.0;
    int len = 0;
    if (live) load_regs(P, e, R);                        // state loads in flight while the table meta is staged
    if (threadIdx.x < NT) s_tm[threadIdx.x] = P.tabmeta[threadIdx.x];
    __syncthreads();
    const LutGlobal lut{P.argidx, P.nT};
    if (live) {
        const int act = decode_action(actions, action_kind, e, (R.a.flags >> 12) & 7);
        if (act < 0) {
            P.err[0] = 1;
            rew[e] = (OUT)NAN;
            done[e] = 0;
        } else {
            const ObsRow<OUT, FM> row(obs, P, e);
            double* irow = (INFO && info) ? info + (size_t)e * PTG_N_INFO : nullptr;
            OUT r;
            term = env_step<OUT, FAST, INFO, FM, PAC>(P, s_tm, lut, R, e, act, row, &r, irow);
            rew[e] = r;
            done[e] = term ? 1 : 0;
            if (term) {
                if (final_obs) {
                    const ObsRow<OUT, FM> frow(final_obs, P, e);
                    frow.copy_row_from(row, P.F);
                }
                ret = R.b.cum;
                if (P.track_changes) { need_c(P, e, R); ret -= (double)R.c.nchg * P.setc[(R.a.flags >> 15) & 3].y; }
                len = R.a.k;
                reset_env<OUT, FAST, FM, PAC>(P, R, e, row);
            }
            store_regs(P, e, R);
        }
    }
    push_finished(P, live && term, e, ret, len);
}

template <typename OUT, bool FAST, bool FM, int PAC>
__global__ void k_reset(const DevParams P, const uint8_t* __restrict__ mask, OUT* __restrict__ obs)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.N) return;
    if (mask && !mask[e]) return;
    Regs R;
    load_regs(P, e, R);
    reset_env<OUT, FAST, FM, PAC>(P, R, e, ObsRow<OUT, FM>(obs, P, e));
    store_regs(P, e, R);
}

__global__ void k_init_state(const DevParams P, int first_ptr_mod, int have_plan)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.N) return;
    if (have_plan) {
        P.st_c[e].epp = P.E > 0 ? (int)(((long long)first_ptr_mod + e) % P.E) : 0;
        return;
    }
    StA a; a.i = 0; a.j = 0; a.k = 0;
    a.flags = 1u | (1u << 12) | ((unsigned)P.key_init << 17);   // cooldown, current_action = 'cooldown' (:143)
    StB b; b.cum = 0.0; b.act_d = 0; b.nctr = 0;
    StC c; c.nchg = 0; c.epp = 0;
    P.st_a[e] = a; P.st_b[e] = b; P.st_c[e] = c;
}

__global__ void k_zero_noise_count(const DevParams P)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < P.N) P.st_b[e].nctr = 0;
}

// fast-path records from the window records: reward coefficients (:280-334) and normalised features (:212-217)
__global__ void k_build_fast(const DevParams P, const Rec* __restrict__ in, RecFast* __restrict__ out, int n)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const Rec rec = in[g];
    const double H2 = rec.m[0], CH4 = rec.m[1], H2r = rec.m[2], H2O = rec.m[3], heat = rec.m[4];
    const double f = P.sim_step_d / 3600;
    const double Q_ch4 = CH4 * P.c_mol * P.Hu_ch4 * 1000;
    const double Q_h2r = H2r * P.c_mol * P.Hu_h2 * 1000;
    const double Q_steam = H2O * P.dt_cp_evap / 3600;
    const double h2_vol = H2 * P.c_mol;
    const double o2_rev = 0.5 * h2_vol * 3600 * P.o2_price;
    const double load = h2_vol / P.max_h2;
    double eta;
    if (load < P.min_load) {
        eta = 0.02;
    } else {
        const double l2 = load * load, inv = 1.0 / load;
        eta = 0.598 - 0.325 * l2 + 0.218 * (l2 * load) + 0.01 * inv - P.c_m2 * (inv * inv) + P.c_m3 * (inv * inv * inv);
    }
    const double water = (H2O + H2 * P.M_h2o / 1000 * 3600) / P.rho * P.water_price;
    RecFast o;
    o.base = (Q_steam * P.heat_price + o2_rev - water) * f;
    o.ch4 = CH4;
    o.c_gas = (Q_ch4 + Q_h2r) * f;
    o.c_el = (heat / 1000 + h2_vol * P.Hu_h2 * 1000 / eta) * f;
    o.feat[0] = (float)((rec.T - P.T_lo) / P.T_rng);
    o.feat[1] = (float)((H2 - P.h2_lo) / P.h2_rng);
    o.feat[2] = (float)((CH4 - P.ch4_lo) / P.ch4_rng);
    o.feat[3] = (float)((H2r - P.h2r_lo) / P.h2r_rng);
    o.feat[4] = (float)((H2O - P.h2o_lo) / P.h2o_rng);
    o.feat[5] = (float)((heat - P.heat_lo) / P.heat_rng);
    o.tkey = rec.tkey;
    o.pad = 0;
    out[g] = o;
}


// ================================================================================================ hot kernels
// Dedicated float32 kernels for the common case: 13-hour look-ahead, no info rows, and NO env terminating inside the launch
// (the host tracks the common step count of a synchronised batch and routes the one step per episode that terminates --
// and everything unusual -- through the generic kernels above).  Same arithmetic as env_step<float, FAST>, laid out for a
// machine that runs few waves per SIMD at N = 65 536, where little hides latency but the order of the instruction stream:
//   * few scalars: HotParams carries only what every step needs; ladder thresholds, table meta data and the _get_index
//     lookup sit in LDS;
//   * every address is (uniform base) + (32-bit lane byte offset): no per-feature 64-bit address registers;
//   * the integer state machine is branch-free (select chains in the reference's if / elif priority order); only
//     wave-uniform branches remain (some lane draws noise; some lane switches partial <-> full load);
//   * k_step_hot (one vector step per launch) issues ALL loads of the step in one burst, then the stores;
//   * k_rollout_pc (T fused steps) splits the step between producer waves (state machine + a 2-byte key gather: the only
//     loop-carried chain) and consumer waves (record gather, reward, stores), see below.
// Measured and rejected on MI355X: a single-role software-pipelined rollout kernel (2.0 us per step against
// 1.55), a "broadcast" form that stores the 28 features a synchronised wave shares as 7 dwordx4, 32 / 16 envs per wave.
template <typename T>
__device__ __forceinline__ T ld_off(const void* base, unsigned byte_off) { return *(const T*)((const char*)base + byte_off); }
typedef unsigned uv4 __attribute__((ext_vector_type(4)));
typedef double vd2 __attribute__((ext_vector_type(2)));
// Output rows (observations, rewards, done flags) are written once and never read back by these kernels: non-temporal
// stores keep the write stream from allocating in L2, where it evicts the window records and market series every step
// re-reads (measured at N = 65 536: 2.86 -> 1.90 us per fused step, and the producer / consumer form 2.96 -> 1.51).
template <typename T>
__device__ __forceinline__ void st_off(void* base, unsigned byte_off, T v)
{
#ifdef PTG_PLAIN_STORES
    *(T*)((char*)base + byte_off) = v;
#else
    __builtin_nontemporal_store(v, (T*)((char*)base + byte_off));
#endif
}

enum { LAD_T1_START_P_F = 0, LAD_T2_START_F_P, LAD_T_P_F, LAD_T_F_P, LAD_T1_P_F_P, LAD_T2_P_F_P, LAD_T3_P_F_P, LAD_T34_P_F_P,
       LAD_T4_P_F_P, LAD_T45_P_F_P, LAD_T5_P_F_P, LAD_T1_F_P_F, LAD_T2_F_P_F, LAD_T23_F_P_F, LAD_T3_F_P_F, LAD_T34_F_P_F,
       LAD_T4_F_P_F, LAD_T45_F_P_F, LAD_T5_F_P_F, LAD_I_FULL, LAD_J_FULL, LAD_N };

struct RewC {                    // reward (:280-334) and normalisation (:206-217) constants, named as in DevParams (reward_ref reads either)
    double c_mol, Hu_ch4, Hu_h2, dt_cp_evap, heat_price, o2_price, eeg, eta_chp, one_m_eta_chp, M_co2, M_h2o,
           rho, water_price, min_load, max_h2, c_m2, c_m3, sim_step_d;
    double T_lo, T_rng, h2_lo, h2_rng, ch4_lo, ch4_rng, h2r_lo, h2r_rng, h2o_lo, h2o_rng, heat_lo, heat_rng;
};

struct HotParams {
    int N, S, sim_step, eps_sim_steps, F, nT, tape_len, track_changes, flat;
    int fm_pitch;                                     // feature-major outputs: elements between two feature planes (>= N; ptg_set_feature_pitch)
    int key_cold_max, key_hot_min, key_standby_max, n_hours, n_days, hstride, dstride;
    unsigned off_featB, off_gasn, off_euan, off_sc;   // element offsets into pool32 (featA at 0; sin/cos pairs at off_sc)
    unsigned off_gas, off_eua;                        // element offsets into pool64 (el at 0)
    unsigned o64_featA, o64_featB, o64_gasn, o64_euan, o64_sc;   // float64 outputs: the same feature series as doubles, in pool64
    unsigned long long noise_seed;
    long long env_offset;
    double noise_sigma, k_chp, k_eua;
    const RecFast* recf;
    const Rec* rec;                                   // float64 outputs: raw window means, reward in the reference's operand order
    const double *pot_raw, *pf_raw;                   // un-normalised Pot_Reward / Part_Full series (info rows of the float64 rollout)
    RewC rc;
    const double* tape;
    const float* pool32;
    const double* pool64;
    const double2* setc;
    const int* argidx;
    const int2* tabmeta;
    const int* ladder;                                // [LAD_N]
    StA* st_a; StB* st_b; StC* st_c;
    int* err;
    int* term_flag;
};

struct HotLds {                  // per-workgroup LDS image
    int2 tm[NT + 1];
    int lad[LAD_N + 3];
};

struct HotRegs {                 // per-env state in registers
    int i, j, k;
    unsigned flags;
    double cum;
    int act_d, nctr;
};

// float32 outputs: strength-reduced records (RecFast) and float32 feature series; float64 outputs: the raw window records (Rec),
// float64 series, reward and normalisation evaluated as the reference writes them (reward_ref)
template <typename OUT> struct HotTypes;
template <> struct HotTypes<float> { typedef RecFast rec_t; typedef float2 sc_t; };
template <> struct HotTypes<double> { typedef Rec rec_t; typedef double2 sc_t; };

template <typename OUT>
struct HotLoads {                // everything front() fetched for one step
    OUT fa[13], fb[13];          // Pot_Reward, Part_Full ('raw': Elec_Price; Gas_Price[2], EUA_Price[2] in fb[0..3])
    typename HotTypes<OUT>::sc_t sc;   // Temp_hour_enc_sin / cos
    unsigned hb4, db4, kk8;      // byte offsets (of 4-byte elements) of the step's hour / day / step-count entries in the pools
    double el, gas, eua;
    typename HotTypes<OUT>::rec_t rec;
    bool changed;
};

__device__ __forceinline__ const RecFast* rec_table(const HotParams& P, const RecFast*) { return P.recf; }
__device__ __forceinline__ const Rec* rec_table(const HotParams& P, const Rec*) { return P.rec; }
// reward of the step (before the state-change penalty): price-linear form / reference form
__device__ __forceinline__ double hot_reward(const HotParams& P, const RecFast& r, double el, double gas, double eua, double b_s3)
{
    return r.base + r.ch4 * (b_s3 * P.k_chp + P.k_eua * eua) + r.c_gas * gas - r.c_el * el;
}
__device__ __forceinline__ double hot_reward(const HotParams& P, const Rec& r, double el, double gas, double eua, double b_s3)
{
    return reward_ref(P.rc, r.m[0], r.m[1], r.m[2], r.m[3], r.m[4], el, gas, eua, b_s3).rew;
}
// normalised T_cat, H2, CH4, H2_res, H2O, el_heating (:212-217): pre-computed float32 / evaluated in float64
template <int Q> __device__ __forceinline__ float rec_feat(const HotParams&, const RecFast& r) { return r.feat[Q]; }
template <int Q> __device__ __forceinline__ double rec_feat(const HotParams& P, const Rec& r)
{
    const RewC& c = P.rc;
    return Q == 0 ? (r.T - c.T_lo) / c.T_rng : Q == 1 ? (r.m[0] - c.h2_lo) / c.h2_rng : Q == 2 ? (r.m[1] - c.ch4_lo) / c.ch4_rng
         : Q == 3 ? (r.m[2] - c.h2r_lo) / c.h2r_rng : Q == 4 ? (r.m[3] - c.h2o_lo) / c.h2o_rng : (r.m[4] - c.heat_lo) / c.heat_rng;
}
// the six record features into an observation sink, columns o + 1 .. o + 6
template <bool U, class Sink, class RecT>
__device__ __forceinline__ void put_rec_feats(const HotParams& P, const Sink& row, int o, const RecT& r)
{
    if (U) {
        row.put_u(o + 1, rec_feat<0>(P, r)); row.put_u(o + 2, rec_feat<1>(P, r)); row.put_u(o + 3, rec_feat<2>(P, r));
        row.put_u(o + 4, rec_feat<3>(P, r)); row.put_u(o + 5, rec_feat<4>(P, r)); row.put_u(o + 6, rec_feat<5>(P, r));
    } else {
        row.put(o + 1, rec_feat<0>(P, r)); row.put(o + 2, rec_feat<1>(P, r)); row.put(o + 3, rec_feat<2>(P, r));
        row.put(o + 4, rec_feat<3>(P, r)); row.put(o + 5, rec_feat<4>(P, r)); row.put(o + 6, rec_feat<5>(P, r));
    }
}
// feature series of the element type: float32 pool / the float64 copies behind the prices in pool64
__device__ __forceinline__ const float* series(const HotParams& P, const float*, int which)
{
    return P.pool32 + (which == 0 ? 0u : which == 1 ? P.off_featB : which == 2 ? P.off_gasn : which == 3 ? P.off_euan : P.off_sc);
}
__device__ __forceinline__ const double* series(const HotParams& P, const double*, int which)
{
    return P.pool64 + (which == 0 ? P.o64_featA : which == 1 ? P.o64_featB : which == 2 ? P.o64_gasn : which == 3 ? P.o64_euan : P.o64_sc);
}

enum { NOISE_NONE = 0, NOISE_TAPE = 1, NOISE_RNG = 2 };

// Select helpers of the state machine.  Both arms are VALUES computed before the call, so the front end emits an IR `select` and the
// backend a v_cndmask: written as nested `?:` with arithmetic in the arms (or as a chain of `== constant` tests, which becomes a switch)
// the same code compiled into exec-mask regions -- 87 s_and_saveexec / s_or / s_xor per producer step out of 450 instructions, and with one
// producer wave per SIMD every instruction, scalar or not, is one 4-cycle issue slot of the loop-carried chain (round 3).
__device__ __forceinline__ int sel(bool c, int a, int b) { return c ? a : b; }
__device__ __forceinline__ int nib(unsigned table, int i) { return (int)((table >> (4 * i)) & 15u); }      // entry i of a table of 4-bit values
// table ids of the six partial-load / full-load tables by their 3-bit state codes (part_tid / full_tid as register tables)
__device__ __forceinline__ int part_tid_r(int p) { return nib(0xCBA985u, p); }                 // 5, 8, 9, 10, 11, 12
__device__ __forceinline__ int full_tid_r(int q) { return 6 + nib(0xA98710u, q); }             // 6, 7, 13, 14, 15, 16

// The ladder thresholds as registers: loaded from the workgroup's LDS image once (k_rollout_pc: once per launch, by the producers) with
// six 16-byte reads instead of up to 21 conditional 4-byte reads inside every step that switches between partial and full load.
struct Ladder { int v[LAD_N + 3]; };
__device__ __forceinline__ void load_ladder(const HotLds& L, Ladder& G)
{
    const int4* q = (const int4*)L.lad;                      // (HotLds::lad sits at a 16-byte offset: int2 tm[18] = 144 bytes before it)
#pragma unroll
    for (int k = 0; k < (LAD_N + 3) / 4; k++) { const int4 x = q[k]; G.v[4 * k] = x.x; G.v[4 * k + 1] = x.y; G.v[4 * k + 2] = x.z; G.v[4 * k + 3] = x.w; }
}

// Integer state machine (:339-440, :525-757), branch-free.  Returns the record index.
// ZPRE (tape mode, the rollout's producers): the env's next tape entry was fetched one step ahead and arrives in z_pre
template <int NOISE, bool ZPRE = false>
__device__ __forceinline__ int hot_ints(const HotParams& P, const HotLds& L, const Ladder& G, const unsigned short* lut16, bool lds_lut,
                                        HotRegs& R, int act, int e, bool& changed, double z_pre = 0.0)
{
    const unsigned f = R.flags;
    const int s = f & 7, sb = (f >> 4) & 1, su = (f >> 5) & 1, pp = (f >> 6) & 7, fq = (f >> 9) & 7;
    const unsigned mset = (f >> 15) & 3;
    const int tkey = (int)(f >> 17);
    const int hot = sel(tkey <= P.key_cold_max, 0, sel(tkey >= P.key_hot_min, 1, (int)((f >> 3) & 1)));      // :339-342
    const int S = P.S, i0 = R.i, j0 = R.j;
    const int time_op = i0 + j0 * S;
    // :368-440 dispatch
    const bool k1 = (act == 0) & (s != 0);          // _standby
    const bool k2 = (act == 1) & (s != 1);          // _cooldown
    const bool k3 = (act == 2) & (s <= 1);          // _startup
    const bool k4 = (act == 3) & (s == 4);          // _partial
    const bool k5 = (act == 4) & (s == 3);          // _full
    const bool noisy = k1 | k2 | k3, k45 = k4 | k5;
    const int sb_new = (tkey <= P.key_standby_max) ? 1 : 0;
    const int d_sb = 2 - sb_new, d_su = 3 + hot;
    const int dest = sel(k1, d_sb, sel(k3, d_su, sel(k4, 5, 0)));       // cooldown 0, standby_up 1, standby_down 2, startup_cold 3, startup_hot 4, op1 5
    const unsigned li = (unsigned)(dest * P.nT + tkey);
    const int idx = lds_lut ? (int)lut16[li] : P.argidx[li];
    int i_noisy = 0;
    if (__ballot(noisy)) {                           // :584-585 int(max(idx + normal(0, noise), 0))
        double z = 0.0;
        if (NOISE == NOISE_TAPE) {
            if (ZPRE) z = z_pre;
            else if (noisy) z = P.tape[(size_t)((unsigned)R.nctr % (unsigned)P.tape_len) * P.N + e];
        }
        else if (NOISE == NOISE_RNG) z = noise_draw(P.noise_seed, P.env_offset + e, (unsigned)R.nctr, P.noise_sigma);
        R.nctr += noisy ? 1 : 0;
        double x = (double)idx + z;
        x = (0 > x) ? 0 : x;
        i_noisy = (int)x;
    }
    // _cont (:559-570): the table the env is in -- by state: standby, cooldown, startup, partial load, full load
    const int t_sb = PTG_T_STANDBY_DOWN + sb, t_pp = part_tid_r(pp), t_fq = full_tid_r(fq);
    const int t_cont = sel(s >= 4, t_fq, sel(s >= 3, t_pp, sel(s >= 2, su, sel(s >= 1, PTG_T_COOLDOWN, t_sb))));
    const int t_sbn = PTG_T_STANDBY_DOWN + sb_new;
    const int t_noisy = sel(k1, t_sbn, sel(k2, PTG_T_COOLDOWN, hot));     // startup_cold 0 / startup_hot 1
    int pp_l = 5, fq_l = 1, i_l = 0, j_l = 1;        // ladder results (_partial :627-691, _full :693-757)
    if (__ballot(k45)) {
        const int* lad = G.v;
        const int t = time_op, j0p = j0 + 1, i_full = lad[LAD_I_FULL], j_full = lad[LAD_J_FULL];
        int i_p, j_p, i_f, j_f;
        {   // _partial, by full_op: the reference's if / elif chain, first match wins (applied last below)
            const bool a0 = t < lad[LAD_T2_START_F_P];
            const int T1 = lad[LAD_T1_P_F_P], T2 = lad[LAD_T2_P_F_P], TPF = lad[LAD_T_P_F], T34 = lad[LAD_T34_P_F_P],
                      T45 = lad[LAD_T45_P_F_P], T5 = lad[LAD_T5_P_F_P];
            const bool r1 = t < T1, r2 = (T1 < t) & (t < T2), r3 = (T2 < t) & (t < TPF), r4 = (TPF < t) & (t < T34),
                       r5 = (T34 < t) & (t < T45), r6 = (T45 < t) & (t < T5);
            const int pp1 = sel(r1, 5, sel(r2 | r3, 1, sel(r4, 2, sel(r5, 3, sel(r6, 4, 5)))));
            const int i1 = sel(r1, i_full, sel(r2, i0, sel(r3, T2, sel(r4, lad[LAD_T3_P_F_P], sel(r5, lad[LAD_T4_P_F_P], sel(r6, T5, 0))))));
            const int j1 = sel(r1, j_full, sel(r2, j0p, 1));
            const bool f0 = fq == 0, f1 = fq == 1;
            pp_l = sel(f0, sel(a0, 0, 5), sel(f1, pp1, 5));
            i_p = sel(f0, sel(a0, idx, 0), sel(f1, i1, 0));
            j_p = sel(f1, j1, 1);
        }
        {   // _full, by part_op
            const bool b0 = t < lad[LAD_T1_START_P_F];
            const int T1 = lad[LAD_T1_F_P_F], TFP = lad[LAD_T_F_P], T23 = lad[LAD_T23_F_P_F], T34 = lad[LAD_T34_F_P_F],
                      T45 = lad[LAD_T45_F_P_F], T5 = lad[LAD_T5_F_P_F];
            const bool q1 = t < T1, q2 = (T1 < t) & (t < TFP), q3 = (TFP < t) & (t < T23), q4 = (T23 < t) & (t < T34),
                       q5 = (T34 < t) & (t < T45), q6 = (T45 < t) & (t < T5);
            const int fq1 = sel(q1, 1, sel(q2 | q3, 2, sel(q4, 3, sel(q5, 4, sel(q6, 5, 1)))));
            const int i1 = sel(q1, i_full, sel(q2, i0, sel(q3, lad[LAD_T2_F_P_F], sel(q4, lad[LAD_T3_F_P_F], sel(q5, lad[LAD_T4_F_P_F], sel(q6, T5, 0))))));
            const int j1 = sel(q1, j_full, sel(q2, j0p, 1));
            const bool p0 = pp == 0, p5 = pp == 5;
            fq_l = sel(p0, sel(b0, 0, 1), sel(p5, fq1, 1));
            i_f = sel(p5, i1, 0);
            j_f = sel(p5, j1, 1);
        }
        i_l = sel(k4, i_p, sel(k5, i_f, 0));
        j_l = sel(k4, j_p, sel(k5, j_f, 1));
    }
    const int pp_n = sel(k3, 0, sel(k4, pp_l, pp));
    const int fq_n = sel(k3, 0, sel(k5, fq_l, fq));
    const int s_ev = sel(k1, 0, sel(k2, 1, sel(k3, 2, sel(k4, 3, sel(k5, 4, s)))));
    const int j0n = j0 + 1;
    int i_n = sel(noisy, i_noisy, sel(k45, i_l, i0));
    int j_n = sel(noisy, 1, sel(k45, j_l, j0n));
    const int t_pl = part_tid_r(pp_l), t_fl = full_tid_r(fq_l);
    const int table = sel(noisy, t_noisy, sel(k4, t_pl, sel(k5, t_fl, t_cont)));
    const int sb_n = sel(k1, sb_new, sb), su_n = sel(k3, hot, su);
    // _perform_sim_step (:525-557) against the virtual table [rows | splice-or-last-row padding]
    const int2 tm = L.tm[table];
    const int n = tm.x;
    const int start = i_n + (j_n - 1) * S;
    const int over = start + S - n;
    const bool inside = over < 0;                    // start + S < n
    const bool is_su = table <= PTG_T_STARTUP_HOT;   // change_operation: startup -> partial load
    const bool head = over < S;
    const int r_min = min(start, n), r_su = sel(head, start, n);
    const int r = sel(inside, start, sel(is_su, r_su, r_min));
    const bool splice = (!inside) & is_su, sh = splice & head;
    const int s_n = sel(splice, 3, s_ev);
    i_n = sel(sh, over, i_n);
    j_n = sel(sh, 0, j_n);
    changed = (s != s_n);
    R.i = i_n; R.j = j_n;
    R.flags = (unsigned)s_n | (hot << 3) | (sb_n << 4) | (su_n << 5) | (pp_n << 6) | (fq_n << 9) | ((unsigned)act << 12) |
              (mset << 15) | ((unsigned)tkey << 17);
    return tm.y + r;
}

// front half of a step: clock (:442-447), market loads, state machine, record gather -- every load of the step in one burst
// market + clock features of one step: 13 + 13 (or 13 + 4) + 2 float32 loads, merged by the backend into dwordx4
template <bool MOD, typename OUT>
__device__ __forceinline__ void hot_load_market(const HotParams& P, HotLoads<OUT>& Q)
{
    constexpr unsigned W = sizeof(OUT) / 4;                 // byte offsets are kept for 4-byte elements
    const OUT* pA = series(P, (const OUT*)nullptr, 0);
#pragma unroll
    for (int q = 0; q < 13; q++) Q.fa[q] = ld_off<OUT>(pA + q, Q.hb4 * W);   // uniform (base + q) + one lane offset
    if (MOD) {
        const OUT* pB = series(P, (const OUT*)nullptr, 1);
#pragma unroll
        for (int q = 0; q < 13; q++) Q.fb[q] = ld_off<OUT>(pB + q, Q.hb4 * W);
    } else {
        const OUT* pG = series(P, (const OUT*)nullptr, 2);
        const OUT* pU = series(P, (const OUT*)nullptr, 3);
        Q.fb[0] = ld_off<OUT>(pG, Q.db4 * W); Q.fb[1] = ld_off<OUT>(pG + 1, Q.db4 * W);
        Q.fb[2] = ld_off<OUT>(pU, Q.db4 * W); Q.fb[3] = ld_off<OUT>(pU + 1, Q.db4 * W);
    }
    Q.sc = ld_off<typename HotTypes<OUT>::sc_t>(series(P, (const OUT*)nullptr, 4), Q.kk8 * W);
}

// k1 = step count after this step: UNIFORM (the hot kernels only run on a synchronised batch), so the clock arithmetic of
// :442-445 is scalar; only the episode offset act_ep_d differs between envs
template <bool MOD, int NOISE, typename OUT>
__device__ __forceinline__ void hot_front(const HotParams& P, const HotLds& L, const Ladder& G, const unsigned short* lut16, bool lds_lut,
                                          HotRegs& R, int act, int e, int k1, HotLoads<OUT>& Q)
{
    const unsigned mset = (R.flags >> 15) & 3;
    const int secs = k1 * P.sim_step;
    const int hs = secs / 3600, ds = secs / 86400;
    int H = R.act_d * 24 + hs, D = R.act_d + ds;
    const bool oob = (H + 13 > P.n_hours) | (D + 2 > P.n_days) | (H < 0) | (D < 0);
    if (__ballot(oob)) {
        if (oob) { P.err[1] = 1; H = max(0, min(H, P.n_hours - 13)); D = max(0, min(D, P.n_days - 2)); }
    }
    const unsigned hb4 = (mset * (unsigned)P.hstride + (unsigned)H) * 4u, db4 = (mset * (unsigned)P.dstride + (unsigned)D) * 4u;
    Q.hb4 = hb4; Q.db4 = db4; Q.kk8 = (unsigned)min(k1, P.eps_sim_steps) * 8u;
    hot_load_market<MOD, OUT>(P, Q);
    Q.el = ld_off<double>(P.pool64, hb4 * 2u);
    Q.gas = ld_off<double>(P.pool64 + P.off_gas, db4 * 2u);
    Q.eua = ld_off<double>(P.pool64 + P.off_eua, db4 * 2u);
    const int ridx = hot_ints<NOISE>(P, L, G, lut16, lds_lut, R, act, e, Q.changed);
    typedef typename HotTypes<OUT>::rec_t rec_t;
    Q.rec = ld_off<rec_t>(rec_table(P, (const rec_t*)nullptr), (unsigned)ridx * 64u);
}

// back half: reward (:280-334 in its price-linear form) and bookkeeping
template <typename OUT>
__device__ __forceinline__ OUT hot_back(const HotParams& P, HotRegs& R, const HotLoads<OUT>& Q, const double2 setc, int e, bool live)
{
    R.flags = (R.flags & 0x1FFFFu) | ((unsigned)Q.rec.tkey << 17);           // Meth_T_cat = op[-1, 1] (:452)
    double rew = hot_reward(P, Q.rec, Q.el, Q.gas, Q.eua, setc.x);
    R.cum += rew;
    rew -= Q.changed ? setc.y : 0.0;                                         // :332 (setc.y = r_0 * penalty, 0 by default)
    if (P.track_changes) { if (Q.changed && live) P.st_c[e].nchg += 1; }     // lanes past N shadow env N-1: no side effects
    return (OUT)rew;
}

template <bool FM, bool MOD = true, typename OUT = float>
struct HotRow {                  // observation row addressing: uniform base + 32-bit lane byte offset (feature q adds q * qbytes)
    static constexpr unsigned B = sizeof(OUT);
    char* base; unsigned boff; unsigned qbytes; bool flat;
    __device__ __forceinline__ HotRow(OUT* b, const HotParams& P, int e)
        : base((char*)b), boff(FM ? (unsigned)e * B : (unsigned)e * (unsigned)P.F * B), qbytes(FM ? (unsigned)P.fm_pitch * B : B), flat(!FM && P.flat) {}
    __device__ __forceinline__ void put(int q, OUT v) const;
    // same address as (uniform pointer advanced by SALU) + (the one lane offset): no per-feature offset registers
    __device__ __forceinline__ void put_u(int q, OUT v) const;
    __device__ __forceinline__ void put_idx(unsigned, unsigned) const {}
};

// SB3_FLAT layout (price_ahead = 13): canonical column q -> column of the flat row, sub-spaces in sorted key order
// ('mod': CH4_syn, Elec_Heating, H2O_DE, H2_in, H2_res, METH_STATUS x6, Part_Full x13, Pot_Reward x13, T_CAT, cos, sin;
//  'raw': CH4_syn, EUA_Price x2, Elec_Heating, Elec_Price x13, Gas_Price x2, H2O_DE, H2_in, H2_res, METH_STATUS x6, T_CAT, cos, sin).
// ptg_create checks this table against the general host-built map.
template <bool MOD>
__host__ __device__ constexpr int flat_col(int q)
{
    if (MOD) {
        return q < 13 ? 24 + q : q < 26 ? 11 + (q - 13)
             : q == 26 ? 5 : q == 27 ? 37 : q == 28 ? 3 : q == 29 ? 0 : q == 30 ? 4 : q == 31 ? 2 : q == 32 ? 1 : q == 33 ? 39 : 38;
    }
    return q < 13 ? 4 + q : q < 15 ? 17 + (q - 13) : q < 17 ? 1 + (q - 15)
         : q == 17 ? 22 : q == 18 ? 28 : q == 19 ? 20 : q == 20 ? 0 : q == 21 ? 21 : q == 22 ? 19 : q == 23 ? 3 : q == 24 ? 30 : 29;
}

template <bool FM, bool MOD, typename OUT>
__device__ __forceinline__ void HotRow<FM, MOD, OUT>::put(int q, OUT v) const
{
    constexpr int QS = MOD ? 26 : 17;
    if (!FM && flat) {
        if (q == QS) {
#pragma unroll
            for (int j = 0; j < 6; j++) st_off<OUT>(base, boff + (unsigned)(flat_col<MOD>(QS) + j) * B, ((int)v == j) ? (OUT)1 : (OUT)0);
        } else st_off<OUT>(base, boff + (unsigned)flat_col<MOD>(q) * B, v);
    } else st_off<OUT>(base, boff + (unsigned)q * qbytes, v);
}
template <bool FM, bool MOD, typename OUT>
__device__ __forceinline__ void HotRow<FM, MOD, OUT>::put_u(int q, OUT v) const
{
    if (!FM && flat) put(q, v);
    else st_off<OUT>(base + (size_t)q * qbytes, boff, v);
}

// Row-major observations ([N][F], the reference's layout): a lane's row is 140 B away from its neighbour's, so per-feature
// stores would be 64 scattered 4-byte writes each.  Instead every wave transposes its 64 rows through a private LDS tile and
// writes them back as ONE contiguous block of 64 * F floats with dwordx4 stores (64 * 140 B = 70 full 128-byte lines).
typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool MOD, bool FLAT, typename OUT = float, bool SPLIT = false>
struct RowTile {
    static constexpr int FC = MOD ? 35 : 26;               // canonical row width
    static constexpr int FMAX = FC + 5;                    // SB3_FLAT: METH_STATUS one-hot (6 columns for 1)
    static constexpr int F = SPLIT ? 16 : FLAT ? FMAX : FC;   // SPLIT: one-hot status (6), 8 env features, hour / day series index
    static constexpr int QS = MOD ? 26 : 17;               // canonical column of METH_STATUS
    static constexpr int EPP = 16 / (int)sizeof(OUT);      // elements per 16-byte piece (4 floats / 2 doubles)
    static constexpr int N4 = 64 * F / EPP;                // 16-byte pieces per wave block (float: 560 / 416, flat 640 / 496; double: 1120 / 832)
    // LDS row pitch: F = 40 would put lanes l and l + 4 on the same bank for every column (16-way conflicts on each of the 40
    // writes: measured 2.15 us per step); 41 is conflict-free, and because 40 is a multiple of 4 a float4 of the output
    // image never straddles two tile rows
    // (float64 rows: pitch 35 doubles = 70 dwords -- lanes l, l + 1 start 6 banks apart, an 8-byte store per lane is conflict-free)
    static constexpr int PITCH = (F == 40) ? 41 : (F == 16) ? 17 : F;
    static constexpr int TILE = 64 * PITCH;                // elements per wave
    typedef __attribute__((address_space(3))) OUT lds_out;
    OUT* t;                                                // this wave's [64][PITCH] tile
    int lane;
    lds_out* r;                                            // this lane's row of the tile, as an LDS byte address held in ONE register:
    // left to the compiler the address is (constant offset of the tiles in LDS) + (lane part), and the constant does not fit the 8-bit
    // offsets of the paired ds_write2_b32 a row is written with -- it then spends one v_add per pair and step (18 of a consumer's ~245
    // instructions) re-adding it.  Laundered through an empty asm the sum is opaque and the columns become immediate offsets.
    __device__ __forceinline__ RowTile(OUT* tiles, int wave) : t(tiles + wave * TILE), lane(threadIdx.x & 63)
    {
        unsigned a = (unsigned)(size_t)(lds_out*)(t + lane * PITCH);
        asm volatile("" : "+v"(a));
        r = (lds_out*)(size_t)a;
    }
    __device__ __forceinline__ void put(int q, OUT v) const
    {
        if (SPLIT) {
            if (q < QS) return;
            if (q == QS) {
#pragma unroll
                for (int j = 0; j < 6; j++) r[j] = ((int)v == j) ? (OUT)1 : (OUT)0;
            } else r[5 + (q - QS)] = v;
        } else if (FLAT) {
            if (q == QS) {
#pragma unroll
                for (int j = 0; j < 6; j++) r[flat_col<MOD>(QS) + j] = ((int)v == j) ? (OUT)1 : (OUT)0;
            } else r[flat_col<MOD>(q)] = v;
        } else r[q] = v;
    }
    __device__ __forceinline__ void put_u(int q, OUT v) const { put(q, v); }
    __device__ __forceinline__ void put_idx(unsigned hour_idx, unsigned day_idx) const
    {
        if (SPLIT) { r[14] = (OUT)hour_idx; r[15] = (OUT)day_idx; }
    }
    // rows = address of the wave's first row; all 64 lanes of the wave must be live
    // PARTIAL = false: all 64 rows exist.  PARTIAL = true: only the first n_valid rows do (the batch's last, ragged wave): whole
    // 16-byte pieces of that prefix, then its last few elements one by one -- the rows are one contiguous block either way, so
    // a ragged wave needs no second, per-lane addressing scheme (that second path cost the masked kernels ~100 spilled registers)
    template <bool PARTIAL = false>
    __device__ __forceinline__ void flush(OUT* rows, int n_valid = 64) const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int n_el = n_valid * F, n16 = PARTIAL ? n_el / EPP : N4;
        if (PITCH == F) {
#pragma unroll
            for (int j = 0; j < (N4 + 63) / 64; j++) {
                const int g = lane + 64 * j;
                if (g < N4 && (!PARTIAL || g < n16)) __builtin_nontemporal_store(((const vf4*)t)[g], (vf4*)rows + g);
            }
        } else {                                            // padded pitch (SB3_FLAT / SPLIT rows): EPP element-sized LDS reads per piece; two pieces
#pragma unroll 2                                            // in flight keep the kernel inside its register budget
            for (int j = 0; j < (N4 + 63) / 64; j++) {
                const int g = lane + 64 * j;
                if (g < N4 && (!PARTIAL || g < n16)) {
                    const OUT* src = t + (g / (F / EPP)) * PITCH + (g % (F / EPP)) * EPP;      // F is a multiple of EPP: a piece never straddles two rows
                    if constexpr (EPP == 4) __builtin_nontemporal_store(vf4{(float)src[0], (float)src[1], (float)src[2], (float)src[3]}, (vf4*)rows + g);
                    else __builtin_nontemporal_store(vd2{(double)src[0], (double)src[1]}, (vd2*)rows + g);
                }
            }
        }
        if (PARTIAL && PITCH == F) {                        // (padded pitch: F is a multiple of 4, no tail)
            const int q = n16 * EPP + lane;
            if (lane < EPP && q < n_el) __builtin_nontemporal_store(t[q], rows + q);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // flush of a full tile with THREE 16-byte pieces in flight.  The plain loop above compiles to read - wait - store nine times over (the
    // compiler reuses one register quartet): nine LDS round trips per step, exposed where no second wave fills the SIMD -- small batches:
    // 0.65 -> 0.60 us per fused step at 4 096 envs.  At 65 536 envs (two waves per SIMD) it bought nothing and the first launch after a
    // reset measured ~1 us slower (profiles/r03_prologue_ab.txt), so k_rollout_pc takes it for small workgroups only.
    __device__ __forceinline__ void flush3(OUT* rows) const
    {
        if constexpr (PITCH != F) { flush(rows); return; }
        else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            constexpr int NJ = (N4 + 63) / 64;
#pragma unroll
            for (int j0 = 0; j0 < NJ; j0 += 3) {
                vf4 v[3];
#pragma unroll
                for (int u = 0; u < 3; u++) { const int g = lane + 64 * (j0 + u); if (j0 + u < NJ && g < N4) v[u] = ((const vf4*)t)[g]; }
#pragma unroll
                for (int u = 0; u < 3; u++) { const int g = lane + 64 * (j0 + u); if (j0 + u < NJ && g < N4) __builtin_nontemporal_store(v[u], (vf4*)rows + g); }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
};

template <bool MOD, class Sink, typename OUT>
__device__ __forceinline__ void hot_store_obs(const HotParams& P, const Sink& row, const HotLoads<OUT>& Q, int s)
{
#pragma unroll
    for (int q = 0; q < 13; q++) row.put(q, Q.fa[q]);
#pragma unroll
    for (int q = 0; q < (MOD ? 13 : 4); q++) row.put(13 + q, Q.fb[q]);
    constexpr int o = MOD ? 26 : 17;
    row.put(o + 0, (OUT)s);
    put_rec_feats<false>(P, row, o, Q.rec);
    row.put(o + 7, Q.sc.x);
    row.put(o + 8, Q.sc.y);
    row.put_idx(Q.hb4 >> 2, Q.db4 >> 2);
}

__device__ __forceinline__ void hot_stage_lds(const HotParams& P, HotLds& L)
{
    if (threadIdx.x < NT) L.tm[threadIdx.x] = P.tabmeta[threadIdx.x];
    else if (threadIdx.x >= 32 && threadIdx.x < 32 + LAD_N) L.lad[threadIdx.x - 32] = P.ladder[threadIdx.x - 32];
}

// raw action -> action id; an invalid discrete action flags the error word and is replaced by the previous action
__device__ __forceinline__ int hot_decode(int actk, const HotParams& P, int raw_i, float raw_f, unsigned flags)
{
    const int prev = (flags >> 12) & 7;
    if (actk == PTG_ACT_F32) return decode_continuous(raw_f, prev);
    const bool bad = (raw_i < -5) | (raw_i > 4);
    if (__ballot(bad)) { if (bad) P.err[0] = 1; }
    return bad ? prev : (raw_i < 0 ? raw_i + 5 : raw_i);
}

// actk (PTG_ACT_*) is a launch-uniform kernel argument: the element type of the action buffer costs one scalar branch, not a
// third of the kernel instantiations
__device__ __forceinline__ void hot_fetch(int actk, const void* actions, size_t g, int& ri, float& rf)
{
    if (actk == PTG_ACT_F32) rf = ((const float*)actions)[g];
    else if (actk == PTG_ACT_I64) { const long long v = ((const long long*)actions)[g]; ri = (v < -5 || v > 4) ? 99 : (int)v; }
    else ri = ((const int*)actions)[g];
}

#ifdef PTG_STAMPS      // diagnostic build (tools/stamps.py, tools/step_stamps.py): 100 MHz wall-clock stamps of the phases of a launch, per workgroup and role
__device__ unsigned long long g_stamps[256][2][8];
#define PTG_STAMP(slot_) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && ((threadIdx.x >> 6) == 0 || (int)(threadIdx.x >> 6) == (NP >> 6))) \
        g_stamps[blockIdx.x][(threadIdx.x >> 6) == 0 ? 0 : 1][slot_] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define ST_STAMP(slot_) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_stamps[blockIdx.x][0][slot_] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define ST_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define PTG_STAMP(slot_) do {} while (0)
#define ST_STAMP(slot_) do {} while (0)
#define ST_DRAIN() do {} while (0)
#endif

// one vector step, no env terminates (host-guaranteed); lanes past N shadow env N-1 and store nothing
// LAY: 0 row-major, 1 feature-major, 2 SB3_FLAT rows (the PTG_OBS_* values)
template <int LAY, bool MOD, int NOISE, typename OUT>
__global__ void __launch_bounds__(256)
k_step_hot(const HotParams P, const void* __restrict__ actions, int actk, OUT* __restrict__ obs, OUT* __restrict__ rew,
           uint8_t* __restrict__ done, int skip_term)
{
    constexpr bool FM = LAY == PTG_OBS_FEATURE_MAJOR, FLAT = LAY == PTG_OBS_SB3_FLAT, SPLIT = LAY == PTG_OBS_SPLIT;
    __shared__ HotLds L;
    __shared__ __attribute__((aligned(16))) OUT s_tile[FM ? 4 : 4 * RowTile<MOD, FLAT, OUT, SPLIT>::TILE];   // row-major / flat / split: one tile per wave
    const int e_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e_raw < P.N;
    const int e = live ? e_raw : P.N - 1;
    ST_STAMP(0);
    const StA a = P.st_a[e]; const StB b = P.st_b[e];     // state + action loads in flight while LDS is staged
    int ri = 0; float rf = 0.f;
    hot_fetch(actk, actions, (size_t)e, ri, rf);
    hot_stage_lds(P, L);
    __syncthreads();
    ST_STAMP(1); ST_DRAIN(); ST_STAMP(2);
    HotRegs R;
    R.i = a.i; R.j = a.j; R.k = a.k; R.flags = a.flags; R.cum = b.cum; R.act_d = b.act_d; R.nctr = b.nctr;
    // The common step count comes from the STATE, not from a kernel argument: a launch captured into a hipGraph (policy forward + this
    // step, say) can be replayed step after step.  What a replay cannot do is route the one terminating step of an episode to the generic
    // kernel -- the host does that for eager calls --, so a hot kernel that finds itself on that step raises the sequence flag (ptg_note_replays).
    const int k0 = __builtin_amdgcn_readfirstlane(a.k);
    if (skip_term) {                                        // captured form: tell the generic kernel enqueued behind this one whether the step is its
        const bool term = k0 >= P.eps_sim_steps - 6;
        if (blockIdx.x == 0 && threadIdx.x == 0) *P.term_flag = term ? 1 : 0;
        if (term) return;
    } else if (k0 >= P.eps_sim_steps - 6) { if (threadIdx.x == 0) P.err[2] = 1; }
    const double2 setc = P.setc[(R.flags >> 15) & 3];
    const int act = hot_decode(actk, P, ri, rf, R.flags);
    HotLoads<OUT> Q;
    Ladder G;
    load_ladder(L, G);
    hot_front<MOD, NOISE, OUT>(P, L, G, nullptr, false, R, act, e, k0 + 1, Q);
    ST_STAMP(3);
    const OUT r = hot_back<OUT>(P, R, Q, setc, e, live);
#ifdef PTG_STAMPS
    if (r == (OUT)123456.789) ST_STAMP(7);                  // keeps stamp 4 behind the reward (the record has arrived)
#endif
    ST_STAMP(4);
    const int e_wave = __builtin_amdgcn_readfirstlane(e_raw);                 // the wave's first env
    if (!FM && e_wave < P.N) {                                                   // rows: through the wave's LDS tile, out as one block
        OUT* rows = obs + (size_t)e_wave * P.F;
        const RowTile<MOD, FLAT, OUT, SPLIT> tile(s_tile, threadIdx.x >> 6);
        hot_store_obs<MOD>(P, tile, Q, R.flags & 7);
        if (e_wave + 63 < P.N) tile.flush(rows);
        else tile.template flush<true>(rows, P.N - e_wave);
    }
    // no `if (live)`: a lane past the batch shadows env N - 1 exactly (same state, same action, same noise stream), so what it
    // stores is the value the real lane stores to the same address -- harmless, and it keeps the stores out of divergent code
    if (FM) hot_store_obs<MOD>(P, HotRow<true, MOD, OUT>(obs, P, e), Q, R.flags & 7);
    st_off<OUT>(rew, (unsigned)e * (unsigned)sizeof(OUT), r);
    st_off<uint8_t>(done, (unsigned)e, 0);
    StA na; na.i = R.i; na.j = R.j; na.k = k0 + 1; na.flags = R.flags;
    StB nb; nb.cum = R.cum; nb.act_d = R.act_d; nb.nctr = R.nctr;
    P.st_a[e] = na; P.st_b[e] = nb;
    ST_STAMP(5); ST_DRAIN(); ST_STAMP(6);
}

// Producer / consumer form of the fused rollout ("split gather").  Half of every workgroup's waves (producers) run ONLY the
// integer state machine: its loop-carried dependence on memory is the temperature key of the window just entered, so the
// producers gather 2 bytes per env and step (rkey[], the keys of all window records as one uint16 array) and hand
// {record index, METH_STATUS, state-changed} to the other half as one LDS word.  The consumers own everything that is not on
// that chain: the 64-byte record gather, the reward and cum_rew, the 26 market features and three prices (functions of the
// hour -- reloaded when the hour changes, one step ahead of use), the clock features (scalar loads: the step count is
// uniform) and ALL stores.  A consumer issues the gather of step t, then finishes step t-1 (whose record was requested one
// iteration earlier), so no wave ever waits on a load it has just issued.  One LDS-only barrier per step.
struct PcSlot { unsigned w[256]; };      // record index | METH_STATUS << 24 | changed << 27 | current action << 28 | hot_cold << 31

template <typename OUT>
struct PcMarket {
    OUT fa[13], fb[13];
    double el, gas, eua;
};

template <bool MOD, typename OUT>
__device__ __forceinline__ void pc_load_market(const HotParams& P, unsigned hb4, unsigned db4, PcMarket<OUT>& M)
{
    constexpr unsigned W = sizeof(OUT) / 4;
    const OUT* pA = series(P, (const OUT*)nullptr, 0);
#pragma unroll
    for (int q = 0; q < 13; q++) M.fa[q] = ld_off<OUT>(pA + q, hb4 * W);
    if (MOD) {
        const OUT* pB = series(P, (const OUT*)nullptr, 1);
#pragma unroll
        for (int q = 0; q < 13; q++) M.fb[q] = ld_off<OUT>(pB + q, hb4 * W);
    } else {
        const OUT* pG = series(P, (const OUT*)nullptr, 2);
        const OUT* pU = series(P, (const OUT*)nullptr, 3);
        M.fb[0] = ld_off<OUT>(pG, db4 * W); M.fb[1] = ld_off<OUT>(pG + 1, db4 * W);
        M.fb[2] = ld_off<OUT>(pU, db4 * W); M.fb[3] = ld_off<OUT>(pU + 1, db4 * W);
    }
    M.el = ld_off<double>(P.pool64, hb4 * 2u);
    M.gas = ld_off<double>(P.pool64 + P.off_gas, db4 * 2u);
    M.eua = ld_off<double>(P.pool64 + P.off_eua, db4 * 2u);
}

// The table refresher.  The records and keys a step gathers are served from the XCD's L2 or the Infinity Cache as long as somebody
// touched their lines recently -- true for a batch whose envs are spread over the tables, false for a SYNCHRONISED batch: after a
// reset (or under a policy that drives every env alike) the envs walk through the tables as a front, every step gathers lines
// nobody has touched since the output stream (9.8 MB per step at 65 536 envs) flushed them from the Infinity Cache, and each
// of those misses goes to DRAM underneath a saturated write stream: measured 2.5 us per step instead of 1.5 for the first
// ~270 steps after a reset (tools/fresh20.py, tools/phase_pmc.sh: same instruction counts, read latency + 40 %, TCP pending-line
// stalls + 50 %; touching the tables right before a launch removes it).  k_refresh runs BESIDE k_rollout_pc, on a stream of its
// own (one 64-lane workgroup per CU, no LDS, a handful of registers: it fits next to the rollout's workgroup): it re-reads its
// 1 / gridDim share of the records and keys `passes` times, one pass per `period` ticks of the 100 MHz clock.  The loads are
// inline asm into one never-read register quartet -- fire and forget.  It shares no vmcnt queue with the rollout's waves (an
// in-order queue: a slow refresh load would hold back the gathers behind it; and a ninth wave inside the rollout's workgroup
// caps the kernel at 168 registers, which spills), and it ends by itself after passes x period: no flag, no polling.
// this part's 1 / parts share of the [records | keys] image, as 1 KiB rows (64 lanes x 16 bytes), `waves` waves taking rows in turn
__device__ __forceinline__ void refresh_rows(const void* __restrict__ recs, const unsigned short* __restrict__ rkey, unsigned n_rec, unsigned part,
                                             unsigned parts, unsigned wave, unsigned waves, unsigned lane, uv4& sink)
{
    const unsigned n16_rec = n_rec * 4u, n16 = n16_rec + (n_rec * 2u + 15u) / 16u;                       // 16-byte pieces: records, then keys
    const unsigned rows = (n16 + 63u) / 64u, per = (rows + parts - 1) / parts;
    const unsigned r_lo = min(rows, part * per), r_hi = min(rows, r_lo + per);
    for (unsigned row = r_lo + wave; row < r_hi; row += waves) {
        const unsigned g = min(row * 64u + lane, n16 - 1u);
        const char* src = g < n16_rec ? (const char*)recs + (size_t)g * 16u : (const char*)rkey + (size_t)(g - n16_rec) * 16u;
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(src) : "memory");
    }
}

// Pass p (p = p0 .. passes - 1) starts p periods after the kernel did.  p0 = 1: the launch's head pass is done by the rollout kernel
// itself, in its prologue (k_rollout_pc, `refresh_rec`), and this kernel -- forked from the rollout's stream right before it, so it
// starts when the rollout does -- only adds the rolling passes of a long launch over a batch that is still a front.
__global__ void __launch_bounds__(64)
k_refresh(const void* __restrict__ recf, const unsigned short* __restrict__ rkey, int n_rec, int p0, int passes, unsigned period)
{
    uv4 sink = {0u, 0u, 0u, 0u};
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int p = p0; p < passes; p++) {
        if (p > 0) {                                        // pace (the clock only runs forward)
            const unsigned long long target = t0 + (unsigned long long)p * period;
            while (__builtin_amdgcn_s_memrealtime() < target) __builtin_amdgcn_s_sleep(64);
        }
        refresh_rows(recf, rkey, (unsigned)n_rec, blockIdx.x, gridDim.x, 0u, 1u, threadIdx.x, sink);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" :: "v"(sink));
}

// What the NEXT hour adds to a consumer's market registers: the 13-hour windows slide by one (window(H + 1)[q] = window(H)[q + 1]),
// so one new element per series and the three prices are all that has to be fetched ahead of the hour change -- not a second
// copy of the 26 features (58 registers of float64 features that made the float64 kernel spill).
template <bool MOD, typename OUT>
struct PcNext {
    OUT fa12, fb12;              // Pot_Reward / Part_Full ('raw': Elec_Price) of hour H + 13
    OUT fbr[MOD ? 1 : 4];        // 'raw': Gas_Price[2], EUA_Price[2] of the next hour's day
    double el, gas, eua;
};

template <bool MOD, typename OUT>
__device__ __forceinline__ void pc_load_next(const HotParams& P, unsigned hb4, unsigned db4, PcNext<MOD, OUT>& X)
{
    constexpr unsigned W = sizeof(OUT) / 4;
    X.fa12 = ld_off<OUT>(series(P, (const OUT*)nullptr, 0) + 12, hb4 * W);
    if (MOD) X.fb12 = ld_off<OUT>(series(P, (const OUT*)nullptr, 1) + 12, hb4 * W);
    else {
        const OUT* pG = series(P, (const OUT*)nullptr, 2);
        const OUT* pU = series(P, (const OUT*)nullptr, 3);
        X.fbr[0] = ld_off<OUT>(pG, db4 * W); X.fbr[1] = ld_off<OUT>(pG + 1, db4 * W);
        X.fbr[2] = ld_off<OUT>(pU, db4 * W); X.fbr[3] = ld_off<OUT>(pU + 1, db4 * W);
    }
    X.el = ld_off<double>(P.pool64, hb4 * 2u);
    X.gas = ld_off<double>(P.pool64 + P.off_gas, db4 * 2u);
    X.eua = ld_off<double>(P.pool64 + P.off_eua, db4 * 2u);
}


template <int LAY, bool MOD, int NOISE, bool LDSLUT, bool FULL, typename OUT, bool INFO = false>
__global__ void __launch_bounds__(512)
k_rollout_pc(const HotParams P, const void* __restrict__ actions, int actk, int T, OUT* __restrict__ obs, OUT* __restrict__ rew,
             uint8_t* __restrict__ done, const unsigned short* __restrict__ lut16, const unsigned short* __restrict__ rkey, int e_base, int vec_rows,
             double* __restrict__ info, int refresh_rec)
{
    constexpr bool FM = LAY == PTG_OBS_FEATURE_MAJOR, FLAT = LAY == PTG_OBS_SB3_FLAT, SPLIT = LAY == PTG_OBS_SPLIT;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const int NP = blockDim.x / 2;                          // envs per workgroup (64, 128 or 256): NP producer lanes, NP consumer lanes
    const int nwork = blockDim.x;
    PTG_STAMP(0);
    HotLds& L = *(HotLds*)s_dyn;
    PcSlot* slot = (PcSlot*)(s_dyn + 16 * ((sizeof(HotLds) + 15) / 16));
    typedef typename HotTypes<OUT>::rec_t rec_t;
    constexpr unsigned B = sizeof(OUT);
    // The head pass of the table refresher (see k_refresh), inside this kernel's own duration: workgroup b re-reads its 1 / gridDim share
    // of the records + keys (33 KiB at 256 workgroups) with fire-and-forget loads, the first thing it does -- they fly while the lookup
    // table and the action rows are staged and have landed in the L2s / the Infinity Cache before the first gather (the prologue's
    // barrier waits for them).  refresh_rec = 0: the host found nothing to refresh (little written since the last pass).
    uv4 rsink = {0u, 0u, 0u, 0u};
    if (refresh_rec)
        refresh_rows(rec_table(P, (const rec_t*)nullptr), rkey, (unsigned)refresh_rec, blockIdx.x, gridDim.x, threadIdx.x >> 6, blockDim.x >> 6,
                     threadIdx.x & 63u, rsink);
    OUT* s_tiles = (OUT*)((unsigned char*)slot + 2 * sizeof(PcSlot));                   // row-major: one [64][F] tile per consumer wave
    // float64 outputs: the 26 (17) per-env market features of the current hour live in LDS, [feature][env] -- as registers they
    // are 52 of the 256 a lane may have, and the kernel spilled.  The two 13-hour windows are rings: slot of element q = (q + head) mod 13
    constexpr bool MLDS = sizeof(OUT) == 8;
    constexpr int NFM = MOD ? 26 : 17;
    OUT* s_mkt = (OUT*)((unsigned char*)s_tiles + (FM ? 0 : (size_t)(NP >> 6) * RowTile<MOD, FLAT, OUT, SPLIT>::TILE * B));
    unsigned char* s_act = (unsigned char*)s_mkt + (MLDS ? (size_t)NFM * NP * B : 0);      // [T][NP] decoded actions
    unsigned short* s_lut = (unsigned short*)(s_act + 16 * (((size_t)T * NP + 15) / 16));
    const bool producer = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < NP;     // wave-uniform: NP is a multiple of 64
    const int lx = producer ? threadIdx.x : threadIdx.x - NP;
    const int e_raw = e_base + blockIdx.x * NP + lx;        // this launch covers envs [e_base, e_base + gridDim.x * NP)
    // FULL: the batch fills every workgroup -- no divergent region around the stores, so the backend's vmcnt bookkeeping stays
    // exact (a skipped-stores path makes it assume the worst and drain the queue every step)
    const bool live = FULL || e_raw < P.N;
    const int e = live ? e_raw : P.N - 1;
    const int e_wave = __builtin_amdgcn_readfirstlane(e_raw);                            // the wave's first env
    const bool wave_full = FULL || e_wave + 63 < P.N;                                    // all 64 envs of this wave exist
    const StA a = P.st_a[e]; const StB b = P.st_b[e];
    HotRegs R;
    R.i = a.i; R.j = a.j; R.k = a.k; R.flags = a.flags; R.cum = b.cum; R.act_d = b.act_d; R.nctr = b.nctr;
    if (LDSLUT) {
        // The 58 KB lookup goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: one 1 KiB row per wave instruction, no register
        // round trip, nothing to wait for before the barrier's own vmcnt(0)); the 32 workgroups of an XCD start at different rows so
        // that they do not walk the same L2 channel in lock-step.  Issued BEFORE the action rows are requested: the two latencies overlap.  (Staged through registers, four 16-byte loads in flight per
        // lane, this copy took 2.7 us of every launch: in-kernel stamps, profiles/r02_tsweep.txt.)
        const int n16 = (N_DEST * P.nT * 2 + 15) / 16, rows = (n16 + 63) / 64;
        const int wave = threadIdx.x >> 6, nw = nwork >> 6, lane = threadIdx.x & 63;
        const int rot = (int)(((blockIdx.x >> 3) & 31u) * (unsigned)(rows / 32));
        for (int r0 = wave; r0 < rows; r0 += nw) {
            int r = r0 + rot;
            r = r >= rows ? r - rows : r;
            const uint4* src = (const uint4*)lut16 + min(r * 64 + lane, n16 - 1);      // the last row's spare lanes re-read the last piece
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)((unsigned char*)s_lut + (size_t)r * 1024), 16, 0, 0);
        }
    }
    // The whole launch's action rows are staged into LDS up front, decoded to one byte each (7 = "keep the previous action":
    // no threshold fired, :351-355, or an invalid discrete action, which also raises the error flag).  A fresh action row
    // comes from HBM; fetched step by step its latency under the write stream -- vmcnt retires in order, so one slow load holds
    // back every younger one -- cost more than the whole rest of the step (2.9 vs 1.55 us per step at N = 65 536).
    {
        const int sh_np = __ffs(NP) - 1, total = T << sh_np;
        const int e_wg = e_base + blockIdx.x * NP;
        bool bad_any = false;           // flagged once after the loops: a ballot inside would keep them from unrolling
        // The element type is hoisted out of the loops (one instantiation per type): with the type test inside, every
        // unrolled trip was its own basic block ending in `global_load; s_waitcnt vmcnt(0)` -- T / 8 serialised HBM round trips
        // at the head of every launch.  Here each trip issues its four row loads back to back and decodes afterwards.
        auto stage = [&](auto kind_tag) {
            constexpr int AK = decltype(kind_tag)::value;
            auto code_i = [&](int ri) -> int {
                const bool bad = (ri < -5) | (ri > 4);
                bad_any |= bad;
                return bad ? 7 : (ri < 0 ? ri + 5 : ri);
            };
            auto pack = [&](const int* c) -> unsigned { return (unsigned)c[0] | ((unsigned)c[1] << 8) | ((unsigned)c[2] << 16) | ((unsigned)c[3] << 24); };
            if (FULL && vec_rows) {      // four envs per lane: one dwordx4 (two for int64 actions) per quad, one packed LDS word
                const int nq = total >> 2, bd = nwork;
                for (int q0 = threadIdx.x; q0 < nq; q0 += 4 * bd) {
                    if (AK == PTG_ACT_I64) {
                        longlong2 v[4][2];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int q = min(q0 + u * bd, nq - 1), idx = q << 2, t = idx >> sh_np, x = idx & (NP - 1);
                            const long long* src = (const long long*)actions + ((size_t)t * P.N + e_wg + x);
                            v[u][0] = *(const longlong2*)src; v[u][1] = *(const longlong2*)(src + 2);
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int q = q0 + u * bd;
                            const long long w[4] = {v[u][0].x, v[u][0].y, v[u][1].x, v[u][1].y};
                            int c[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) c[j] = code_i((w[j] < -5 || w[j] > 4) ? 99 : (int)w[j]);
                            if (q < nq) *(unsigned*)(s_act + (q << 2)) = pack(c);
                        }
                    } else {
                        int4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int q = min(q0 + u * bd, nq - 1), idx = q << 2, t = idx >> sh_np, x = idx & (NP - 1);
                            v[u] = *(const int4*)((const int*)actions + ((size_t)t * P.N + e_wg + x));
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int q = q0 + u * bd;
                            int c[4];
                            if (AK == PTG_ACT_F32) {
                                c[0] = decode_continuous(__int_as_float(v[u].x), 7); c[1] = decode_continuous(__int_as_float(v[u].y), 7);
                                c[2] = decode_continuous(__int_as_float(v[u].z), 7); c[3] = decode_continuous(__int_as_float(v[u].w), 7);
                            } else {
                                c[0] = code_i(v[u].x); c[1] = code_i(v[u].y); c[2] = code_i(v[u].z); c[3] = code_i(v[u].w);
                            }
                            if (q < nq) *(unsigned*)(s_act + (q << 2)) = pack(c);
                        }
                    }
                }
            } else {
#pragma unroll 4
                for (int idx = threadIdx.x; idx < total; idx += nwork) {
                    const int t = idx >> sh_np, x = idx & (NP - 1);
                    const int eg = min(e_wg + x, P.N - 1);
                    const size_t g = (size_t)t * P.N + eg;
                    int code;
                    if (AK == PTG_ACT_F32) code = decode_continuous(((const float*)actions)[g], 7);
                    else if (AK == PTG_ACT_I64) { const long long w = ((const long long*)actions)[g]; code = code_i((w < -5 || w > 4) ? 99 : (int)w); }
                    else code = code_i(((const int*)actions)[g]);
                    s_act[idx] = (unsigned char)code;
                }
            }
        };
        if (actk == PTG_ACT_F32) stage(std::integral_constant<int, PTG_ACT_F32>{});
        else if (actk == PTG_ACT_I64) stage(std::integral_constant<int, PTG_ACT_I64>{});
        else stage(std::integral_constant<int, PTG_ACT_I32>{});
        if (__ballot(bad_any)) { if (bad_any) P.err[0] = 1; }
    }
    PTG_STAMP(1);
    hot_stage_lds(P, L);
    PTG_STAMP(2);
    if (refresh_rec) {                                      // (uniform) the refresh loads have landed: their target registers are free again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(rsink));
    }
    __syncthreads();
    PTG_STAMP(3);
    // step count of the launch's first step: from the state (a captured launch can be replayed, see k_step_hot).  Read HERE, behind the
    // prologue's barrier, where the state has long arrived: right behind its load the wait would hold up the staging loads (+0.5 us per launch)
    const int k0 = __builtin_amdgcn_readfirstlane(R.k);
    if (k0 + T > P.eps_sim_steps - 6) { if (threadIdx.x == 0) P.err[2] = 1; }      // the launch would run over the episode's terminating step
    const unsigned short* lut = LDSLUT ? s_lut : nullptr;
    const unsigned mset = (R.flags >> 15) & 3;
    // series offsets of step count k1 for this env (:442-447); uniform except for the episode offset
    auto offsets = [&](int k1, unsigned& hb4, unsigned& db4) {
        const int secs = k1 * P.sim_step;
        int H = R.act_d * 24 + secs / 3600, D = R.act_d + secs / 86400;
        const bool oob = (H + 13 > P.n_hours) | (D + 2 > P.n_days) | (H < 0) | (D < 0);
        if (__ballot(oob)) {
            if (oob) { P.err[1] = 1; H = max(0, min(H, P.n_hours - 13)); D = max(0, min(D, P.n_days - 2)); }
        }
        hb4 = (mset * (unsigned)P.hstride + (unsigned)H) * 4u;
        db4 = (mset * (unsigned)P.dstride + (unsigned)D) * 4u;
    };
    PcMarket<OUT> M;                                        // consumer: market data of the step being finished ...
    PcNext<MOD, OUT> Mn;                                    // ... and what the next hour adds to it
    double2 setc = make_double2(0.0, 0.0);
    int hour_cur = 0, head = 0;                             // head: first slot of the feature rings (uniform: the hour is)
    unsigned hb_cur = 0, db_cur = 0;                        // SPLIT layout: series indices of the current hour / day
    auto mkt_to_lds = [&]() {                               // the window just loaded into M -> LDS, rings at head 0
        if (MLDS) {
#pragma unroll
            for (int q = 0; q < 13; q++) s_mkt[q * NP + lx] = M.fa[q];
#pragma unroll
            for (int q = 0; q < (MOD ? 13 : 4); q++) s_mkt[(13 + q) * NP + lx] = M.fb[q];
            head = 0;
        }
    };
    auto mk = [&](const int q) -> OUT {                     // canonical market column q of the current hour
        if (!MLDS) return q < 13 ? M.fa[q] : M.fb[q - 13];
        if (q >= 13 && !MOD) return s_mkt[q * NP + lx];
        int r = (q < 13 ? q : q - 13) + head;
        r = r >= 13 ? r - 13 : r;
        return s_mkt[((q < 13 ? 0 : 13) + r) * NP + lx];
    };
    unsigned tk = 0;                                        // producer: key of the window entered by the previous step (in flight)
    double z_next = 0.0;                                    // producer, tape mode: the env's next tape entry (in flight)
    rec_t recA, recB; unsigned wA = 0, wB = 0;              // consumer: records in flight (ping-pong: no copies of pending loads)
    if (!producer) {
        setc = P.setc[mset];
        unsigned hb4, db4;
        offsets(k0 + 1, hb4, db4);
        hour_cur = ((k0 + 1) * P.sim_step) / 3600;
        pc_load_market<MOD, OUT>(P, hb4, db4, M);
        mkt_to_lds();
        hb_cur = hb4 >> 2; db_cur = db4 >> 2;
    }
    const unsigned NF4 = (unsigned)(FM ? P.fm_pitch : P.N) * (unsigned)P.F * B;      // bytes of one step's observation block
    char* obs_t = (char*)obs; char* rew_t = (char*)rew; char* done_t = (char*)done;      // consumer: rows of the step being finished
    double* info_t = info;
    // hand-off barrier: only the LDS traffic has to be complete.  __syncthreads() would also drain vmcnt -- the consumers'
    // stores and the gathers just issued -- once per step, which serialises exactly what this kernel overlaps
    auto handoff = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#ifdef PTG_ABLATE_KEYLAG      // TIMING-ONLY ablation (wrong results): the state machine consumes the key gathered TWO steps ago (two registers used in
    unsigned tkB = 0;         // turn), so the newest gather is never on the loop-carried chain: the upper bound of what any key run-ahead / speculation
#endif                        // scheme could gain for chain-bound shapes (small batches, SPLIT rows); tools/r03_keylag.sh
    Ladder G;                                               // producers: the ladder thresholds, in registers for the whole launch
    int code_nx = 0;                                        // producers: the NEXT step's action byte, read one step ahead of its use --
    if (producer) { load_ladder(L, G); code_nx = s_act[lx]; }      // its address depends on nothing, its LDS latency need not sit on the chain
    auto produce_ = [&](const int it, unsigned& tk) {       // state machine of step `it`
        if (it > 0) R.flags = (R.flags & 0x1FFFFu) | (tk << 17);                  // Meth_T_cat = op[-1, 1] (:452)
        const int code = code_nx;
        code_nx = s_act[min(it + 1, T - 1) * NP + lx];
        const int act = (code == 7) ? (int)((R.flags >> 12) & 7) : code;
        bool changed;
        const int nctr0 = R.nctr;
        const int ridx = hot_ints<NOISE, NOISE == NOISE_TAPE>(P, L, G, lut, LDSLUT, R, act, e, changed, z_next);
        tk = ld_off<unsigned short>(rkey, (unsigned)ridx * 2u);
        // tape mode: the draw the env will consume next is fetched as soon as the previous one is used up -- steps ahead of its use,
        // as a rule (a load inside the step that needs it is a second dependent round trip: 2.25 us per step against 1.5)
        if (NOISE == NOISE_TAPE) {
            const bool used = R.nctr != nctr0;
            if (__ballot(used)) { if (used) z_next = P.tape[(size_t)((unsigned)R.nctr % (unsigned)P.tape_len) * P.N + e]; }
        }
        slot[it & 1].w[lx] = (unsigned)ridx | ((R.flags & 7u) << 24) | (changed ? (1u << 27) : 0u) |
                             (((R.flags >> 12) & 7u) << 28) | (((R.flags >> 3) & 1u) << 31);       // + action, hot / cold: the info rows' fields
    };
    auto produce = [&](const int it) { produce_(it, tk); };
    auto request = [&](const int t, rec_t& rec, unsigned& w) {                    // consumer: record gather of step t
        w = slot[t & 1].w[lx];
        rec = ld_off<rec_t>(rec_table(P, (const rec_t*)nullptr), (w & 0xFFFFFFu) * 64u);
    };
    // consumer: finish step t (record already requested), optionally requesting step t+1 in between.  Issue order = retire
    // order (vmcnt): clock + next-hour loads, then the gather, then the stores -- nothing the stores need sits behind the gather
    auto finish = [&](const int t, const rec_t& rec, const unsigned w, const bool more, rec_t& recN, unsigned& wN) {
        const int k1 = k0 + t + 1;
        const int hour_next = ((k1 + 1) * P.sim_step) / 3600;
        const bool reload = (hour_next != hour_cur) && (t + 1 < T);
        const auto sc = ld_off<typename HotTypes<OUT>::sc_t>(series(P, (const OUT*)nullptr, 4), (unsigned)min(k1, P.eps_sim_steps) * 2u * B);
        const bool slide = hour_next == hour_cur + 1;       // uniform (always, for steps of at most an hour)
        unsigned hb4n = 0, db4n = 0;
        if (reload) {                                       // uniform branch
            offsets(k1 + 1, hb4n, db4n);
            if (slide) pc_load_next<MOD, OUT>(P, hb4n, db4n, Mn);
        }
        if (more) request(t + 1, recN, wN);
        const bool changed = (w >> 27) & 1u;
        double rw;
        RewTerms rt;
        if constexpr (sizeof(OUT) == 8) { rt = reward_ref(P.rc, rec.m[0], rec.m[1], rec.m[2], rec.m[3], rec.m[4], M.el, M.gas, M.eua, setc.x); rw = rt.rew; }
        else rw = hot_reward(P, rec, M.el, M.gas, M.eua, setc.x);
        R.cum += rw;
        rw -= changed ? setc.y : 0.0;                                            // :332
        if constexpr (INFO && sizeof(OUT) == 8) {           // its own instantiation: six inlined copies of these stores in every float64
            if (info_t) {                                   // kernel cost 8 % at 20 steps per launch (instruction footprint).  _get_info (:251-278)
                double* ir = info_t + (size_t)e * PTG_N_INFO;
                ir[0] = (double)(k1 - 1); ir[1] = M.el; ir[2] = M.gas; ir[3] = M.eua;
                ir[4] = (double)((w >> 24) & 7u); ir[5] = (double)((w >> 28) & 7u); ir[6] = (double)(w >> 31);
                ir[7] = rec.T; ir[8] = rec.m[0]; ir[9] = rec.m[1]; ir[10] = rec.m[3]; ir[11] = rec.m[4];
                ir[12] = rt.ch4_rev; ir[13] = rt.steam_rev; ir[14] = rt.o2_rev; ir[15] = rt.eua_rev; ir[16] = rt.chp_rev;
                ir[17] = -rt.cost_heat; ir[18] = -rt.cost_elz; ir[19] = -rt.cost_water; ir[20] = rw; ir[21] = R.cum;
                ir[22] = P.pot_raw[hb_cur]; ir[23] = P.pf_raw[hb_cur];
                info_t += (size_t)P.N * PTG_N_INFO;
            }
        }
        if (P.track_changes) { if (changed && live) P.st_c[e].nchg += 1; }
        auto emit = [&](const auto& row) {
#pragma unroll
            for (int q = 0; q < (MOD ? 26 : 17); q++) row.put_u(q, mk(q));
            constexpr int o = MOD ? 26 : 17;
            row.put_u(o + 0, (OUT)((w >> 24) & 7u));
            put_rec_feats<true>(P, row, o, rec);
            row.put_u(o + 7, sc.x);
            row.put_u(o + 8, sc.y);
            row.put_idx(hb_cur, db_cur);
        };
        if (!FM && (FULL || e_wave < P.N)) {                // row-major: transpose the wave's 64 rows through LDS, one contiguous block out
            OUT* rows = (OUT*)obs_t + (size_t)e_wave * P.F;
            const int cw = (int)(threadIdx.x >> 6) - (NP >> 6);
            const RowTile<MOD, FLAT, OUT, SPLIT> tile(s_tiles, cw);
            emit(tile);
            if (wave_full) { if (NP < 256) tile.flush3(rows); else tile.flush(rows); }      // (uniform: the workgroup size of the launch)
            else tile.template flush<true>(rows, P.N - e_wave);
        }
        // lanes past the batch shadow env N - 1 and store what its real lane stores, to the same addresses: no divergent region
        // around the stores (the masked form made the backend drain the whole vmcnt queue every step: 3.9 vs 1.5 us per step)
        if (FM) emit(HotRow<true, MOD, OUT>((OUT*)obs_t, P, e));
        st_off<OUT>(rew_t, (unsigned)e * B, (OUT)rw);
        st_off<uint8_t>(done_t, (unsigned)e, 0);
        obs_t += NF4; rew_t += (size_t)P.N * B; done_t += P.N;
        if (reload) {
            if (slide && MLDS) {                            // the new elements take the slots of the ones that left the windows
                s_mkt[head * NP + lx] = Mn.fa12;
                if (MOD) s_mkt[(13 + head) * NP + lx] = Mn.fb12;
                else {
#pragma unroll
                    for (int q = 0; q < 4; q++) s_mkt[(13 + q) * NP + lx] = Mn.fbr[q];
                }
                head = head + 1 == 13 ? 0 : head + 1;
                M.el = Mn.el; M.gas = Mn.gas; M.eua = Mn.eua;
            } else if (slide) {
#pragma unroll
                for (int q = 0; q < 12; q++) M.fa[q] = M.fa[q + 1];
                M.fa[12] = Mn.fa12;
                if (MOD) {
#pragma unroll
                    for (int q = 0; q < 12; q++) M.fb[q] = M.fb[q + 1];
                    M.fb[12] = Mn.fb12;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) M.fb[q] = Mn.fbr[q];
                }
                M.el = Mn.el; M.gas = Mn.gas; M.eua = Mn.eua;
            } else { pc_load_market<MOD, OUT>(P, hb4n, db4n, M); mkt_to_lds(); }      // steps longer than an hour: a plain (waited-for) reload
            hour_cur = hour_next;
            hb_cur = hb4n >> 2; db_cur = db4n >> 2;
        }
    };
    // iteration `it`: producers run step it (while it < T); consumers request step it-1 and finish step it-2.  The two roles
    // run SEPARATE loops with the same number (T + 1) of barriers: the role is wave-uniform, and with one role per loop the
    // backend's vmcnt bookkeeping sees only that role's loads and stores (a shared loop with the roles as exec-masked regions
    // made it drain the whole queue at the top of every consumer iteration).
    if (producer) {
        if (NOISE == NOISE_TAPE) z_next = P.tape[(size_t)((unsigned)R.nctr % (unsigned)P.tape_len) * P.N + e];
#ifdef PTG_ABLATE_KEYLAG
        {
            int it = 0;
            for (; it + 1 < T; it += 2) { produce_(it, tk); handoff(); produce_(it + 1, tkB); handoff(); }
            if (it < T) { produce_(it, tk); handoff(); }
        }
#else
        for (int it = 0; it < T; it++) { produce(it); handoff(); if (it == 0) PTG_STAMP(4); if (it == 1) PTG_STAMP(5); }
#endif
        PTG_STAMP(6);
        handoff();
    } else {
        handoff();
        request(0, recA, wA);
        handoff();
        PTG_STAMP(4);
        if (T == 1) {
            finish(0, recA, wA, false, recB, wB);
        } else {
            // the first finish is peeled so that the loop is entered in the state its back edge leaves (a gather followed by
            // a step's stores): the backend then derives the exact vmcnt for "record landed", not the prologue's small one
            finish(0, recA, wA, true, recB, wB);
            PTG_STAMP(5);
            handoff();
            int it = 3;
            for (; it + 1 <= T; it += 2) {                  // two steps per trip: the records ping-pong, no register copies
                finish(it - 2, recB, wB, true, recA, wA);
                handoff();
                finish(it - 1, recA, wA, true, recB, wB);
                handoff();
            }
            if (it <= T) {                                  // it == T: one request left
                finish(it - 2, recB, wB, true, recA, wA);
                handoff();
                finish(T - 1, recA, wA, false, recB, wB);
            } else {
                finish(T - 1, recB, wB, false, recA, wA);
            }
        }
    }
#ifdef PTG_STAMPS
    if (producer) PTG_STAMP(7);
    else { PTG_STAMP(6); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PTG_STAMP(7); }
#endif
    if (producer) {                                         // (shadow lanes: the same values to env N - 1's slots)
        R.flags = (R.flags & 0x1FFFFu) | (tk << 17);
        StA na; na.i = R.i; na.j = R.j; na.k = k0 + T; na.flags = R.flags;
        P.st_a[e] = na;
        *(int2*)((char*)&P.st_b[e] + 8) = make_int2(R.act_d, R.nctr);
    } else {
        *(double*)&P.st_b[e] = R.cum;
    }
}

// METH_STATUS of every env's observation row as one contiguous byte array (ptg_step_host's "status" section): the NumPy side of a
// VecEnv needs it as an int64 vector, and gathering column 26 of 65 536 rows of 140 bytes on the host costs more than the step.
// c0 = the column that holds it (row / feature-major), or the first of its six one-hot columns (SB3_FLAT / SPLIT rows).
template <typename OUT>
__global__ void __launch_bounds__(256) k_pack_status(const OUT* __restrict__ obs, int N, int F, int c0, int fm, int onehot, uint8_t* __restrict__ status, int pitch)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    int v;
    if (onehot) {
        v = 0;
#pragma unroll
        for (int j = 1; j < 6; j++) v += obs[(size_t)e * F + c0 + j] != (OUT)0 ? j : 0;
    } else v = (int)(fm ? obs[(size_t)c0 * pitch + e] : obs[(size_t)e * F + c0]);
    status[e] = (uint8_t)v;
}

__global__ void k_extract_keys(const RecFast* __restrict__ recf, unsigned short* __restrict__ rkey, int n)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) rkey[g] = (unsigned short)recf[g].tkey;
}

// ================================================================================== VecNormalize reward normalisation
// stable-baselines3 2.0.0a13 (the reference's pin, requirements.txt:5; un-vendored), vec_env/vec_normalize.py + running_mean_std.py,
// as the reference uses it: VecNormalize(env, norm_obs=False) (src/rl_utils.py:453).  Per vector step:
//   returns = returns * gamma + reward;  ret_rms.update(returns)   [batch mean / population variance over the envs, merged into
//   the running moments];  reward_out = clip(reward / sqrt(ret_rms.var + epsilon), +-clip_reward);  returns[done] = 0.
// Over a [T][N] reward matrix that is: a per-env recurrence with per-step moments over the envs (k_vn_moments, k_vn_merge), a
// T-step scalar scan of the running moments (k_vn_scan) and an elementwise pass (k_vn_norm).  Moments travel as
// (count, mean, M2) and are merged with Chan's formula -- across waves here, across GPUs in rl_ptg_amd/dist.py.
__device__ __forceinline__ void chan_merge(double& ca, double& ma, double& Ma, double cb, double mb, double Mb)
{
    if (cb == 0.0) return;
    if (ca == 0.0) { ca = cb; ma = mb; Ma = Mb; return; }
    const double tot = ca + cb, delta = mb - ma;
    ma = ma + delta * cb / tot;
    Ma = Ma + Mb + delta * delta * ca * cb / tot;
    ca = tot;
}

// Per-env recurrence + per-step moments of every wave (one wave per workgroup).  Cross-lane reductions per step would
// dominate (a float64 butterfly is 12 dependent ds_bpermute or DPP stages: measured 0.33-0.46 us per step at one wave per SIMD),
// so the work is transposed instead: for 64 steps at a time lane e runs the recurrence of ITS env and parks the 64 returns in
// an LDS tile [step][env]; then lane t sums row t -- the moments of step t over the wave's envs -- in a private loop (two
// passes: mean, then squared deviations) and writes that step's partial.  No cross-lane instruction at all.
template <typename OUT>
__global__ void __launch_bounds__(64)
k_vn_moments(const OUT* __restrict__ rew, const uint8_t* __restrict__ done, int N, int T, double gamma, double* __restrict__ returns,
             double* __restrict__ partials, int nW)
{
    constexpr int TS = 64, PITCH = 65;                      // 65: row t starts 2 banks after row t-1
    __shared__ double tile[TS * PITCH];
    const int lane = threadIdx.x, w = blockIdx.x;           // w = wave index = workgroup index
    const int e_raw = w * 64 + lane;
    const bool live = e_raw < N;
    const int e = live ? e_raw : N - 1;
    const int n_live = min(64, N - w * 64);
    double ret = live ? returns[e] : 0.0;
    for (int t0 = 0; t0 < T; t0 += TS) {
        const int nt = min(TS, T - t0);
        for (int tb = 0; tb < nt; tb += 8) {                // recurrence, loads batched eight steps at a time
            OUT r[8]; uint8_t d[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const size_t g = (size_t)(t0 + min(tb + j, nt - 1)) * N + e;
                r[j] = rew[g]; d[j] = done[g];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (tb + j < nt) {
                    ret = ret * gamma + (double)r[j];       // _update_reward
                    tile[(tb + j) * PITCH + lane] = ret;
                    ret = d[j] ? 0.0 : ret;                 // self.returns[dones] = 0
                }
            }
        }
        __syncthreads();
        if (lane < nt) {                                    // lane t: moments of step t0 + t over this wave's envs
            const double* row = tile + lane * PITCH;
            double s = 0.0;
            for (int q = 0; q < n_live; q++) s += row[q];
            const double mean = s / (double)n_live;         // np.mean
            double m2 = 0.0;
            for (int q = 0; q < n_live; q++) { const double dv = row[q] - mean; m2 += dv * dv; }
            double* p = partials + ((size_t)(t0 + lane) * nW + w) * 3;
            p[0] = (double)n_live; p[1] = mean; p[2] = m2;
        }
        __syncthreads();
    }
    if (live) returns[e] = ret;
}

__global__ void __launch_bounds__(64)
k_vn_merge(const double* __restrict__ partials, int nW, double* __restrict__ moments)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    double c = 0.0, m = 0.0, M = 0.0;
    for (int w = lane; w < nW; w += 64) {
        const double* p = partials + ((size_t)t * nW + w) * 3;
        chan_merge(c, m, M, p[0], p[1], p[2]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double cb = __shfl_xor(c, off, 64), mb = __shfl_xor(m, off, 64), Mb = __shfl_xor(M, off, 64);
        // both partners must end with the same value: merge in a fixed (lower lane first) order
        double ca = c, ma = m, Ma = M;
        if (lane & off) { double tc = cb, tm = mb, tM = Mb; chan_merge(tc, tm, tM, ca, ma, Ma); c = tc; m = tm; M = tM; }
        else { chan_merge(ca, ma, Ma, cb, mb, Mb); c = ca; m = ma; M = Ma; }
    }
    if (lane == 0) { moments[t * 3 + 0] = c; moments[t * 3 + 1] = m; moments[t * 3 + 2] = M; }
}

// RunningMeanStd.update_from_moments over the T steps of a launch; den[t] = sqrt(var + epsilon) AFTER the update of step t
// (step_wait updates before it normalises).  update_from_moments IS Chan's merge of (count, mean, var * count), which is
// associative: one wave runs it as a prefix scan (per-lane chunks of the steps, a 6-stage scan of the lane totals, then the
// chunks again) -- a chain of ~2 T / 64 + 6 merges instead of T, each a float64 division.  training == 0: frozen statistics.
__global__ void __launch_bounds__(64)
k_vn_scan(const double* __restrict__ moments, int T, int training, double epsilon, double* __restrict__ stats, double* __restrict__ den)
{
    const int lane = threadIdx.x;
    const double mean0 = stats[0], var0 = stats[1], count0 = stats[2];
    const int chunk = (T + 63) / 64, t_lo = min(T, lane * chunk), t_hi = min(T, t_lo + chunk);
    if (!training) {
        for (int t = t_lo; t < t_hi; t++) den[t] = sqrt(var0 + epsilon);
        return;
    }
    double c = 0.0, m = 0.0, M = 0.0;                       // this lane's chunk, merged
    for (int t = t_lo; t < t_hi; t++) chan_merge(c, m, M, moments[t * 3 + 0], moments[t * 3 + 1], moments[t * 3 + 2]);
    double ic = c, im = m, iM = M;                          // inclusive scan over the lanes (earlier steps first)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double pc = __shfl_up(ic, off, 64), pm = __shfl_up(im, off, 64), pM = __shfl_up(iM, off, 64);
        if (lane >= off) { chan_merge(pc, pm, pM, ic, im, iM); ic = pc; im = pm; iM = pM; }
    }
    double rc = __shfl_up(ic, 1, 64), rm = __shfl_up(im, 1, 64), rM = __shfl_up(iM, 1, 64);      // exclusive prefix
    if (lane == 0) { rc = 0.0; rm = 0.0; rM = 0.0; }
    double qc = count0, qm = mean0, qM = var0 * count0;     // running moments before this lane's first step
    chan_merge(qc, qm, qM, rc, rm, rM);
    for (int t = t_lo; t < t_hi; t++) {
        chan_merge(qc, qm, qM, moments[t * 3 + 0], moments[t * 3 + 1], moments[t * 3 + 2]);
        den[t] = sqrt(qM / qc + epsilon);
    }
    if (lane == 63) { stats[0] = qm; stats[1] = qM / qc; stats[2] = qc; }      // lane 63's prefix + chunk = all T steps
}

template <typename OUT>
__global__ void __launch_bounds__(256)
k_vn_norm(const OUT* __restrict__ rew, OUT* __restrict__ out, const double* __restrict__ den, int N, size_t total, double clip)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const double v = (double)rew[g] / den[g / (size_t)N];
    out[g] = (OUT)fmin(fmax(v, -clip), clip);               // np.clip
}

}  // namespace

// ================================================================================================= host side
struct ptg_env {
    ptg_config cfg;
    int n = 0, device = 0, n_sets = 0, F = 0, S = 0;
    bool reset_done = false;
    DevParams P;
    std::vector<void*> allocs;
    std::vector<double> Tvals;
    std::vector<int> tab_rows, rec_base;
    size_t rec_total = 0;
    bool fast = false, fm = false, flat = false, split = false;
    double* d_tape = nullptr;
    unsigned short* d_lut16 = nullptr;
    unsigned short* d_rkey = nullptr;   // temperature keys of all window records (k_rollout_pc producers)
    float* d_pool32 = nullptr; double* d_pool64 = nullptr;
    unsigned off_featB = 0, off_gasn = 0, off_euan = 0, off_gas = 0, off_eua = 0, off_sc = 0;
    unsigned o64_featA = 0, o64_featB = 0, o64_gasn = 0, o64_euan = 0, o64_sc = 0;
    std::vector<float> pool32_host;
    std::vector<double> pool64_host;
    int* d_ladder = nullptr;
    int sync_k = -1;             // common step count k of all envs when the batch is known to be synchronised, else -1
    int step_skip_term = 0;      // argument of the next k_step_hot launch: 1 while ptg_step is being captured (see ptg_step)
    int replay_proof = 0;        // ptg_set_replay_proof: a captured ptg_step is enqueued as hot kernel + predicated generic kernel
    // VecNormalize reward normalisation (ptg_vn_*): per-env discounted returns, running (mean, var, count), scratch
    double *vn_returns = nullptr, *vn_stats = nullptr, *vn_partials = nullptr, *vn_den = nullptr, *vn_moments = nullptr;
    size_t vn_partials_cap = 0; int vn_T_cap = 0;
    double vn_gamma = 0.99, vn_eps = 1e-8, vn_clip = 10.0;
    bool fin_maybe = false;      // a generic step ran since the last ptg_finished_episodes: only those can finish episodes
    void* fin_stage = nullptr; size_t fin_stage_bytes = 0;      // pinned staging of ptg_finished_episodes
    unsigned long long fin_dropped = 0;      // finished episodes never handed out: ring overflow, or a query whose cap was too small
    int tape_len = 0;
    double *d_pot_raw = nullptr, *d_pf_raw = nullptr;
    int* d_eps_ind = nullptr;
    // experiment knobs, read from the environment ONCE in ptg_create (PTG_NO_HOT_KERNELS, PTG_NO_LDS_LUT, PTG_NO_REFRESH, PTG_REFRESH_ALWAYS, PTG_PC_CHUNK, PTG_BLOCK)
    bool knob_no_hot = false, knob_no_lds_lut = false, knob_no_refresh = false, knob_refresh_always = false;
    bool knob_capture_fork = false;      // PTG_REFRESH_CAPTURE_FORK: capture k_refresh as a forked branch of the graph (measured slower: the branches replay serially)
    int knob_refresh_mode = 0;   // PTG_REFRESH_MODE: 0 head pass in the rollout's prologue + forked rolling passes (default), 1 "legacy" (round 2:
                                 // k_refresh enqueued ahead of the rollout, unordered), 2 "head" (no rolling passes)
    int front_horizon = 0;       // steps after a synchronised reset during which the table refresher keeps rolling (k_refresh)
    hipStream_t ref_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;      // the refresher's stream is forked from / joined to the caller's (also under capture)
    double unrefreshed_bytes = 1e18;      // written by this handle's kernels since the tables were last re-read (first launch: refresh)
    int n_cu = 256;
    std::vector<const void*> attr_done;   // kernels whose dynamic-LDS limit has been raised on this handle's device
    int* err_host = nullptr;     // DevParams::err as the host sees it
    // ptg_step_host: device staging for batches too large for zero-copy, and the classification of the caller's buffers
    void *hs_act = nullptr, *hs_out = nullptr, *hs_final = nullptr; double* hs_info = nullptr;
    struct HostStep {            // a ptg_step_host call between its phases (begin .. tail .. end)
        bool active = false, zc = false, tail_done = false;
        void* out_host = nullptr; void* final_host = nullptr; double* info_host = nullptr;
        hipStream_t st = nullptr;
        int n_done = 0;
    } hs;
    int status_col_flat = 5;           // SB3_FLAT rows: first of the six one-hot METH_STATUS columns (ptg_create, from the column map)
    hipEvent_t ev_tail = nullptr;      // recorded behind the copy of [rewards | done flags | status] (+ info rows): the part the caller needs first
    struct HostPtr { const void* host = nullptr; void* dev = nullptr; };      // dev == nullptr: not device-mapped (pageable, or not host memory)
    HostPtr hs_map[8]; int hs_next = 0;      // classification of the caller's buffers by address: a small round-robin cache (a VecEnv rotates 4 blocks)
    int knob_chunk = 65536, knob_block = 0;
    // per-launch timing (ptg_profile): kernel-attached start / stop events of the launches since profiling was switched on
    double* rollout_info = nullptr;   // set by ptg_rollout_info around its hot launches: the [T][N][24] info matrix (float64 kernels only)
    bool profiling = false;
    struct ProfRec { hipEvent_t e0 = nullptr, e1 = nullptr, h0 = nullptr, h1 = nullptr; };      // the launch's events; its helper's (k_refresh), if any
    std::vector<ProfRec> prof_used;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;
    std::string err;
};

namespace {

thread_local std::string g_create_err;

int set_err(ptg_env* h, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(h, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return set_err(h, PTG_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int dev_alloc(ptg_env* h, T** p, size_t count)
{
    void* q = nullptr;
    HIP_TRY(h, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    h->allocs.push_back(q);
    *p = (T*)q;
    return 0;
}

template <typename T>
int dev_upload(ptg_env* h, T** p, const T* src, size_t count)
{
    int rc = dev_alloc(h, p, count);
    if (rc) return rc;
    if (count) HIP_TRY(h, hipMemcpy(*p, src, count * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// math.sin / math.cos are two separate libm calls in the reference (:449-450)
__attribute__((noinline)) double host_sin(double x) { return std::sin(x); }
__attribute__((noinline)) double host_cos(double x) { return std::cos(x); }

inline int grid_for(long long n, int block) { return (int)((n + block - 1) / block); }

int launch_check(ptg_env* h, const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(h, PTG_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

// ptg_profile: a (start, stop) event pair attached to the next kernel launch (hipExtLaunchKernelGGL stamps them at the kernel's
// own begin / end, like the profiler's dispatch timestamps: no host launch latency inside the interval); nulls when off
bool prof_events(ptg_env* h, hipEvent_t& e0, hipEvent_t& e1)
{
    std::pair<hipEvent_t, hipEvent_t> p{nullptr, nullptr};
    if (!h->prof_free.empty()) { p = h->prof_free.back(); h->prof_free.pop_back(); }
    else if (hipEventCreate(&p.first) != hipSuccess || hipEventCreate(&p.second) != hipSuccess) { (void)hipGetLastError(); return false; }
    e0 = p.first; e1 = p.second;
    return true;
}

void prof_pair(ptg_env* h, hipEvent_t& e0, hipEvent_t& e1)
{
    e0 = e1 = nullptr;
    if (!h->profiling || !prof_events(h, e0, e1)) return;
    ptg_env::ProfRec r;
    r.e0 = e0; r.e1 = e1;
    h->prof_used.push_back(r);
}

// the event pair of a helper kernel (k_refresh) that runs beside the NEXT profiled launch: remembered until that launch's record exists
thread_local hipEvent_t g_helper0 = nullptr, g_helper1 = nullptr;

// raise the dynamic-LDS limit of a kernel once per handle (= once per device the handle lives on)
void lds_attr_once(ptg_env* h, const void* kfn, int bytes)
{
    for (const void* q : h->attr_done) if (q == kfn) return;
    (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    h->attr_done.push_back(kfn);
}

int build_tables(ptg_env* h, const ptg_tables* tb)
{
    const int S = h->S;
    // distinct catalyst temperatures (every table's T column + the initial temperature), sorted
    std::vector<double>& Tv = h->Tvals;
    size_t total_rows = 0;
    for (int t = 0; t < NT; t++) {
        if (tb->rows[t] < 1 || !tb->data_host[t]) return set_err(h, PTG_E_INVALID, "table %d is empty", t);
        total_rows += tb->rows[t];
        for (int r = 0; r < tb->rows[t]; r++) {
            double T = tb->data_host[t][(size_t)r * NC + 1];
            if (!(T == T)) return set_err(h, PTG_E_INVALID, "NaN temperature in table %d row %d", t, r);
            Tv.push_back(T);
        }
    }
    if (tb->rows[PTG_T_OP1_START_P] < S) return set_err(h, PTG_E_INVALID, "op1_start_p is shorter than one step");
    Tv.push_back(h->cfg.t_cat_initial);
    std::sort(Tv.begin(), Tv.end());
    Tv.erase(std::unique(Tv.begin(), Tv.end()), Tv.end());
    const int nT = (int)Tv.size();
    if (nT >= (1 << 15)) return set_err(h, PTG_E_INVALID, "more than 32767 distinct catalyst temperatures (%d)", nT);
    auto key_of = [&](double T) { return (int)(std::lower_bound(Tv.begin(), Tv.end(), T) - Tv.begin()); };

    // raw tables + per-row keys on the device (only needed while building)
    std::vector<double> raw(total_rows * NC);
    std::vector<int> rowkey(total_rows), raw_base(NT);
    h->tab_rows.resize(NT); h->rec_base.resize(NT);
    size_t off = 0, rec_total = 0;
    for (int t = 0; t < NT; t++) {
        raw_base[t] = (int)off;
        h->tab_rows[t] = tb->rows[t];
        h->rec_base[t] = (int)rec_total;
        memcpy(&raw[off * NC], tb->data_host[t], sizeof(double) * NC * tb->rows[t]);
        for (int r = 0; r < tb->rows[t]; r++) rowkey[off + r] = key_of(tb->data_host[t][(size_t)r * NC + 1]);
        off += tb->rows[t];
        rec_total += (size_t)tb->rows[t] + 1;
    }
    double* d_raw = nullptr; int* d_key = nullptr; double* d_T = nullptr;
    struct Scratch {                                        // build-time device scratch, released on every return path
        double*& a; int*& b;
        ~Scratch() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); }
    } scratch{d_raw, d_key};
    HIP_TRY(h, hipMalloc((void**)&d_raw, raw.size() * sizeof(double)));
    HIP_TRY(h, hipMalloc((void**)&d_key, rowkey.size() * sizeof(int)));
    HIP_TRY(h, hipMemcpy(d_raw, raw.data(), raw.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(d_key, rowkey.data(), rowkey.size() * sizeof(int), hipMemcpyHostToDevice));
    int rc;
    if ((rc = dev_upload(h, &d_T, Tv.data(), Tv.size()))) return rc;
    Rec* d_rec; int* d_arg; int2* d_meta;
    h->rec_total = rec_total;
    if ((rc = dev_alloc(h, &d_rec, rec_total))) return rc;
    if ((rc = dev_alloc(h, &d_arg, (size_t)N_DEST * nT))) return rc;
    std::vector<int2> meta(NT);
    for (int t = 0; t < NT; t++) meta[t] = make_int2(h->tab_rows[t], h->rec_base[t]);
    if ((rc = dev_upload(h, &d_meta, meta.data(), meta.size()))) return rc;
    const double* d_op1 = d_raw + (size_t)raw_base[PTG_T_OP1_START_P] * NC;
    const int* d_op1key = d_key + raw_base[PTG_T_OP1_START_P];
    for (int t = 0; t < NT; t++) {
        const int n = h->tab_rows[t];
        hipLaunchKernelGGL(k_build_records, dim3(grid_for(n + 1, 128)), dim3(128), 0, 0, d_raw + (size_t)raw_base[t] * NC,
                           d_key + raw_base[t], n, d_op1, d_op1key, t <= PTG_T_STARTUP_HOT ? 1 : 0, S, d_rec + h->rec_base[t]);
        if ((rc = launch_check(h, "k_build_records"))) return rc;
    }
    for (int d = 0; d < N_DEST; d++) {
        const int t = DEST_TID[d];
        hipLaunchKernelGGL(k_build_argmin, dim3(grid_for(nT, 4)), dim3(256), 0, 0, d_raw + (size_t)raw_base[t] * NC,
                           h->tab_rows[t], d_T, nT, d_arg + (size_t)d * nT);
        if ((rc = launch_check(h, "k_build_argmin"))) return rc;
    }
    HIP_TRY(h, hipDeviceSynchronize());
    DevParams& P = h->P;
    P.rec = d_rec; P.argidx = d_arg; P.tabmeta = d_meta; P.Tvals = d_T; P.nT = nT;
    P.key_init = key_of(h->cfg.t_cat_initial);
    P.key_cold_max = (int)(std::upper_bound(Tv.begin(), Tv.end(), h->cfg.t_cat_startup_cold) - Tv.begin()) - 1;
    P.key_hot_min = key_of(h->cfg.t_cat_startup_hot);
    P.key_standby_max = (int)(std::upper_bound(Tv.begin(), Tv.end(), h->cfg.t_cat_standby) - Tv.begin()) - 1;
    // reset state (:117-122): i = argmin |cooldown.T - T_initial|, flows of that single row
    HIP_TRY(h, hipMemcpy(&P.i_reset, d_arg + P.key_init, sizeof(int), hipMemcpyDeviceToHost));
    for (int c = 0; c < 5; c++) P.reset_flow[c] = tb->data_host[PTG_T_COOLDOWN][(size_t)P.i_reset * NC + 2 + c];
    P.T_init = h->cfg.t_cat_initial;
    {   // uint16 copy of the lookup for LDS residency in k_rollout (every table has < 65536 rows; else stay with int32 in L2)
        std::vector<int> lut((size_t)N_DEST * nT);
        HIP_TRY(h, hipMemcpy(lut.data(), d_arg, sizeof(int) * lut.size(), hipMemcpyDeviceToHost));
        bool fits = true;
        for (int v : lut) fits = fits && v >= 0 && v < 65536;
        if (fits) {
            std::vector<unsigned short> l16((lut.size() + 15) / 8 * 8, 0);      // whole 16-byte pieces (k_rollout_pc stages it as uint4)
            for (size_t q = 0; q < lut.size(); q++) l16[q] = (unsigned short)lut[q];
            if ((rc = dev_upload(h, &h->d_lut16, l16.data(), l16.size()))) return rc;
        }
    }
    return 0;
}

int build_market(ptg_env* h, const ptg_market* sets, int n_sets)
{
    const ptg_config& c = h->cfg;
    DevParams& P = h->P;
    const int nh = sets[0].n_hours, nd = sets[0].n_days;
    if (nh < c.price_ahead || nd < 2) return set_err(h, PTG_E_INVALID, "market series too short");
    for (int s = 0; s < n_sets; s++) {
        if (sets[s].n_hours != nh || sets[s].n_days != nd) return set_err(h, PTG_E_INVALID, "market sets differ in length");
        if (!sets[s].el_host || !sets[s].pot_rew_host || !sets[s].part_full_host || !sets[s].gas_host || !sets[s].eua_host)
            return set_err(h, PTG_E_INVALID, "market set %d has a null series", s);
    }
    std::vector<double> el((size_t)n_sets * nh), fa((size_t)n_sets * nh), fb((size_t)n_sets * nh), pot((size_t)n_sets * nh),
        pf((size_t)n_sets * nh), gas((size_t)n_sets * nd), eua((size_t)n_sets * nd), gasn((size_t)n_sets * nd), euan((size_t)n_sets * nd);
    std::vector<double2> setc(n_sets);
    for (int s = 0; s < n_sets; s++) {
        const ptg_market& m = sets[s];
        for (int t = 0; t < nh; t++) {
            const size_t g = (size_t)s * nh + t;
            el[g] = m.el_host[t]; pot[g] = m.pot_rew_host[t]; pf[g] = m.part_full_host[t];
            if (c.raw_modified) {      // :208 pot_rew_n ; Part_Full stays raw (:239)
                fa[g] = (m.pot_rew_host[t] - m.rew_l_b) / (m.rew_u_b - m.rew_l_b);
                fb[g] = m.part_full_host[t];
            } else {                   // :209 el_n
                fa[g] = (m.el_host[t] - c.el_l_b) / (c.el_u_b - c.el_l_b);
                fb[g] = 0.0;
            }
        }
        for (int d = 0; d < nd; d++) {
            const size_t g = (size_t)s * nd + d;
            gas[g] = m.gas_host[d]; eua[g] = m.eua_host[d];
            gasn[g] = (m.gas_host[d] - c.gas_l_b) / (c.gas_u_b - c.gas_l_b);     // :210
            euan[g] = (m.eua_host[d] - c.eua_l_b) / (c.eua_u_b - c.eua_l_b);     // :211
        }
        setc[s] = make_double2(m.scenario == 3 ? 1.0 : 0.0, m.r_0 * c.state_change_penalty);   // :76-77, :332
    }
    int rc;
    double *d_el, *d_fa, *d_fb, *d_gas, *d_eua, *d_gasn, *d_euan; double2* d_setc;
    if ((rc = dev_upload(h, &d_el, el.data(), el.size()))) return rc;
    if ((rc = dev_upload(h, &d_fa, fa.data(), fa.size()))) return rc;
    if ((rc = dev_upload(h, &d_fb, fb.data(), fb.size()))) return rc;
    if ((rc = dev_upload(h, &d_gas, gas.data(), gas.size()))) return rc;
    if ((rc = dev_upload(h, &d_eua, eua.data(), eua.size()))) return rc;
    if ((rc = dev_upload(h, &d_gasn, gasn.data(), gasn.size()))) return rc;
    if ((rc = dev_upload(h, &d_euan, euan.data(), euan.size()))) return rc;
    if ((rc = dev_upload(h, &h->d_pot_raw, pot.data(), pot.size()))) return rc;
    if ((rc = dev_upload(h, &h->d_pf_raw, pf.data(), pf.size()))) return rc;
    if ((rc = dev_upload(h, &d_setc, setc.data(), setc.size()))) return rc;
    {   // float32 copies of the pre-normalised feature series for the fast path (one rounding of the float64 value),
        // pooled so that the fast kernels address them as one base pointer + element offsets
        std::vector<float> pool;
        pool.insert(pool.end(), fa.begin(), fa.end());
        h->off_featB = (unsigned)pool.size(); pool.insert(pool.end(), fb.begin(), fb.end());
        h->off_gasn = (unsigned)pool.size(); pool.insert(pool.end(), gasn.begin(), gasn.end());
        h->off_euan = (unsigned)pool.size(); pool.insert(pool.end(), euan.begin(), euan.end());
        h->pool32_host = pool;                            // uploaded by ptg_create once the sin/cos table is appended
        // pool64 = [el | gas | eua | featA | featB | gas_n | eua_n | sin,cos pairs]: prices for every hot kernel, the float64 feature
        // series for the float64-output ones (uploaded by ptg_create once the sin/cos table is appended)
        std::vector<double>& p64 = h->pool64_host;
        p64.insert(p64.end(), el.begin(), el.end());
        h->off_gas = (unsigned)p64.size(); p64.insert(p64.end(), gas.begin(), gas.end());
        h->off_eua = (unsigned)p64.size(); p64.insert(p64.end(), eua.begin(), eua.end());
        h->o64_featA = (unsigned)p64.size(); p64.insert(p64.end(), fa.begin(), fa.end());
        h->o64_featB = (unsigned)p64.size(); p64.insert(p64.end(), fb.begin(), fb.end());
        h->o64_gasn = (unsigned)p64.size(); p64.insert(p64.end(), gasn.begin(), gasn.end());
        h->o64_euan = (unsigned)p64.size(); p64.insert(p64.end(), euan.begin(), euan.end());
    }
    P.pot_raw = h->d_pot_raw; P.pf_raw = h->d_pf_raw;
    P.el = d_el; P.featA = d_fa; P.featB = d_fb; P.gas = d_gas; P.eua = d_eua; P.gas_n = d_gasn; P.eua_n = d_euan; P.setc = d_setc;
    P.n_hours = nh; P.n_days = nd; P.hstride = nh; P.dstride = nd;
    return 0;
}

HotParams make_hot_params(const ptg_env* h)
{
    const DevParams& P = h->P;
    HotParams F;
    memset(&F, 0, sizeof F);
    F.fm_pitch = P.fm_pitch;
    F.N = P.N; F.S = P.S; F.sim_step = P.sim_step; F.eps_sim_steps = P.eps_sim_steps; F.F = P.F; F.nT = P.nT; F.tape_len = P.tape_len;
    F.flat = h->flat ? 1 : 0; F.track_changes = P.track_changes; F.key_cold_max = P.key_cold_max; F.key_hot_min = P.key_hot_min; F.key_standby_max = P.key_standby_max;
    F.n_hours = P.n_hours; F.n_days = P.n_days; F.hstride = P.hstride; F.dstride = P.dstride;
    F.off_featB = h->off_featB; F.off_gasn = h->off_gasn; F.off_euan = h->off_euan; F.off_sc = h->off_sc; F.off_gas = h->off_gas; F.off_eua = h->off_eua;
    F.noise_seed = P.noise_seed; F.env_offset = P.env_offset; F.k_chp = P.k_chp; F.k_eua = P.k_eua;
    // no noise source configured (neither tape nor RNG): the draws are 0 -- the RNG kernels with sigma 0 (0 x finite = 0 exactly;
    // the draw counter advances as in every mode), one third fewer kernel instantiations to build
    F.noise_sigma = (P.tape_len > 0 || P.noise_inline) ? P.noise_sigma : 0.0;
    F.o64_featA = h->o64_featA; F.o64_featB = h->o64_featB; F.o64_gasn = h->o64_gasn; F.o64_euan = h->o64_euan; F.o64_sc = h->o64_sc;
    F.rec = P.rec; F.pot_raw = P.pot_raw; F.pf_raw = P.pf_raw;
    RewC& c = F.rc;
    c.c_mol = P.c_mol; c.Hu_ch4 = P.Hu_ch4; c.Hu_h2 = P.Hu_h2; c.dt_cp_evap = P.dt_cp_evap; c.heat_price = P.heat_price; c.o2_price = P.o2_price;
    c.eeg = P.eeg; c.eta_chp = P.eta_chp; c.one_m_eta_chp = P.one_m_eta_chp; c.M_co2 = P.M_co2; c.M_h2o = P.M_h2o; c.rho = P.rho;
    c.water_price = P.water_price; c.min_load = P.min_load; c.max_h2 = P.max_h2; c.c_m2 = P.c_m2; c.c_m3 = P.c_m3; c.sim_step_d = P.sim_step_d;
    c.T_lo = P.T_lo; c.T_rng = P.T_rng; c.h2_lo = P.h2_lo; c.h2_rng = P.h2_rng; c.ch4_lo = P.ch4_lo; c.ch4_rng = P.ch4_rng;
    c.h2r_lo = P.h2r_lo; c.h2r_rng = P.h2r_rng; c.h2o_lo = P.h2o_lo; c.h2o_rng = P.h2o_rng; c.heat_lo = P.heat_lo; c.heat_rng = P.heat_rng;
    F.recf = P.recf; F.tape = P.tape; F.pool32 = h->d_pool32; F.pool64 = h->d_pool64; F.setc = P.setc; F.argidx = P.argidx;
    F.tabmeta = P.tabmeta; F.ladder = h->d_ladder; F.st_a = P.st_a; F.st_b = P.st_b; F.st_c = P.st_c; F.err = P.err;
    F.term_flag = P.term_flag;
    return F;
}

// elements of one step's observation block: F rows of N (row-major) or F planes of fm_pitch (feature-major)
size_t obs_step_elems(const ptg_env* h) { return (size_t)h->F * (size_t)(h->fm ? h->P.fm_pitch : h->n); }

// the hot kernels apply to a 13-hour look-ahead, a synchronised batch, and steps on which no env terminates
bool hot_eligible(const ptg_env* h)
{
    const unsigned long long osz = h->cfg.out_dtype == PTG_OUT_F64 ? 8 : 4;
    return h->cfg.price_ahead == 13 && h->sync_k >= 0 && !h->knob_no_hot && (unsigned long long)obs_step_elems(h) * osz < 0xFFFFFFFFull;
}

int noise_mode(const ptg_env* h) { return h->P.tape_len > 0 ? NOISE_TAPE : (h->P.noise_inline ? NOISE_RNG : NOISE_NONE); }

}  // namespace

// The launchers of the hot kernels have external linkage: the build spreads their instantiations (6 layout x dtype combinations x
// 2 feature sets x 2 noise sources, each a family of kernels) over several translation units compiled in parallel, see PTG_PART below.
namespace ptg_hot __attribute__((visibility("hidden"))) {

template <int LAY, bool MOD, int NOISE, typename OUT>
void launch_step_hot(ptg_env* h, hipStream_t st, const void* actions, int kind, OUT* obs, OUT* rew, uint8_t* done)
{
    const HotParams hp = make_hot_params(h);
    const dim3 grid(grid_for(h->n, 256)), block(256);
    if (h->profiling) {
        hipEvent_t e0, e1;
        prof_pair(h, e0, e1);
        hipExtLaunchKernelGGL((k_step_hot<LAY, MOD, NOISE, OUT>), grid, block, 0, st, e0, e1, 0, hp, actions, kind, obs, rew, done, h->step_skip_term);
    } else
        hipLaunchKernelGGL((k_step_hot<LAY, MOD, NOISE, OUT>), grid, block, 0, st, hp, actions, kind, obs, rew, done, h->step_skip_term);
}

}  // namespace ptg_hot

namespace {

// Launch geometry of the fused hot rollout.  One launch covers <= 65 536 envs (one 512-thread workgroup per CU) and as many
// steps as its LDS action stage holds; larger batches / longer rollouts run as consecutive launches over env slices and
// step segments.
struct PcPlan { int chunk, block, t_cap; bool lds_lut; size_t fixed, lut_bytes, lds_max; };

PcPlan pc_plan(const ptg_env* h)
{
    PcPlan pl;
    pl.chunk = std::max(256, h->knob_chunk / 256 * 256);
    {   // a batch wider than one launch is cut into EQUAL slices (100 000 envs: 2 x ~50 000, not 65 536 + 34 464): every launch then
        // runs at the same bandwidth-bound pace instead of a full one followed by a half-empty, latency-bound one
        const int slices = (h->n + pl.chunk - 1) / pl.chunk;
        pl.chunk = std::max(256, ((h->n + slices - 1) / slices + 255) / 256 * 256);
    }
    pl.fixed = 16 * ((sizeof(HotLds) + 15) / 16) + 2 * sizeof(PcSlot);     // + the row-major tiles, below
    pl.lut_bytes = 1024 * ((((size_t)N_DEST * h->Tvals.size() * 2 + 15) / 16 + 63) / 64);      // whole 1 KiB rows (k_rollout_pc's LDS-DMA staging)
    pl.lds_max = 160 * 1024 - 512;
    // half producers, half consumers: the smallest workgroup (64, 128 or 256 envs) whose grid still fits the chip in ONE round of
    // workgroups (one per CU: the LDS stage allows no second one) -- small batches spread over many CUs, large ones do not queue
    pl.block = 128;
    while (pl.block < 512 && (long long)grid_for(std::min(pl.chunk, h->n), pl.block / 2) > 256) pl.block *= 2;
    if (h->knob_block) pl.block = h->knob_block;
    const int np = pl.block / 2;
    if (!h->fm) pl.fixed += (size_t)np * (h->F == 40 ? 41 : h->F == 16 ? 17 : h->F) * (h->cfg.out_dtype == PTG_OUT_F64 ? 8 : 4);      // one [64][pitch] tile per consumer wave (RowTile::PITCH)
    if (h->cfg.out_dtype == PTG_OUT_F64) pl.fixed += (size_t)(h->P.mod ? 26 : 17) * np * 8;      // the market features' LDS home (k_rollout_pc, MLDS)
    // the _get_index lookup goes to LDS when that still leaves room for >= 64 staged steps
    pl.lds_lut = h->d_lut16 && pl.fixed + pl.lut_bytes + (size_t)64 * np + 64 <= pl.lds_max && !h->knob_no_lds_lut;
    const size_t avail = pl.lds_max - pl.fixed - (pl.lds_lut ? pl.lut_bytes : 0) - 64;
    pl.t_cap = (int)std::min<size_t>(512, avail / np);
    return pl;
}

// The table refresher around a hot rollout launch of m envs x tn steps that starts at step count k0 (see k_refresh).  Returns the
// `refresh_rec` argument of the rollout kernel: the record count when the launch is to re-read the tables in its prologue, else 0.
//   head pass  -- whenever this handle's kernels have written more than ~64 MB since the tables were last re-read (every launch at
//                 65 536 envs; every ~100 steps at 4 096): inside the rollout kernel, so inside its measured duration -- and inside a
//                 captured graph.
//   rolling    -- one pass per ~192 MB of output (the Infinity Cache holds 256 MB), only while a synchronised batch is still walking
//                 the tables as a front (the first front_horizon steps of an episode; PTG_REFRESH_ALWAYS: a policy that keeps the envs
//                 in lock-step) and only in launches long enough to need one: k_refresh on a stream FORKED from `st` (event on `st`
//                 before the rollout launch -> the refresher starts when the rollout does, whatever else `st` was busy with).
//                 NOT while `st` is being captured: the fork / join captures fine (a parallel branch of the graph), but the runtime
//                 replays the two branches one after the other -- 150 steps from reset took 420 us as a graph against 253 us eager
//                 (tests/test_batch_edges.py, round 3) -- so a captured launch keeps the head pass only (+ 8 % on the first ~250 steps
//                 after a reset, profiles/r03_refresh_ab.txt).  PTG_REFRESH_CAPTURE_FORK=1 brings the captured branch back.
struct RefreshPlan { int head_rec = 0; bool forked = false; };

RefreshPlan launch_refresher(ptg_env* h, hipStream_t st, int m, int tn, int k0)
{
    RefreshPlan rp;
    const int osz = h->cfg.out_dtype == PTG_OUT_F64 ? 8 : 4;
    const double step_bytes = (double)m * (h->F * osz + osz + 1), launch_bytes = step_bytes * tn;
    if (h->knob_no_refresh) return rp;
    const bool legacy = h->knob_refresh_mode == 1;
    const double pass_steps = std::max(8.0, 192e6 / step_bytes);
    const bool roll = (h->knob_refresh_always || k0 < h->front_horizon) && h->knob_refresh_mode != 2;
    int passes;                                             // k_refresh's pass count (pass 0 = the head pass, only "legacy" runs it there)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    const bool capturing = cs != hipStreamCaptureStatusNone;
    const bool may_roll = roll && !(capturing && !h->knob_capture_fork);
    if (legacy) passes = roll ? std::max(1, (int)std::ceil(tn / pass_steps)) : 1;
    else {
        const bool head = h->unrefreshed_bytes + launch_bytes >= 64e6;
        if (head) { rp.head_rec = (int)h->rec_total; h->unrefreshed_bytes = 0.0; }
        passes = (may_roll && tn > 4) ? 1 + (int)((tn - 4) / pass_steps) : 1;      // a pass with fewer than 4 steps left to serve is not issued
        h->unrefreshed_bytes += launch_bytes - (passes - 1) * pass_steps * step_bytes;
    }
    if (!h->ref_stream || (!legacy && passes < 2)) return rp;
    if (capturing && (legacy || !h->knob_capture_fork)) return rp;
    if (!legacy) {                                          // fork
        if (!h->ev_fork || hipEventRecord(h->ev_fork, st) != hipSuccess || hipStreamWaitEvent(h->ref_stream, h->ev_fork, 0) != hipSuccess) {
            (void)hipGetLastError();
            return rp;
        }
        rp.forked = capturing;
    }
    const double step_us = std::max(0.9, step_bytes / 6.2e6);                 // the rollout's pace: its output stream at ~6.2 TB/s, >= the producers' chain
    const unsigned period = (unsigned)std::min(4.0e6, pass_steps * step_us * 100.0);      // ticks of the 100 MHz clock; <= 40 ms
    const long long n16 = (long long)h->rec_total * 4 + ((long long)h->rec_total * 2 + 15) / 16;
    const int grid = (int)std::min<long long>(h->n_cu, (n16 + 63) / 64);
    const void* recs = osz == 8 ? (const void*)h->P.rec : (const void*)h->P.recf;
    hipEvent_t h0 = nullptr, h1 = nullptr;
    if (h->profiling && !capturing && prof_events(h, h0, h1)) {
        g_helper0 = h0; g_helper1 = h1;
        hipExtLaunchKernelGGL(k_refresh, dim3(grid), dim3(64), 0, h->ref_stream, h0, h1, 0, recs, (const unsigned short*)h->d_rkey, (int)h->rec_total,
                              legacy ? 0 : 1, passes, period);
    } else
        hipLaunchKernelGGL(k_refresh, dim3(grid), dim3(64), 0, h->ref_stream, recs, (const unsigned short*)h->d_rkey, (int)h->rec_total,
                           legacy ? 0 : 1, passes, period);
    return rp;
}

void join_refresher(ptg_env* h, hipStream_t st, const RefreshPlan& rp)
{
    if (!rp.forked) return;                                 // eager: k_refresh ends by itself, nothing waits for it
    if (hipEventRecord(h->ev_join, h->ref_stream) != hipSuccess || hipStreamWaitEvent(st, h->ev_join, 0) != hipSuccess) (void)hipGetLastError();
}

// attach the helper's events (if one was just launched) to the profile record of the launch it runs beside
void prof_attach_helper(ptg_env* h)
{
    if (g_helper0 && !h->prof_used.empty()) { h->prof_used.back().h0 = g_helper0; h->prof_used.back().h1 = g_helper1; }
    g_helper0 = g_helper1 = nullptr;
}

}  // namespace

namespace ptg_hot __attribute__((visibility("hidden"))) {

template <int LAY, bool MOD, int NOISE, typename OUT>
void launch_rollout_hot(ptg_env* h, hipStream_t st, const void* actions, int kind, int T, OUT* obs, OUT* rew, uint8_t* done)
{
    const HotParams hp = make_hot_params(h);
    const PcPlan pl = pc_plan(h);
    const int chunk = pl.chunk, bs_all = pl.block, np = pl.block / 2, t_cap = pl.t_cap;
    const bool ll = pl.lds_lut;
    const size_t asz = kind == PTG_ACT_I64 ? 8 : 4;
    const size_t fixed = pl.fixed, lut_bytes = pl.lut_bytes, lds_max = pl.lds_max;
    for (int ts = 0; ts < T; ts += t_cap) {
        const int tn = std::min(t_cap, T - ts);
        const char* a_s = (const char*)actions + (size_t)ts * h->n * asz;
        OUT* o_s = obs + (size_t)ts * obs_step_elems(h);
        OUT* r_s = rew + (size_t)ts * h->n;
        uint8_t* d_s = done + (size_t)ts * h->n;
        double* i_s = h->rollout_info ? h->rollout_info + (size_t)ts * h->n * PTG_N_INFO : nullptr;
        const int k0 = h->sync_k + ts;
        const int vec_rows = (h->n % 4 == 0) && ((uintptr_t)a_s % 16 == 0) && (chunk % 4 == 0);     // whole-row vector loads are aligned
        const size_t sh = fixed + 16 * (((size_t)tn * np + 15) / 16) + (ll ? lut_bytes : 0);
        for (int e0 = 0; e0 < h->n; e0 += chunk) {
            const int m = std::min(chunk, h->n - e0);
            const dim3 grid(grid_for(m, np)), block(bs_all);
            const bool full = m % np == 0;
            const RefreshPlan rp = launch_refresher(h, st, m, tn, k0);
            const int rr = rp.head_rec;
#define PTG_PC2(LL, FULL)                                                                                             \
    do {                                                                                                              \
        auto kfn = k_rollout_pc<LAY, MOD, NOISE, LL, FULL, OUT>;                                                      \
        lds_attr_once(h, (const void*)kfn, (int)lds_max);      /* > 64 KiB of dynamic LDS needs the attribute */      \
        if (h->profiling) {                                                                                           \
            hipEvent_t pe0, pe1;                                                                                      \
            prof_pair(h, pe0, pe1);                                                                                   \
            prof_attach_helper(h);                                                                                    \
            hipExtLaunchKernelGGL(kfn, grid, block, (unsigned)sh, st, pe0, pe1, 0, hp, (const void*)a_s, kind, tn, o_s, r_s, d_s, \
                                  (const unsigned short*)h->d_lut16, (const unsigned short*)h->d_rkey, e0, vec_rows, i_s, rr); \
        } else                                                                                                        \
            hipLaunchKernelGGL(kfn, grid, block, sh, st, hp, (const void*)a_s, kind, tn, o_s, r_s, d_s, h->d_lut16, h->d_rkey, e0, vec_rows, i_s, rr); \
    } while (0)
#define PTG_PC(LL) do { if (full) PTG_PC2(LL, true); else PTG_PC2(LL, false); } while (0)
            bool launched = false;
            if constexpr (std::is_same<OUT, double>::value) {
                if (i_s) {                                  // the eval info stream: one variant (lookup in global memory, any batch size)
                    auto kfn = k_rollout_pc<LAY, MOD, NOISE, false, false, double, true>;
                    lds_attr_once(h, (const void*)kfn, (int)lds_max);
                    hipLaunchKernelGGL(kfn, grid, block, sh, st, hp, (const void*)a_s, kind, tn, o_s, r_s, d_s, h->d_lut16, h->d_rkey, e0, vec_rows, i_s, rr);
                    launched = true;
                }
            }
            if (!launched) { if (ll) PTG_PC(true); else PTG_PC(false); }
            if (g_helper0) { h->prof_free.push_back({g_helper0, g_helper1}); g_helper0 = g_helper1 = nullptr; }      // (a helper nobody's record took: back to the pool)
            join_refresher(h, st, rp);
#undef PTG_PC
#undef PTG_PC2
        }
    }
}

}  // namespace ptg_hot

// PTG_PART (the parallel build, rl_ptg_amd/_lib.py): part 0 holds the C ABI, the generic kernels and everything else, and only
// DECLARES the hot launchers' instantiations; parts 1..8 each define those of one (layout, dtype).  Without PTG_PART this file is one
// self-contained translation unit (hipcc -shared ptg_env.hip: the diagnostic builds, anybody's quick build).
#define PTG_HOT_INST1(X, LAY, OUT, MOD, NZ)                                                                                              \
    X template void ptg_hot::launch_step_hot<LAY, MOD, NZ, OUT>(ptg_env*, hipStream_t, const void*, int, OUT*, OUT*, uint8_t*);         \
    X template void ptg_hot::launch_rollout_hot<LAY, MOD, NZ, OUT>(ptg_env*, hipStream_t, const void*, int, int, OUT*, OUT*, uint8_t*);
#define PTG_HOT_INST(X, LAY, OUT)                                                                                                        \
    PTG_HOT_INST1(X, LAY, OUT, true, NOISE_TAPE) PTG_HOT_INST1(X, LAY, OUT, false, NOISE_TAPE)                                           \
    PTG_HOT_INST1(X, LAY, OUT, true, NOISE_RNG) PTG_HOT_INST1(X, LAY, OUT, false, NOISE_RNG)
#define PTG_NOTHING
#if defined(PTG_PART)
#if PTG_PART == 0
PTG_HOT_INST(extern, PTG_OBS_ROW_MAJOR, float) PTG_HOT_INST(extern, PTG_OBS_FEATURE_MAJOR, float) PTG_HOT_INST(extern, PTG_OBS_SB3_FLAT, float)
PTG_HOT_INST(extern, PTG_OBS_SPLIT, float) PTG_HOT_INST(extern, PTG_OBS_ROW_MAJOR, double) PTG_HOT_INST(extern, PTG_OBS_FEATURE_MAJOR, double)
PTG_HOT_INST(extern, PTG_OBS_SB3_FLAT, double) PTG_HOT_INST(extern, PTG_OBS_SPLIT, double)
#elif PTG_PART == 1
PTG_HOT_INST(PTG_NOTHING, PTG_OBS_ROW_MAJOR, float)
#elif PTG_PART == 2
PTG_HOT_INST(PTG_NOTHING, PTG_OBS_FEATURE_MAJOR, float)
#elif PTG_PART == 3
PTG_HOT_INST(PTG_NOTHING, PTG_OBS_SB3_FLAT, float)
#elif PTG_PART == 4
PTG_HOT_INST(PTG_NOTHING, PTG_OBS_SPLIT, float)
#elif PTG_PART == 5
PTG_HOT_INST(PTG_NOTHING, PTG_OBS_ROW_MAJOR, double)
#elif PTG_PART == 6
PTG_HOT_INST(PTG_NOTHING, PTG_OBS_FEATURE_MAJOR, double)
#elif PTG_PART == 7
PTG_HOT_INST(PTG_NOTHING, PTG_OBS_SB3_FLAT, double)
#elif PTG_PART == 8
PTG_HOT_INST(PTG_NOTHING, PTG_OBS_SPLIT, double)
#else
#error "PTG_PART must be 0..8"
#endif
#endif

#if !defined(PTG_PART) || PTG_PART == 0      // ---- from here to the end of the file: part 0 only

namespace {

using ptg_hot::launch_step_hot;
using ptg_hot::launch_rollout_hot;

#define PTG_HOT_DISPATCH3(FN, LAY_, OUT_, ...)                                                           \
    do {                                                                                                \
        if (nm_ == NOISE_TAPE) { if (mod_) FN<LAY_, true, NOISE_TAPE, OUT_>(__VA_ARGS__); else FN<LAY_, false, NOISE_TAPE, OUT_>(__VA_ARGS__); }      \
        else { if (mod_) FN<LAY_, true, NOISE_RNG, OUT_>(__VA_ARGS__); else FN<LAY_, false, NOISE_RNG, OUT_>(__VA_ARGS__); }   /* no noise source = the RNG kernels with sigma 0 (make_hot_params) */ \
    } while (0)
// obs / rew arrive as void*: the element type is the handle's out_dtype
#define PTG_HOT_DISPATCH(FN, H_, ST_, A_, KIND_, CNT_ARGS_, OBS_, REW_, DONE_)                           \
    do {                                                                                                \
        const int nm_ = noise_mode(H_);                                                                 \
        const bool mod_ = (H_)->P.mod != 0;                                                             \
        if ((H_)->cfg.out_dtype == PTG_OUT_F64) {                                                       \
            if ((H_)->fm) PTG_HOT_DISPATCH3(FN, PTG_OBS_FEATURE_MAJOR, double, H_, ST_, A_, KIND_ CNT_ARGS_, (double*)(OBS_), (double*)(REW_), DONE_); \
            else if ((H_)->flat) PTG_HOT_DISPATCH3(FN, PTG_OBS_SB3_FLAT, double, H_, ST_, A_, KIND_ CNT_ARGS_, (double*)(OBS_), (double*)(REW_), DONE_); \
            else if ((H_)->split) PTG_HOT_DISPATCH3(FN, PTG_OBS_SPLIT, double, H_, ST_, A_, KIND_ CNT_ARGS_, (double*)(OBS_), (double*)(REW_), DONE_);   \
            else PTG_HOT_DISPATCH3(FN, PTG_OBS_ROW_MAJOR, double, H_, ST_, A_, KIND_ CNT_ARGS_, (double*)(OBS_), (double*)(REW_), DONE_);               \
        } else if ((H_)->fm) PTG_HOT_DISPATCH3(FN, PTG_OBS_FEATURE_MAJOR, float, H_, ST_, A_, KIND_ CNT_ARGS_, (float*)(OBS_), (float*)(REW_), DONE_);  \
        else if ((H_)->flat) PTG_HOT_DISPATCH3(FN, PTG_OBS_SB3_FLAT, float, H_, ST_, A_, KIND_ CNT_ARGS_, (float*)(OBS_), (float*)(REW_), DONE_);       \
        else if ((H_)->split) PTG_HOT_DISPATCH3(FN, PTG_OBS_SPLIT, float, H_, ST_, A_, KIND_ CNT_ARGS_, (float*)(OBS_), (float*)(REW_), DONE_);         \
        else PTG_HOT_DISPATCH3(FN, PTG_OBS_ROW_MAJOR, float, H_, ST_, A_, KIND_ CNT_ARGS_, (float*)(OBS_), (float*)(REW_), DONE_);                      \
    } while (0)
#define PTG_NOARG
#define PTG_COMMA_ARG(x) , x

hipStream_t as_stream(void* s) { return (hipStream_t)s; }

int check_error_flags(ptg_env* h)          // after the stream has been synchronised
{
    // each word is exchanged with 0 on its own, and only the one being reported: a flag that another stream's kernel raises between
    // the read and the clear of the OTHER word is not lost, and a pending RANGE error is reported by the next call
    int* e = h->err_host;
    if (__atomic_load_n(&e[0], __ATOMIC_RELAXED)) {
        __atomic_exchange_n(&e[0], 0, __ATOMIC_RELAXED);
        const bool also = __atomic_load_n(&e[1], __ATOMIC_RELAXED) != 0;
        return set_err(h, PTG_E_ACTION, "a discrete action outside [-5, 4] was passed (the reference raises IndexError)%s", also ?
                       "; a price-index error (PTG_E_RANGE) is pending as well and will be reported by the next call" : "");
    }
    if (__atomic_load_n(&e[1], __ATOMIC_RELAXED)) {
        __atomic_exchange_n(&e[1], 0, __ATOMIC_RELAXED);
        return set_err(h, PTG_E_RANGE, "a price index left the market series (episode longer than the data)");
    }
    if (__atomic_load_n(&e[2], __ATOMIC_RELAXED)) {
        __atomic_exchange_n(&e[2], 0, __ATOMIC_RELAXED);
        return set_err(h, PTG_E_INVALID, "a hot kernel ran on the terminating step of an episode: a captured launch was replayed past "
                       "ptg_steps_to_episode_end (ptg_set_replay_proof(env, 1) before capturing ptg_step lifts that; a fused ptg_rollout cannot cross "
                       "an episode end), or replays were not reported with ptg_note_replays");
    }
    return 0;
}

// Wait for `st`: poll for up to ~200 us (hipStreamQuery: no interrupt, no wake-up latency -- a blocking synchronise returns 20-30 us
// after a short kernel has ended), then block.  A step or a short rollout is over long before the polling budget runs out.
int wait_stream(ptg_env* h, hipStream_t st)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return set_err(h, PTG_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
        (void)hipGetLastError();                            // hipErrorNotReady is sticky in hipGetLastError()
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
    }
    HIP_TRY(h, hipStreamSynchronize(st));
    return 0;
}

int collect_error(ptg_env* h, hipStream_t st)
{
    const int rc = wait_stream(h, st);
    if (rc) return rc;
    return check_error_flags(h);
}

}  // namespace

extern "C" {

int ptg_abi_version(void) { return PTG_ABI_VERSION; }

const char* ptg_last_error(const ptg_env* env) { return env ? env->err.c_str() : g_create_err.c_str(); }

int ptg_num_envs(const ptg_env* env) { return env ? env->n : PTG_E_INVALID; }
int ptg_obs_dim(const ptg_env* env) { return env ? env->F : PTG_E_INVALID; }

void ptg_destroy(ptg_env* env)
{
    if (!env) return;
    (void)hipSetDevice(env->device);
    (void)hipDeviceSynchronize();                           // nothing of this handle (incl. the refresher's stream) is in flight any more
    if (env->ref_stream) (void)hipStreamDestroy(env->ref_stream);
    if (env->ev_fork) (void)hipEventDestroy(env->ev_fork);
    if (env->ev_join) (void)hipEventDestroy(env->ev_join);
    if (env->ev_tail) (void)hipEventDestroy(env->ev_tail);
    for (void* p : env->allocs) (void)hipFree(p);
    if (env->d_tape) (void)hipFree(env->d_tape);
    if (env->d_eps_ind) (void)hipFree(env->d_eps_ind);
    if (env->vn_partials) (void)hipFree(env->vn_partials);
    if (env->vn_den) (void)hipFree(env->vn_den);
    if (env->vn_moments) (void)hipFree(env->vn_moments);
    if (env->err_host) (void)hipHostFree(env->err_host);
    if (env->fin_stage) (void)hipHostFree(env->fin_stage);
    for (void* q : {env->hs_act, env->hs_out, env->hs_final, (void*)env->hs_info}) if (q) (void)hipFree(q);
    for (auto& r : env->prof_used)
        for (hipEvent_t e : {r.e0, r.e1, r.h0, r.h1}) if (e) (void)hipEventDestroy(e);
    for (auto& p : env->prof_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    delete env;
}

int ptg_create(const ptg_config* cfg, const ptg_tables* tables, const ptg_market* sets, int n_sets, int n_envs,
               int device_id, ptg_env** out)
{
    if (!cfg || !tables || !sets || !out) return set_err(nullptr, PTG_E_INVALID, "ptg_create: null argument");
    if (n_envs <= 0 || n_sets < 1 || n_sets > PTG_MAX_MARKET_SETS) return set_err(nullptr, PTG_E_INVALID, "ptg_create: bad n_envs / n_sets");
    if (cfg->time_step_op <= 0 || cfg->sim_step <= 0 || cfg->price_ahead < 1 || cfg->price_ahead > 64)
        return set_err(nullptr, PTG_E_INVALID, "ptg_create: bad sim_step / time_step_op / price_ahead");
    if (cfg->out_dtype != PTG_OUT_F32 && cfg->out_dtype != PTG_OUT_F64) return set_err(nullptr, PTG_E_INVALID, "ptg_create: bad out_dtype");
    if (cfg->obs_layout < PTG_OBS_ROW_MAJOR || cfg->obs_layout > PTG_OBS_SPLIT) return set_err(nullptr, PTG_E_INVALID, "ptg_create: bad obs_layout");
    if (cfg->eps_sim_steps < 7) return set_err(nullptr, PTG_E_INVALID, "ptg_create: eps_sim_steps must be >= 7");
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0)
        return set_err(nullptr, PTG_E_HIP, "no HIP device available (%s): libptg_env has no CPU path", hipGetErrorString(he));
    if (device_id < 0 || device_id >= ndev) return set_err(nullptr, PTG_E_INVALID, "device %d out of range (%d devices)", device_id, ndev);
    he = hipSetDevice(device_id);
    if (he != hipSuccess) return set_err(nullptr, PTG_E_HIP, "hipSetDevice failed: %s", hipGetErrorString(he));

    ptg_env* h = new ptg_env();
    h->cfg = *cfg; h->n = n_envs; h->device = device_id; h->n_sets = n_sets;
    h->knob_no_hot = getenv("PTG_NO_HOT_KERNELS") != nullptr; h->knob_no_lds_lut = getenv("PTG_NO_LDS_LUT") != nullptr;
    h->knob_no_refresh = getenv("PTG_NO_REFRESH") != nullptr; h->knob_refresh_always = getenv("PTG_REFRESH_ALWAYS") != nullptr;
    h->knob_capture_fork = getenv("PTG_REFRESH_CAPTURE_FORK") != nullptr;
    if (const char* v = getenv("PTG_REFRESH_MODE")) h->knob_refresh_mode = !strcmp(v, "legacy") ? 1 : !strcmp(v, "head") ? 2 : 0;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
        else (void)hipGetLastError();
    }
    if (const char* v = getenv("PTG_PC_CHUNK")) h->knob_chunk = atoi(v);
    if (const char* v = getenv("PTG_BLOCK")) { const int b = atoi(v); if (b == 128 || b == 256 || b == 512) h->knob_block = b; }
    h->S = (int)((double)cfg->sim_step / (double)cfg->time_step_op);     // :66
    h->F = cfg->raw_modified ? 2 * cfg->price_ahead + 9 : cfg->price_ahead + 4 + 9;
    h->fast = (cfg->out_dtype == PTG_OUT_F32);          // float32 outputs without info rows run the strength-reduced kernels
    h->fm = (cfg->obs_layout == PTG_OBS_FEATURE_MAJOR);
    memset(&h->P, 0, sizeof h->P);
    DevParams& P = h->P;
    int rc = 0;
    auto fail = [&](int code) { g_create_err = h->err; ptg_destroy(h); return code; };
    if (h->S < 1) { set_err(h, PTG_E_INVALID, "step_size < 1"); return fail(PTG_E_INVALID); }
    if ((rc = build_tables(h, tables))) return fail(rc);
    {   // mixing time of a synchronised batch, in steps: twice the longest table's walk plus a margin (measured at sim_step 600:
        // the slow phase ends 265-285 steps after a reset; this gives 364)
        int longest = 0;
        for (int t = 0; t < NT; t++) longest = std::max(longest, h->tab_rows[t]);
        h->front_horizon = 2 * ((longest + h->S - 1) / h->S) + 64;
    }
    if ((rc = build_market(h, sets, n_sets))) return fail(rc);

    if (cfg->obs_layout == PTG_OBS_SB3_FLAT) {            // rows in SB3's flattened form: sub-spaces by sorted key, METH_STATUS one-hot
        struct Sub { const char* key; int q0, width; };
        const int PA = cfg->price_ahead;
        std::vector<Sub> subs;
        int o;
        if (cfg->raw_modified) { subs.push_back({"Pot_Reward", 0, PA}); subs.push_back({"Part_Full", PA, PA}); o = 2 * PA; }
        else { subs.push_back({"Elec_Price", 0, PA}); subs.push_back({"Gas_Price", PA, 2}); subs.push_back({"EUA_Price", PA + 2, 2}); o = PA + 4; }
        const char* tail[9] = {"METH_STATUS", "T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow",
                               "Elec_Heating", "Temp_hour_enc_sin", "Temp_hour_enc_cos"};
        for (int q = 0; q < 9; q++) subs.push_back({tail[q], o + q, 1});
        std::sort(subs.begin(), subs.end(), [](const Sub& a, const Sub& b) { return strcmp(a.key, b.key) < 0; });    // Python sorted() of ASCII keys
        std::vector<int> cmap(h->F, 0);
        int c = 0;
        for (const Sub& sb : subs) {
            for (int j = 0; j < sb.width; j++) cmap[sb.q0 + j] = c + j;
            c += (sb.q0 == o) ? 6 : sb.width;
        }
        if (PA == 13)
            for (int q = 0; q < h->F; q++)
                if (cmap[q] != (cfg->raw_modified ? flat_col<true>(q) : flat_col<false>(q))) { set_err(h, PTG_E_INVALID, "internal: flat column table mismatch at %d", q); return fail(PTG_E_INVALID); }
        int* d_cmap;
        if ((rc = dev_upload(h, &d_cmap, cmap.data(), cmap.size()))) return fail(rc);
        P.cmap = d_cmap; P.q_stat = o;
        h->status_col_flat = cmap[o];
        h->F += 5; h->flat = true;
    }
    if (cfg->obs_layout == PTG_OBS_SPLIT) {               // env part of the flat row + series indices (include/ptg_env.h)
        P.split = 1; P.q_stat = cfg->raw_modified ? 2 * cfg->price_ahead : cfg->price_ahead + 4;
        h->F = 16; h->split = true;
    }
    P.fm_pitch = n_envs;
    P.N = n_envs; P.S = h->S; P.sim_step = cfg->sim_step; P.eps_sim_steps = cfg->eps_sim_steps; P.PA = cfg->price_ahead;
    P.F = h->F; P.mod = cfg->raw_modified; P.eps_len_d = cfg->eps_len_d; P.E = 0; P.ep_stride = 0; P.tape_len = 0;
    P.noise_inline = 0; P.noise_seed = 0; P.env_offset = 0; P.noise_sigma = cfg->noise;
    P.track_changes = (cfg->state_change_penalty != 0.0) ? 1 : 0;
    P.t1_start_p_f = cfg->time1_start_p_f; P.t2_start_f_p = cfg->time2_start_f_p; P.t_p_f = cfg->time_p_f; P.t_f_p = cfg->time_f_p;
    P.t1_p_f_p = cfg->time1_p_f_p; P.t2_p_f_p = cfg->time2_p_f_p; P.t3_p_f_p = cfg->time3_p_f_p; P.t34_p_f_p = cfg->time34_p_f_p;
    P.t4_p_f_p = cfg->time4_p_f_p; P.t45_p_f_p = cfg->time45_p_f_p; P.t5_p_f_p = cfg->time5_p_f_p; P.t1_f_p_f = cfg->time1_f_p_f;
    P.t2_f_p_f = cfg->time2_f_p_f; P.t23_f_p_f = cfg->time23_f_p_f; P.t3_f_p_f = cfg->time3_f_p_f; P.t34_f_p_f = cfg->time34_f_p_f;
    P.t4_f_p_f = cfg->time4_f_p_f; P.t45_f_p_f = cfg->time45_f_p_f; P.t5_f_p_f = cfg->time5_f_p_f;
    P.i_full = cfg->i_fully_developed; P.j_full = cfg->j_fully_developed;
    P.c_mol = cfg->convert_mol_to_Nm3; P.Hu_ch4 = cfg->H_u_CH4; P.Hu_h2 = cfg->H_u_H2;
    P.dt_cp_evap = cfg->dt_water * cfg->cp_water + cfg->h_H2O_evap;      // :296
    P.heat_price = cfg->heat_price; P.o2_price = cfg->o2_price; P.eeg = cfg->eeg_el_price; P.eta_chp = cfg->eta_CHP;
    P.one_m_eta_chp = 1 - cfg->eta_CHP; P.M_co2 = cfg->Molar_mass_CO2; P.M_h2o = cfg->Molar_mass_H2O; P.rho = cfg->rho_water;
    P.water_price = cfg->water_price; P.min_load = cfg->min_load_electrolyzer; P.max_h2 = cfg->max_h2_volumeflow;
    P.c_m2 = 1.68 * std::pow(10.0, -3.0);                               // python: 1.68 * 10 ** (-3) (:316)
    P.c_m3 = 2.51 * std::pow(10.0, -5.0);                               // python: 2.51 * 10 ** (-5) (:317)
    P.sim_step_d = (double)cfg->sim_step;
    P.T_lo = cfg->T_l_b; P.T_rng = cfg->T_u_b - cfg->T_l_b; P.h2_lo = cfg->h2_l_b; P.h2_rng = cfg->h2_u_b - cfg->h2_l_b;
    P.ch4_lo = cfg->ch4_l_b; P.ch4_rng = cfg->ch4_u_b - cfg->ch4_l_b; P.h2r_lo = cfg->h2_res_l_b; P.h2r_rng = cfg->h2_res_u_b - cfg->h2_res_l_b;
    P.h2o_lo = cfg->h2o_l_b; P.h2o_rng = cfg->h2o_u_b - cfg->h2o_l_b; P.heat_lo = cfg->heat_l_b; P.heat_rng = cfg->heat_u_b - cfg->heat_l_b;

    // temporal encoding (:442,449-450) for every step count of an episode, with the reference's growing argument
    std::vector<double2> sc((size_t)cfg->eps_sim_steps + 1);
    for (int k1 = 0; k1 <= cfg->eps_sim_steps; k1++) {
        const double clock_hours = (double)((long long)k1 * cfg->sim_step) / 3600;
        sc[k1] = make_double2(host_sin(2 * M_PI * clock_hours), host_cos(2 * M_PI * clock_hours));
    }
    double2* d_sc;
    if ((rc = dev_upload(h, &d_sc, sc.data(), sc.size()))) return fail(rc);
    P.sincos = d_sc;
    {
        std::vector<float2> sc32(sc.size());
        for (size_t q = 0; q < sc.size(); q++) sc32[q] = make_float2((float)sc[q].x, (float)sc[q].y);
        float2* d_sc32;
        if ((rc = dev_upload(h, &d_sc32, sc32.data(), sc32.size()))) return fail(rc);
        P.sincos32 = d_sc32;
        // pool32 = [featA | featB | gas_n | eua_n | sin,cos pairs]: one base pointer for every float32 lookup of the fast kernels
        std::vector<float>& pool = h->pool32_host;
        h->off_sc = (unsigned)pool.size();
        for (size_t q = 0; q < sc32.size(); q++) { pool.push_back(sc32[q].x); pool.push_back(sc32[q].y); }
        pool.resize(pool.size() + 16, 0.f);
        float* d_pool;
        if ((rc = dev_upload(h, &d_pool, pool.data(), pool.size()))) return fail(rc);
        h->d_pool32 = d_pool;
        P.featA32 = d_pool; P.featB32 = d_pool + h->off_featB; P.gas_n32 = d_pool + h->off_gasn; P.eua_n32 = d_pool + h->off_euan;
        pool.clear(); pool.shrink_to_fit();
        std::vector<double>& p64 = h->pool64_host;
        h->o64_sc = (unsigned)p64.size();
        for (size_t q = 0; q < sc.size(); q++) { p64.push_back(sc[q].x); p64.push_back(sc[q].y); }
        p64.resize(p64.size() + 16, 0.0);
        double* d_p64;
        if ((rc = dev_upload(h, &d_p64, p64.data(), p64.size()))) return fail(rc);
        h->d_pool64 = d_p64;
        p64.clear(); p64.shrink_to_fit();
    }
    // fast-path records: reward coefficients per window start (k_build_fast)
    {
        const double f = P.sim_step_d / 3600;
        P.k_chp = cfg->convert_mol_to_Nm3 * cfg->H_u_CH4 * 1000 * (cfg->eta_CHP * cfg->eeg_el_price + (1 - cfg->eta_CHP) * cfg->heat_price) * f;
        P.k_eua = cfg->Molar_mass_CO2 / 1000 / 1000 * 3600 * 100 * f;
        RecFast* d_recf;
        if ((rc = dev_alloc(h, &d_recf, h->rec_total))) return fail(rc);
        hipLaunchKernelGGL(k_build_fast, dim3(grid_for((long long)h->rec_total, 256)), dim3(256), 0, 0, P, P.rec, d_recf, (int)h->rec_total);
        if ((rc = launch_check(h, "k_build_fast"))) return fail(rc);
        P.recf = d_recf;
        if ((rc = dev_alloc(h, &h->d_rkey, h->rec_total + 8))) return fail(rc);
        hipLaunchKernelGGL(k_extract_keys, dim3(grid_for((long long)h->rec_total, 256)), dim3(256), 0, 0, d_recf, h->d_rkey, (int)h->rec_total);
        if ((rc = launch_check(h, "k_extract_keys"))) return fail(rc);
    }

    if ((rc = dev_alloc(h, &P.st_a, n_envs)) || (rc = dev_alloc(h, &P.st_b, n_envs)) || (rc = dev_alloc(h, &P.st_c, n_envs)))
        return fail(rc);
    P.fin_cap = std::max(2 * n_envs, 1024);
    if ((rc = dev_alloc(h, &P.fin_ret, P.fin_cap)) || (rc = dev_alloc(h, &P.fin_len, P.fin_cap)) ||
        (rc = dev_alloc(h, &P.fin_env, P.fin_cap)) || (rc = dev_alloc(h, &P.fin_count, 1)))
        return fail(rc);
    {   // the error words live in pinned host memory the kernels can write (see DevParams::err)
        void* dp = nullptr;
        if (hipHostMalloc((void**)&h->err_host, 4 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&dp, h->err_host, 0) != hipSuccess) { set_err(h, PTG_E_HIP, "hipHostMalloc of the error words failed"); return fail(PTG_E_HIP); }
        h->err_host[0] = 0; h->err_host[1] = 0; h->err_host[2] = 0; h->err_host[3] = 0;
        P.err = (int*)dp;
    }
    if ((rc = dev_alloc(h, &P.term_flag, 4))) return fail(rc);
    if (hipMemset(P.term_flag, 0, 4 * sizeof(int)) != hipSuccess) { set_err(h, PTG_E_HIP, "hipMemset failed"); return fail(PTG_E_HIP); }
    {   // pinned staging of ptg_finished_episodes, sized for the whole ring (allocated here: a first query pays no hipHostMalloc)
        const size_t need = (size_t)P.fin_cap * (sizeof(double) + 2 * sizeof(int));
        if (hipHostMalloc(&h->fin_stage, need, hipHostMallocDefault) == hipSuccess) h->fin_stage_bytes = need;
        else { (void)hipGetLastError(); h->fin_stage = nullptr; }
    }
    if (hipMemset(P.fin_count, 0, sizeof(int)) != hipSuccess) {
        set_err(h, PTG_E_HIP, "hipMemset failed");
        return fail(PTG_E_HIP);
    }
    {   // load-change ladder thresholds of the hot kernels (staged into LDS by every workgroup)
        const int lad[LAD_N] = {P.t1_start_p_f, P.t2_start_f_p, P.t_p_f, P.t_f_p, P.t1_p_f_p, P.t2_p_f_p, P.t3_p_f_p, P.t34_p_f_p, P.t4_p_f_p,
                                P.t45_p_f_p, P.t5_p_f_p, P.t1_f_p_f, P.t2_f_p_f, P.t23_f_p_f, P.t3_f_p_f, P.t34_f_p_f, P.t4_f_p_f, P.t45_f_p_f,
                                P.t5_f_p_f, P.i_full, P.j_full};
        if ((rc = dev_upload(h, &h->d_ladder, lad, LAD_N))) return fail(rc);
    }
    hipLaunchKernelGGL(k_init_state, dim3(grid_for(n_envs, 256)), dim3(256), 0, 0, P, 0, 0);
    if ((rc = launch_check(h, "k_init_state"))) return fail(rc);
    if (!h->knob_no_refresh) {                              // the refresher's own stream: highest priority, never blocks on the NULL stream
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&h->ref_stream, hipStreamNonBlocking, hi) != hipSuccess) { h->ref_stream = nullptr; (void)hipGetLastError(); }
        if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (h->ref_stream) { (void)hipStreamDestroy(h->ref_stream); h->ref_stream = nullptr; }     // no fork / join events: no rolling passes
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) { set_err(h, PTG_E_HIP, "device synchronize failed after init"); return fail(PTG_E_HIP); }
    *out = h;
    return 0;
}

int ptg_set_market_assignment(ptg_env* h, const uint8_t* set_of_env_host)
{
    if (!h || !set_of_env_host) return set_err(h, PTG_E_INVALID, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    std::vector<StA> a(h->n);
    HIP_TRY(h, hipMemcpy(a.data(), h->P.st_a, sizeof(StA) * h->n, hipMemcpyDeviceToHost));
    for (int e = 0; e < h->n; e++) {
        if (set_of_env_host[e] >= h->n_sets) return set_err(h, PTG_E_INVALID, "env %d: market set %d out of range", e, set_of_env_host[e]);
        a[e].flags = (a[e].flags & ~(3u << 15)) | ((unsigned)set_of_env_host[e] << 15);
    }
    HIP_TRY(h, hipMemcpy(h->P.st_a, a.data(), sizeof(StA) * h->n, hipMemcpyHostToDevice));
    return 0;
}

int ptg_set_episode_plan(ptg_env* h, const double* eps_ind_host, int n, int64_t first_ptr, int64_t stride)
{
    if (!h || n < 0 || (n > 0 && !eps_ind_host) || first_ptr < 0 || stride < 0) return set_err(h, PTG_E_INVALID, "bad episode plan");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->d_eps_ind) { (void)hipFree(h->d_eps_ind); h->d_eps_ind = nullptr; }
    h->P.E = n; h->P.eps_ind = nullptr; h->P.ep_stride = 0;
    if (n > 0) {
        std::vector<int> ei(n);
        for (int q = 0; q < n; q++) ei[q] = (int)eps_ind_host[q];            // int(eps_ind[..] * eps_len_d) for integral entries (:60-61)
        HIP_TRY(h, hipMalloc((void**)&h->d_eps_ind, sizeof(int) * n));
        HIP_TRY(h, hipMemcpy(h->d_eps_ind, ei.data(), sizeof(int) * n, hipMemcpyHostToDevice));
        h->P.eps_ind = h->d_eps_ind;
        h->P.ep_stride = (int)(stride % n);
    }
    hipLaunchKernelGGL(k_init_state, dim3(grid_for(h->n, 256)), dim3(256), 0, 0, h->P, n > 0 ? (int)(first_ptr % n) : 0, 1);
    int rc = launch_check(h, "k_init_state");
    if (rc) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    return 0;
}

static int set_tape_len(ptg_env* h, int L)
{
    if (L != h->tape_len) {
        if (h->d_tape) { (void)hipFree(h->d_tape); h->d_tape = nullptr; }
        h->tape_len = 0;
        if (L > 0) {
            HIP_TRY(h, hipMalloc((void**)&h->d_tape, sizeof(double) * (size_t)h->n * L));
            h->tape_len = L;
        }
    }
    h->P.tape = h->d_tape; h->P.tape_len = h->tape_len;
    return 0;
}

int ptg_set_noise_tape(ptg_env* h, const double* tape_host, int per_env_len)
{
    if (!h || per_env_len < 0 || (per_env_len > 0 && !tape_host)) return set_err(h, PTG_E_INVALID, "bad noise tape");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = set_tape_len(h, per_env_len);
    if (rc) return rc;
    h->P.noise_inline = 0;
    if (per_env_len > 0) {                                     // env-major host rows -> the draw-major device tape
        double* tmp = nullptr;
        const size_t bytes = sizeof(double) * (size_t)h->n * per_env_len;
        HIP_TRY(h, hipMalloc((void**)&tmp, bytes));
        hipError_t e = hipMemcpy(tmp, tape_host, bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_transpose, dim3((per_env_len + 31) / 32, (h->n + 31) / 32), dim3(32, 8), 0, 0, tmp, h->d_tape, h->n, per_env_len);
            e = hipDeviceSynchronize();
        }
        (void)hipFree(tmp);
        if (e != hipSuccess) return set_err(h, PTG_E_HIP, "noise tape upload failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_zero_noise_count, dim3(grid_for(h->n, 256)), dim3(256), 0, 0, h->P);
    if ((rc = launch_check(h, "k_zero_noise_count"))) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    return 0;
}

int ptg_fill_noise_tape(ptg_env* h, uint64_t seed, int per_env_len, void* stream)
{
    if (!h || per_env_len <= 0) return set_err(h, PTG_E_INVALID, "bad noise tape length");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = set_tape_len(h, per_env_len);
    if (rc) return rc;
    const long long total = (long long)h->n * per_env_len;
    hipLaunchKernelGGL(k_fill_noise, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), h->d_tape, h->n, per_env_len,
                       (unsigned long long)seed, h->P.env_offset, h->cfg.noise);
    h->P.noise_inline = 0; h->P.noise_seed = (unsigned long long)seed;
    if ((rc = launch_check(h, "k_fill_noise"))) return rc;
    hipLaunchKernelGGL(k_zero_noise_count, dim3(grid_for(h->n, 256)), dim3(256), 0, as_stream(stream), h->P);
    return launch_check(h, "k_zero_noise_count");
}

int ptg_set_noise_rng(ptg_env* h, uint64_t seed)
{
    if (!h) return PTG_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    int rc = set_tape_len(h, 0);
    if (rc) return rc;
    h->P.noise_inline = 1; h->P.noise_seed = (unsigned long long)seed;
    hipLaunchKernelGGL(k_zero_noise_count, dim3(grid_for(h->n, 256)), dim3(256), 0, 0, h->P);
    if ((rc = launch_check(h, "k_zero_noise_count"))) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    return 0;
}

int ptg_set_feature_pitch(ptg_env* h, int64_t pitch)
{
    if (!h) return PTG_E_INVALID;
    if (!h->fm) return set_err(h, PTG_E_INVALID, "ptg_set_feature_pitch: the handle's obs_layout is not PTG_OBS_FEATURE_MAJOR");
    if (pitch < h->n || pitch > (int64_t)h->n + (1 << 20)) return set_err(h, PTG_E_INVALID, "ptg_set_feature_pitch: pitch must be in [n_envs, n_envs + 2^20]");
    if (h->hs.active) return set_err(h, PTG_E_INVALID, "ptg_set_feature_pitch: a host step is in flight");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    h->P.fm_pitch = (int)pitch;
    for (void** q : {&h->hs_out, &h->hs_final}) if (*q) { (void)hipFree(*q); *q = nullptr; }      // staged host steps: sized by the old pitch
    return 0;
}

int ptg_set_global_env_offset(ptg_env* h, int64_t offset)
{
    if (!h || offset < 0) return set_err(h, PTG_E_INVALID, "bad env offset");
    h->P.env_offset = (long long)offset;
    return 0;
}

int ptg_get_noise_tape(ptg_env* h, double* tape_host)
{
    if (!h || !tape_host) return set_err(h, PTG_E_INVALID, "null argument");
    if (h->tape_len <= 0) return set_err(h, PTG_E_INVALID, "no noise tape set");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    double* tmp = nullptr;                                     // the draw-major device tape -> env-major host rows
    const size_t bytes = sizeof(double) * (size_t)h->n * h->tape_len;
    HIP_TRY(h, hipMalloc((void**)&tmp, bytes));
    hipLaunchKernelGGL(k_transpose, dim3((h->n + 31) / 32, (h->tape_len + 31) / 32), dim3(32, 8), 0, 0, (const double*)h->d_tape, tmp, h->tape_len, h->n);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(tape_host, tmp, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (e != hipSuccess) return set_err(h, PTG_E_HIP, "noise tape download failed: %s", hipGetErrorString(e));
    return 0;
}

int ptg_reset(ptg_env* h, const uint8_t* mask_host, void* obs_dev, void* stream)
{
    if (!h) return PTG_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = as_stream(stream);
    uint8_t* d_mask = nullptr;
    if (mask_host) {
        HIP_TRY(h, hipMalloc((void**)&d_mask, h->n));
        HIP_TRY(h, hipMemcpyAsync(d_mask, mask_host, h->n, hipMemcpyHostToDevice, st));
    }
    const dim3 grid(grid_for(h->n, 256)), block(256);
    if (h->cfg.out_dtype == PTG_OUT_F64) {
        if (h->fm) hipLaunchKernelGGL((k_reset<double, false, true, 0>), grid, block, 0, st, h->P, d_mask, (double*)obs_dev);
        else hipLaunchKernelGGL((k_reset<double, false, false, 0>), grid, block, 0, st, h->P, d_mask, (double*)obs_dev);
    } else {
        if (h->fm) hipLaunchKernelGGL((k_reset<float, true, true, 0>), grid, block, 0, st, h->P, d_mask, (float*)obs_dev);
        else hipLaunchKernelGGL((k_reset<float, true, false, 0>), grid, block, 0, st, h->P, d_mask, (float*)obs_dev);
    }
    int rc = launch_check(h, "k_reset");
    if (d_mask) { (void)hipStreamSynchronize(st); (void)hipFree(d_mask); }
    if (rc) return rc;
    bool all = !mask_host;
    if (mask_host) { all = true; for (int e = 0; e < h->n && all; e++) all = mask_host[e] != 0; }
    if (all) { h->reset_done = true; h->sync_k = 0; }      // (a mask that selects every env is a full reset)
    else h->sync_k = -1;                                 // a partial reset de-synchronises the batch: generic kernels from here on
    return 0;
}

// generic (any configuration, handles termination + auto-reset) launch of one vector step
static int launch_step_generic(ptg_env* h, hipStream_t st, const void* actions_dev, int action_kind, void* obs_dev, void* rew_dev,
                               uint8_t* done_dev, void* final_obs_dev, double* info_dev, int only_at_k = -1)
{
    const dim3 grid(grid_for(h->n, 256)), block(256);
    const bool f64 = h->cfg.out_dtype == PTG_OUT_F64;
    h->fin_maybe = true;
#define PTG_LAUNCH_STEP_(OUT, FAST, INFO, FM, PAC)                                                                     \
    hipLaunchKernelGGL((k_step<OUT, FAST, INFO, FM, PAC>), grid, block, 0, st, h->P, actions_dev, action_kind, (OUT*)obs_dev, \
                       (OUT*)rew_dev, done_dev, (OUT*)final_obs_dev, info_dev, only_at_k)
#define PTG_LAUNCH_STEP(OUT, FAST, INFO, FM)                                                                          \
    do { if (h->cfg.price_ahead == 13) PTG_LAUNCH_STEP_(OUT, FAST, INFO, FM, 13); else PTG_LAUNCH_STEP_(OUT, FAST, INFO, FM, 0); } while (0)
    if (f64) {
        if (info_dev) { if (h->fm) PTG_LAUNCH_STEP(double, false, true, true); else PTG_LAUNCH_STEP(double, false, true, false); }
        else { if (h->fm) PTG_LAUNCH_STEP(double, false, false, true); else PTG_LAUNCH_STEP(double, false, false, false); }
    } else if (info_dev) {      // info rows need the un-reduced reward terms: float32 outputs of the float64 formula
        if (h->fm) PTG_LAUNCH_STEP(float, false, true, true); else PTG_LAUNCH_STEP(float, false, true, false);
    } else {
        if (h->fm) PTG_LAUNCH_STEP(float, true, false, true); else PTG_LAUNCH_STEP(float, true, false, false);
    }
#undef PTG_LAUNCH_STEP
#undef PTG_LAUNCH_STEP_
    return launch_check(h, "k_step");
}

// generic rollout = T generic step launches (measured faster than a fused generic kernel, whose register pressure spills):
// float64 outputs, unusual price_ahead, de-synchronised batches, and the one terminating step per episode take this route
static int launch_rollout_generic(ptg_env* h, hipStream_t st, const void* actions_dev, int action_kind, int n_steps, void* obs_dev,
                                  void* rew_dev, uint8_t* done_dev, double* info_dev = nullptr)
{
    const size_t asz = action_kind == PTG_ACT_I64 ? 8 : 4, osz = h->cfg.out_dtype == PTG_OUT_F64 ? 8 : 4;
    for (int t = 0; t < n_steps; t++) {
        const int rc = launch_step_generic(h, st, (const char*)actions_dev + (size_t)t * h->n * asz, action_kind,
                                           (char*)obs_dev + (size_t)t * obs_step_elems(h) * osz, (char*)rew_dev + (size_t)t * h->n * osz,
                                           done_dev + (size_t)t * h->n, nullptr, info_dev ? info_dev + (size_t)t * h->n * PTG_N_INFO : nullptr);
        if (rc) return rc;
    }
    return 0;
}

int ptg_step(ptg_env* h, const void* actions_dev, int action_kind, void* obs_dev, void* rew_dev, uint8_t* done_dev,
             void* final_obs_dev, double* info_dev, void* stream)
{
    if (!h) return PTG_E_INVALID;
    if (!actions_dev || !obs_dev || !rew_dev || !done_dev) return set_err(h, PTG_E_INVALID, "ptg_step: null buffer");
    if (action_kind < PTG_ACT_I32 || action_kind > PTG_ACT_I64) return set_err(h, PTG_E_INVALID, "ptg_step: bad action_kind");
    if ((h->cfg.action_type == 1) != (action_kind == PTG_ACT_F32))
        return set_err(h, PTG_E_INVALID, "ptg_step: action_kind does not match cfg.action_type");
    if (!h->reset_done) return set_err(h, PTG_E_INVALID, "ptg_step: envs must be reset first");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = as_stream(stream);
    const int k_term = h->cfg.eps_sim_steps - 6;               // the step taken at k == k_term terminates (:508-511)
    if (hot_eligible(h) && !info_dev && h->sync_k != k_term) {
        // Being captured into a hipGraph with ptg_set_replay_proof(env, 1), the step is enqueued in its replay-proof form: the hot kernel,
        // which does nothing when it finds the batch on the terminating step, and behind it the generic kernel, which does nothing unless
        // the hot kernel skipped.  Every replay then takes the right one by itself -- across episode ends, auto-reset and finished-episode
        // list included -- for the price of one empty launch per step (+1.5-2 us; folding the generic step into the hot kernel instead cost
        // EVERY launch 0.5 us: 4.95 -> 5.45 us, a second parameter block and scratch).  Default: the hot kernel alone, which flags a replay
        // that reaches the terminating step.  (Eager calls are routed on the host as before.)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        const bool capturing = cs != hipStreamCaptureStatusNone && h->replay_proof;
        h->step_skip_term = capturing ? 1 : 0;
        PTG_HOT_DISPATCH(launch_step_hot, h, st, actions_dev, action_kind, PTG_NOARG, obs_dev, rew_dev, done_dev);
        h->step_skip_term = 0;
        int rc = launch_check(h, "k_step_hot");
        if (!rc && capturing) rc = launch_step_generic(h, st, actions_dev, action_kind, obs_dev, rew_dev, done_dev, final_obs_dev, nullptr, k_term);
        h->sync_k += 1;
        return rc;
    }
    const int rc = launch_step_generic(h, st, actions_dev, action_kind, obs_dev, rew_dev, done_dev, final_obs_dev, info_dev);
    if (h->sync_k >= 0) h->sync_k = (h->sync_k == k_term) ? 0 : h->sync_k + 1;   // a synchronised batch resets together
    return rc;
}

// ptg_rollout / ptg_rollout_info: hot segments between the (rare) terminating steps; generic kernels for everything else.  Info rows
// come out of the fused kernel for float64 outputs (it evaluates the reference-order reward terms anyway), else of generic steps.
static int rollout_impl(ptg_env* h, const char* what, const void* actions_dev, int action_kind, int n_steps, void* obs_dev, void* rew_dev,
                        uint8_t* done_dev, double* info_dev, void* stream)
{
    if (!h) return PTG_E_INVALID;
    if (!actions_dev || !obs_dev || !rew_dev || !done_dev || n_steps < 1) return set_err(h, PTG_E_INVALID, "%s: bad argument", what);
    if (action_kind < PTG_ACT_I32 || action_kind > PTG_ACT_I64) return set_err(h, PTG_E_INVALID, "%s: bad action_kind", what);
    if ((h->cfg.action_type == 1) != (action_kind == PTG_ACT_F32))
        return set_err(h, PTG_E_INVALID, "%s: action_kind does not match cfg.action_type", what);
    if (!h->reset_done) return set_err(h, PTG_E_INVALID, "%s: envs must be reset first", what);
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = as_stream(stream);
    const int k_term = h->cfg.eps_sim_steps - 6;
    const size_t asz = action_kind == PTG_ACT_I64 ? 8 : 4, osz = h->cfg.out_dtype == PTG_OUT_F64 ? 8 : 4;
    const bool hot_ok_info = !info_dev || h->cfg.out_dtype == PTG_OUT_F64;
    int t0 = 0;
    while (t0 < n_steps) {
        const char* a_t = (const char*)actions_dev + (size_t)t0 * h->n * asz;
        char* o_t = (char*)obs_dev + (size_t)t0 * obs_step_elems(h) * osz;
        char* r_t = (char*)rew_dev + (size_t)t0 * h->n * osz;
        uint8_t* d_t = done_dev + (size_t)t0 * h->n;
        double* i_t = info_dev ? info_dev + (size_t)t0 * h->n * PTG_N_INFO : nullptr;
        int rc = 0, cnt;
        if (hot_eligible(h) && hot_ok_info && h->sync_k != k_term) {
            cnt = std::min(n_steps - t0, k_term - h->sync_k);
            h->rollout_info = i_t;
            PTG_HOT_DISPATCH(launch_rollout_hot, h, st, a_t, action_kind, PTG_COMMA_ARG(cnt), o_t, r_t, d_t);
            h->rollout_info = nullptr;
            rc = launch_check(h, "k_rollout_pc");
            h->sync_k += cnt;
        } else if (h->sync_k >= 0) {                            // the terminating step of a synchronised batch (or a float32 info stream)
            cnt = 1;
            rc = launch_rollout_generic(h, st, a_t, action_kind, 1, o_t, r_t, d_t, i_t);
            h->sync_k = (h->sync_k == k_term) ? 0 : h->sync_k + 1;
        } else {
            cnt = n_steps - t0;
            rc = launch_rollout_generic(h, st, a_t, action_kind, cnt, o_t, r_t, d_t, i_t);
        }
        if (rc) return rc;
        t0 += cnt;
    }
    return 0;
}

int ptg_rollout(ptg_env* h, const void* actions_dev, int action_kind, int n_steps, void* obs_dev, void* rew_dev,
                uint8_t* done_dev, void* stream)
{
    return rollout_impl(h, "ptg_rollout", actions_dev, action_kind, n_steps, obs_dev, rew_dev, done_dev, nullptr, stream);
}

int ptg_rollout_info(ptg_env* h, const void* actions_dev, int action_kind, int n_steps, void* obs_dev, void* rew_dev,
                     uint8_t* done_dev, double* info_dev, void* stream)
{
    if (h && !info_dev) return set_err(h, PTG_E_INVALID, "ptg_rollout_info: bad argument");
    return rollout_impl(h, "ptg_rollout_info", actions_dev, action_kind, n_steps, obs_dev, rew_dev, done_dev, info_dev, stream);
}

int ptg_rollout_launches(ptg_env* h, int n_steps)
{
    if (!h || n_steps < 1) return PTG_E_INVALID;
    const int k_term = h->cfg.eps_sim_steps - 6;
    int k = h->sync_k, t0 = 0, launches = 0;
    while (t0 < n_steps) {                                  // mirrors the segment loop of ptg_rollout
        if (hot_eligible(h) && k >= 0 && k != k_term) {
            const PcPlan pl = pc_plan(h);
            const int cnt = std::min(n_steps - t0, k_term - k);
            launches += ((cnt + pl.t_cap - 1) / pl.t_cap) * ((h->n + pl.chunk - 1) / pl.chunk);
            k += cnt; t0 += cnt;
        } else if (k >= 0) {
            launches += 1; k = (k == k_term) ? 0 : k + 1; t0 += 1;
        } else {
            launches += n_steps - t0; t0 = n_steps;
        }
    }
    return launches;
}

// The hot kernels take the step count from the device state, so a captured ptg_step / ptg_rollout can be replayed; the host's own count
// (which routes the terminating step of an episode to the generic kernel) only sees eager calls and the capture itself.
int ptg_note_replays(ptg_env* h, int n_steps)
{
    if (!h || n_steps < 0) return set_err(h, PTG_E_INVALID, "ptg_note_replays: bad argument");
    if (h->sync_k < 0) return set_err(h, PTG_E_INVALID, "ptg_note_replays: the batch is not synchronised (captured hot launches do not exist for it)");
    // a captured ptg_step carries its own terminating-step kernel: replays may run across episode ends (the step count wraps at k_term + 1).
    // (A captured ptg_rollout must not: its kernel flags the overrun, see check_error_flags.)
    const long long period = (long long)(h->cfg.eps_sim_steps - 6) + 1;
    h->sync_k = (int)(((long long)h->sync_k + n_steps) % period);
    h->fin_maybe = true;
    return 0;
}

int ptg_set_replay_proof(ptg_env* h, int enable)
{
    if (!h) return PTG_E_INVALID;
    h->replay_proof = enable ? 1 : 0;
    return 0;
}

int ptg_steps_to_episode_end(ptg_env* h, int* steps)
{
    if (!h || !steps) return set_err(h, PTG_E_INVALID, "bad argument");
    *steps = h->sync_k >= 0 ? (h->cfg.eps_sim_steps - 6) - h->sync_k + 1 : 0;
    return 0;
}

// ---- the SB3-facing form of a step: host buffers in, host buffers out, one call ---------------------------------------
int ptg_host_layout_ex(const ptg_env* h, size_t* off_rew, size_t* off_done, size_t* off_status, size_t* total)
{
    if (!h) return PTG_E_INVALID;
    const size_t osz = h->cfg.out_dtype == PTG_OUT_F64 ? 8 : 4;
    const size_t o_rew = (obs_step_elems(h) * osz + 15) / 16 * 16, o_done = (o_rew + (size_t)h->n * osz + 15) / 16 * 16;
    const size_t o_status = (o_done + (size_t)h->n + 15) / 16 * 16;
    if (off_rew) *off_rew = o_rew;
    if (off_done) *off_done = o_done;
    if (off_status) *off_status = o_status;
    if (total) *total = (o_status + (size_t)h->n + 15) / 16 * 16;
    return 0;
}

int ptg_host_layout(const ptg_env* h, size_t* off_rew, size_t* off_done, size_t* total) { return ptg_host_layout_ex(h, off_rew, off_done, nullptr, total); }

namespace {

// where METH_STATUS sits in an observation row of this handle's layout (k_pack_status)
void status_column(const ptg_env* h, int& c0, int& onehot)
{
    const int q_stat = h->cfg.raw_modified ? 2 * h->cfg.price_ahead : h->cfg.price_ahead + 4;      // canonical column (:219-249)
    onehot = (h->flat || h->split) ? 1 : 0;
    c0 = h->split ? 0 : q_stat;
    if (h->flat) c0 = h->status_col_flat;                     // sorted-key order ('mod': CH4, Elec_Heating, H2O, H2_in, H2_res come first: column 5)
}

int wait_event(ptg_env* h, hipEvent_t ev)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return set_err(h, PTG_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        (void)hipGetLastError();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
    }
    HIP_TRY(h, hipEventSynchronize(ev));
    return 0;
}

}  // namespace

int ptg_step_host_begin(ptg_env* h, const void* actions_host, int action_kind, void* out_host, void* final_obs_host, double* info_host, void* stream)
{
    if (!h) return PTG_E_INVALID;
    if (!actions_host || !out_host) return set_err(h, PTG_E_INVALID, "ptg_step_host: null buffer");
    if (h->hs.active) return set_err(h, PTG_E_INVALID, "ptg_step_host_begin: the previous host step has not been ended (ptg_step_host_end)");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = as_stream(stream);
    const size_t osz = h->cfg.out_dtype == PTG_OUT_F64 ? 8 : 4, asz = action_kind == PTG_ACT_I64 ? 8 : 4;
    size_t o_rew, o_done, o_status, total;
    ptg_host_layout_ex(h, &o_rew, &o_done, &o_status, &total);
    const size_t final_bytes = obs_step_elems(h) * osz, info_bytes = (size_t)h->n * PTG_N_INFO * sizeof(double);
    // Small batches: the kernels read the actions from and write their outputs to the caller's pinned buffers directly (zero
    // copy: no DMA descriptors, one launch + one synchronise per step).  Large ones: device staging; the outputs come back as TWO
    // copies -- [rewards | done flags | status] (+ info rows) first, with an event behind them, then the observations -- so that the
    // caller's work on the small part (ptg_step_host_tail) overlaps the 9-18 MB of observations still crossing PCIe.
    // A buffer's classification is cached by its address (8 entries): the caller keeps a buffer registered / allocated for as long as
    // it passes it here, and calls ptg_host_buffers_changed() before re-using an ADDRESS for memory of another kind.
    const void* ptrs[4] = {actions_host, out_host, final_obs_host, info_host};
    void* devp[4] = {nullptr, nullptr, nullptr, nullptr};
    bool zc = total + final_bytes + (info_host ? info_bytes : 0) + (size_t)h->n * asz <= (256u << 10);
    for (int q = 0; q < 4 && zc; q++) {
        if (!ptrs[q]) continue;
        const ptg_env::HostPtr* hit = nullptr;
        for (const auto& m : h->hs_map) if (m.host == ptrs[q]) { hit = &m; break; }
        if (!hit) {
            hipPointerAttribute_t at;
            ptg_env::HostPtr& m = h->hs_map[h->hs_next];
            h->hs_next = (h->hs_next + 1) % 8;
            m.host = ptrs[q]; m.dev = nullptr;
            if (hipPointerGetAttributes(&at, ptrs[q]) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) m.dev = at.devicePointer;
            else (void)hipGetLastError();
            hit = &m;
        }
        devp[q] = hit->dev;
        zc = zc && hit->dev != nullptr;
    }
    if (!zc) {
        if (!h->hs_act) HIP_TRY(h, hipMalloc(&h->hs_act, (size_t)h->n * 8));
        if (!h->hs_out) HIP_TRY(h, hipMalloc(&h->hs_out, total));
        if (!h->hs_final) HIP_TRY(h, hipMalloc(&h->hs_final, final_bytes));
        if (info_host && !h->hs_info) HIP_TRY(h, hipMalloc((void**)&h->hs_info, info_bytes));
        if (!h->ev_tail) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_tail, hipEventDisableTiming));
    }
    const void* d_act = zc ? devp[0] : h->hs_act;
    char* d_out = (char*)(zc ? devp[1] : h->hs_out);
    void* d_final = final_obs_host ? (zc ? devp[2] : h->hs_final) : nullptr;
    double* d_info = info_host ? (zc ? (double*)devp[3] : h->hs_info) : nullptr;
    if (!zc) HIP_TRY(h, hipMemcpyAsync(h->hs_act, actions_host, (size_t)h->n * asz, hipMemcpyHostToDevice, st));
    int rc = ptg_step(h, d_act, action_kind, d_out, d_out + o_rew, (uint8_t*)(d_out + o_done), d_final, d_info, stream);
    if (rc) return rc;
    if (!zc) {
        int c0, onehot;
        status_column(h, c0, onehot);
        const dim3 grid(grid_for(h->n, 256)), block(256);
        if (osz == 8) hipLaunchKernelGGL(k_pack_status<double>, grid, block, 0, st, (const double*)d_out, h->n, h->F, c0, h->fm ? 1 : 0, onehot, (uint8_t*)(d_out + o_status), h->P.fm_pitch);
        else hipLaunchKernelGGL(k_pack_status<float>, grid, block, 0, st, (const float*)d_out, h->n, h->F, c0, h->fm ? 1 : 0, onehot, (uint8_t*)(d_out + o_status), h->P.fm_pitch);
        if ((rc = launch_check(h, "k_pack_status"))) return rc;
        HIP_TRY(h, hipMemcpyAsync((char*)out_host + o_rew, h->hs_out ? (char*)h->hs_out + o_rew : nullptr, total - o_rew, hipMemcpyDeviceToHost, st));
        if (info_host) HIP_TRY(h, hipMemcpyAsync(info_host, h->hs_info, info_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipEventRecord(h->ev_tail, st));
        HIP_TRY(h, hipMemcpyAsync(out_host, h->hs_out, o_rew, hipMemcpyDeviceToHost, st));
    }
    ptg_env::HostStep& hs = h->hs;
    hs.active = true; hs.zc = zc; hs.tail_done = false; hs.out_host = out_host; hs.final_host = final_obs_host; hs.info_host = info_host;
    hs.st = st; hs.n_done = 0;
    return 0;
}

int ptg_step_host_tail(ptg_env* h, int* n_done)
{
    if (!h) return PTG_E_INVALID;
    if (!h->hs.active) return set_err(h, PTG_E_INVALID, "ptg_step_host_tail: no host step in flight (ptg_step_host_begin)");
    HIP_TRY(h, hipSetDevice(h->device));
    ptg_env::HostStep& hs = h->hs;
    int rc;
    if (!hs.tail_done) {
        if ((rc = hs.zc ? wait_stream(h, hs.st) : wait_event(h, h->ev_tail))) { hs.active = false; return rc; }
        size_t o_rew, o_done, o_status, total;
        ptg_host_layout_ex(h, &o_rew, &o_done, &o_status, &total);
        const uint8_t* dn = (const uint8_t*)hs.out_host + o_done;
        int cnt = 0;
        size_t e = 0;
        for (; e + 8 <= (size_t)h->n; e += 8) {                     // done flags are 0 / 1 bytes: sum eight at a time
            uint64_t w; memcpy(&w, dn + e, 8);
            if (w) cnt += __builtin_popcountll(w);
        }
        for (; e < (size_t)h->n; e++) cnt += dn[e] != 0;
        hs.n_done = cnt;
        if (hs.zc) {                                                // tiny batch, written in place: the status bytes from the rows themselves
            int c0, onehot;
            status_column(h, c0, onehot);
            uint8_t* stt = (uint8_t*)hs.out_host + o_status;
            const bool f64 = h->cfg.out_dtype == PTG_OUT_F64;
            auto at = [&](size_t idx) -> double { return f64 ? ((const double*)hs.out_host)[idx] : (double)((const float*)hs.out_host)[idx]; };
            for (int q = 0; q < h->n; q++) {
                int v = 0;
                if (onehot) { for (int j = 1; j < 6; j++) v += at((size_t)q * h->F + c0 + j) != 0.0 ? j : 0; }
                else v = (int)(h->fm ? at((size_t)c0 * h->P.fm_pitch + q) : at((size_t)q * h->F + c0));
                stt[q] = (uint8_t)v;
            }
        }
        hs.tail_done = true;
    }
    if (n_done) *n_done = hs.n_done;
    return 0;
}

int ptg_step_host_end(ptg_env* h)
{
    if (!h) return PTG_E_INVALID;
    if (!h->hs.active) return set_err(h, PTG_E_INVALID, "ptg_step_host_end: no host step in flight (ptg_step_host_begin)");
    ptg_env::HostStep& hs = h->hs;
    int rc = hs.tail_done ? 0 : ptg_step_host_tail(h, nullptr);
    hs.active = false;                                          // whatever happens below, the step is over
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if ((rc = wait_stream(h, hs.st))) return rc;
    if ((rc = check_error_flags(h))) return rc;
    if (hs.n_done && !hs.zc && hs.final_host) {                 // rare: the terminal observations of the episodes that just ended
        const size_t final_bytes = obs_step_elems(h) * (h->cfg.out_dtype == PTG_OUT_F64 ? 8 : 4);
        HIP_TRY(h, hipMemcpyAsync(hs.final_host, h->hs_final, final_bytes, hipMemcpyDeviceToHost, hs.st));
        HIP_TRY(h, hipStreamSynchronize(hs.st));
    }
    return 0;
}

int ptg_step_host_finish(ptg_env* h, int* n_done)      // tail + end in one call (small batches: nothing to overlap)
{
    const int rc = ptg_step_host_tail(h, n_done);
    if (rc) return rc;
    return ptg_step_host_end(h);
}

int ptg_step_host(ptg_env* h, const void* actions_host, int action_kind, void* out_host, void* final_obs_host, double* info_host,
                  int* n_done, void* stream)
{
    if (!h) return PTG_E_INVALID;
    if (!n_done) return set_err(h, PTG_E_INVALID, "ptg_step_host: null buffer");
    int rc = ptg_step_host_begin(h, actions_host, action_kind, out_host, final_obs_host, info_host, stream);
    if (rc) return rc;
    if ((rc = ptg_step_host_tail(h, n_done))) return rc;
    return ptg_step_host_end(h);
}

int ptg_host_buffers_changed(ptg_env* h)
{
    if (!h) return PTG_E_INVALID;
    for (auto& m : h->hs_map) m = ptg_env::HostPtr();
    h->hs_next = 0;
    return 0;
}

// ---- VecNormalize(norm_obs=False) on the device ---------------------------------------------------------------------
int ptg_vn_init(ptg_env* h, double gamma, double epsilon, double clip_reward)
{
    if (!h) return PTG_E_INVALID;
    if (!(gamma >= 0.0) || !(epsilon >= 0.0) || !(clip_reward > 0.0)) return set_err(h, PTG_E_INVALID, "ptg_vn_init: bad gamma / epsilon / clip_reward");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if (!h->vn_returns) {
        if ((rc = dev_alloc(h, &h->vn_returns, (size_t)h->n)) || (rc = dev_alloc(h, &h->vn_stats, 3))) return rc;
    }
    h->vn_gamma = gamma; h->vn_eps = epsilon; h->vn_clip = clip_reward;
    const double st[3] = {0.0, 1.0, 1e-4};                  // RunningMeanStd(epsilon=1e-4): mean 0, var 1, count 1e-4
    HIP_TRY(h, hipMemset(h->vn_returns, 0, sizeof(double) * h->n));
    HIP_TRY(h, hipMemcpy(h->vn_stats, st, sizeof st, hipMemcpyHostToDevice));
    return 0;
}

static int vn_scratch(ptg_env* h, int T)
{
    const int nW = (h->n + 63) / 64;
    const size_t need = (size_t)T * nW * 3;
    if (need > h->vn_partials_cap) {
        if (h->vn_partials) (void)hipFree(h->vn_partials);
        h->vn_partials = nullptr; h->vn_partials_cap = 0;
        if (hipMalloc((void**)&h->vn_partials, need * sizeof(double)) != hipSuccess) return set_err(h, PTG_E_HIP, "hipMalloc of %zu bytes failed", need * sizeof(double));
        h->vn_partials_cap = need;
    }
    if (T > h->vn_T_cap) {
        if (h->vn_den) (void)hipFree(h->vn_den);
        if (h->vn_moments) (void)hipFree(h->vn_moments);
        h->vn_den = h->vn_moments = nullptr; h->vn_T_cap = 0;
        if (hipMalloc((void**)&h->vn_den, sizeof(double) * T) != hipSuccess || hipMalloc((void**)&h->vn_moments, sizeof(double) * 3 * T) != hipSuccess)
            return set_err(h, PTG_E_HIP, "hipMalloc failed");
        h->vn_T_cap = T;
    }
    return 0;
}

int ptg_vn_batch_moments(ptg_env* h, const void* rew_dev, const uint8_t* done_dev, int n_steps, double* moments_dev, void* stream)
{
    if (!h || !rew_dev || !done_dev || n_steps < 1) return set_err(h, PTG_E_INVALID, "ptg_vn_batch_moments: bad argument");
    if (!h->vn_returns) return set_err(h, PTG_E_INVALID, "ptg_vn_batch_moments: call ptg_vn_init first");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if ((rc = vn_scratch(h, n_steps))) return rc;
    hipStream_t st = as_stream(stream);
    const int nW = (h->n + 63) / 64;
    const dim3 grid(nW), block(64);
    if (h->cfg.out_dtype == PTG_OUT_F64)
        hipLaunchKernelGGL(k_vn_moments<double>, grid, block, 0, st, (const double*)rew_dev, done_dev, h->n, n_steps, h->vn_gamma, h->vn_returns, h->vn_partials, nW);
    else
        hipLaunchKernelGGL(k_vn_moments<float>, grid, block, 0, st, (const float*)rew_dev, done_dev, h->n, n_steps, h->vn_gamma, h->vn_returns, h->vn_partials, nW);
    hipLaunchKernelGGL(k_vn_merge, dim3(n_steps), dim3(64), 0, st, h->vn_partials, nW, moments_dev ? moments_dev : h->vn_moments);
    return launch_check(h, "k_vn_moments");
}

int ptg_vn_apply(ptg_env* h, const void* rew_dev, int n_steps, const double* moments_dev, void* rew_out_dev, int training, void* stream)
{
    if (!h || !rew_dev || !rew_out_dev || n_steps < 1) return set_err(h, PTG_E_INVALID, "ptg_vn_apply: bad argument");
    if (!h->vn_returns) return set_err(h, PTG_E_INVALID, "ptg_vn_apply: call ptg_vn_init first");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    if ((rc = vn_scratch(h, n_steps))) return rc;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(k_vn_scan, dim3(1), dim3(64), 0, st, moments_dev ? moments_dev : h->vn_moments, n_steps, training, h->vn_eps, h->vn_stats, h->vn_den);
    const size_t total = (size_t)n_steps * h->n;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (h->cfg.out_dtype == PTG_OUT_F64)
        hipLaunchKernelGGL(k_vn_norm<double>, grid, block, 0, st, (const double*)rew_dev, (double*)rew_out_dev, h->vn_den, h->n, total, h->vn_clip);
    else
        hipLaunchKernelGGL(k_vn_norm<float>, grid, block, 0, st, (const float*)rew_dev, (float*)rew_out_dev, h->vn_den, h->n, total, h->vn_clip);
    return launch_check(h, "k_vn_norm");
}

int ptg_vn_get(ptg_env* h, double* stats3_host, double* returns_host)
{
    if (!h || !h->vn_returns) return set_err(h, PTG_E_INVALID, "ptg_vn_get: not initialised");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    if (stats3_host) HIP_TRY(h, hipMemcpy(stats3_host, h->vn_stats, sizeof(double) * 3, hipMemcpyDeviceToHost));
    if (returns_host) HIP_TRY(h, hipMemcpy(returns_host, h->vn_returns, sizeof(double) * h->n, hipMemcpyDeviceToHost));
    return 0;
}

int ptg_vn_set(ptg_env* h, const double* stats3_host, const double* returns_host)
{
    if (!h || !h->vn_returns) return set_err(h, PTG_E_INVALID, "ptg_vn_set: not initialised");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    if (stats3_host) HIP_TRY(h, hipMemcpy(h->vn_stats, stats3_host, sizeof(double) * 3, hipMemcpyHostToDevice));
    if (returns_host) HIP_TRY(h, hipMemcpy(h->vn_returns, returns_host, sizeof(double) * h->n, hipMemcpyHostToDevice));
    return 0;
}

int ptg_profile(ptg_env* h, int enable)
{
    if (!h) return PTG_E_INVALID;
    if (enable && !h->profiling) {                          // a fresh collection
        for (auto& r : h->prof_used) {
            h->prof_free.push_back({r.e0, r.e1});
            if (r.h0) h->prof_free.push_back({r.h0, r.h1});
        }
        h->prof_used.clear();
        while (h->prof_free.size() < 16) {                  // event pairs created here, not inside the first timed launch
            std::pair<hipEvent_t, hipEvent_t> p{nullptr, nullptr};
            if (hipEventCreate(&p.first) != hipSuccess || hipEventCreate(&p.second) != hipSuccess) { (void)hipGetLastError(); break; }
            h->prof_free.push_back(p);
        }
    }
    h->profiling = enable != 0;
    return 0;
}

int ptg_profile_read_ex(ptg_env* h, double* us_host, double* helper_us_host, double* span_us_host, int cap, int* count)
{
    if (!h || !count || cap < 0 || (cap > 0 && !us_host)) return set_err(h, PTG_E_INVALID, "ptg_profile_read: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    int n = 0;
    for (auto& r : h->prof_used) {
        HIP_TRY(h, hipEventSynchronize(r.e1));
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, r.e0, r.e1));
        double us = (double)ms * 1e3, helper = 0.0, span = us;
        if (r.h0) {                                         // the union of [launch] and [its helper]: from the earlier start to the later end
            HIP_TRY(h, hipEventSynchronize(r.h1));
            float hm = 0.f, d0 = 0.f, d1 = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&hm, r.h0, r.h1));
            HIP_TRY(h, hipEventElapsedTime(&d0, r.e0, r.h0));      // helper start - launch start (negative: the helper started first)
            HIP_TRY(h, hipEventElapsedTime(&d1, r.e1, r.h1));      // helper end - launch end
            helper = (double)hm * 1e3;
            span = us + std::max(0.0, (double)d1 * 1e3) - std::min(0.0, (double)d0 * 1e3);
            h->prof_free.push_back({r.h0, r.h1});
        }
        if (n < cap) {
            us_host[n] = us;
            if (helper_us_host) helper_us_host[n] = helper;
            if (span_us_host) span_us_host[n] = span;
        }
        n++;
        h->prof_free.push_back({r.e0, r.e1});
    }
    h->prof_used.clear();
    *count = std::min(n, cap);
    return 0;
}

int ptg_profile_read(ptg_env* h, double* us_host, int cap, int* count) { return ptg_profile_read_ex(h, us_host, nullptr, nullptr, cap, count); }

int ptg_sync(ptg_env* h, void* stream)
{
    if (!h) return PTG_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    return collect_error(h, as_stream(stream));
}

int ptg_get_state(ptg_env* h, int field, void* out_host)
{
    if (!h || !out_host) return set_err(h, PTG_E_INVALID, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    const int n = h->n;
    const DevParams& P = h->P;
    std::vector<StA> a(n); std::vector<StB> b(n); std::vector<StC> c(n);
    HIP_TRY(h, hipMemcpy(a.data(), P.st_a, sizeof(StA) * n, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(b.data(), P.st_b, sizeof(StB) * n, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(c.data(), P.st_c, sizeof(StC) * n, hipMemcpyDeviceToHost));
    int* oi = (int*)out_host;
    double* od = (double*)out_host;
    for (int e = 0; e < n; e++) {
        const unsigned f = a[e].flags;
        const int pp = (f >> 6) & 7, fq = (f >> 9) & 7;
        switch (field) {
        case PTG_F_I: oi[e] = a[e].i; break;
        case PTG_F_J: oi[e] = a[e].j; break;
        case PTG_F_K: oi[e] = a[e].k; break;
        case PTG_F_ACT_EP_D: oi[e] = b[e].act_d; break;
        case PTG_F_EP_PTR: oi[e] = c[e].epp; break;
        case PTG_F_NOISE_COUNT: oi[e] = b[e].nctr; break;
        case PTG_F_N_STATE_CHANGES: oi[e] = c[e].nchg; break;
        case PTG_F_CUM_REW: od[e] = b[e].cum; break;
        case PTG_F_METH_STATE: oi[e] = f & 7; break;
        case PTG_F_HOT_COLD: oi[e] = (f >> 3) & 1; break;
        case PTG_F_STANDBY_TID: oi[e] = ((f >> 4) & 1) ? PTG_T_STANDBY_UP : PTG_T_STANDBY_DOWN; break;
        case PTG_F_STARTUP_TID: oi[e] = ((f >> 5) & 1) ? PTG_T_STARTUP_HOT : PTG_T_STARTUP_COLD; break;
        case PTG_F_PARTIAL_TID: oi[e] = pp == 0 ? PTG_T_OP1_START_P : 7 + pp; break;
        case PTG_F_FULL_TID: oi[e] = fq == 0 ? PTG_T_OP2_START_F : (fq == 1 ? PTG_T_OP3_P_F : 11 + fq); break;
        case PTG_F_CURRENT_ACTION: oi[e] = (f >> 12) & 7; break;
        case PTG_F_MARKET_SET: oi[e] = (f >> 15) & 3; break;
        case PTG_F_T_CAT: od[e] = h->Tvals[std::min<size_t>(f >> 17, h->Tvals.size() - 1)]; break;
        default: return set_err(h, PTG_E_INVALID, "unknown state field %d", field);
        }
    }
    return 0;
}

int ptg_set_state(ptg_env* h, int field, const void* in_host)
{
    if (!h || !in_host) return set_err(h, PTG_E_INVALID, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    const int n = h->n;
    DevParams& P = h->P;
    std::vector<StA> a(n); std::vector<StB> b(n); std::vector<StC> c(n);
    HIP_TRY(h, hipMemcpy(a.data(), P.st_a, sizeof(StA) * n, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(b.data(), P.st_b, sizeof(StB) * n, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(c.data(), P.st_c, sizeof(StC) * n, hipMemcpyDeviceToHost));
    const int* ii = (const int*)in_host;
    const double* id = (const double*)in_host;
    auto put = [](unsigned f, int shift, unsigned mask, unsigned v) { return (f & ~(mask << shift)) | ((v & mask) << shift); };
    for (int e = 0; e < n; e++) {
        unsigned f = a[e].flags;
        switch (field) {
        case PTG_F_I: a[e].i = ii[e]; break;
        case PTG_F_J: a[e].j = ii[e]; break;
        case PTG_F_K: a[e].k = ii[e]; break;
        case PTG_F_ACT_EP_D: b[e].act_d = ii[e]; break;
        case PTG_F_EP_PTR:
            if (ii[e] < 0 || (P.E > 0 && ii[e] >= P.E)) return set_err(h, PTG_E_INVALID, "episode pointer out of range");
            c[e].epp = ii[e]; break;
        case PTG_F_NOISE_COUNT: b[e].nctr = ii[e]; break;
        case PTG_F_N_STATE_CHANGES: c[e].nchg = ii[e]; break;
        case PTG_F_CUM_REW: b[e].cum = id[e]; break;
        case PTG_F_METH_STATE: if (ii[e] < 0 || ii[e] > 4) return set_err(h, PTG_E_INVALID, "bad meth_state"); f = put(f, 0, 7, ii[e]); break;
        case PTG_F_HOT_COLD: f = put(f, 3, 1, ii[e] != 0); break;
        case PTG_F_STANDBY_TID: f = put(f, 4, 1, ii[e] == PTG_T_STANDBY_UP); break;
        case PTG_F_STARTUP_TID: f = put(f, 5, 1, ii[e] == PTG_T_STARTUP_HOT); break;
        case PTG_F_PARTIAL_TID: {
            int t = ii[e];
            if (t != PTG_T_OP1_START_P && (t < PTG_T_OP4_P_F_P_5 || t > PTG_T_OP8_F_P)) return set_err(h, PTG_E_INVALID, "bad partial table id");
            f = put(f, 6, 7, t == PTG_T_OP1_START_P ? 0 : t - 7); break;
        }
        case PTG_F_FULL_TID: {
            int t = ii[e];
            if (t != PTG_T_OP2_START_F && t != PTG_T_OP3_P_F && (t < PTG_T_OP9_F_P_F_5 || t > PTG_T_OP12_F_P_F_20)) return set_err(h, PTG_E_INVALID, "bad full table id");
            f = put(f, 9, 7, t == PTG_T_OP2_START_F ? 0 : (t == PTG_T_OP3_P_F ? 1 : t - 11)); break;
        }
        case PTG_F_CURRENT_ACTION: if (ii[e] < 0 || ii[e] > 4) return set_err(h, PTG_E_INVALID, "bad action"); f = put(f, 12, 7, ii[e]); break;
        case PTG_F_MARKET_SET: if (ii[e] < 0 || ii[e] >= h->n_sets) return set_err(h, PTG_E_INVALID, "bad market set"); f = put(f, 15, 3, ii[e]); break;
        case PTG_F_T_CAT: {
            auto it = std::lower_bound(h->Tvals.begin(), h->Tvals.end(), id[e]);
            if (it == h->Tvals.end() || *it != id[e]) return set_err(h, PTG_E_INVALID, "T_cat %.17g of env %d is not a table temperature", id[e], e);
            f = (f & 0x1FFFFu) | ((unsigned)(it - h->Tvals.begin()) << 17); break;
        }
        default: return set_err(h, PTG_E_INVALID, "unknown state field %d", field);
        }
        a[e].flags = f;
    }
    HIP_TRY(h, hipMemcpy(P.st_a, a.data(), sizeof(StA) * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(P.st_b, b.data(), sizeof(StB) * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(P.st_c, c.data(), sizeof(StC) * n, hipMemcpyHostToDevice));
    if (field == PTG_F_K) {                             // step counts set by hand: synchronised again iff they are all equal
        bool same = true;
        for (int e = 1; e < n; e++) same = same && (a[e].k == a[0].k);
        h->sync_k = (same && a[0].k >= 0 && a[0].k <= h->cfg.eps_sim_steps - 6) ? a[0].k : -1;
    }
    return 0;
}

int ptg_finished_episodes(ptg_env* h, double* returns_host, int32_t* lengths_host, int32_t* env_ids_host, int cap, int* count)
{
    if (!h || !count || cap < 0) return set_err(h, PTG_E_INVALID, "bad argument");
    if (!h->fin_maybe) { *count = 0; return 0; }          // only hot launches since the last query: nothing can have finished
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    h->fin_maybe = false;
    unsigned total = 0;
    HIP_TRY(h, hipMemcpy(&total, h->P.fin_count, sizeof(unsigned), hipMemcpyDeviceToHost));
    const int have = (int)std::min<unsigned>(total, (unsigned)h->P.fin_cap);
    const int n = std::min(have, cap);
    h->fin_dropped += (unsigned long long)(total - (unsigned)have) + (unsigned long long)(have - n);      // the list is cleared below either way
    // entries [total - have, total) are live (ring); hand out the oldest n of them: at most two contiguous pieces.  All pieces go
    // through ONE pinned staging block with asynchronous copies and one synchronise (three pageable hipMemcpy of 65 536 entries cost
    // 230-250 us of the episode-boundary window; bench.py's episode_boundary.finished_query_us)
    if (n > 0) {
        const int cap_r = h->P.fin_cap, s0 = (int)((total - (unsigned)have) % (unsigned)cap_r);
        const int n0 = std::min(n, cap_r - s0), n1 = n - n0;
        const size_t need = (size_t)n * (sizeof(double) + 2 * sizeof(int));
        if (need > h->fin_stage_bytes) {
            if (h->fin_stage) (void)hipHostFree(h->fin_stage);
            h->fin_stage = nullptr; h->fin_stage_bytes = 0;
            if (hipHostMalloc(&h->fin_stage, need, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->fin_stage = nullptr; }
            else h->fin_stage_bytes = need;
        }
        char* stage = (char*)h->fin_stage;
        double* s_ret = stage ? (double*)stage : returns_host;
        int* s_len = stage ? (int*)(stage + (size_t)n * sizeof(double)) : lengths_host;
        int* s_env = stage ? s_len + n : env_ids_host;
        auto pull = [&](void* dst, const void* src0, const void* src_wrap, size_t el) -> hipError_t {
            if (!dst) return hipSuccess;
            hipError_t e = hipMemcpyAsync(dst, src0, el * n0, hipMemcpyDeviceToHost, nullptr);
            if (e == hipSuccess && n1) e = hipMemcpyAsync((char*)dst + el * n0, src_wrap, el * n1, hipMemcpyDeviceToHost, nullptr);
            return e;
        };
        if (returns_host) HIP_TRY(h, pull(s_ret, h->P.fin_ret + s0, h->P.fin_ret, sizeof(double)));
        if (lengths_host) HIP_TRY(h, pull(s_len, h->P.fin_len + s0, h->P.fin_len, sizeof(int)));
        if (env_ids_host) HIP_TRY(h, pull(s_env, h->P.fin_env + s0, h->P.fin_env, sizeof(int)));
        HIP_TRY(h, hipStreamSynchronize(nullptr));
        if (stage) {
            if (returns_host) memcpy(returns_host, s_ret, sizeof(double) * n);
            if (lengths_host) memcpy(lengths_host, s_len, sizeof(int) * n);
            if (env_ids_host) memcpy(env_ids_host, s_env, sizeof(int) * n);
        }
    }
    HIP_TRY(h, hipMemset(h->P.fin_count, 0, sizeof(int)));
    *count = n;
    return 0;
}

int ptg_finished_dropped(ptg_env* h, uint64_t* dropped_total)
{
    if (!h || !dropped_total) return set_err(h, PTG_E_INVALID, "bad argument");
    *dropped_total = h->fin_dropped;
    return 0;
}

int ptg_market_feature_series(ptg_env* h, int which, float* out_host, int cap, int* count)
{
    if (!h || !count || which < 0 || which > 3 || cap < 0) return set_err(h, PTG_E_INVALID, "ptg_market_feature_series: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    const bool hourly = which < 2;
    const int n = h->n_sets * (hourly ? h->P.n_hours : h->P.n_days);
    const unsigned off = which == 0 ? 0u : which == 1 ? h->off_featB : which == 2 ? h->off_gasn : h->off_euan;
    *count = n;
    if (out_host) {
        if (cap < n) return set_err(h, PTG_E_INVALID, "ptg_market_feature_series: buffer too small (%d < %d)", cap, n);
        HIP_TRY(h, hipMemcpy(out_host, h->d_pool32 + off, sizeof(float) * n, hipMemcpyDeviceToHost));
    }
    return 0;
}

#ifdef PTG_STAMPS
int ptg_debug_stamps(unsigned long long* out_host) { return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
#endif

int ptg_debug_get_index_lut(ptg_env* h, double* T_values_host, int32_t* lut_host, int* n_T)
{
    if (!h || !n_T) return set_err(h, PTG_E_INVALID, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    *n_T = (int)h->Tvals.size();
    if (T_values_host) memcpy(T_values_host, h->Tvals.data(), sizeof(double) * h->Tvals.size());
    if (lut_host) HIP_TRY(h, hipMemcpy(lut_host, h->P.argidx, sizeof(int) * N_DEST * h->Tvals.size(), hipMemcpyDeviceToHost));
    return 0;
}

int ptg_debug_window_record(ptg_env* h, int table_id, int start_row, double* out7_host)
{
    if (!h || !out7_host || table_id < 0 || table_id >= NT) return set_err(h, PTG_E_INVALID, "bad argument");
    if (start_row < 0 || start_row > h->tab_rows[table_id]) return set_err(h, PTG_E_INVALID, "start_row out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    Rec r;
    HIP_TRY(h, hipMemcpy(&r, h->P.rec + h->rec_base[table_id] + start_row, sizeof(Rec), hipMemcpyDeviceToHost));
    out7_host[0] = r.T;
    for (int c = 0; c < 5; c++) out7_host[1 + c] = r.m[c];
    out7_host[6] = (double)r.tkey;
    return 0;
}

}  // extern "C"

#endif      // part 0
