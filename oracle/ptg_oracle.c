/*
 * ptg_oracle.c -- CPU ORACLE (test infrastructure, see ptg_oracle.h).
 *
 * Scalar restatement of /root/reference/env/ptg_gym_env.py.  One ptgo_slot per env keeps exactly the
 * attributes the reference env object keeps; every function names the reference lines it follows.
 * Floating-point expressions keep the reference's operand order and use libm pow/sin/cos exactly where
 * CPython's float arithmetic calls them, and column means use NumPy's pairwise summation, so that on the
 * same host this file reproduces the reference bit for bit (checked by tests/test_oracle_golden.py).
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: no FMA contraction).
 */
#include "ptg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NT PTGO_N_TABLES
#define NC PTGO_N_COLS

static __thread char g_err[256];
const char* ptgo_last_error(void) { return g_err; }
#define FAIL(code, ...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return (code); } while (0)

typedef struct ptgo_slot {
    /* env/ptg_gym_env.py:105-138 (_initialize_op_rew) attributes */
    int meth_state, i, j, k, hot_cold;
    int standby_tid, startup_tid, partial_tid, full_tid;   /* self.standby / startup / partial(part_op) / full(full_op) */
    int current_action;                                    /* index into self.actions (:142) */
    int state_change;
    int64_t act_ep_h, act_ep_d;
    double T_cat, H2, CH4, H2_res, H2O, el_heating;
    double cum_rew, rew, eta;
    double clock_hours, sin_h, cos_h;
    int64_t h_idx, d_idx;                                  /* current e_r_b / g_e column */
    /* reward constituents kept for _get_info (:251-278) */
    double ch4_revenues, steam_revenues, o2_revenues, eua_revenues, chp_revenues,
           elec_costs_heating, elec_costs_electrolyzer, water_costs;
    /* normalised observations (:206-217) */
    double T_n, H2_n, CH4_n, H2_res_n, H2O_n, heat_n;
    int64_t noise_count;
    /* snapshot at return of the last step() (before DummyVecEnv's auto-reset) */
    int64_t last_int[PTGO_N_INT];
    double last_f64[PTGO_N_F64];
} ptgo_slot;

struct ptgo_env {
    ptgo_config c;
    int n;
    int S;                         /* step_size = int(sim_step / time_step_op) (:66) */
    double* tab[NT];
    int rows[NT];
    int n_hours, n_days, n_eps_ind;
    double *el, *pot, *pf, *gas, *eua, *eps_ind;
    double prob_thre[6];           /* :153-155 */
    double b_s3;                   /* :76-77 */
    int64_t ep_index;              /* module-global ep_index (:9) */
    double* tape; int tape_len;
    ptgo_slot* s;
    double* win;                   /* scratch [threads][S][7] */
    int n_win;
};

/* math.sin / math.cos call libm's sin and cos separately; keep gcc from fusing the pair into sincos(),
 * whose result can differ from sin() in the last bit. */
static __attribute__((noinline)) double py_sin(double x) { return sin(x); }
static __attribute__((noinline)) double py_cos(double x) { return cos(x); }

/* ---- NumPy pairwise summation (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum; the inner
 * loop np.add.reduce runs for a strided 1-D double array).  np.average(x) = (0.0 + pairwise(x, n)) / n. */
static double pairwise_sum(const double* a, int64_t n, int64_t stride)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i * stride];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int64_t i;
        for (int q = 0; q < 8; q++) r[q] = a[q * stride];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int q = 0; q < 8; q++) r[q] += a[(i + q) * stride];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i * stride];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2, stride) + pairwise_sum(a + n2 * stride, n - n2, stride);
    }
}

double ptgo_pairwise_mean(const double* a, int64_t n, int64_t stride_elems)
{
    double s = 0.0 + pairwise_sum(a, n, stride_elems);
    return s / (double)n;
}

/* env/ptg_gym_env.py:514-523 (_get_index): first index of min |T_col - t_cat| */
static int get_index(const ptgo_env* h, int tid, double t_cat)
{
    const double* t = h->tab[tid];
    int n = h->rows[tid], best = 0;
    double bd = fabs(t[1] - t_cat);
    for (int r = 1; r < n; r++) {
        double d = fabs(t[r * NC + 1] - t_cat);
        if (d < bd) { bd = d; best = r; }
    }
    return best;
}
int32_t ptgo_get_index(const ptgo_env* h, int table_id, double t_cat) { return get_index(h, table_id, t_cat); }

/* :351-355 -- first threshold greater than the (float32) action picks actions[ival-1]; python index -1 wraps */
int32_t ptgo_decode_continuous(const ptgo_env* h, float a, int32_t previous)
{
    double ad = (double)a;
    for (int ival = 0; ival < 6; ival++) {
        if (h->prob_thre[ival] > ad) {
            int idx = ival - 1;
            if (idx < 0) idx += 5;
            return idx;
        }
    }
    return previous;
}

/* :584-585 / :598-599 / :620-621 -- int(max(idx + normal(0, noise), 0)) */
static int noisy_index(ptgo_env* h, int e, int idx)
{
    ptgo_slot* s = &h->s[e];
    double z = 0.0;
    if (h->tape) {
        if (s->noise_count < h->tape_len) z = h->tape[(int64_t)e * h->tape_len + s->noise_count];
        else z = h->tape[(int64_t)e * h->tape_len + (s->noise_count % h->tape_len)];
    }
    s->noise_count++;
    double x = (double)idx + z;
    if (0 > x) x = 0;            /* python max(x, 0) */
    return (int)x;               /* python int(): truncation */
}

/* :525-557 (_perform_sim_step).  Writes the S-row window into win, returns r_state and may update *idx,*j */
static int perform_sim_step(const ptgo_env* h, double* win, int op_tid, int initial_state, int next_tid,
                            int next_state, int* idx, int* j, int change_operation)
{
    const double* op = h->tab[op_tid];
    const int S = h->S;
    int64_t total = h->rows[op_tid];
    int64_t end = (int64_t)*idx + (int64_t)*j * S;
    int r_state;
    if (end < total) {
        r_state = initial_state;
        int64_t start = (int64_t)*idx + (int64_t)(*j - 1) * S;
        memcpy(win, op + start * NC, sizeof(double) * NC * S);
    } else {
        r_state = next_state;
        int64_t over = end - total;
        if (over < S) {
            int64_t start = (int64_t)*idx + (int64_t)(*j - 1) * S;
            int64_t nhead = total - start;
            memcpy(win, op + start * NC, sizeof(double) * NC * nhead);
            if (change_operation) {
                *idx = (int)over;
                *j = 0;
                memcpy(win + nhead * NC, h->tab[next_tid], sizeof(double) * NC * over);
            } else {
                for (int64_t q = 0; q < over; q++)
                    for (int c = 0; c < NC; c++) win[(nhead + q) * NC + c] = 1.0 * op[(total - 1) * NC + c];
            }
        } else {
            for (int64_t q = 0; q < S; q++)
                for (int c = 0; c < NC; c++) win[q * NC + c] = 1.0 * op[(total - 1) * NC + c];
        }
    }
    return r_state;
}

/* :559-570 (_cont) */
static int cont(ptgo_env* h, int e, double* win, int op_tid, int next_tid, int next_state, int change)
{
    ptgo_slot* s = &h->s[e];
    s->j += 1;
    return perform_sim_step(h, win, op_tid, s->meth_state, next_tid, next_state, &s->i, &s->j, change);
}

/* :572-589 */
static int do_standby(ptgo_env* h, int e, double* win)
{
    ptgo_slot* s = &h->s[e];
    s->meth_state = 0;
    s->standby_tid = (s->T_cat <= h->c.t_cat_standby) ? PTGO_T_STANDBY_UP : PTGO_T_STANDBY_DOWN;
    s->i = noisy_index(h, e, get_index(h, s->standby_tid, s->T_cat));
    s->j = 1;
    return perform_sim_step(h, win, s->standby_tid, s->meth_state, s->standby_tid, s->meth_state, &s->i, &s->j, 0);
}

/* :591-603 */
static int do_cooldown(ptgo_env* h, int e, double* win)
{
    ptgo_slot* s = &h->s[e];
    s->meth_state = 1;
    s->i = noisy_index(h, e, get_index(h, PTGO_T_COOLDOWN, s->T_cat));
    s->j = 1;
    return perform_sim_step(h, win, PTGO_T_COOLDOWN, s->meth_state, PTGO_T_COOLDOWN, s->meth_state, &s->i, &s->j, 0);
}

/* :605-625 */
static int do_startup(ptgo_env* h, int e, double* win)
{
    ptgo_slot* s = &h->s[e];
    s->meth_state = 2;
    s->partial_tid = PTGO_T_OP1_START_P;
    s->full_tid = PTGO_T_OP2_START_F;
    s->startup_tid = (s->hot_cold == 0) ? PTGO_T_STARTUP_COLD : PTGO_T_STARTUP_HOT;
    s->i = noisy_index(h, e, get_index(h, s->startup_tid, s->T_cat));
    s->j = 1;
    return perform_sim_step(h, win, s->startup_tid, s->meth_state, s->partial_tid, 3, &s->i, &s->j, 1);
}

/* :627-691 */
static int do_partial(ptgo_env* h, int e, double* win)
{
    ptgo_slot* s = &h->s[e];
    const ptgo_config* c = &h->c;
    s->meth_state = 3;
    int64_t time_op = (int64_t)s->i + (int64_t)s->j * h->S;
    if (s->full_tid == PTGO_T_OP2_START_F) {
        if (time_op < c->time2_start_f_p) {
            s->partial_tid = PTGO_T_OP1_START_P;
            s->i = get_index(h, s->partial_tid, s->T_cat);
            s->j = 1;
        } else {
            s->partial_tid = PTGO_T_OP8_F_P; s->i = 0; s->j = 1;
        }
    } else if (s->full_tid == PTGO_T_OP3_P_F) {
        if (time_op < c->time1_p_f_p) {
            s->partial_tid = PTGO_T_OP8_F_P;
            s->i = c->i_fully_developed;
            s->j = c->j_fully_developed;
            s->T_cat = h->tab[PTGO_T_OP8_F_P][(h->rows[PTGO_T_OP8_F_P] - 1) * NC + 1];
        } else if (c->time1_p_f_p < time_op && time_op < c->time2_p_f_p) {
            s->partial_tid = PTGO_T_OP4_P_F_P_5; s->j += 1;
        } else if (c->time2_p_f_p < time_op && time_op < c->time_p_f) {
            s->partial_tid = PTGO_T_OP4_P_F_P_5; s->i = c->time2_p_f_p; s->j = 1;
        } else if (c->time_p_f < time_op && time_op < c->time34_p_f_p) {
            s->partial_tid = PTGO_T_OP5_P_F_P_10; s->i = c->time3_p_f_p; s->j = 1;
        } else if (c->time34_p_f_p < time_op && time_op < c->time45_p_f_p) {
            s->partial_tid = PTGO_T_OP6_P_F_P_15; s->i = c->time4_p_f_p; s->j = 1;
        } else if (c->time45_p_f_p < time_op && time_op < c->time5_p_f_p) {
            s->partial_tid = PTGO_T_OP7_P_F_P_22; s->i = c->time5_p_f_p; s->j = 1;
        } else {
            s->partial_tid = PTGO_T_OP8_F_P; s->i = 0; s->j = 1;
        }
    } else {
        s->partial_tid = PTGO_T_OP8_F_P; s->i = 0; s->j = 1;
    }
    return perform_sim_step(h, win, s->partial_tid, s->meth_state, s->partial_tid, 3, &s->i, &s->j, 0);
}

/* :693-757 */
static int do_full(ptgo_env* h, int e, double* win)
{
    ptgo_slot* s = &h->s[e];
    const ptgo_config* c = &h->c;
    s->meth_state = 4;
    int64_t time_op = (int64_t)s->i + (int64_t)s->j * h->S;
    if (s->partial_tid == PTGO_T_OP1_START_P) {
        if (time_op < c->time1_start_p_f) { s->full_tid = PTGO_T_OP2_START_F; s->i = 0; s->j = 1; }
        else { s->full_tid = PTGO_T_OP3_P_F; s->i = 0; s->j = 1; }
    } else if (s->partial_tid == PTGO_T_OP8_F_P) {
        if (time_op < c->time1_f_p_f) {
            s->full_tid = PTGO_T_OP3_P_F;
            s->i = c->i_fully_developed;
            s->j = c->j_fully_developed;
            s->T_cat = h->tab[PTGO_T_OP3_P_F][(h->rows[PTGO_T_OP3_P_F] - 1) * NC + 1];
        } else if (c->time1_f_p_f < time_op && time_op < c->time_f_p) {
            s->full_tid = PTGO_T_OP9_F_P_F_5; s->j += 1;
        } else if (c->time_f_p < time_op && time_op < c->time23_f_p_f) {
            s->full_tid = PTGO_T_OP9_F_P_F_5; s->i = c->time2_f_p_f; s->j = 1;
        } else if (c->time23_f_p_f < time_op && time_op < c->time34_f_p_f) {
            s->full_tid = PTGO_T_OP10_F_P_F_10; s->i = c->time3_f_p_f; s->j = 1;
        } else if (c->time34_f_p_f < time_op && time_op < c->time45_f_p_f) {
            s->full_tid = PTGO_T_OP11_F_P_F_15; s->i = c->time4_f_p_f; s->j = 1;
        } else if (c->time45_f_p_f < time_op && time_op < c->time5_f_p_f) {
            s->full_tid = PTGO_T_OP12_F_P_F_20; s->i = c->time5_f_p_f; s->j = 1;
        } else {
            s->full_tid = PTGO_T_OP3_P_F; s->i = 0; s->j = 1;
        }
    } else {
        s->full_tid = PTGO_T_OP3_P_F; s->i = 0; s->j = 1;
    }
    return perform_sim_step(h, win, s->full_tid, s->meth_state, s->full_tid, 4, &s->i, &s->j, 0);
}

/* :206-217 (scalar members; the 13/2-wide price members are produced on the fly in write_obs) */
static void normalize(const ptgo_env* h, ptgo_slot* s)
{
    const ptgo_config* c = &h->c;
    s->T_n = (s->T_cat - c->T_l_b) / (c->T_u_b - c->T_l_b);
    s->H2_n = (s->H2 - c->h2_l_b) / (c->h2_u_b - c->h2_l_b);
    s->CH4_n = (s->CH4 - c->ch4_l_b) / (c->ch4_u_b - c->ch4_l_b);
    s->H2_res_n = (s->H2_res - c->h2_res_l_b) / (c->h2_res_u_b - c->h2_res_l_b);
    s->H2O_n = (s->H2O - c->h2o_l_b) / (c->h2o_u_b - c->h2o_l_b);
    s->heat_n = (s->el_heating - c->heat_l_b) / (c->heat_u_b - c->heat_l_b);
}

/* :219-249 (_get_obs) flattened in the dict's insertion order */
static void write_obs(const ptgo_env* h, const ptgo_slot* s, double* o)
{
    const ptgo_config* c = &h->c;
    const int P = c->price_ahead;
    int q = 0;
    if (c->raw_modified == 0) {
        for (int i = 0; i < P; i++) o[q++] = (h->el[s->h_idx + i] - c->el_l_b) / (c->el_u_b - c->el_l_b);
        for (int i = 0; i < 2; i++) o[q++] = (h->gas[s->d_idx + i] - c->gas_l_b) / (c->gas_u_b - c->gas_l_b);
        for (int i = 0; i < 2; i++) o[q++] = (h->eua[s->d_idx + i] - c->eua_l_b) / (c->eua_u_b - c->eua_l_b);
    } else {
        for (int i = 0; i < P; i++) o[q++] = (h->pot[s->h_idx + i] - c->rew_l_b) / (c->rew_u_b - c->rew_l_b);
        for (int i = 0; i < P; i++) o[q++] = h->pf[s->h_idx + i];
    }
    o[q++] = (double)s->meth_state;
    o[q++] = s->T_n; o[q++] = s->H2_n; o[q++] = s->CH4_n; o[q++] = s->H2_res_n; o[q++] = s->H2O_n;
    o[q++] = s->heat_n; o[q++] = s->sin_h; o[q++] = s->cos_h;
}

/* :251-278 (_get_info) in key order; Meth_Action as its index in self.actions */
static void write_info(const ptgo_env* h, const ptgo_slot* s, double* v)
{
    v[0] = (double)s->k;
    v[1] = h->el[s->h_idx];
    v[2] = h->gas[s->d_idx];
    v[3] = h->eua[s->d_idx];
    v[4] = (double)s->meth_state;
    v[5] = (double)s->current_action;
    v[6] = (double)s->hot_cold;
    v[7] = s->T_cat; v[8] = s->H2; v[9] = s->CH4; v[10] = s->H2O; v[11] = s->el_heating;
    v[12] = s->ch4_revenues; v[13] = s->steam_revenues; v[14] = s->o2_revenues; v[15] = s->eua_revenues;
    v[16] = s->chp_revenues; v[17] = -s->elec_costs_heating; v[18] = -s->elec_costs_electrolyzer;
    v[19] = -s->water_costs; v[20] = s->rew; v[21] = s->cum_rew;
    v[22] = h->pot[s->h_idx]; v[23] = h->pf[s->h_idx];
}

/* :280-334 (_get_reward), operand order as written there */
static double get_reward(const ptgo_env* h, ptgo_slot* s)
{
    const ptgo_config* c = &h->c;
    const double gas = h->gas[s->d_idx], eua = h->eua[s->d_idx], el = h->el[s->h_idx];
    double ch4_volumeflow = s->CH4 * c->convert_mol_to_Nm3;
    double h2_res_volumeflow = s->H2_res * c->convert_mol_to_Nm3;
    double Q_ch4 = ch4_volumeflow * c->H_u_CH4 * 1000;
    double Q_h2_res = h2_res_volumeflow * c->H_u_H2 * 1000;
    s->ch4_revenues = (Q_ch4 + Q_h2_res) * gas;
    double power_chp = Q_ch4 * c->eta_CHP * h->b_s3;
    double Q_chp = Q_ch4 * (1 - c->eta_CHP) * h->b_s3;
    s->chp_revenues = power_chp * c->eeg_el_price;
    double Q_steam = s->H2O * (c->dt_water * c->cp_water + c->h_H2O_evap) / 3600;
    s->steam_revenues = (Q_steam + Q_chp) * c->heat_price;
    double h2_volumeflow = s->H2 * c->convert_mol_to_Nm3;
    double o2_volumeflow = 1.0 / 2 * h2_volumeflow * 3600;
    s->o2_revenues = o2_volumeflow * c->o2_price;
    double co2_mass_flow = s->CH4 * c->Molar_mass_CO2 / 1000;
    s->eua_revenues = co2_mass_flow / 1000 * 3600 * eua * 100;
    s->elec_costs_heating = s->el_heating / 1000 * el;
    double load = h2_volumeflow / c->max_h2_volumeflow;
    if (load < c->min_load_electrolyzer) {
        s->eta = 0.02;
    } else {
        /* python: 0.598 - 0.325*l**2 + 0.218*l**3 + 0.01*l**(-1) - 1.68*10**(-3)*l**(-2) + 2.51*10**(-5)*l**(-3) */
        s->eta = (0.598 - 0.325 * pow(load, 2.0) + 0.218 * pow(load, 3.0) + 0.01 * pow(load, -1.0)
                  - 1.68 * pow(10.0, -3.0) * pow(load, -2.0) + 2.51 * pow(10.0, -5.0) * pow(load, -3.0));
    }
    s->elec_costs_electrolyzer = h2_volumeflow * c->H_u_H2 * 1000 / s->eta * el;
    double elec_costs = s->elec_costs_heating + s->elec_costs_electrolyzer;
    double water_elec = s->H2 * c->Molar_mass_H2O / 1000 * 3600;
    s->water_costs = (s->H2O + water_elec) / c->rho_water * c->water_price;
    s->rew = (s->ch4_revenues + s->chp_revenues + s->steam_revenues + s->eua_revenues + s->o2_revenues
              - elec_costs - s->water_costs) * c->sim_step / 3600;
    s->cum_rew += s->rew;
    if (s->state_change) s->rew -= c->r_0 * c->state_change_penalty;
    return s->rew;
}

static void snapshot(ptgo_slot* s, double reward)
{
    int64_t* a = s->last_int;
    a[0] = s->meth_state; a[1] = s->i; a[2] = s->j; a[3] = s->hot_cold; a[4] = s->standby_tid;
    a[5] = s->startup_tid; a[6] = s->partial_tid; a[7] = s->full_tid; a[8] = s->k; a[9] = s->current_action;
    a[10] = s->act_ep_h; a[11] = s->act_ep_d;
    double* f = s->last_f64;
    f[0] = reward; f[1] = s->cum_rew; f[2] = s->T_cat; f[3] = s->H2; f[4] = s->CH4; f[5] = s->H2_res;
    f[6] = s->H2O; f[7] = s->el_heating;
}

/* :59-64 / :490-495 -- take the next eps_ind entry from the shared counter */
static int take_episode(ptgo_env* h, ptgo_slot* s)
{
    if (h->n_eps_ind > 0) {
        if (h->ep_index < 0 || h->ep_index >= h->n_eps_ind)
            FAIL(-2, "eps_ind exhausted (ep_index=%lld, len=%d): the reference raises IndexError here",
                 (long long)h->ep_index, h->n_eps_ind);
        double v = h->eps_ind[h->ep_index];
        s->act_ep_h = (int64_t)(v * h->c.eps_len_d * 24);
        s->act_ep_d = (int64_t)(v * h->c.eps_len_d);
        h->ep_index += 1;
    } else {
        s->act_ep_h = 0; s->act_ep_d = 0;
    }
    return 0;
}

/* :483-506 (reset) = :81-103 (_initialize_datasets) + :105-138 (_initialize_op_rew) + :206-217 */
static int reset_one(ptgo_env* h, int e, double* obs_row, double* info_row)
{
    ptgo_slot* s = &h->s[e];
    int rc = take_episode(h, s);
    if (rc) return rc;
    s->clock_hours = 0.0 * h->c.sim_step / 3600;
    s->h_idx = s->act_ep_h; s->d_idx = s->act_ep_d;
    s->sin_h = py_sin(2 * M_PI * s->clock_hours);
    s->cos_h = py_cos(2 * M_PI * s->clock_hours);
    s->meth_state = 1;
    s->standby_tid = PTGO_T_STANDBY_DOWN; s->startup_tid = PTGO_T_STARTUP_COLD;
    s->partial_tid = PTGO_T_OP1_START_P; s->full_tid = PTGO_T_OP2_START_F;
    s->T_cat = 16;
    s->i = get_index(h, PTGO_T_COOLDOWN, s->T_cat);
    s->j = 0;
    const double* op = h->tab[PTGO_T_COOLDOWN] + (int64_t)s->i * NC;
    s->H2 = op[2]; s->CH4 = op[3]; s->H2_res = op[4]; s->H2O = op[5]; s->el_heating = op[6];
    s->hot_cold = 0; s->state_change = 0;
    s->ch4_revenues = s->steam_revenues = s->o2_revenues = s->eua_revenues = s->chp_revenues = 0.0;
    s->elec_costs_heating = s->elec_costs_electrolyzer = s->water_costs = 0.0;
    s->rew = 0.0; s->eta = 0.02; s->cum_rew = 0; s->k = 0;
    normalize(h, s);
    if (obs_row) write_obs(h, s, obs_row);
    if (info_row) write_info(h, s, info_row);
    return 0;
}

/* :336-481 (step) for one env, without the wrapper's auto-reset. Returns terminated (0/1) or <0 */
static int step_one(ptgo_env* h, int e, const void* actions, double* win, double* obs_row, double* rew,
                    double* info_row)
{
    ptgo_slot* s = &h->s[e];
    const ptgo_config* c = &h->c;
    int k = s->k;
    /* :339-342 */
    if (s->T_cat <= c->t_cat_startup_cold) s->hot_cold = 0;
    else if (s->T_cat >= c->t_cat_startup_hot) s->hot_cold = 1;
    int previous_state = s->meth_state;
    /* :346-357 */
    if (c->action_type == 0) {
        int a = ((const int32_t*)actions)[e];
        if (a < -5 || a > 4) FAIL(-3, "invalid discrete action %d for env %d (reference: IndexError)", a, e);
        if (a < 0) a += 5;
        s->current_action = a;
    } else {
        s->current_action = ptgo_decode_continuous(h, ((const float*)actions)[e], s->current_action);
    }
    int st = s->meth_state, r_state;
    /* :368-440 */
    switch (s->current_action) {
    case 0:
        if (st == 0) r_state = cont(h, e, win, s->standby_tid, s->standby_tid, s->meth_state, 0);
        else r_state = do_standby(h, e, win);
        break;
    case 1:
        if (st == 1) r_state = cont(h, e, win, PTGO_T_COOLDOWN, PTGO_T_COOLDOWN, s->meth_state, 0);
        else r_state = do_cooldown(h, e, win);
        break;
    case 2:
        if (st == 2) r_state = cont(h, e, win, s->startup_tid, s->partial_tid, 3, 1);
        else if (st == 3) r_state = cont(h, e, win, s->partial_tid, s->partial_tid, 3, 0);
        else if (st == 4) r_state = cont(h, e, win, s->full_tid, s->full_tid, 4, 0);
        else r_state = do_startup(h, e, win);
        break;
    case 3:
        if (st == 0) r_state = cont(h, e, win, s->standby_tid, s->standby_tid, s->meth_state, 0);
        else if (st == 1) r_state = cont(h, e, win, PTGO_T_COOLDOWN, PTGO_T_COOLDOWN, s->meth_state, 0);
        else if (st == 2) r_state = cont(h, e, win, s->startup_tid, s->partial_tid, 3, 1);
        else if (st == 3) r_state = cont(h, e, win, s->partial_tid, s->partial_tid, 3, 0);
        else r_state = do_partial(h, e, win);
        break;
    default:
        if (st == 0) r_state = cont(h, e, win, s->standby_tid, s->standby_tid, s->meth_state, 0);
        else if (st == 1) r_state = cont(h, e, win, PTGO_T_COOLDOWN, PTGO_T_COOLDOWN, s->meth_state, 0);
        else if (st == 2) r_state = cont(h, e, win, s->startup_tid, s->partial_tid, 3, 1);
        else if (st == 4) r_state = cont(h, e, win, s->full_tid, s->full_tid, 4, 0);
        else r_state = do_full(h, e, win);
        break;
    }
    s->meth_state = r_state;
    /* :442-450 */
    s->clock_hours = (double)((int64_t)(k + 1) * c->sim_step) / 3600;
    double clock_days = s->clock_hours / 24;
    int64_t h_step = (int64_t)floor(s->clock_hours), d_step = (int64_t)floor(clock_days);
    s->h_idx = s->act_ep_h + h_step;
    s->d_idx = s->act_ep_d + d_step;
    if (s->h_idx + c->price_ahead > h->n_hours || s->d_idx + 2 > h->n_days)
        FAIL(-4, "price index out of range (env %d, h=%lld, d=%lld)", e, (long long)s->h_idx, (long long)s->d_idx);
    s->sin_h = py_sin(2 * M_PI * s->clock_hours);
    s->cos_h = py_cos(2 * M_PI * s->clock_hours);
    /* :452-458 */
    const int S = h->S;
    s->T_cat = win[(S - 1) * NC + 1];
    s->H2 = ptgo_pairwise_mean(win + 2, S, NC);
    s->CH4 = ptgo_pairwise_mean(win + 3, S, NC);
    s->H2_res = ptgo_pairwise_mean(win + 4, S, NC);
    s->H2O = ptgo_pairwise_mean(win + 5, S, NC);
    s->el_heating = ptgo_pairwise_mean(win + 6, S, NC);
    normalize(h, s);                                     /* :460 */
    s->state_change = (previous_state != s->meth_state); /* :463-466 */
    *rew = get_reward(h, s);                             /* :468 */
    write_obs(h, s, obs_row);                            /* :469 */
    int terminated = (s->k == c->eps_sim_steps - 6);     /* :470, :508-511 */
    if (info_row) write_info(h, s, info_row);            /* :471-474 */
    s->k += 1;                                           /* :476 */
    snapshot(s, *rew);
    return terminated;
}

int ptgo_obs_dim(const ptgo_env* h) { return h->c.raw_modified == 0 ? h->c.price_ahead + 4 + 9 : 2 * h->c.price_ahead + 9; }

static double* dup_d(const double* p, size_t n)
{
    double* q = (double*)malloc(sizeof(double) * (n ? n : 1));
    if (q && p && n) memcpy(q, p, sizeof(double) * n);
    return q;
}

int ptgo_create(const ptgo_config* cfg, const ptgo_tables* tab, const ptgo_market* mkt, int n_envs,
                int64_t ep_index0, ptgo_env** out)
{
    if (!cfg || !tab || !mkt || !out || n_envs <= 0) FAIL(-1, "ptgo_create: bad arguments");
    ptgo_env* h = (ptgo_env*)calloc(1, sizeof *h);
    h->c = *cfg;
    h->n = n_envs;
    h->S = (int)((double)cfg->sim_step / (double)cfg->time_step_op);
    if (h->S <= 0) FAIL(-1, "step_size must be positive");
    for (int t = 0; t < NT; t++) {
        h->rows[t] = tab->rows[t];
        if (tab->rows[t] < 1) FAIL(-1, "table %d is empty", t);
        h->tab[t] = dup_d(tab->data[t], (size_t)tab->rows[t] * NC);
    }
    if (h->rows[PTGO_T_OP1_START_P] < h->S) FAIL(-1, "op1_start_p shorter than one step");
    h->n_hours = mkt->n_hours; h->n_days = mkt->n_days; h->n_eps_ind = mkt->n_eps_ind;
    h->el = dup_d(mkt->el, mkt->n_hours); h->pot = dup_d(mkt->pot_rew, mkt->n_hours);
    h->pf = dup_d(mkt->part_full, mkt->n_hours);
    h->gas = dup_d(mkt->gas, mkt->n_days); h->eua = dup_d(mkt->eua, mkt->n_days);
    h->eps_ind = dup_d(mkt->eps_ind, mkt->n_eps_ind);
    /* :147-155 */
    double act_b0 = -1, act_b1 = 1;
    double prob_ival = (act_b1 - act_b0) / 5;
    for (int ival = 0; ival < 6; ival++) h->prob_thre[ival] = act_b0 + ival * prob_ival;
    h->b_s3 = (cfg->scenario == 3) ? 1 : 0;
    h->ep_index = ep_index0;
    h->s = (ptgo_slot*)calloc(n_envs, sizeof(ptgo_slot));
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    h->n_win = nth;
    h->win = (double*)malloc(sizeof(double) * (size_t)nth * h->S * NC);
    /* __init__ (:28-79): every constructed env consumes one eps_ind entry and starts with current_action = cooldown */
    for (int e = 0; e < n_envs; e++) {
        int rc = take_episode(h, &h->s[e]);
        if (rc) { ptgo_destroy(h); return rc; }
        rc = 0;
        int64_t keep = h->ep_index;
        /* _initialize_op_rew etc. without consuming another episode */
        h->ep_index = keep - (h->n_eps_ind > 0 ? 1 : 0);
        rc = reset_one(h, e, NULL, NULL);
        h->ep_index = keep;
        if (rc) { ptgo_destroy(h); return rc; }
        h->s[e].current_action = 1;     /* :143 */
    }
    *out = h;
    return 0;
}

void ptgo_destroy(ptgo_env* h)
{
    if (!h) return;
    for (int t = 0; t < NT; t++) free(h->tab[t]);
    free(h->el); free(h->pot); free(h->pf); free(h->gas); free(h->eua); free(h->eps_ind);
    free(h->tape); free(h->s); free(h->win); free(h);
}

int ptgo_set_noise_tape(ptgo_env* h, const double* tape, int per_env_len)
{
    free(h->tape); h->tape = NULL; h->tape_len = 0;
    if (tape && per_env_len > 0) {
        h->tape = dup_d(tape, (size_t)h->n * per_env_len);
        h->tape_len = per_env_len;
    }
    for (int e = 0; e < h->n; e++) h->s[e].noise_count = 0;
    return 0;
}

int ptgo_reset(ptgo_env* h, int e, double* obs_out, double* info_out)
{
    const int F = ptgo_obs_dim(h);
    int lo = e < 0 ? 0 : e, hi = e < 0 ? h->n : e + 1;
    if (hi > h->n) FAIL(-1, "env index out of range");
    for (int q = lo; q < hi; q++) {
        int rc = reset_one(h, q, obs_out ? obs_out + (size_t)q * F : NULL,
                           info_out ? info_out + (size_t)q * PTGO_N_INFO : NULL);
        if (rc) return rc;
    }
    return 0;
}

int ptgo_step(ptgo_env* h, const void* actions, double* obs_out, double* rew_out, uint8_t* done_out,
              double* final_obs, double* info_out)
{
    const int F = ptgo_obs_dim(h);
    for (int e = 0; e < h->n; e++) {
        int t = step_one(h, e, actions, h->win, obs_out + (size_t)e * F, rew_out + e,
                         info_out ? info_out + (size_t)e * PTGO_N_INFO : NULL);
        if (t < 0) return t;
        done_out[e] = (uint8_t)t;
        if (t) {
            if (final_obs) memcpy(final_obs + (size_t)e * F, obs_out + (size_t)e * F, sizeof(double) * F);
            int rc = reset_one(h, e, obs_out + (size_t)e * F, NULL);
            if (rc) return rc;
        }
    }
    return 0;
}

int ptgo_step_mt(ptgo_env* h, const void* actions, double* obs_out, double* rew_out, uint8_t* done_out,
                 double* final_obs, double* info_out, int n_threads)
{
    const int F = ptgo_obs_dim(h);
    int err = 0;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > h->n_win) n_threads = h->n_win;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
    for (int e = 0; e < h->n; e++) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        int t = step_one(h, e, actions, h->win + (size_t)tid * h->S * NC, obs_out + (size_t)e * F, rew_out + e,
                         info_out ? info_out + (size_t)e * PTGO_N_INFO : NULL);
        if (t < 0) { err = t; t = 0; }
        done_out[e] = (uint8_t)t;
    }
    if (err) FAIL(err, "ptgo_step_mt: a step failed");
    for (int e = 0; e < h->n; e++) {
        if (done_out[e]) {
            if (final_obs) memcpy(final_obs + (size_t)e * F, obs_out + (size_t)e * F, sizeof(double) * F);
            int rc = reset_one(h, e, obs_out + (size_t)e * F, NULL);
            if (rc) return rc;
        }
    }
    return 0;
}

int ptgo_get_last(const ptgo_env* h, int64_t* ints, double* f64s)
{
    for (int e = 0; e < h->n; e++) {
        if (ints) memcpy(ints + (size_t)e * PTGO_N_INT, h->s[e].last_int, sizeof(int64_t) * PTGO_N_INT);
        if (f64s) memcpy(f64s + (size_t)e * PTGO_N_F64, h->s[e].last_f64, sizeof(double) * PTGO_N_F64);
    }
    return 0;
}

int ptgo_get_state(const ptgo_env* h, int64_t* ints, double* f64s)
{
    for (int e = 0; e < h->n; e++) {
        ptgo_slot tmp = h->s[e];
        snapshot(&tmp, tmp.rew);
        if (ints) memcpy(ints + (size_t)e * PTGO_N_INT, tmp.last_int, sizeof(int64_t) * PTGO_N_INT);
        if (f64s) memcpy(f64s + (size_t)e * PTGO_N_F64, tmp.last_f64, sizeof(double) * PTGO_N_F64);
    }
    return 0;
}

int64_t ptgo_ep_index(const ptgo_env* h) { return h->ep_index; }
int64_t ptgo_noise_count(const ptgo_env* h, int e) { return (e >= 0 && e < h->n) ? h->s[e].noise_count : -1; }
