"""Would letting the producer waves of k_rollout_pc run several steps ahead on the action tape help small batches?
At N = 4 096 a fused step costs 0.86 us = the producers dependent chain: the state machine of step t + 1 needs the temperature key of the
window entered at step t (a 2-byte gather from L2).  The key is only CONSUMED by decisions that depend on the catalyst temperature:
_standby (up / down table + _get_index), _startup (hot / cold table + _get_index), the op2_start_f early branch of _partial; everywhere else
the next window start is i + j * S, known without it.  A wave could therefore issue the gathers of several steps back to back and wait only
when one of its lanes reaches such a decision (vmcnt is per wave).  This script measures, on the CPU oracle with the bench workload
(BS2/OP2, sticky actions p = 1/12, 4 096 envs, steps 200-600 after reset), how often that is.   python tools/runahead_estimate.py"""
import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import ptg_oracle as po
from rl_ptg_amd.prep import synthetic_spec
spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=32)
n, T = 4096, 600
m = spec.markets[0]
consts = dict(spec.consts, scenario=m["scenario"], rew_l_b=m["rew_l_b"], rew_u_b=m["rew_u_b"], r_0=m["r_0"])
env = po.OracleVecEnv(consts, spec.tables, dict(el=m["el"], pot_rew=m["pot_rew"], part_full=m["part_full"], gas=m["gas"], eua=m["eua"], eps_ind=None), n, ep_index0=0)
rng = np.random.default_rng(1)
env.set_noise_tape(rng.normal(0, 10, (n, 512)))
env.reset()
cur = rng.integers(0, 5, n)
need = np.zeros((T, n), bool)
for t in range(T):
    sw = rng.random(n) < 1 / 12.0
    cur = np.where(sw, rng.integers(0, 5, n), cur)
    ints, _ = env.state()
    s = ints[:, 0]; full_tid = ints[:, 7]
    k1 = (cur == 0) & (s != 0); k3 = (cur == 2) & (s <= 1); k4 = (cur == 3) & (s == 4) & (full_tid == 6)   # op2_start_f: the early branch may use _get_index
    need[t] = k1 | k3 | k4
    env.step_reuse(cur.astype(np.int32), n_threads=8)
need = need[200:]          # stationary part
print("per env-step: a key-dependent decision on %.2f %% of the steps" % (100 * need.mean()))
w = need.reshape(need.shape[0], n // 64, 64).any(axis=2)
print("per 64-env wave: some lane needs its key on %.1f %% of the steps" % (100 * w.mean()))
runs = []
for wv in range(w.shape[1]):
    col = w[:, wv]; c = 0
    for x in col:
        if x: runs.append(c); c = 0
        else: c += 1
print("mean run of steps a wave could run ahead without any key: %.2f" % (np.mean(runs) if runs else float('nan')))
