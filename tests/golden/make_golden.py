#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The reference is imported behind the stand-in `gymnasium` of oracle/refharness (gymnasium / SB3 are not
installable here; nothing was denied).  Outputs are data only: inputs (price series, action tapes, recorded
normal draws, constants) and the reference's outputs (integer state, float state, observations, info rows).

Files written
  market_real.npz            the reference's real price series as its loader returns them (3 splits)
  tables_OP{1,2}.npz  ->     written to rl_ptg_amd/data/ (the 17 process tables per load level; product input)
  prep_<mkt>_bs<k>_<op>.npz  per business scenario / load level: series the env sees, pot_rew / part_full,
                             bounds, r_level, T-OPT totals  (reference: load_data + Preprocessing)
  traj_<case>.npz            trajectories of n reference envs stepped in DummyVecEnv order
  units_<op>.npz             _get_index for every distinct T_cat x 6 destination tables
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle", "refharness"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import ref_driver as rd                      # noqa: E402
import ptg_oracle as po                      # noqa: E402
from rl_ptg_amd.synthetic import synthetic_market_csv_units  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "rl_ptg_amd", "data")


def jsonable(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, (np.floating, np.integer)):
            v = v.item()
        out[k] = v
    return out


def synth_markets():
    return {"train": synthetic_market_csv_units(38, 20250614),
            "val": synthetic_market_csv_units(9, 20250615),
            "test": synthetic_market_csv_units(9, 20250616)}


def sticky_tape(rng, K, n, p=1 / 12.0):
    a = np.zeros((K, n), np.int64)
    cur = rng.integers(0, 5, n)
    for t in range(K):
        sw = rng.random(n) < p
        cur = np.where(sw, rng.integers(0, 5, n), cur)
        a[t] = cur
    return a


def toggler_tape(rng, K, n, max_hold=25, warm=None):
    """startup until production, then partial/full toggling with short random holds (+ rare detours)."""
    a = np.zeros((K, n), np.int64)
    for e in range(n):
        t = 0
        w = warm if warm is not None else int(rng.integers(150, 400))
        while t < K:
            a[t:t + w, e] = 2
            t += w
            seg_end = min(K, t + int(rng.integers(300, 900)))
            cur = 3
            first = True
            while t < seg_end:
                hold = int(rng.integers(1, max_hold + 1))
                if first and rng.random() < 0.5:
                    hold = int(rng.integers(30, 80))        # long first partial phase -> op3 branch later
                first = False
                a[t:t + hold, e] = cur
                t += hold
                cur = 7 - cur
            det = int(rng.integers(0, 2))                   # detour: standby or cooldown
            hold = int(rng.integers(5, 60))
            a[t:t + hold, e] = det
            t += hold
            w = int(rng.integers(20, 200))
    return a[:K]


def to_continuous(rng, a):
    """discrete tape -> float32 actions inside the matching interval, with the decode edge cases injected."""
    centers = -1 + 0.4 * (a + 0.5)
    x = (centers + rng.uniform(-0.19, 0.19, a.shape)).astype(np.float32)
    edges = np.array([-1.0, 1.0, np.nan, -1.5, 1.5, -0.6, -0.2, 0.2, 0.6, 0.99999994, -0.99999994,
                      np.nextafter(np.float32(-0.6), np.float32(1)), np.nextafter(np.float32(0.2), np.float32(-1)),
                      np.nextafter(np.float32(0.6), np.float32(-1)), 0.0, -0.0, 5.0, -5.0, np.inf, -np.inf],
                     dtype=np.float32)
    K, n = a.shape
    pos = rng.choice(K, size=min(K // 3, 6 * len(edges)), replace=False)
    for q, t in enumerate(np.sort(pos)):
        x[t, q % n] = edges[q % len(edges)]
    return x


def save_prep(name, setup, extra_meta):
    pre, price = setup.pre, setup.price
    kw = {s: setup.kwargs(s) for s in ("train", "val", "test")}
    arrs = {}
    for s in ("train", "val", "test"):
        arrs[f"el_{s}"] = price[f"el_price_{s}"].astype(np.float64)
        arrs[f"gas_{s}"] = np.asarray(price[f"gas_price_{s}"], dtype=np.float64)
        arrs[f"eua_{s}"] = np.asarray(price[f"eua_price_{s}"], dtype=np.float64)
        arrs[f"pot_rew_{s}"] = pre.dict_pot_r_b[f"pot_rew_{s}"].astype(np.float64)
        arrs[f"part_full_{s}"] = pre.dict_pot_r_b[f"part_full_b_{s}"].astype(np.int8)
        # cross-check the series folding used by every consumer of these fixtures
        c, t, m = po.split_reference_kwargs(kw[s], "train")
        P = kw[s]["price_ahead"]
        assert np.array_equal(m["el"], arrs[f"el_{s}"][:len(m["el"])]) and len(m["el"]) == len(arrs[f"el_{s}"]) - 1
        assert np.array_equal(m["pot_rew"], arrs[f"pot_rew_{s}"][:len(m["pot_rew"])])
        assert np.array_equal(m["part_full"], arrs[f"part_full_{s}"][:len(m["part_full"])].astype(float))
        assert np.array_equal(m["gas"], arrs[f"gas_{s}"]) and np.array_equal(m["eua"], arrs[f"eua_{s}"])
    consts, _, _ = po.split_reference_kwargs(kw["train"], "train")
    meta = dict(consts=jsonable(consts), r_level=float(pre.r_level[0]), n_eps=int(pre.n_eps),
                eps_sim_steps=dict(train=int(pre.eps_sim_steps_train), val=int(pre.eps_sim_steps_val),
                                   test=int(pre.eps_sim_steps_test)),
                rew_l_b=float(kw["train"]["rew_l_b"]), rew_u_b=float(kw["train"]["rew_u_b"]),
                operation=setup.EnvConfig.operation, scenario=int(setup.EnvConfig.scenario),
                seed_train=int(setup.TrainConfig.seed_train), train_steps=int(setup.TrainConfig.train_steps),
                **extra_meta)
    arrs["eps_ind"] = np.asarray(pre.eps_ind, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, f"prep_{name}.npz"), meta=json.dumps(meta), **arrs)
    print(f"  prep_{name}.npz  n_eps={pre.n_eps} eps_ind={len(pre.eps_ind)} r_level={pre.r_level[0]:.8f}")


def save_traj(case, setup, prep_name, split, train_or_eval, actions, seed, kw_over=None, note=""):
    kw = dict(setup.kwargs(split))
    kw.update(kw_over or {})
    out = rd.run_vector(kw, actions, seed=seed, train_or_eval=train_or_eval)
    consts, _, market = po.split_reference_kwargs(kw, train_or_eval)
    meta = dict(case=case, prep=prep_name, split=split, train_or_eval=train_or_eval, n_envs=int(actions.shape[1]),
                seed=int(seed), ep_index0=0, consts=jsonable(consts), operation=setup.EnvConfig.operation,
                int_cols=rd.INT_COLS, f64_cols=rd.F64_COLS, info_keys=rd.INFO_KEYS, note=note)
    arrs = dict(actions=out["actions"], ints=out["ints"].astype(np.int32), f64s=out["f64s"], obs=out["obs"],
                done=out["done"], noise=out["noise"], noise_len=out["noise_len"], n_noise=out["n_noise"].astype(np.int32),
                reset_obs=out["reset_obs"], reset_int=out["reset_int"].astype(np.int32), reset_info=out["reset_info"],
                post_reset_obs=out["post_reset_obs"], post_reset_int=out["post_reset_int"].astype(np.int32),
                post_reset_at=out["post_reset_at"].astype(np.int32), ep_index_end=out["ep_index_end"],
                eps_ind=np.zeros(0) if market["eps_ind"] is None else market["eps_ind"])
    if "infos" in out:
        arrs["infos"] = out["infos"]
    np.savez_compressed(os.path.join(OUT, f"traj_{case}.npz"), meta=json.dumps(meta), **arrs)
    ints = out["ints"]
    pt, ft = set(ints[..., 6].reshape(-1).tolist()), set(ints[..., 7].reshape(-1).tolist())
    print(f"  traj_{case}.npz K={actions.shape[0]} n={actions.shape[1]} dones={int(out['done'].sum())} "
          f"noise={out['noise_len'].tolist()} partial_tids={sorted(pt)} full_tids={sorted(ft)} "
          f"sum_rew={out['f64s'][..., 0].sum():.6f} zero_rew={(out['f64s'][..., 0] == 0).mean():.2f}")
    return out


def save_units(setup, op):
    kw = setup.kwargs("train")
    m = rd._import_reference()
    env = m["ptg"].PTGEnv(kw, "train")
    allT = np.unique(np.concatenate([kw[k][:, 1] for k in rd.TABLE_KEYS] + [np.array([16.0])]))
    dests = ["cooldown", "standby_up", "standby_down", "startup_cold", "startup_hot", "op1_start_p"]
    idx = np.zeros((len(dests), len(allT)), np.int32)
    for d, k in enumerate(dests):
        for q, T in enumerate(allT):
            idx[d, q] = env._get_index(kw[k], T)
    np.savez_compressed(os.path.join(OUT, f"units_{op}.npz"), T=allT, dests=np.array(dests), get_index=idx)
    print(f"  units_{op}.npz distinct_T={len(allT)}")


def save_tables(setup, op):
    os.makedirs(DATA, exist_ok=True)
    np.savez_compressed(os.path.join(DATA, f"tables_{op}.npz"), **{k: setup.op[k] for k in rd.TABLE_KEYS})
    print(f"  rl_ptg_amd/data/tables_{op}.npz rows={sum(len(setup.op[k]) for k in rd.TABLE_KEYS)}")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(12345)

    # ---------------- real market data, default episode length (37 d), sim_step 600 ----------------
    print("real / BS2 / OP2")
    s = rd.RefSetup(dict(scenario=2, operation="OP2"))
    save_tables(s, "OP2")
    save_units(s, "OP2")
    save_prep("real_bs2_OP2", s, dict(market="real", eps_len_d=37, sim_step=600))
    # real price series before scenario overrides: BS1 leaves them untouched -> saved from the BS1 setup below
    save_traj("real_bs2_op2_mod_disc_train", s, "real_bs2_OP2", "train", "train",
              rng.integers(0, 5, (700, 3)), 3654, note="uniform-random actions; pins ep_index order (3 init + 3 reset)")
    save_traj("real_bs2_op2_mod_disc_evalval", s, "real_bs2_OP2", "val", "eval",
              sticky_tape(rng, 1500, 2), 605, note="sticky actions, eval mode (info rows)")
    save_traj("real_bs2_op2_raw_cont_test", s, "real_bs2_OP2", "test", "train",
              to_continuous(rng, sticky_tape(rng, 900, 2, 1 / 6.0)), 11, dict(raw_modified="raw", action_type="continuous"),
              note="raw features + continuous actions with decode edge cases")
    s.close()

    print("real / BS1 / OP1")
    s = rd.RefSetup(dict(scenario=1, operation="OP1"))
    save_tables(s, "OP1")
    save_units(s, "OP1")
    save_prep("real_bs1_OP1", s, dict(market="real", eps_len_d=37, sim_step=600))
    np.savez_compressed(os.path.join(OUT, "market_real.npz"),
                        **{f"{c}_{sp}": np.asarray(s.price[f"{c}_price_{sp}"], dtype=np.float64)
                           for c in ("el", "gas", "eua") for sp in ("train", "val", "test")})
    save_traj("real_bs1_op1_raw_cont_evalval", s, "real_bs1_OP1", "val", "eval",
              to_continuous(rng, sticky_tape(rng, 1500, 2)), 605, dict(raw_modified="raw", action_type="continuous"))
    save_traj("real_bs1_op1_mod_disc_train", s, "real_bs1_OP1", "train", "train", rng.integers(0, 5, (600, 2)), 467)
    s.close()

    print("real / BS3 / OP2")
    s = rd.RefSetup(dict(scenario=3, operation="OP2"))
    save_prep("real_bs3_OP2", s, dict(market="real", eps_len_d=37, sim_step=600))
    save_traj("real_bs3_op2_mod_cont_test", s, "real_bs3_OP2", "test", "eval",
              to_continuous(rng, sticky_tape(rng, 1200, 2)), 7, dict(action_type="continuous"))
    s.close()

    # ---------------- synthetic 38-day market (BASELINE.json configurations) ----------------
    sm = synth_markets()
    for scen, op in ((2, "OP2"), (1, "OP1"), (3, "OP2")):
        print(f"synthetic / BS{scen} / {op} / 32-day episodes")
        s = rd.RefSetup(dict(scenario=scen, operation=op, eps_len_d=32), synthetic_market=sm)
        for sp in ("train", "val", "test"):     # the reference's loader must see exactly what the product generator emits
            el, gas, eua = sm[sp]
            assert np.array_equal(s.price[f"el_price_{sp}"], el / 10)
            if scen == 1:
                assert np.array_equal(s.price[f"gas_price_{sp}"], gas / 10) and np.array_equal(s.price[f"eua_price_{sp}"], eua)
        save_prep(f"synth_bs{scen}_{op}", s, dict(market="synth", eps_len_d=32, sim_step=600))
        save_traj(f"synth_bs{scen}_{op.lower()}_mod_disc_train", s, f"synth_bs{scen}_{op}", "train", "train",
                  sticky_tape(rng, 800, 2), 100 + scen, note="BASELINE.json-style configuration (32-day episode, sticky actions)")
        s.close()

    # short episodes -> terminations, auto-reset order over the shared ep_index, state-change penalty
    print("synthetic / BS2 / OP2 / 2-day episodes, penalty")
    s = rd.RefSetup(dict(scenario=2, operation="OP2", eps_len_d=2, state_change_penalty=0.3), synthetic_market=sm,
                    train_steps=20000)
    out = save_traj("synth_bs2_op2_term_penalty", s, "synth_bs2_OP2", "train", "train", sticky_tape(rng, 900, 5, 1 / 5.0), 3654,
                    note="eps_len_d=2 (283-step episodes), 5 envs sharing ep_index, state_change_penalty=0.3")
    assert out["done"].sum() >= 10
    # an env set whose members terminate on different steps: different ep_index order
    s.close()

    # sim_step = 60 s (step_size 30): reaches every rung of the _partial / _full ladders
    for scen, op in ((2, "OP2"), (1, "OP1")):
        print(f"synthetic / BS{scen} / {op} / sim_step 60")
        s = rd.RefSetup(dict(scenario=scen, operation=op, eps_len_d=1, sim_step=60), synthetic_market=sm, train_steps=200000)
        out = save_traj(f"synth_bs{scen}_{op.lower()}_s60_toggle", s, f"synth_bs{scen}_{op}", "train", "train",
                        toggler_tape(rng, 2600, 2), 42 + scen, note="sim_step=60: all _partial/_full ladder rungs; eps_len_d=1 (1435-step episodes)")
        pt, ft = set(out["ints"][..., 6].reshape(-1).tolist()), set(out["ints"][..., 7].reshape(-1).tolist())
        assert pt >= {5, 8, 9, 10, 11, 12} and ft >= {6, 7, 13, 14, 15, 16}, (pt, ft)
        s.close()
    print("done")


if __name__ == "__main__":
    main()
