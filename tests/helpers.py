"""Shared fixture loading for the parity tests (test infrastructure; may import the oracle)."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ptg_oracle as po  # noqa: E402

TRAJ_CASES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLD, "traj_*.npz")))
PREP_CASES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLD, "prep_*.npz")))

_cache = {}


def load_npz(path):
    if path not in _cache:
        z = np.load(path, allow_pickle=False)
        d = {k: z[k] for k in z.files}
        if "meta" in d:
            d["meta"] = json.loads(str(d["meta"]))
        _cache[path] = d
    return _cache[path]


def load_tables(op):
    z = load_npz(os.path.join(ROOT, "rl_ptg_amd", "data", f"tables_{op}.npz"))
    return {k: z[k] for k in po.TABLE_KEYS}


def load_prep(name):
    return load_npz(os.path.join(GOLD, f"prep_{name}.npz"))


def load_traj(case):
    """-> (traj dict, consts, tables, market) with market series taken from the referenced prep fixture."""
    tr = load_npz(os.path.join(GOLD, f"traj_{case}.npz"))
    meta = tr["meta"]
    prep = load_prep(meta["prep"])
    sp = meta["split"]
    market = dict(el=prep[f"el_{sp}"], pot_rew=prep[f"pot_rew_{sp}"], part_full=prep[f"part_full_{sp}"].astype(np.float64),
                  gas=prep[f"gas_{sp}"], eua=prep[f"eua_{sp}"],
                  eps_ind=tr["eps_ind"] if len(tr["eps_ind"]) else None)
    return tr, dict(meta["consts"]), load_tables(meta["operation"]), market


def make_oracle(case):
    tr, consts, tables, market = load_traj(case)
    env = po.OracleVecEnv(consts, tables, market, tr["meta"]["n_envs"], ep_index0=tr["meta"]["ep_index0"])
    env.set_noise_tape(tr["noise"])
    return tr, env


def market_for_engine(consts, market, prep_meta=None):
    """oracle-style market dict -> HipEngine market dict (adds the scenario-dependent scalars)."""
    m = {k: market[k] for k in ("el", "pot_rew", "part_full", "gas", "eua")}
    m.update(scenario=consts["scenario"], rew_l_b=consts["rew_l_b"], rew_u_b=consts["rew_u_b"], r_0=consts["r_0"])
    return m


def make_engine(case, out_dtype="float64", n_envs=None, obs_layout="row"):
    """HIP engine on cuda:0 configured exactly like the reference run that produced the fixture."""
    from rl_ptg_amd.engine import HipEngine
    tr, consts, tables, market = load_traj(case)
    n = tr["meta"]["n_envs"] if n_envs is None else n_envs
    eng = HipEngine(consts, tables, market_for_engine(consts, market), n, device=0, out_dtype=out_dtype, obs_layout=obs_layout)
    eng.set_noise_tape(tr["noise"])
    if market["eps_ind"] is not None:
        # DummyVecEnv order: n constructions consume eps_ind[0:n], the first vector reset takes eps_ind[n + e]
        eng.set_episode_plan(market["eps_ind"], first_ptr=n, stride=n)
    return tr, eng


def kwargs_from_fixture(case):
    """A reference-style env kwargs dict (src/rl_utils.py:337-405, price data as 1-D series) rebuilt from a trajectory fixture."""
    tr, consts, tables, market = load_traj(case)
    kw = {k: v for k, v in consts.items() if k not in ("raw_modified", "action_type", "train_or_eval", "r_0")}
    kw.update(raw_modified="mod" if consts["raw_modified"] else "raw",
              action_type="continuous" if consts["action_type"] else "discrete",
              reward_level=np.array([consts["r_0"]]), parallel="Singleprocessing", n_eps_loops=0,
              eps_ind=None if market["eps_ind"] is None else market["eps_ind"].astype(int),
              el_series=market["el"], pot_rew_series=market["pot_rew"], part_full_series=market["part_full"],
              gas_series=market["gas"], eua_series=market["eua"])
    kw.update({f"ptg_{k}": i for i, k in enumerate(["standby", "cooldown", "startup", "partial_load", "full_load"])})
    kw.update(tables)
    return tr, kw
