"""The fused rollout on the reference's REAL training set instead of the synthetic one-episode trace: 36 552 hourly prices, 41 episodes of
37 days drawn through the shuffled eps_ind (src/rl_utils.py:315-335), so the envs of a batch sit in DIFFERENT episodes -- their market windows
are per-lane gathers, not one broadcast line per wave as in the bench's trace (n_eps = 1).  Price series: tests/golden/market_real.npz (data of
the reference repository, as its loader returns them).  python tools/realdata_bench.py [envs] [scenario] [operation]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rl_ptg_amd import dist as ptg_dist
from rl_ptg_amd.config import EnvConfig
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import EnvSpec, Preprocessing, synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device
from rl_ptg_amd.tables import load_op_tables

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
scenario = int(sys.argv[2]) if len(sys.argv) > 2 else 1
operation = sys.argv[3] if len(sys.argv) > 3 else "OP1"
dev = torch.device("cuda", 0)
z = np.load(os.path.join(ROOT, "tests", "golden", "market_real.npz"))
prices = {k: z[k] for k in z.files}


def run(tag, spec, plan):
    for dtype, layout, B in (("float32", "row", 149), ("float64", "row", 293)):
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=dtype, obs_layout=layout)
        eng.set_episode_plan(spec.eps_ind, *plan)
        eng.set_noise_rng(seed=20250614)
        T = 400
        actions = sticky_actions_device(3 * T, n, seed=1234, device=dev, p_switch=1.0 / 12.0)
        obs = eng.alloc_obs(T)
        rew = torch.zeros((T, n), dtype=eng.out_dtype, device=dev)
        done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
        eng.reset()
        eng.rollout(actions[:T], obs, rew, done)
        eng.sync()
        eng.profile(True)
        for q in (1, 2):
            eng.rollout(actions[q * T:(q + 1) * T], obs, rew, done)
        eng.sync()
        us, hp, sp = eng.profile_read_ex()
        per = sum(sp) / (2 * T)
        ad = eng.get_state("act_ep_d")
        print(f"{tag:34s} {dtype} {layout}: {per:.3f} us per fused step = {B * n / per / 1e3:.0f} GB/s = {B * n / per / 8e6:.3f} of 8 TB/s"
              f"   (distinct episode offsets in the batch: {len(np.unique(ad))})", flush=True)
        eng.close()


cfg = EnvConfig(scenario=scenario, operation=operation)                      # eps_len_d = 37 as in the reference's config_env.yaml
pre = Preprocessing(prices, load_op_tables(operation), cfg, seed_train=3654, train_steps=1500000, action_type="discrete")
spec = EnvSpec.from_dict_input(pre.dict_env_kwargs("train"), "train")
print(f"# N = {n}, BS{scenario}/{operation}; real training set: {len(prices['el_train'])} hours, n_eps = {pre.n_eps}, eps_ind {len(spec.eps_ind)} entries, "
      f"eps_sim_steps {spec.consts['eps_sim_steps']}; steady state: 2 x 400 fused steps after 400, kernel-attached events", flush=True)
run("real data, DummyVecEnv episode order", spec, (n, n))                   # env e takes eps_ind[n + e + m n]: every env its own draw
syn, _ = synthetic_spec(scenario=scenario, operation=operation, eps_len_d=32)
run("synthetic 38-day trace (bench)", syn, ptg_dist.episode_plan(n, 1, 0))
