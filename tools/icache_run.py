"""Workload for tools/icache_pmc.sh: stationary batch, then launches of 1, 1, 2, 5, 20, 20, 100, 400 steps (row-major float32, N = 65 536)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rl_ptg_amd import dist as ptg_dist
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device
n = 65536
TS = [400, 1, 1, 2, 5, 20, 20, 100, 400]
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
first_ptr, stride = ptg_dist.episode_plan(n, 1, 0)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
eng.set_noise_rng(seed=20250614)
actions = sticky_actions_device(sum(TS), n, seed=1234, device=dev, p_switch=1.0 / 12.0)
R = max(TS)
obs = torch.zeros((R, n, eng.obs_dim), dtype=torch.float32, device=dev)
rew = torch.zeros((R, n), dtype=torch.float32, device=dev)
done = torch.zeros((R, n), dtype=torch.uint8, device=dev)
eng.reset()
t0 = 0
for T in TS:
    eng.rollout(actions[t0:t0 + T], obs[:T], rew[:T], done[:T])
    eng.sync()
    t0 += T
eng.close()
