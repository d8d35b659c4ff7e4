"""Workload for tools/phase_pmc.sh: fresh handle -> reset -> 5 steps -> 45 launches of 20 steps (early phase vs stationary phase)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rl_ptg_amd import dist as ptg_dist
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device
n, W, K, M = 65536, 5, 20, 45
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
first_ptr, stride = ptg_dist.episode_plan(n, 1, 0)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout=os.environ.get("TS_LAYOUT", "feature"))
eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
eng.set_noise_rng(seed=20250614)
actions = sticky_actions_device(W + K * M, n, seed=1234, device=dev, p_switch=1.0 / 12.0)
F = eng.obs_dim
obs = torch.zeros((K, F, n) if eng.feature_major else (K, n, F), dtype=torch.float32, device=dev)
rew = torch.zeros((K, n), dtype=torch.float32, device=dev)
done = torch.zeros((K, n), dtype=torch.uint8, device=dev)
eng.reset()
eng.rollout(actions[:W], obs[:W], rew[:W], done[:W])
eng.sync()
for q in range(M):
    eng.rollout(actions[W + q * K:W + (q + 1) * K], obs, rew, done)
    if q in (0, 1, 2, 5, 10, 20, 30, 40, 44):
        import numpy as np
        eng.sync()
        st = eng.get_state("meth_state")
        print("after launch", q, "state mix", np.round(np.bincount(st, minlength=5) / n, 3), "distinct i", len(np.unique(eng.get_state("i"))),
              "distinct T", len(np.unique(eng.get_state("T_cat"))), flush=True)
eng.close()
