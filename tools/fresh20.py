"""The driver's bench condition, repeated: fresh handle -> reset -> W untimed steps -> sync -> K-step launch, kernel-attached events.
python tools/fresh20.py [envs] [W] [K] [follow-up launches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rl_ptg_amd import dist as ptg_dist
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
M = int(sys.argv[4]) if len(sys.argv) > 4 else 12
layout = os.environ.get("TS_LAYOUT", "feature")
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
first_ptr, stride = ptg_dist.episode_plan(n, 1, 0)
for rep in range(4):
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=os.environ.get("TS_DTYPE", "float32"), obs_layout=layout)
    eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
    eng.set_noise_rng(seed=20250614)
    actions = sticky_actions_device(W + K * (M + 1), n, seed=1234 + rep, device=dev, p_switch=1.0 / 12.0)
    F = eng.obs_dim
    obs = torch.zeros((K, F, n) if eng.feature_major else (K, n, F), dtype=eng.out_dtype, device=dev)
    rew = torch.zeros((K, n), dtype=eng.out_dtype, device=dev)
    done = torch.zeros((K, n), dtype=torch.uint8, device=dev)
    eng.reset()
    eng.rollout(actions[:W], obs[:W], rew[:W], done[:W])
    eng.sync()
    torch.cuda.synchronize()
    if os.environ.get("F20_SLEEP"):
        time.sleep(float(os.environ["F20_SLEEP"]))
    eng.profile(True)
    for q in range(M + 1):
        eng.rollout(actions[W + q * K:W + (q + 1) * K], obs, rew, done)
        if q == 0:
            torch.cuda.synchronize()
    us = eng.profile_read()
    print(f"rep {rep}: fresh handle, {W} warm-up steps, then {K}-step launches [us]: " + " ".join("%.1f" % u for u in us), flush=True)
    eng.close()
