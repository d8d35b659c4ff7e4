"""Device time of the reward-normalisation pass over a fused rollout's [T][N] rewards: python tools/bench_vecnorm.py [envs] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 400
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="feature")
eng.set_episode_plan(spec.eps_ind, n, n)
eng.set_noise_rng(1)
eng.reset()
acts = sticky_actions_device(T, n, seed=1, device=torch.device("cuda", 0))
_, r, d = eng.rollout(acts)
eng.vn_init()
out = torch.empty_like(r)
for _ in range(3):
    eng.vn_normalize(r, d, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    eng.vn_normalize(r, d, out=out)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
byt = T * n * (4 + 1 + 4 + 4)          # moments pass reads reward + done, normalise pass reads reward and writes the result
print("vn_normalize N=%d T=%d: %.1f us per call = %.3f us per vector step; %.0f GB/s of its 13 B per env-step" % (n, T, us, us / T, byt / us * 1e-3))
