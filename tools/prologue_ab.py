"""Launch duration against steps per launch, in the driver's condition (fresh handle -> reset -> 5 untimed steps), kernel-attached events.
Run once per library (PTG_LIB_PATH = an experiment build, unset = the product) to compare prologue variants on one box.
python tools/prologue_ab.py [envs]      TS_LAYOUT=row|feature|split  TS_DTYPE=float32|float64"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rl_ptg_amd import dist as ptg_dist
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
layout = os.environ.get("TS_LAYOUT", "row")
dtype = os.environ.get("TS_DTYPE", "float32")
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
first_ptr, stride = ptg_dist.episode_plan(n, 1, 0)
TS = [20, 20, 20, 1, 1, 2, 2, 5, 5, 10, 10, 20, 20, 50, 100, 400, 400]
W = 5
tag = os.path.basename(os.environ.get("PTG_LIB_PATH", "product"))
print(f"# {tag}: N={n} layout={layout} dtype={dtype}; T sequence after reset + {W} steps", flush=True)
for rep in range(3):
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=dtype, obs_layout=layout)
    eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
    eng.set_noise_rng(seed=20250614)
    actions = sticky_actions_device(W + sum(TS), n, seed=1234 + rep, device=dev, p_switch=1.0 / 12.0)
    F = eng.obs_dim
    R = max(TS)
    obs = torch.zeros((R, F, n) if eng.feature_major else (R, n, F), dtype=eng.out_dtype, device=dev)
    rew = torch.zeros((R, n), dtype=eng.out_dtype, device=dev)
    done = torch.zeros((R, n), dtype=torch.uint8, device=dev)
    eng.reset()
    eng.rollout(actions[:W], obs[:W], rew[:W], done[:W])
    eng.sync()
    torch.cuda.synchronize()
    eng.profile(True)
    t0 = W
    out = []
    for T in TS:
        eng.rollout(actions[t0:t0 + T], obs[:T], rew[:T], done[:T])
        torch.cuda.synchronize()
        us, hp, sp = eng.profile_read_ex()
        out.append("T%d:%.1f" % (T, sum(sp)))
        t0 += T
    print(f"rep {rep}: " + " ".join(out), flush=True)
    eng.profile(False)
    eng.close()
