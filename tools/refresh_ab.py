"""Table-refresher modes side by side, in the driver's bench condition and in a long launch from reset.
Run once per mode (the knobs are read at ptg_create):   PTG_REFRESH_MODE=legacy|head  PTG_NO_REFRESH=1  (default: head pass in the
rollout's prologue + forked rolling passes).  Prints, per repetition on a fresh handle:
  A  reset -> W untimed steps -> M x K-step launches: kernel us / helper us / union us of each launch
  B  reset -> one 400-step launch (the front phase), then a second one (stationary)
python tools/refresh_ab.py [envs] [W] [K] [M]"""
import os, sys
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
M = int(sys.argv[4]) if len(sys.argv) > 4 else 6
layout = os.environ.get("TS_LAYOUT", "row")
dtype = os.environ.get("TS_DTYPE", "float32")
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
first_ptr, stride = ptg_dist.episode_plan(n, 1, 0)
mode = "none" if os.environ.get("PTG_NO_REFRESH") else os.environ.get("PTG_REFRESH_MODE", "default")
print(f"# refresher mode: {mode}; N={n} layout={layout} dtype={dtype}", flush=True)
LONG = 400
for rep in range(3):
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=dtype, obs_layout=layout)
    eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
    eng.set_noise_rng(seed=20250614)
    actions = sticky_actions_device(W + K * M + 2 * LONG, n, seed=1234 + rep, device=dev, p_switch=1.0 / 12.0)
    F = eng.obs_dim
    R = max(K, LONG)
    obs = torch.zeros((R, F, n) if eng.feature_major else (R, n, F), dtype=eng.out_dtype, device=dev)
    rew = torch.zeros((R, n), dtype=eng.out_dtype, device=dev)
    done = torch.zeros((R, n), dtype=torch.uint8, device=dev)
    eng.reset()
    eng.rollout(actions[:W], obs[:W], rew[:W], done[:W])
    eng.sync()
    torch.cuda.synchronize()
    eng.profile(True)
    for q in range(M):
        eng.rollout(actions[W + q * K:W + (q + 1) * K], obs[:K], rew[:K], done[:K])
        if q == 0:
            torch.cuda.synchronize()
    us, hp, sp = eng.profile_read_ex()
    print(f"A rep {rep}: kernel " + " ".join("%.1f" % u for u in us) + " | helper " + " ".join("%.1f" % u for u in hp) +
          " | union " + " ".join("%.1f" % u for u in sp), flush=True)
    eng.reset()
    torch.cuda.synchronize()
    t0 = W + K * M
    for q in range(2):
        eng.rollout(actions[t0 + q * LONG:t0 + (q + 1) * LONG], obs[:LONG], rew[:LONG], done[:LONG])
    us, hp, sp = eng.profile_read_ex()
    print(f"B rep {rep}: 400 steps from reset: kernel {us[0]:.1f} helper {hp[0]:.1f} union {sp[0]:.1f} us; next 400: kernel {us[1]:.1f} helper {hp[1]:.1f} union {sp[1]:.1f}", flush=True)
    eng.profile(False)
    eng.close()
