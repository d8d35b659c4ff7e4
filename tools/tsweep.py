"""Fixed vs per-step cost of a fused rollout launch: python tools/tsweep.py [envs]   (run under rocprofv3 --kernel-trace to get
the kernels' own durations next to the HIP-event figures printed here; tools/tsweep_parse.py joins the two)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rl_ptg_amd import dist as ptg_dist
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
layout = os.environ.get("TS_LAYOUT", "feature")
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
first_ptr, stride = ptg_dist.episode_plan(n, 1, 0)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype=os.environ.get("TS_DTYPE", "float32"), obs_layout=layout)
eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
eng.set_noise_rng(seed=20250614)
TS = [int(x) for x in os.environ.get("TS_T", "1,2,5,10,20,25,50,100,200,400").split(",")]
REPS = int(os.environ.get("TS_REPS", "6"))
TMAX = max(TS)
actions = sticky_actions_device(600 + TMAX, n, seed=1234, device=dev, p_switch=1.0 / 12.0)
F = eng.obs_dim
obs = torch.zeros((TMAX, F, n) if eng.feature_major else (TMAX, n, F), dtype=eng.out_dtype, device=dev)
rew = torch.zeros((TMAX, n), dtype=eng.out_dtype, device=dev)
done = torch.zeros((TMAX, n), dtype=torch.uint8, device=dev)
eng.reset()
eng.rollout(actions[:400], obs[:400] if TMAX >= 400 else None, None, None)      # stationary state mix
eng.sync()
print("order " + " ".join(f"{T}x{REPS}" for T in TS), flush=True)
for T in TS:
    evs = []
    eng.profile(True)
    for r in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.rollout(actions[400:400 + T], obs[:T], rew[:T], done[:T])
        e1.record()
        evs.append((e0, e1))
        if os.environ.get("TS_SYNC_EACH"):
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    us = [a.elapsed_time(b) * 1e3 for a, b in evs]
    ku = eng.profile_read()
    eng.profile(False)
    print(f"T {T:4d} events us: " + " ".join("%.1f" % u for u in us) + f"   min/T {min(us) / T:.3f}   | kernel-attached: " + " ".join("%.1f" % u for u in ku), flush=True)
eng.close()
