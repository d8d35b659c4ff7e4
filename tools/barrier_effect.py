"""Does a torch.distributed barrier (an RCCL kernel) right before a 20-step rollout launch change the launch's duration?
python tools/barrier_effect.py   (one rank, nccl backend, 127.0.0.1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import numpy as np, torch, torch.distributed as dist
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n, T = 65536, 20
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
eng.set_episode_plan(spec.eps_ind, n, n); eng.set_noise_rng(1)
acts = sticky_actions_device(400 + 64 * T, n, seed=1, device=dev)
eng.reset(); eng.rollout(acts[:400]); eng.sync()
obs = torch.zeros((T, n, 35), device=dev); rew = torch.zeros((T, n), device=dev); done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
x = torch.zeros(2, dtype=torch.float64, device=dev)
eng.profile(True)
r = 0
for mode in ("plain", "barrier", "allreduce", "plain", "barrier", "sleep"):
    us = []
    for rep in range(8):
        if mode == "barrier": dist.barrier()
        elif mode == "allreduce": dist.all_reduce(x, op=dist.ReduceOp.MAX)
        elif mode == "sleep": torch.cuda.synchronize(); import time; time.sleep(0.002)
        torch.cuda.synchronize()
        eng.rollout(acts[400 + r * T:400 + (r + 1) * T], obs, rew, done); r += 1
        us.append(float(eng.profile_read()[0]))
    print(f"{mode:10s} 20-step launch us: " + " ".join(f"{u:.1f}" for u in us), flush=True)
eng.close()
dist.destroy_process_group()
