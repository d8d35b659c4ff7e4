"""The collectives of the N > 1 path on the RCCL backend itself.  Only one GPU is available to the tests, so the process group has
ONE rank: what this pins is that every tensor the path hands to torch.distributed (dtype, device, the flat all-gather form) is
accepted by RCCL and comes back unchanged -- the arithmetic across ranks is covered by the world_size-2 gloo test."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys
sys.path.insert(0, os.environ["PTG_ROOT"])
import numpy as np, torch, torch.distributed as dist
from rl_ptg_amd import dist as ptg_dist
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
try:
    r, l = np.array([1.5, -2.25, 3.0]), np.array([7, 8, 9])
    ra, la = ptg_dist.all_gather_finished(r, l, device=dev)
    assert ra.tolist() == r.tolist() and la.tolist() == l.tolist()
    ra, la = ptg_dist.all_gather_finished(np.zeros(0), np.zeros(0, np.int64), device=dev)       # nobody finished anything
    assert len(ra) == 0 and len(la) == 0
    mom = torch.tensor([[4.0, 1.0, 2.0], [4.0, -1.0, 0.5]], dtype=torch.float64, device=dev)
    out = ptg_dist.all_merge_moments(mom)
    assert torch.equal(out.cpu(), mom.cpu())
    # the engine's reward normalisation with the group in place: same numbers as without a group (one rank)
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=8)
    n, T = 512, 12
    res = []
    for grouped in (True, False):
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
        eng.set_episode_plan(spec.eps_ind, *ptg_dist.episode_plan(n, 1, 0))
        eng.set_noise_rng(3)
        eng.vn_init()
        eng.reset()
        acts = np.random.default_rng(5).integers(0, 5, (T, n)).astype(np.int32)
        obs, rew, done = eng.rollout(acts)
        if grouped:
            out = eng.vn_normalize(rew, done)
        else:
            dist.destroy_process_group()
            out = eng.vn_normalize(rew, done)
        eng.sync()
        res.append(out.cpu().numpy().copy())
        eng.close()
    assert np.array_equal(res[0], res[1])
    print("RCCL_OK")
finally:
    if dist.is_initialized():
        dist.destroy_process_group()
'''


def test_collectives_on_the_rccl_backend_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PTG_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
