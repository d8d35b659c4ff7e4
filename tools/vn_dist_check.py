"""Two (or more) ranks, each a shard of one env batch, normalise rewards with the merged moments of all ranks; rank 0 also runs the
whole batch in one engine and compares.  Launch: PTG_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr
127.0.0.1 --master-port 29541 tools/vn_dist_check.py   (gloo lets several ranks share one GPU; nccl on a multi-GPU node)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from rl_ptg_amd import dist as ptg_dist
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("PTG_BACKEND", "nccl")
dev_id = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
torch.cuda.set_device(dev_id)
dist.init_process_group(backend, rank=rank, world_size=world)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=1, train_steps=400000)
n_total, K = 1024, 200
n = n_total // world
acts = np.random.default_rng(3).integers(0, 5, (K, n_total)).astype(np.int32)


def make(n_envs, r, w):
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n_envs, device=dev_id, out_dtype="float32", obs_layout="feature")
    first_ptr, stride = ptg_dist.episode_plan(n_total, w, r)
    eng.set_global_env_offset(first_ptr - n_total)
    eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
    eng.set_noise_rng(9)
    eng.reset()
    return eng


eng = make(n, rank, world)
lo, hi = ptg_dist.shard_range(n_total, world, rank)
_, r, d = eng.rollout(acts[:, lo:hi])
eng.vn_init()
out = eng.vn_normalize(r, d)                      # all-gathers and merges the per-step moments of all ranks
eng.sync()
st, _ = eng.vn_get()
stats = torch.tensor([st["mean"], st["var"], st["count"]], dtype=torch.float64)
allst = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
dist.all_gather(allst, stats if backend != "nccl" else stats.cuda())
allst = [a.cpu() for a in allst]
assert all(torch.equal(allst[0], a) for a in allst), "ranks disagree on the running statistics"
if rank == 0:
    full = make(n_total, 0, 1)
    _, rf, df = full.rollout(acts)
    full.vn_init()
    # single-engine reference without the process group: call the two phases directly
    import ctypes as C
    mom = torch.empty((K, 3), dtype=torch.float64, device=rf.device)
    full._chk(full._L.ptg_vn_batch_moments(full._h, C.c_void_p(rf.data_ptr()), C.c_void_p(df.data_ptr()), K, C.c_void_p(mom.data_ptr()), full._stream()))
    of = torch.empty_like(rf)
    full._chk(full._L.ptg_vn_apply(full._h, C.c_void_p(rf.data_ptr()), K, C.c_void_p(mom.data_ptr()), C.c_void_p(of.data_ptr()), 1, full._stream()))
    full.sync()
    sf, _ = full.vn_get()
    np.testing.assert_allclose([st["mean"], st["var"], st["count"]], [sf["mean"], sf["var"], sf["count"]], rtol=1e-11)
    np.testing.assert_allclose(out.cpu().numpy(), of[:, lo:hi].cpu().numpy(), rtol=1e-6, atol=1e-30)
    assert torch.equal(r, rf[:, lo:hi])
    print("vn_dist_check ok: %d ranks (%s), statistics of the whole batch on every rank: mean %.6f var %.6f count %.4f" % (world, backend, st["mean"], st["var"], st["count"]))
dist.barrier()
dist.destroy_process_group()
