"""What the SB3-facing observation hand-off buys end to end: a collect loop  obs -> policy -> action -> env step  with a
policy of the shape the reference trains (SB3 "MultiInputPolicy": CombinedExtractor's 40 columns -> 64 -> 64 -> 5 logits, tanh; random
weights, greedy actions), four ways:
  flat    PTG_OBS_SB3_FLAT rows stay on the device: torch MLP on the [N, 40] tensor the kernel wrote, ptg_step on the action tensor it produced
  graph   the flat loop captured once (policy forward + ptg_step) and replayed as one hipGraph per step (ptg_note_replays afterwards)
  split   PTG_OBS_SPLIT rows (16 columns) + policy_split.FirstLayerSplit for the first layer
  host    the drop-in route of the reference's loop: PtGVecEnv.step(numpy actions) -> dict of NumPy arrays -> flattened on the host ->
          torch.as_tensor(...).cuda() -> policy -> actions.cpu().numpy()   (what SB3's collect_rollouts does around a VecEnv)
Not product code; the policy is the caller's.  python tools/policy_loop.py [envs] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rl_ptg_amd import dist as ptg_dist
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.policy_split import FirstLayerSplit
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.vec_env import PtGVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
first_ptr, stride = ptg_dist.episode_plan(n, 1, 0)
torch.manual_seed(0)
l1, l2, l3 = torch.nn.Linear(40, 64).to(dev), torch.nn.Linear(64, 64).to(dev), torch.nn.Linear(64, 5).to(dev)
print(f"# N = {n}, {K} timed steps after 50 warm-up steps; policy 40 -> 64 -> 64 -> 5 (tanh), greedy; float32; wall clock, torch.cuda.synchronize() on both sides", flush=True)


def engine(layout):
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout=layout)
    eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
    eng.set_noise_rng(seed=20250614)
    return eng


def timed(step_fn, obs):
    with torch.no_grad():
        for _ in range(50):
            obs = step_fn(obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            obs = step_fn(obs)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


# --- flat rows on the device
eng = engine("sb3_flat")
obs = eng.reset()
def step_flat(o):
    a = l3(torch.tanh(l2(torch.tanh(l1(o))))).argmax(dim=1).to(torch.int32)
    return eng.step(a, want_final=False)[0]
dt = timed(step_flat, obs)
print(f"flat   (device-resident, [N, 40] rows):            {dt * 1e6:8.1f} us per vector step = {n / dt:.3e} env-steps/s", flush=True)
eng.close()

# --- the same loop captured once (policy forward + ptg_step) and replayed as a hipGraph: the hot kernels read the step count from the state
eng = engine("sb3_flat")
obs_static = eng.reset()
a_static = torch.zeros(n, dtype=torch.int32, device=dev)
def body():
    a_static.copy_(l3(torch.tanh(l2(torch.tanh(l1(obs_static))))).argmax(dim=1))
    eng.step(a_static, want_final=False)                     # writes the next observations into obs_static (= eng.obs); captured, the step
                                                             # is replay-proof: hot kernel + generic kernel, one of them a no-op
with torch.no_grad():
    for _ in range(3):
        body()                                              # (torch wants the ops warm before a capture)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            body()
    torch.cuda.current_stream().wait_stream(side)
dt = timed(lambda o: g.replay(), None)
eng.note_replays(50 + K - 1)                                 # the capture counted as one step; every replay beyond the first is reported
eng.sync()
assert eng.steps_to_episode_end() == int(spec.consts["eps_sim_steps"]) - 5 - (3 + 50 + K)
print(f"graph  (flat rows, policy + ptg_step as one hipGraph): {dt * 1e6:6.1f} us per vector step = {n / dt:.3e} env-steps/s", flush=True)
eng.close()

# --- split rows + first layer from projection tables
eng = engine("split")
fls = FirstLayerSplit(eng.market_feature_series(), "mod", device=dev).prepare(l1.weight.detach(), l1.bias.detach())
obs = eng.reset()
def step_split(o):
    a = l3(torch.tanh(l2(torch.tanh(fls(o))))).argmax(dim=1).to(torch.int32)
    return eng.step(a, want_final=False)[0]
dt = timed(step_split, obs)
print(f"split  (device-resident, [N, 16] rows + G_hour):   {dt * 1e6:8.1f} us per vector step = {n / dt:.3e} env-steps/s", flush=True)
eng.close()

# --- the VecEnv route (NumPy dict observations over PCIe, as SB3's loop sees them)
env = PtGVecEnv(spec, n_envs=n, seed=3654, noise="device", out_dtype="float32")
keys = sorted(env.observation_space.spaces.keys()) if hasattr(env.observation_space, "spaces") else None
od = env.reset()
def flatten(od):
    cols = []
    for k in sorted(od.keys()):
        v = od[k]
        if k == "METH_STATUS":
            v = (np.asarray(v).reshape(-1, 1) == np.arange(6)[None, :])
        cols.append(np.asarray(v, dtype=np.float32).reshape(n, -1))
    return np.concatenate(cols, axis=1)
def step_host(od):
    x = torch.as_tensor(flatten(od)).to(dev)
    a = l3(torch.tanh(l2(torch.tanh(l1(x))))).argmax(dim=1).cpu().numpy()
    return env.step(a)[0]
dt = timed(step_host, od)
print(f"host   (PtGVecEnv, NumPy dict obs, H2D / D2H):     {dt * 1e6:8.1f} us per vector step = {n / dt:.3e} env-steps/s", flush=True)
env.close()
print("# reference pipeline (SB3 PPO, 6 envs, its own published TensorBoard log): 166 steps/s (BASELINE.md); reference env alone 1-2e4 steps/s per core", flush=True)
