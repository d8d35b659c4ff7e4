"""Step-path probe: K ptg_step launches replayed as one hipGraph, action rows either fresh from a long tape or cycling through
R cache-resident rows (what a policy that has just written its actions looks like): python tools/step_probe.py [envs] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
for R in (K, 4):
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="feature")
    eng.set_episode_plan(spec.eps_ind, n, n)
    eng.set_noise_rng(1)
    acts = sticky_actions_device(K + 200, n, seed=1, device=dev)
    eng.reset()
    for t in range(200):
        eng.step(acts[t], want_final=False)
    eng.sync()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for t in range(K):
                eng.step(acts[200 + (t % R)], want_final=False)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("N=%d K=%d action rows cycling through %d: %.3f us per step" % (n, K, R, e0.elapsed_time(e1) * 1e3 / K))
    eng.close()
