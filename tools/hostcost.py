import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device
n, T = 65536, 20
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
eng.set_episode_plan(spec.eps_ind, n, n); eng.set_noise_rng(1)
acts = sticky_actions_device(400 + 40 * T, n, seed=1, device=dev)
eng.reset(); eng.rollout(acts[:400]); eng.sync()
obs = torch.zeros((T, n, 35), device=dev); rew = torch.zeros((T, n), device=dev); done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
for prof in (False, True):
    eng.profile(prof)
    ts = []
    for r in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout(acts[400 + r * T:400 + (r + 1) * T], obs, rew, done)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    if prof: eng.profile_read()
    ts = ts[5:]
    print("profile", prof, "host time of eng.rollout (launch only) median %.1f us; launch + synchronize %.1f us" % (sorted(t[0] for t in ts)[len(ts)//2], sorted(t[1] for t in ts)[len(ts)//2]))
import cProfile, pstats
eng.profile(False)
pr = cProfile.Profile(); pr.enable()
for r in range(200): eng.rollout(acts[400:400 + T], obs, rew, done)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumtime").print_stats(12)

# the pieces of bench.py's timed region (wall clock, microseconds)
import numpy as np
from rl_ptg_amd import dist as ptg_dist
eng.profile(True)
rows = []
for r in range(12):
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = [time.perf_counter()]
    ev0.record(); t.append(time.perf_counter())
    eng.rollout(acts[400 + r * T:400 + (r + 1) * T], obs, rew, done); t.append(time.perf_counter())
    ev1.record(); t.append(time.perf_counter())
    fr, fl, _ = eng.finished_episodes(); t.append(time.perf_counter())
    ptg_dist.all_gather_finished(fr, fl, device=torch.device("cpu")); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    rows.append([(b - a) * 1e6 for a, b in zip(t[:-1], t[1:])] + [(t[-1] - t[0]) * 1e6])
eng.profile_read()
med = np.median(np.array(rows[4:]), axis=0)
print("timed-region pieces (median us): ev0.record %.1f | rollout launch %.1f | ev1.record %.1f | finished_episodes %.1f | all_gather_finished %.1f | synchronize %.1f | total %.1f" % tuple(med))
