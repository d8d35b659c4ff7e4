"""Wall clock around ONE 20-step rollout launch at 65 536 envs: launch call, launch + wait, by wait flavour and with / without the
kernel-attached event pair (ptg_profile).  Where do the 75-90 us of bench.py's 20-step window go when the kernel takes 33?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device
n, T, R = 65536, int(os.environ.get("HC_T", "20")), 40
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
eng.set_episode_plan(spec.eps_ind, n, n); eng.set_noise_rng(1)
acts = sticky_actions_device(400 + T, n, seed=1, device=dev)
eng.reset(); eng.rollout(acts[:400]); eng.sync()
obs = torch.zeros((T, n, 35), device=dev); rew = torch.zeros((T, n), device=dev); done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
a = acts[400:400 + T]
for prof in (False, True):
    for wait in ("torch.cuda.synchronize", "ptg_sync (poll)", "stream.query spin"):
        eng.profile(prof)
        ts = []
        st = torch.cuda.current_stream(dev)
        for r in range(R):
            torch.cuda.synchronize()
            time.sleep(0.0005)
            t0 = time.perf_counter()
            eng.rollout(a, obs, rew, done)
            t1 = time.perf_counter()
            if wait.startswith("torch"):
                torch.cuda.synchronize()
            elif wait.startswith("ptg"):
                eng.sync()
            else:
                while not st.query():
                    pass
            t2 = time.perf_counter()
            ts.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
        k = eng.profile_read() if prof else None
        ts = np.array(ts[8:])
        print(f"events {'on ' if prof else 'off'} | wait = {wait:<24} | launch call {np.median(ts[:, 0]):5.1f} us | launch + wait: median {np.median(ts[:, 1]):5.1f}, min {ts[:, 1].min():5.1f}, p90 {np.percentile(ts[:, 1], 90):5.1f} us"
              + (f" | kernel {np.median(k):.1f} us" if k is not None else ""), flush=True)
eng.close()
