"""PCIe-inclusive rate of the NumPy-facing VecEnv path (reported in DESIGN.md; never the bench `value`)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.vec_env import PtGVecEnv

for n, dt, lay in [(65536, "float64", "row"), (65536, "float32", "feature"), (6, "float64", "row")]:
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
    env = PtGVecEnv(spec, n, seed=1, out_dtype=dt, obs_layout=lay, noise="device")
    env.reset()
    rng = np.random.default_rng(0)
    acts = [rng.integers(0, 5, n) for _ in range(60)]
    for a in acts[:10]:
        env.step(a)
    t0 = time.perf_counter()
    for a in acts[10:]:
        env.step(a)
    dt_s = time.perf_counter() - t0
    print(f"PtGVecEnv.step N={n} {dt} {lay}: {50 * n / dt_s:.3e} env-steps/s  ({dt_s / 50 * 1e3:.3f} ms per vector step, host buffers / NumPy dicts included)")
    env.close()
