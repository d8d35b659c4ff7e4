"""PCIe-inclusive rate of the NumPy-facing VecEnv path (reported in DESIGN.md; never the bench `value`)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.vec_env import PtGVecEnv

for n, dt, lay, mode in [(65536, "float32", "row", "train"), (65536, "float32", "feature", "train"), (65536, "float64", "row", "train"),
                         (4096, "float64", "row", "train"), (6, "float64", "row", "train"), (6, "float32", "row", "train"), (1, "float64", "row", "eval"),
                         (65536, "float32", "row", "eval")]:
    spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
    env = PtGVecEnv(spec, n, train_or_eval=mode, seed=1, out_dtype=dt, obs_layout=lay, noise="device")
    env.reset()
    rng = np.random.default_rng(0)
    K = 300 if n <= 4096 else 60
    acts = [rng.integers(0, 5, n) for _ in range(K + 10)]
    for a in acts[:10]:
        env.step(a)
    t0 = time.perf_counter()
    for a in acts[10:]:
        env.step(a)
    dt_s = time.perf_counter() - t0
    print(f"PtGVecEnv.step N={n} {dt} {lay} {mode}: {K * n / dt_s:.3e} env-steps/s  ({dt_s / K * 1e6:.1f} us per vector step, host buffers / NumPy dicts included;"
          f" zero_copy={env._copy_out})", flush=True)
    env.close()
