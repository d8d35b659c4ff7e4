import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTG_DEBUG_FLAGS"] = "16"
import numpy as np, torch
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device
n, T = 65536, 200
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, out_dtype="float32", obs_layout="feature")
eng.set_episode_plan(spec.eps_ind, n, n); eng.set_noise_rng(1)
a = sticky_actions_device(T, n, 1, torch.device("cuda"))
eng.reset()
eng.rollout(a); eng.sync()
eng.rollout(a); eng.sync()
out = np.zeros(64, np.int64)
eng._L.ptg_debug_read_counters(eng._h, out.ctypes.data_as(C.POINTER(C.c_int64)))
for w in range(8):
    tA, tB, tC, TT = out[w*4:w*4+4]
    if TT: print(f"wave {w} ({'producer' if w < 4 else 'consumer'}): phaseA {tA/TT:.0f}  work(total before barrier) {tB/TT:.0f}  barrier wait {tC/TT:.0f}  cycles/step (memtime ticks)")
