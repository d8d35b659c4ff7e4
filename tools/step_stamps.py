"""Phases of one k_step_hot launch from in-kernel 100 MHz stamps (diagnostic build: hipcc ... -DPTG_STAMPS -o tools/libptg_stamps.so
rl_ptg_amd/csrc/ptg_env.hip, loaded through PTG_LIB_PATH=tools/libptg_stamps.so): wave 0 of every workgroup stamps 0 entry, 1 past the LDS-stage barrier, 2 state + action arrived,
3 look-up done and record gather issued, 4 record arrived / reward done, 5 stores issued, 6 stores retired."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rl_ptg_amd import _lib
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device
n = 65536
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout=os.environ.get("TS_LAYOUT", "row"))
eng.set_episode_plan(spec.eps_ind, n, n)
eng.set_noise_rng(1)
acts = sticky_actions_device(480, n, seed=1, device=dev)
eng.reset()
eng.rollout(acts[:400])
L = _lib.lib()
L.ptg_debug_stamps.argtypes = [C.c_void_p]
names = ["entry", "barrier", "state in", "lut done", "rec in", "stores issued", "retired"]
for rep in range(6):
    for t in range(8):                                        # back-to-back launches; the stamps of the last one survive
        eng.step(acts[400 + rep * 8 + t], want_final=False)
    eng.sync()
    buf = np.zeros((256, 2, 8), np.uint64)
    L.ptg_debug_stamps(buf.ctypes.data_as(C.c_void_p))
    t = buf[:, 0, :7].astype(np.int64)
    rel = (t - t[:, 0].min()) / 100.0
    print(f"rep {rep}: us since first entry, median over workgroups [min..max]: " +
          "  ".join(f"{names[q]} {np.median(rel[:, q]):.2f} [{rel[:, q].min():.2f}..{rel[:, q].max():.2f}]" for q in range(7)), flush=True)
eng.close()
