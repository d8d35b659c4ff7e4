"""Phases of one fused-rollout launch from in-kernel 100 MHz stamps (diagnostic build -DPTG_STAMPS, loaded through PTG_LIB_PATH):
python tools/stamps.py [T].  Stamps per workgroup, wave 0 (producer) / first consumer wave: 0 kernel entry, 1 actions staged,
2 LDS staging issued, 3 past the barrier, 4 first hand-off (producer: step 0 produced; consumer: step 0 requested), 5 second step
(consumer: first finish issued), 6 loop done (producer: before the last barrier), 7 end (consumer: all stores retired)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rl_ptg_amd import _lib
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 65536
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout=os.environ.get("TS_LAYOUT", "row"))
eng.set_episode_plan(spec.eps_ind, n, n)
eng.set_noise_rng(1)
acts = sticky_actions_device(400 + 8 * T, n, seed=1, device=dev)
eng.reset()
eng.rollout(acts[:400])
F = eng.obs_dim
obs = torch.zeros((T, n, F) if not eng.feature_major else (T, F, n), device=dev)
rew = torch.zeros((T, n), device=dev); done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
L = _lib.lib()
L.ptg_debug_stamps.argtypes = [C.c_void_p]
for rep in range(4):
    eng.rollout(acts[400 + rep * T:400 + (rep + 1) * T], obs, rew, done)
    eng.sync()
    torch.cuda.synchronize()
    buf = np.zeros((256, 2, 8), np.uint64)
    L.ptg_debug_stamps(buf.ctypes.data_as(C.c_void_p))
    t = buf.astype(np.int64)
    t0 = t[:, :, 0].min()
    rel = (t - t0) / 100.0                                  # us since the first workgroup entered
    names = ["entry", "acts staged", "lds issued", "past barrier", "1st handoff", "2nd", "loop done", "end"]
    print(f"rep {rep}: T = {T}; us since first entry, median over workgroups [min..max]")
    for role, rn in ((0, "producer"), (1, "consumer")):
        print("  " + rn + ": " + "  ".join(f"{names[q]} {np.median(rel[:, role, q]):.2f} [{rel[:, role, q].min():.2f}..{rel[:, role, q].max():.2f}]" for q in range(8)))
    for role, rn in ((0, "producer"), (1, "consumer")):     # the XCDs' counters are offset against each other: differences inside a workgroup
        d = (t[:, role, :] - t[:, role, 0:1]) / 100.0
        print("  " + rn + " since own entry: " + "  ".join(f"{names[q]} +{np.median(d[:, q]):.2f}" for q in range(1, 8)))
    if rep == 3:
        ent = rel[:, 0, 0]
        print("  entry by blockIdx (every 8th = one XCD): " + " ".join("%.2f" % ent[b] for b in range(0, 256, 8)))
        print("  entry of blocks 0..15: " + " ".join("%.2f" % ent[b] for b in range(16)))
eng.close()
