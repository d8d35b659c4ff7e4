"""Why are the first few hundred fused steps after an idle period slow?  python tools/coldstart.py [envs] [steps_per_launch]

Separates the candidate causes of the round-1 "cold start" (Next #1) with the evidence each predicts:
  clocks / power state   a pure write stream (no tables, no state) shows the same ramp from idle; an in-kernel
                         s_memtime / s_memrealtime probe reads a lower shader clock; a busy burst right before removes it
  caches (L2 / MALL)     the ramp follows the data: replaying the SAME steps from the SAME state is fast the second time
  env state mix          the ramp follows the state: restoring a saved state and replaying is slow again even when hot
Everything is timed with HIP events on the launch stream; nothing here is product code.
"""
import ctypes as C
import glob
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rl_ptg_amd import dist as ptg_dist
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = int(sys.argv[2]) if len(sys.argv) > 2 else 25
IDLE = float(os.environ.get("CS_IDLE", "0.3"))
dev = torch.device("cuda", 0)

so = os.path.join(ROOT, "tools", "libclockprobe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "clockprobe.hip")])
P = C.CDLL(so)
P.clk_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
P.bw_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
P.bw_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]


def stream():
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def dpm(tag):
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_*clk")):
        try:
            act = [l.strip() for l in open(f).read().splitlines() if "*" in l]
            out.append(os.path.basename(f)[7:] + "=" + ("|".join(act) if act else "?"))
        except OSError as e:
            out.append(os.path.basename(f) + "=ERR")
    print(f"[dpm {tag}] " + "  ".join(out), flush=True)


NWG = 256
clk_bufs = []


def clk_probe():
    """enqueue a ~4 us probe; returns a thunk that reads the median shader clock (MHz) after a sync"""
    b = torch.zeros(NWG * 4, dtype=torch.int64, device=dev)
    clk_bufs.append(b)
    P.clk_probe(stream(), C.c_void_p(b.data_ptr()), NWG, 6000)

    def read():
        a = b.cpu().numpy().reshape(NWG, 4).astype(np.float64)
        dt, dr = a[:, 1] - a[:, 0], a[:, 3] - a[:, 2]
        return float(np.median(dt / np.maximum(dr, 1)) * 100.0)
    return read


def idle(sec=IDLE):
    torch.cuda.synchronize()
    time.sleep(sec)


big = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
sink = torch.zeros(4, dtype=torch.float32, device=dev)


def prime(ms):
    """keep the chip busy with a write stream for about `ms` milliseconds (256 MiB per launch ~ 45 us)"""
    for _ in range(max(1, int(ms * 1000 / 45))):
        P.bw_write(stream(), C.c_void_p(big.data_ptr()), big.numel())


def timed_series(fn, count, probe_every=1):
    evs, clks = [], []
    for q in range(count):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(q)
        e1.record()
        evs.append((e0, e1))
        if q % probe_every == 0:
            clks.append(clk_probe())
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 for a, b in evs], [c() for c in clks]


def fmt(v, f="%.1f"):
    return " ".join(f % x for x in v)


print(f"N = {n}, {S} steps per launch, idle = {IDLE} s", flush=True)
subprocess.call("rocm-smi --showperflevel --showclocks 2>/dev/null | grep -v '^$' | head -40", shell=True)
idle(); dpm("idle")

# ---- E1: a pure write stream from idle
for rep in range(2):
    idle()
    us, clk = timed_series(lambda q: P.bw_write(stream(), C.c_void_p(big.data_ptr()), big.numel()), 48, probe_every=4)
    print(f"E1 write 256 MiB x48 from idle (rep {rep}): us " + fmt(us), flush=True)
    print("   GB/s " + fmt([big.numel() / u / 1e3 for u in us], "%.0f"))
    print("   shader clock MHz (every 4th) " + fmt(clk, "%.0f"), flush=True)
dpm("after write burst")
idle()
us, clk = timed_series(lambda q: P.bw_read(stream(), C.c_void_p(big.data_ptr()), big.numel(), C.c_void_p(sink.data_ptr())), 48, probe_every=4)
print("E1r read 256 MiB x48 from idle: GB/s " + fmt([big.numel() / u / 1e3 for u in us], "%.0f"))
print("   shader clock MHz " + fmt(clk, "%.0f"), flush=True)

# ---- engine
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
first_ptr, stride = ptg_dist.episode_plan(n, 1, 0)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout=os.environ.get("CS_LAYOUT", "feature"))
eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
eng.set_noise_rng(seed=20250614)
NSEG = 16
actions = sticky_actions_device(64 * S + 1024, n, seed=1234, device=dev, p_switch=1.0 / 12.0)
F = eng.obs_dim
oshape = (S, F, n) if eng.feature_major else (S, n, F)
obs = torch.zeros(oshape, dtype=torch.float32, device=dev)
rew = torch.zeros((S, n), dtype=torch.float32, device=dev)
done = torch.zeros((S, n), dtype=torch.uint8, device=dev)


def seg(a0):
    return lambda q: eng.rollout(actions[a0 + q * S:a0 + (q + 1) * S], obs, rew, done)


def report(tag, us, clk):
    print(f"{tag}: us/step " + fmt([u / S for u in us], "%.2f"))
    print("   shader clock MHz " + fmt(clk, "%.0f"), flush=True)


eng.reset(); idle()
report("E2 fresh engine, from reset, after idle", *timed_series(seg(0), NSEG))
eng.reset(); idle(); prime(5.0)
report("E3 from reset, after idle + 5 ms write burst", *timed_series(seg(0), NSEG))
eng.reset()
report("E4 from reset, no idle (right after E3)", *timed_series(seg(0), NSEG))
# continue to a stationary state mix
for q in range(NSEG, 40):
    seg(0)(q)
eng.sync()
sd = eng.state_dict()
a_st = 40 * S
report("E5a stationary state, hot (right after 600 more steps)", *timed_series(seg(a_st), 8))
eng.load_state_dict(sd); idle()
report("E5b SAME state + SAME actions replayed after idle", *timed_series(seg(a_st), 8))
eng.load_state_dict(sd); idle(); prime(5.0)
report("E5c SAME state + actions after idle + 5 ms write burst", *timed_series(seg(a_st), 8))
eng.load_state_dict(sd); idle(); prime(5.0); idle(0.0)
report("E5d ... burst, then a host-side sync (no sleep)", *timed_series(seg(a_st), 8))

# ---- E6: how much priming is needed / E7: how fast does it decay
for ms in (0.1, 0.3, 1.0, 3.0, 10.0, 30.0):
    eng.load_state_dict(sd); idle(); prime(ms)
    us, clk = timed_series(seg(a_st), 2)
    print(f"E6 prime {ms:5.1f} ms -> first launches us/step {us[0] / S:.2f} {us[1] / S:.2f}  clk {clk[0]:.0f}", flush=True)
for gap in (0.0, 0.0002, 0.001, 0.005, 0.02, 0.1, 0.5):
    eng.load_state_dict(sd); prime(10.0); torch.cuda.synchronize(); time.sleep(gap)
    us, clk = timed_series(seg(a_st), 2)
    print(f"E7 prime 10 ms, host gap {gap * 1e3:6.1f} ms -> us/step {us[0] / S:.2f} {us[1] / S:.2f}  clk {clk[0]:.0f}", flush=True)
dpm("end")
eng.close()
