"""Is a rollout replayed as a hipGraph slower than the eager launch?  Eager x3, then one captured graph replayed x3 (timing only), each
timed by stream events behind a filler; run under `rocprofv3 --kernel-trace` the dispatch durations of k_rollout_pc tell the same story
from the device's side.  python tools/graph_probe.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rl_ptg_amd.engine import HipEngine
from rl_ptg_amd.prep import synthetic_spec
from rl_ptg_amd.synthetic import sticky_actions_device

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 65536
dev = torch.device("cuda", 0)
spec, _ = synthetic_spec(scenario=1, operation="OP1", eps_len_d=32)
eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
eng.set_episode_plan(spec.eps_ind, n, n)
eng.set_noise_rng(1)
eng.reset()
acts = sticky_actions_device(8 * T, n, seed=7, device=dev, p_switch=1.0 / 12.0)
obs = torch.zeros((T, n, eng.obs_dim), device=dev); rew = torch.zeros((T, n), device=dev); done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
filler = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def timed(fn, label):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        filler.zero_()
    ev0.record(); fn(); ev1.record()
    torch.cuda.synchronize()
    print(f"{label}: {ev0.elapsed_time(ev1) * 1e3:.1f} us", flush=True)


k = 0
for q in range(3):
    timed(lambda: eng.rollout(acts[k * T:(k + 1) * T], obs, rew, done), f"eager {q} ({T} steps)")
    k += 1
eng.profile(True)
eng.rollout(acts[k * T:(k + 1) * T], obs, rew, done); k += 1
print("eager, kernel-attached events:", eng.profile_read_ex(), flush=True)
eng.profile(False)
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        eng.rollout(acts[k * T:(k + 1) * T], obs, rew, done)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
for q in range(3):
    timed(lambda: g.replay(), f"graph replay {q}")
with torch.cuda.stream(side):
    for q in range(2):
        timed(lambda: g.replay(), f"graph replay on the capture stream {q}")
# a raw hipGraph without torch: capture on a plain non-blocking stream through the C ABI only
import ctypes as C
hip = C.CDLL("libamdhip64.so")
st = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
graph, gexec = C.c_void_p(), C.c_void_p()
assert hip.hipStreamBeginCapture(st, 2) == 0          # hipStreamCaptureModeRelaxed
L = eng._L
a = acts[k * T:(k + 1) * T]
rc = L.ptg_rollout(eng._h, C.c_void_p(a.data_ptr()), 0, T, C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()), st)
assert rc == 0
assert hip.hipStreamEndCapture(st, C.byref(graph)) == 0
assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
for q in range(3):
    torch.cuda.synchronize()
    hip.hipEventRecord(e0, st); hip.hipGraphLaunch(gexec, st); hip.hipEventRecord(e1, st)
    hip.hipStreamSynchronize(st)
    ms = C.c_float()
    hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    print(f"raw hipGraphLaunch {q} on its own stream (host latency inside): {ms.value * 1e3:.1f} us", flush=True)
eng.close()
