import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
import numpy as np, torch, torch.distributed as dist
from rl_ptg_amd import dist as ptg_dist
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
side = torch.cuda.Stream(device=dev)
def t(f, n=20):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    return np.median(ts[5:])
z = np.zeros(0); zi = np.zeros(0, np.int64)
seq = []
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(side): ptg_dist.all_gather_finished(z, zi, device=dev)
    seq.append((time.perf_counter() - t0) * 1e6)
print("first six calls on the side stream: " + " ".join("%.0f" % v for v in seq) + " us")
print("all_gather_finished (empty), current stream: %.1f us" % t(lambda: ptg_dist.all_gather_finished(z, zi, device=dev)))
def on_side():
    with torch.cuda.stream(side): ptg_dist.all_gather_finished(z, zi, device=dev)
print("all_gather_finished (empty), side stream: %.1f us" % t(on_side))
print("torch.tensor([[0]], device) : %.1f us" % t(lambda: torch.tensor([[0]], dtype=torch.int64, device=dev)))
x = torch.zeros(1, dtype=torch.int64, device=dev); out = torch.zeros(1, dtype=torch.int64, device=dev)
print("all_gather_into_tensor: %.1f us" % t(lambda: dist.all_gather_into_tensor(out, x)))
print("out.cpu(): %.1f us" % t(lambda: out.cpu()))
print("barrier: %.1f us" % t(lambda: dist.barrier()))
ph = torch.zeros(1, dtype=torch.int64).pin_memory()
def pinned():
    ph[0] = 3; x.copy_(ph, non_blocking=True); dist.all_gather_into_tensor(out, x); ph.copy_(out, non_blocking=True); torch.cuda.current_stream().synchronize()
print("pinned H2D + all_gather + pinned D2H + stream sync: %.1f us" % t(pinned))
dist.destroy_process_group()
