"""Floor of a launch-per-step path: K dependent launches of the step kernel's shape (256 x 256 threads) replayed as one hipGraph, moving
nothing / writing the step's bytes / reading one sixth and writing the rest.  python tools/chain_probe.py"""
import ctypes as C, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
so = os.path.join(ROOT, "tools", "libclockprobe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "clockprobe.hip")])
P = C.CDLL(so)
P.chain_probe.restype = C.c_double
P.chain_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
a = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
b = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for name, kind, nbytes in [("empty kernel", 0, 0), ("write 13.96 MB (213 B x 65 536: ptg_step's compulsory bytes, float32)", 1, 213 * 65536),
                           ("read 1/6 + write 13.96 MB", 2, 213 * 65536), ("write 23.4 MB (357 B x 65 536: float64)", 1, 357 * 65536),
                           ("write 27.9 MB (213 B x 131 072)", 1, 213 * 131072)]:
    r = [P.chain_probe(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), nbytes // 16 * 16, kind, 1000) for _ in range(3)]
    print(f"{name}: {min(r):.2f} .. {max(r):.2f} us per launch" + (f"  = {nbytes / min(r) / 1e3:.0f} GB/s" if nbytes else ""), flush=True)
