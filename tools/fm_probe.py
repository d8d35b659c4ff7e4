"""Feature-major store pattern vs column stride and column order: python tools/fm_probe.py  (see k_fm in tools/clockprobe.hip)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
so = os.path.join(ROOT, "tools", "libclockprobe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "clockprobe.hip")])
P = C.CDLL(so)
P.fm_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
P.fm_probe.restype = C.c_double
dev = torch.device("cuda", 0)
F, T = 35, 200
buf = torch.zeros(T * F * 65536 * 8 // 4, dtype=torch.float32, device=dev)       # room for the largest case
torch.cuda.synchronize()
for W in (8, 4):
    for N in (65536, 65280, 61440, 49152):
        line = f"W = {W} B, N = {N:6d} (column stride {N * W:7d} B):"
        for rot in (0, 1, 3, 9, 12):
            us = P.fm_probe(C.c_void_p(buf.data_ptr()), N, F, T, W, rot)
            line += f"  rot {rot}: {us:6.3f} us/step = {N * F * W / us / 1e6:5.2f} TB/s"
        print(line, flush=True)
