"""How long does the dispatcher take to start all workgroups of a grid?  python tools/entry_probe.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
so = os.path.join(ROOT, "tools", "libclockprobe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "clockprobe.hip")])
P = C.CDLL(so)
P.entry_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
dev = torch.device("cuda", 0)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for n_wg, block, lds in [(256, 256, 36 * 1024), (256, 128, 18 * 1024), (128, 256, 36 * 1024), (512, 64, 9 * 1024), (256, 64, 9 * 1024), (256, 256, 0), (256, 128, 0), (256, 512, 0), (256, 512, 65536), (256, 512, 100 * 1024), (256, 512, 150 * 1024), (512, 256, 75 * 1024), (1024, 128, 37 * 1024),
                         (256, 256, 150 * 1024), (1024, 256, 0), (256, 1024, 150 * 1024)]:
    res = []
    for rep in range(5):
        b = torch.zeros(n_wg, dtype=torch.int64, device=dev)
        P.entry_probe(st, C.c_void_p(b.data_ptr()), n_wg, block, lds)
        torch.cuda.synchronize()
        a = b.cpu().numpy()
        res.append((a.max() - a.min()) / 100.0)
    print(f"{n_wg} workgroups x {block} threads, {lds // 1024} KiB dynamic LDS: first -> last entry {min(res):.2f} .. {max(res):.2f} us", flush=True)
