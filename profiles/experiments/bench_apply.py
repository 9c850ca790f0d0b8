"""bn_apply launch time at one-frame sizes (and at 15 frames) for a given library build:  python bench_apply.py"""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import torch
from gpu_util import DEV, P, call, stream
for B in (1, 15):
    for rows, Cc, res in ((33153, 64, False), (33153, 256, True), (8385, 128, False), (8385, 256, False), (8385, 512, False), (8385, 1024, True), (8385, 2048, True)):
        M = rows * B
        raw = torch.randn(M * Cc, device=DEV).to(torch.bfloat16); r = torch.randn(M * Cc, device=DEV).to(torch.bfloat16) if res else None
        out = torch.empty_like(raw)
        sc = torch.rand(B * Cc, device=DEV) + 0.5; sh = torch.randn(B * Cc, device=DEV)
        f = lambda: call("diga_bn_apply", P(raw), P(sc), P(sh), P(r) if res else None, None, None, P(out), M, Cc, 1, B if B > 1 else 0, stream())
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        nb = M * Cc * 2 * (3 if res else 2)
        print(f"B={B:2d} rows {rows:6d} C {Cc:5d} res {int(res)}: {us:7.2f} us  {nb / us / 1e3:7.1f} GB/s", flush=True)
