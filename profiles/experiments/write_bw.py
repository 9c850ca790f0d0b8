"""How fast can the chip WRITE?  (the floor of conv3, whose raw output is 4x its input)"""
import torch
dev = torch.device("cuda:0")
for mb in (257, 1024):
    x = torch.empty(mb * 1024 * 1024 // 2, dtype=torch.bfloat16, device=dev)
    y = torch.empty_like(x)
    for name, fn, nbytes in (("fill", lambda: x.zero_(), x.numel() * 2), ("copy", lambda: y.copy_(x), 2 * x.numel() * 2)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{name} {mb} MiB: {us:.1f} us, {nbytes / us / 1e6:.2f} TB/s", flush=True)
