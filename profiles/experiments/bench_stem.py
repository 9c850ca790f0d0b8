"""Time of the stem launch at B frames of 1024x512:  python profiles/experiments/bench_stem.py [B]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import torch
from gpu_util import DEV, P, call, stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 15
H, W = 512, 1024
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, 3, H, W, generator=g) * 50).to(DEV)
wt = (torch.randn(64, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5).to(DEV)
shift = torch.zeros(64, device=DEV)
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
nu = call("diga_stem_conv7x7s2", None, None, None, None, None, B, H, W, None)
out = torch.empty(B * Ho * Wo * 64, dtype=torch.bfloat16, device=DEV)
slab = torch.empty(nu * 2 * 64, dtype=torch.float32, device=DEV)
def launch():
    call("diga_stem_conv7x7s2", P(x), P(wt), P(out), P(slab), P(shift), B, H, W, stream())
for _ in range(3): launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): launch()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 50
gf = 2.0 * B * Ho * Wo * 147 * 64 / 1e9
gb = (x.numel() * 4 + out.numel() * 2) / 1e9
print(f"stem B={B}: {us:.1f} us per launch, {gf / us * 1e3:.1f} TFLOP/s (algorithmic), {gb / us * 1e6:.0f} GB/s")
