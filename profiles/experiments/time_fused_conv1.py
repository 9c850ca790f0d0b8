"""Launch time of the fused-residual-input conv1 of layer3 (AMODE 2, tile 27) at B frames with the library given by with_lib.py:
    python profiles/experiments/with_lib.py LIB.so profiles/experiments/time_fused_conv1.py [B]"""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import torch
from gpu_util import DEV, P, call, conv_desc, stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 15
H, W, Cin, Cout = 65, 129, 1024, 256
g = torch.Generator().manual_seed(1)
raw = torch.randn(B * H * W * Cin, generator=g).to(torch.bfloat16).to(DEV)
res = torch.randn(B * H * W * Cin, generator=g).to(torch.bfloat16).to(DEV)
act = torch.empty_like(raw)
sc = (torch.rand(B * Cin, generator=g) + 0.5).to(DEV); sh = torch.randn(B * Cin, generator=g).to(DEV)
w = (torch.randn(Cout * Cin, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
shift = torch.zeros(Cout, dtype=torch.float32, device=DEV)
d = conv_desc(B, H, W, Cin, Cout, 1, False, 1, tile=27)
nu = call("diga_conv_stat_units", C.byref(d))
out = torch.empty(B * H * W * Cout, dtype=torch.bfloat16, device=DEV)
slab = torch.empty(nu * 2 * Cout, dtype=torch.float32, device=DEV)
def launch():
    call("diga_conv_igemm_bf16_fused_in", C.byref(d), P(raw), P(sc), P(sh), P(res), P(act), B, P(w), P(out), P(slab), P(shift), stream())
for _ in range(5): launch()
torch.cuda.synchronize()
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): launch()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 50)
byts = B * H * W * (Cin * 2 * 3 + Cout * 2)
print(f"fused conv1 tile 27 B={B}: {min(ts):.1f} us per launch ({byts / min(ts) / 1e6:.2f} TB/s of algorithmic bytes)  runs {[round(t, 1) for t in ts]}")
