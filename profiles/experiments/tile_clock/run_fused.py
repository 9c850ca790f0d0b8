"""Phase clock of the fused-residual-input conv1 (AMODE 2) of layer3 at B frames:  python run_fused.py TILE [B]"""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "..", "tests"))
import diga_amd._lib as L
L.LIB_PATH = os.path.join(HERE, os.environ.get("CLOCK_LIB", "libdiga_hip_clock.so"))
import numpy as np, torch
from gpu_util import DEV, P, call, conv_desc, stream
tile = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 15
H, W, Cin, Cout = 65, 129, 1024, 256
g = torch.Generator().manual_seed(1)
raw = torch.randn(B * H * W * Cin, generator=g).to(torch.bfloat16).to(DEV)
res = torch.randn(B * H * W * Cin, generator=g).to(torch.bfloat16).to(DEV)
act = torch.empty_like(raw)
sc = (torch.rand(B * Cin, generator=g) + 0.5).to(DEV); sh = torch.randn(B * Cin, generator=g).to(DEV)
w = (torch.randn(Cout * Cin, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
shift = torch.zeros(Cout, dtype=torch.float32, device=DEV)
d = conv_desc(B, H, W, Cin, Cout, 1, False, 1, tile=tile)
nu = call("diga_conv_stat_units", C.byref(d))
out = torch.empty(B * H * W * Cout, dtype=torch.bfloat16, device=DEV)
slab = torch.empty(nu * 2 * Cout, dtype=torch.float32, device=DEV)
def launch():
    call("diga_conv_igemm_bf16_fused_in", C.byref(d), P(raw), P(sc), P(sh), P(res), P(act), B, P(w), P(out), P(slab), P(shift), stream())
for _ in range(5): launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): launch()
e1.record(); torch.cuda.synchronize()
print(f"fused conv1 tile {tile} B={B}: {e0.elapsed_time(e1) * 100:.1f} us per launch")
lib = C.CDLL(L.LIB_PATH)
from bench_conv import TILE_MN
bm, bn = TILE_MN[tile % 10]
nblk = B * -(-(H * W) // bm) * (Cout // bn)
buf = np.zeros((min(nblk, 16384), 16), dtype=np.uint64)
assert lib.diga_debug_tile_clock(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.shape[0])) == 0
t = buf[:, :8].astype(np.int64); us = (t - t[:, 0].min()) / 100.0
names = ["entry->prologue issued", "->(unused)", "->K loop done", "->stats", "->LDS staged", "->stores issued", "->stores done"]
for i, n in enumerate(names):
    dlt = us[:, i + 1] - us[:, i]
    print(f"  {n:28s} mean {dlt.mean():7.2f}  p10 {np.percentile(dlt, 10):7.2f}  p90 {np.percentile(dlt, 90):7.2f} us")
kl = us[:, 3] - us[:, 2]
print(f"  K loop: {np.mean(buf[:, 10].astype(np.float64) / kl):.0f} MHz, {kl.mean() / (Cin // 64):.3f} us per 64-channel step; blocks {nblk}, span {us[:, 7].max():.1f} us")
