"""Per-tile phase clock of the conv kernel (library built by build.sh with -DDIGA_TILE_CLOCK).

    python profiles/experiments/tile_clock/run.py SHAPE TILE [BATCH]
Prints, over the blocks of ONE launch: the time between phase marks (us, 100 MHz clock), and per CU the gap between one block's
end and the next block's start.
"""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "..", "tests"))
import diga_amd._lib as L
L.LIB_PATH = os.path.join(HERE, os.environ.get("CLOCK_LIB", "libdiga_hip_clock.so"))
import numpy as np, torch
from gpu_util import DEV, P, call, conv_desc, stream
from bench_conv import SHAPES, TILE_MN

name, tile = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 15
H, W, Cin, Cout, stride, k3, dil = SHAPES[name]
g = torch.Generator().manual_seed(1)
x = torch.randn(B * H * W * Cin, generator=g).to(torch.bfloat16).to(DEV)
taps = 9 if k3 else 1
w = (torch.randn(Cout * taps * Cin, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
shift = torch.zeros(Cout, dtype=torch.float32, device=DEV)
d = conv_desc(B, H, W, Cin, Cout, stride, k3, dil, tile=tile)
bm, bn = TILE_MN[tile % 10]
M = B * d.Ho * d.Wo
nu = call("diga_conv_stat_units", C.byref(d))
out = torch.empty(M * Cout, dtype=torch.bfloat16, device=DEV)
slab = torch.empty(nu * 2 * Cout, dtype=torch.float32, device=DEV)
nblk = B * -(-(d.Ho * d.Wo) // bm) * (Cout // bn)
for _ in range(5):
    call("diga_conv_igemm_bf16", C.byref(d), P(x), P(w), P(out), P(slab), P(shift), stream())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
call("diga_conv_igemm_bf16", C.byref(d), P(x), P(w), P(out), P(slab), P(shift), stream())
e1.record()
torch.cuda.synchronize()
lib = C.CDLL(L.LIB_PATH)
buf = np.zeros((min(nblk, 16384), 16), dtype=np.uint64)
assert lib.diga_debug_tile_clock(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.shape[0])) == 0
t = buf[:, :8].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["entry->prologue issued", "->first K step landed", "->K loop done", "->stats", "->LDS staged", "->stores issued", "->stores done"]
print(f"{name} tile {tile} B={B}: {nblk} blocks, launch {e0.elapsed_time(e1) * 1e3:.1f} us (event pair incl. ~5 us), span by clock {us[:, 7].max():.1f} us")
for i, n in enumerate(names):
    dlt = us[:, i + 1] - us[:, i]
    print(f"  {n:28s} mean {dlt.mean():7.2f}  p10 {np.percentile(dlt, 10):7.2f}  p90 {np.percentile(dlt, 90):7.2f} us")
kl = us[:, 3] - us[:, 2]
print(f"  K loop: shader clock {np.mean(buf[:, 10].astype(np.float64) / kl):.0f} MHz (s_memtime cycles / 100 MHz wall clock), {kl.mean() / max(1, d.Cin * taps // 64):.3f} us per 64-channel K step")
tot = us[:, 7] - us[:, 0]
print(f"  block lifetime               mean {tot.mean():7.2f}  p10 {np.percentile(tot, 10):7.2f}  p90 {np.percentile(tot, 90):7.2f} us")
hw = buf[:, 8].astype(np.int64); xcc = buf[:, 9].astype(np.int64) & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)     # cu_id, sh_id, se_id, xcc
gaps = []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    idx = idx[np.argsort(us[idx, 0])]
    for a, b in zip(idx[:-1], idx[1:]):
        gaps.append(us[b, 0] - us[a, 7])
print(f"  distinct CU keys {len(np.unique(cu))}; first-block start spread p90 {np.percentile(us[:, 0], 10):.2f}..; blocks per CU {nblk / len(np.unique(cu)):.2f}")
if gaps:
    gaps = np.array(gaps)
    print(f"  gap block end -> next block entry on the same CU: mean {gaps.mean():.2f} p10 {np.percentile(gaps, 10):.2f} p90 {np.percentile(gaps, 90):.2f} us  (negative = co-resident)")
print(f"  first entry {us[:, 0].min():.2f}, median entry of the first 256 blocks {np.median(np.sort(us[:, 0])[:256]):.2f}, last end {us[:, 7].max():.2f}")
