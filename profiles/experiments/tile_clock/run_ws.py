"""Phase clock of the weights-stationary 1x1 kernel (tile 91): per wave, shader cycles in the K loop (incl. waits for the fragment
loads), in statistics + staging, in LDS read-back + stores; library built by build.sh."""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "..", "tests"))
import diga_amd._lib as L
L.LIB_PATH = os.path.join(HERE, os.environ.get("CLOCK_LIB", "libdiga_hip_clock.so"))
import numpy as np, torch
from gpu_util import DEV, P, call, conv_desc, stream
from bench_conv import SHAPES
name = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 15
H, W, Cin, Cout, stride, k3, dil = SHAPES[name]
g = torch.Generator().manual_seed(1)
x = torch.randn(B * H * W * Cin, generator=g).to(torch.bfloat16).to(DEV)
w = (torch.randn(Cout * Cin, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
shift = torch.zeros(Cout, dtype=torch.float32, device=DEV)
d = conv_desc(B, H, W, Cin, Cout, stride, k3, dil, tile=91)
nu = call("diga_conv_stat_units", C.byref(d))
out = torch.empty(B * H * W * Cout, dtype=torch.bfloat16, device=DEV)
slab = torch.empty(nu * 2 * Cout, dtype=torch.float32, device=DEV)
for _ in range(5):
    call("diga_conv_igemm_bf16", C.byref(d), P(x), P(w), P(out), P(slab), P(shift), stream())
torch.cuda.synchronize()
lib = C.CDLL(L.LIB_PATH)
nblk = min(2048, (256 // (Cout // 128)) * (Cout // 128))
buf = np.zeros((nblk * 4, 16), dtype=np.uint64)       # 8 waves x 8 words per block = 4 rows of 16
assert lib.diga_debug_tile_clock(buf.ctypes.data_as(C.c_void_p), C.c_int(nblk * 4)) == 0
v = buf.reshape(nblk, 8, 8).astype(np.float64)
units = v[:, :, 5]
ok = units > 0
print(f"{name} B={B}: waves with work {ok.sum()}, units per wave {units[ok].mean():.1f}")
for i, n in ((0, "K loop (MFMA + fragment waits)"), (6, "statistics + slab stores"), (1, "statistics + staging writes"), (2, "read-back + stores")):
    print(f"  {n:32s} {np.mean(v[:, :, i][ok] / units[ok]):8.0f} cycles per unit")
life = v[:, :, 4][ok] / 100
print(f"  weights load + barrier before the loop: {v[:, :, 7][ok].mean() / 100:.1f} us; loop time per wave: mean {life.mean():.1f} p10 {np.percentile(life, 10):.1f} p90 {np.percentile(life, 90):.1f} max {life.max():.1f} us")
print(f"  wave lifetime {v[:, :, 3][ok].mean():.0f} cycles = {v[:, :, 4][ok].mean() / 100:.1f} us ({np.mean(v[:, :, 3][ok] / v[:, :, 4][ok] * 100):.0f} MHz); MFMA issue alone = {2 * 8 * (Cin // 32) * 16} cycles per unit")
