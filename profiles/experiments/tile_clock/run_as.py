"""Phase clock of the A-stationary fused conv3 (tile 93): per wave and interval, shader cycles in each phase; library built by
build_variant.sh clock -DDIGA_TILE_CLOCK (CLOCK_LIB = its path)."""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "..", "tests"))
import diga_amd._lib as L
L.LIB_PATH = os.path.abspath(os.environ["CLOCK_LIB"])
import numpy as np, torch
from gpu_util import DEV, P, call, conv_desc, stream
H, W, Cin, Cout = 65, 129, 256, 1024
B = int(sys.argv[1]) if len(sys.argv) > 1 else 15
g = torch.Generator().manual_seed(1)
M = B * H * W
raw = torch.randn(M * Cin, generator=g).to(torch.bfloat16).to(DEV)
w = (torch.randn(Cout * Cin, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
sc = (torch.rand(B * Cin, generator=g) + 0.5).to(DEV)
sh = (torch.randn(B * Cin, generator=g) * 0.3).to(DEV)
shift = torch.zeros(Cout, dtype=torch.float32, device=DEV)
out = torch.empty(M * Cout, dtype=torch.bfloat16, device=DEV)
d = conv_desc(B, H, W, Cin, Cout, 1, False, 1, tile=93)
slab = torch.empty(call("diga_conv_stat_units", C.byref(d)) * 2 * Cout, dtype=torch.float32, device=DEV)
for _ in range(5):
    call("diga_conv_igemm_bf16_fused_in", C.byref(d), P(raw), P(sc), P(sh), None, None, B, P(w), P(out), P(slab), P(shift), stream())
torch.cuda.synchronize()
lib = C.CDLL(L.LIB_PATH)
nblk = 256
buf = np.zeros((nblk * 4, 16), dtype=np.uint64)
assert lib.diga_debug_tile_clock(buf.ctypes.data_as(C.c_void_p), C.c_int(nblk * 4)) == 0
v = buf.reshape(nblk, 8, 8).astype(np.float64)
names = ["wait + barrier", "DMA issue", "unit load issue", "late epilogue", "transform (waits for the loads)", "MFMA slice", "early epilogue"]
for grp, sel in (("early waves 0-3", slice(0, 4)), ("late waves 4-7", slice(4, 8))):
    q = v[:, sel, 7]
    ok = q > 0
    print(grp, "intervals per wave", q[ok].mean())
    for i, n in enumerate(names):
        print(f"   {n:34s} {np.mean(v[:, sel, i][ok] / q[ok]):8.0f} cycles per interval")
    print(f"   total {np.mean(v[:, sel, :7].sum(-1)[ok] / q[ok]):8.0f}")
