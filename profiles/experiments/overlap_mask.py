"""conv on a CU-masked stream || bn_apply on the complementary CU mask (experiment)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from gpu_util import DEV, P, call, conv_desc

torch.cuda.init(); torch.zeros(1, device=DEV)
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return st
B, H, W = 15, 65, 129
M = B * H * W
reps = 30
def mk_conv(Cin, Cout, k3, dil, tile):
    x = torch.randn(M * Cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(Cout * (9 if k3 else 1) * Cin, device=DEV) * 0.05).to(torch.bfloat16)
    d = conv_desc(B, H, W, Cin, Cout, 1, k3, dil, tile=tile)
    out = torch.empty(M * Cout, dtype=torch.bfloat16, device=DEV)
    nu = call("diga_conv_stat_units", C.byref(d))
    slab = torch.empty(nu * 2 * Cout, dtype=torch.float32, device=DEV)
    shift = torch.zeros(Cout, dtype=torch.float32, device=DEV)
    return lambda st: call("diga_conv_igemm_bf16", C.byref(d), P(x), P(w), P(out), P(slab), P(shift), st)
def mk_apply(Cc):
    raw = torch.randn(M * Cc, device=DEV).to(torch.bfloat16)
    res = torch.randn(M * Cc, device=DEV).to(torch.bfloat16)
    out = torch.empty_like(raw)
    sc = torch.ones(B * Cc, device=DEV); sh = torch.zeros(B * Cc, device=DEV)
    return lambda st: call("diga_bn_apply", P(raw), P(sc), P(sh), P(res), None, None, P(out), M, Cc, 1, B, st)
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6 / reps
conv = mk_conv(256, 256, True, 2, 48)
conv1 = mk_conv(1024, 256, False, 1, 48)
app = mk_apply(1024)
for name, abits in (("apply on bits 0..63", set(range(64))), ("apply on bits 0..31", set(range(32))), ("apply on every 4th bit", set(range(0, 256, 4))),
                    ("apply on every 8th bit", set(range(0, 256, 8))), ("apply on bits 0..95", set(range(96)))):
    sa = masked_stream(abits)
    sc_ = masked_stream(set(range(256)) - abits)
    for _ in range(2):
        ta = timed(lambda: [app(sa) for _ in range(reps)])
        tc = timed(lambda: [conv(sc_) for _ in range(reps)])
        tc1 = timed(lambda: [conv1(sc_) for _ in range(reps)])
        tb = timed(lambda: [(conv(sc_), app(sa)) for _ in range(reps)])
        tb1 = timed(lambda: [(conv1(sc_), app(sa)) for _ in range(reps)])
    print(f"{name} ({len(abits)} CUs): apply alone {ta:6.1f} us | conv3x3 on the rest {tc:6.1f}, both {tb:6.1f} | conv1x1 on the rest {tc1:6.1f}, both {tb1:6.1f}", flush=True)
