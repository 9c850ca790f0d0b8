"""Does a memory-bound bn_apply on one HIP stream overlap a compute-bound conv on another?  (experiment, not a test)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from gpu_util import DEV, P, call, conv_desc

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
    return lambda st: call("diga_conv_igemm_bf16", C.byref(d), P(x), P(w), P(out), P(slab), P(shift), C.c_void_p(st.cuda_stream))
def mk_apply(Cc):
    raw = torch.randn(M * Cc, device=DEV).to(torch.bfloat16)
    res = torch.randn(M * Cc, device=DEV).to(torch.bfloat16)
    out = torch.empty_like(raw)
    sc = torch.ones(B * Cc, device=DEV); sh = torch.zeros(B * Cc, device=DEV)
    return lambda st: call("diga_bn_apply", P(raw), P(sc), P(sh), P(res), None, None, P(out), M, Cc, 1, B, C.c_void_p(st.cuda_stream))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(fa, fb, label):
    def timed(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6 / reps
    for _ in range(2):
        ta = timed(lambda: [fa(s1) for _ in range(reps)])
        tb = timed(lambda: [fb(s2) for _ in range(reps)])
        tab = timed(lambda: [(fa(s1), fb(s2)) for _ in range(reps)])
    print(f"{label}: A alone {ta:7.1f} us  B alone {tb:7.1f} us  both streams {tab:7.1f} us  (sum {ta+tb:7.1f}, max {max(ta,tb):7.1f})", flush=True)
for tile in (48, 38, 27, 24):
    run(mk_conv(256, 256, True, 2, tile), mk_apply(1024), f"conv3x3 256->256 tile {tile} || bn_apply<1> C=1024")
run(mk_conv(1024, 256, False, 1, 48), mk_apply(1024), "conv1x1 1024->256 tile 48 || bn_apply<1> C=1024")
run(mk_apply(1024), mk_apply(1024), "bn_apply || bn_apply")
run(mk_conv(256, 256, True, 2, 48), mk_conv(256, 256, True, 2, 48), "conv || conv")
