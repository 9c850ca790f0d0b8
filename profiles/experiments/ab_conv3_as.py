"""A/B (round 3): conv3 of a bottleneck as  diga_bn_apply(raw2 -> act2) + diga_conv_igemm_bf16 (tile 91, weights-stationary)
against  diga_conv_igemm_bf16_fused_in (tile 93, A-stationary, transform fused), 15 frames, isolated launches, HIP events.
    python profiles/experiments/ab_conv3_as.py [--batch 15]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, P, call, conv_desc, stream  # noqa: E402

SHAPES = {"l1.c3": (129, 257, 64, 256), "l2.c3": (65, 129, 128, 512), "l3.c3": (65, 129, 256, 1024)}


def timed(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=15)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
B = a.batch
for name, (H, W, Cin, Cout) in SHAPES.items():
    g = torch.Generator().manual_seed(1)
    M = B * H * W
    raw = torch.randn(M * Cin, generator=g).to(torch.bfloat16).to(DEV)
    act = torch.empty_like(raw)
    w = (torch.randn(Cout * Cin, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    sc = (torch.rand(B * Cin, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(B * Cin, generator=g) * 0.3).to(DEV)
    shift = torch.zeros(Cout, dtype=torch.float32, device=DEV)
    out = torch.empty(M * Cout, dtype=torch.bfloat16, device=DEV)
    d91, d93 = conv_desc(B, H, W, Cin, Cout, 1, False, 1, tile=91), conv_desc(B, H, W, Cin, Cout, 1, False, 1, tile=93)
    slab = torch.empty(call("diga_conv_stat_units", C.byref(d91)) * 2 * Cout, dtype=torch.float32, device=DEV)
    t_apply = timed(lambda: call("diga_bn_apply", P(raw), P(sc), P(sh), None, None, None, P(act), M, Cin, 1, B, stream()), a.reps)
    t_ws = timed(lambda: call("diga_conv_igemm_bf16", C.byref(d91), P(act), P(w), P(out), P(slab), P(shift), stream()), a.reps)
    t_as = timed(lambda: call("diga_conv_igemm_bf16_fused_in", C.byref(d93), P(raw), P(sc), P(sh), None, None, B, P(w), P(out), P(slab), P(shift),
                              stream()), a.reps)
    fl = 2.0 * M * Cin * Cout
    print(f"{name}: bn_apply {t_apply:6.1f} us + ws(91) {t_ws:6.1f} us = {t_apply + t_ws:6.1f} us | as(93) {t_as:6.1f} us "
          f"({fl / t_as / 1e6:5.0f} TFLOP/s; ws alone {fl / t_ws / 1e6:5.0f})", flush=True)
