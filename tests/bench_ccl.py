"""Timing of the optional connected-component / instance-pooling kernels (not a pytest file).

    python tests/bench_ccl.py [--reps 20]
Workloads follow BASELINE.json configs[4] ("high instance counts"): label maps with >= 2000 components per frame at the
feature grids of the 1280x720 / 1024x512 / 2048x1024 crops and at full label resolution.  Prints the mean time per call
(HIP events over `reps` back-to-back calls) and the algorithmic bandwidth: labels read + instance map written (8 B per
pixel) for the labelling, one read of the bf16 feature map for the pooling.
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from diga_amd import instances
from diga_amd._lib import call


def labels(rng, B, h, w, Cn, cell, noise):
    coarse = rng.integers(0, Cn, (B, -(-h // cell), -(-w // cell)))
    lab = np.repeat(np.repeat(coarse, cell, 1), cell, 2)[:, :h, :w].copy()
    flip = rng.random((B, h, w)) < noise
    lab[flip] = rng.integers(0, Cn, int(flip.sum()))
    lab[rng.random((B, h, w)) < 0.04] = 255
    return torch.from_numpy(lab.astype(np.int32)).cuda()


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    P = lambda t: C.c_void_p(t.data_ptr())
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    Cn = 19
    print("labelling (8-connectivity)")
    for name, B, h, w, cell, noise in (("feat 65x129 x1", 1, 65, 129, 4, 0.10), ("feat 65x129 x15", 15, 65, 129, 4, 0.10),
                                       ("feat 91x161 x1", 1, 91, 161, 4, 0.10), ("feat 129x257 x4", 4, 129, 257, 5, 0.08),
                                       ("full 512x1024 x1", 1, 512, 1024, 12, 0.004), ("full 720x1280 x1", 1, 720, 1280, 14, 0.003),
                                       ("full 1024x2048 x4", 4, 1024, 2048, 20, 0.002)):
        lab = labels(rng, B, h, w, Cn, cell, noise)
        work = torch.empty(call("diga_ccl_work_bytes", B, h, w, Cn), dtype=torch.uint8, device="cuda")
        inst = torch.empty(B, h, w, dtype=torch.int32, device="cuda")
        ncomp = torch.empty(B, Cn, dtype=torch.int32, device="cuda")
        us = timed(lambda: call("diga_ccl_label", P(lab), 0, B, h, w, Cn, 8, P(inst), P(ncomp), P(work), st()), a.reps)
        print(f"  {name:24s} {int(ncomp.sum()) / B:8.0f} components/frame  {us:8.1f} us  {B * h * w * 8 / us / 1e3:7.1f} GB/s", flush=True)
    print("instance pooling (CH = 2048, bf16 features)")
    for name, B, h, w, cell, noise in (("feat 65x129 x1", 1, 65, 129, 4, 0.10), ("feat 91x161 x1", 1, 91, 161, 4, 0.10),
                                       ("feat 65x129 x15", 15, 65, 129, 4, 0.10), ("feat 129x257 x4", 4, 129, 257, 5, 0.08)):
        lab = labels(rng, B, h, w, Cn, cell, noise)
        inst, ncomp = instances.ccl_label(lab, Cn, 8)
        total = int(ncomp.sum())
        max_inst = -(-total // 1024) * 1024
        CH = 2048
        feat = torch.randn(B * h * w, CH, device="cuda").to(torch.bfloat16)
        work = torch.empty(call("diga_instance_pool_work_bytes", B, h * w, Cn, CH, max_inst), dtype=torch.uint8, device="cuda")
        protos = torch.empty(max_inst, CH, dtype=torch.float32, device="cuda")
        meta = torch.empty(3, max_inst, dtype=torch.int32, device="cuda")
        nt = torch.empty(1, dtype=torch.int32, device="cuda")
        us = timed(lambda: call("diga_instance_pool", P(feat), P(lab), 0, P(inst), P(ncomp), B, h * w, Cn, CH, max_inst, P(protos),
                                P(meta[0]), P(meta[1]), P(meta[2]), P(nt), P(work), st()), a.reps)
        print(f"  {name:24s} {total / B:8.0f} components/frame  {us:8.1f} us  {B * h * w * CH * 2 / us / 1e3:7.1f} GB/s (feature read)",
              flush=True)


if __name__ == "__main__":
    main()
