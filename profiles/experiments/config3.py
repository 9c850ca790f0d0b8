"""BASELINE configs[3] for the record (not a bench line): batch 4 at 2048x1024, the reference's batch semantics
(half_first DAM statistics over the batch, prototypes pooled over the batch), eager launches with autotuned tiles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from diga_amd import synth
from diga_amd.engine import Engine, NetWeights

CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)
dev = torch.device("cuda:0")
w = NetWeights(synth.synth_params(0), dev)
B, H, W = 4, 1024, 2048
eng = Engine(w, B, H, W, autotune=True)
x, y = synth.synth_frame(0, B, H, W)
x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
for _ in range(3):
    eng.forward_and_adapt(x, y, CFG); eng.tail(y)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng.forward_and_adapt(x, y, CFG); eng.tail(y)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"configs[3]: batch {B} at {W}x{H}: {dt * 1e3:.2f} ms/step, {B / dt:.1f} frames/s, {eng.conv_flops / dt / 1e12:.0f} TFLOP/s of conv work", flush=True)
