"""Same-box A/B: Engine(fuse_finapply=True) vs False at K frames per step (default 1), HIP-graph replay."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from diga_amd import synth
from diga_amd.engine import Engine, NetWeights
CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)
dev = torch.device("cuda:0")
w = NetWeights(synth.synth_params(0), dev)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W = 512, 1024
xy = [synth.synth_frame(1000 + t, 1, H, W) for t in range(K)]
x = torch.from_numpy(np.concatenate([a for a, _ in xy])).to(dev); y = torch.from_numpy(np.concatenate([b for _, b in xy])).to(dev)
for fa in (False, True, False, True):
    e = Engine(w, K, H, W, stream=K > 1, autotune=True, fuse_apply=False, fuse_finapply=fa)
    for _ in range(2):
        e.forward_and_adapt(x, y, CFG); e.tail(y)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e.forward_and_adapt(x, y, CFG); e.tail(y)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200 if K == 1 else 30
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"fuse_finapply={fa}: {dt*1e3:.3f} ms/step, {K/dt:.1f} frames/s", flush=True)
    del e, g
