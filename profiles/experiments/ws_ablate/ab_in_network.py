"""In-network A/B of the weights-stationary variants: the engine's autotune (which times every candidate of a conv shape on its real
buffers, in the first eager pass, i.e. with the operand just written by bn_apply) with tiles 94 / 95 / 96 added to the candidates;
prints what it picked for the conv3 shapes and the per-candidate times it saw.
    python profiles/experiments/ws_ablate/ab_in_network.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from diga_amd import synth, engine
from diga_amd.engine import NetWeights, Engine
CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)
engine.Engine.TILE_CANDIDATES = {**engine.Engine.TILE_CANDIDATES, 94: 128, 95: 128, 96: 128}
seen = {}
orig = engine.Engine._tune
def tune(self, d, launch, reps=3):
    import ctypes as C
    res = {}
    if d.ntaps == 1 and d.stride == 1 and d.Cin in (64, 128, 256) and d.Cout % 128 == 0:
        for tile in (91, 94, 95, 96, 66, 67, 28, 48):
            d.tile = tile
            try:
                ts = []
                for r in range(6):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                res[tile] = round(min(ts[1:]), 1)
            except Exception:
                pass
        seen[(d.H, d.W, d.Cin, d.Cout)] = res
    return orig(self, d, launch, reps)
engine.Engine._tune = tune
w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
K = 15
e = Engine(w, K, 512, 1024, stream=True, autotune=True)
fr = [synth.synth_frame(t, 1, 512, 1024) for t in range(K)]
x = torch.from_numpy(np.concatenate([a for a, _ in fr])).cuda(); y = torch.from_numpy(np.concatenate([b for _, b in fr])).cuda()
e.forward_and_adapt(x, y, CFG); torch.cuda.synchronize()
for k, v in seen.items():
    print(k, v)
