"""Soak: the bench configuration (2 windows x 15 frames, autotuned tiles, HIP-graph replay) run twice for 150 steps (4 500 frames each);
the signatures of logits / features / prototypes / confusion matrix taken every 10 steps must be the same bits in both runs.
    python profiles/experiments/soak_determinism.py"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from diga_amd import synth
from diga_amd.engine import NetWeights
from diga_amd.pipeline import StreamWindows
CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)
w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
K, NW, steps = 15, 2, 150
frames = [synth.synth_frame(t, 1, 512, 1024) for t in range(K * NW)]
wins = []
for i in range(NW):
    fr = frames[i * K:(i + 1) * K]
    wins.append((torch.from_numpy(np.concatenate([a for a, _ in fr])).cuda(), torch.from_numpy(np.concatenate([b for _, b in fr])).cuda()))
outs = []
for run in range(2):
    sw = StreamWindows(w, K, NW, 512, 1024, bn_lambda=0.8, autotune=True)
    sw.step(wins, CFG); sw.step(wins, CFG); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sw.step(wins, CFG)
    sig = []
    for s in range(steps):
        g.replay()
        if s % 10 == 9:
            e0, e1 = sw.engines
            sig.append(torch.stack([e0.logits.double().nan_to_num(-1).sum(), e1.logits.double().nan_to_num(-1).sum(), e0.run_proto.double().sum(), e0.hist.double().sum(),
                                    e0.feat.view(torch.int16).long().sum().double(), e1.feat.view(torch.int16).long().sum().double()]).cpu())
    torch.cuda.synchronize()
    outs.append(torch.stack(sig))
    print("run", run, outs[-1][-1].tolist(), flush=True)
print("identical:", bool(torch.equal(outs[0], outs[1])), "frames per run:", steps * K * NW)
assert torch.equal(outs[0], outs[1])
# round 3, final tree: identical: True, frames per run: 4500
