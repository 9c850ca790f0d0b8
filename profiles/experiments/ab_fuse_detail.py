"""Per-conv event timing (eager) of layer3's non-first blocks: fused-input convs vs plain convs, 15 frames in flight."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from diga_amd import synth
from diga_amd.engine import Engine, NetWeights
CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)
dev = torch.device("cuda:0")
w = NetWeights(synth.synth_params(0), dev)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 15
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H, W = 512, 1024
xy = [synth.synth_frame(1000 + t, 1, H, W) for t in range(K)]
x = torch.from_numpy(np.concatenate([a for a, _ in xy])).to(dev); y = torch.from_numpy(np.concatenate([b for _, b in xy])).to(dev)
for fuse in (False, True):
    e = Engine(w, K, H, W, stream=K > 1, autotune=tile == 0, conv_tile=tile, fuse_apply=fuse)
    for _ in range(2):
        e.forward_and_adapt(x, y, CFG)
    torch.cuda.synchronize()
    acc = {}
    for rep in range(3):
        e.conv_events = []
        e.forward_and_adapt(x, y, CFG)
        torch.cuda.synchronize()
        for name, fl, s, en in e.conv_events:
            if name.startswith("layer3.") and not name.startswith("layer3.0."):
                k = name.split(".")[2]
                acc.setdefault(k, []).append(s.elapsed_time(en) * 1e3)
        e.conv_events = None
    print(f"fuse_apply={fuse}:", {k: round(float(np.median(v)) - 5.0, 1) for k, v in acc.items()}, "us (median per launch, minus ~5 us event overhead); tiles",
          {(k[2], k[5], k[8], k[9]): v for k, v in e.tile_cache.items() if k[0] == 65 and k[2] in (256, 1024) and k[5] in (256, 1024)}, flush=True)
    del e
