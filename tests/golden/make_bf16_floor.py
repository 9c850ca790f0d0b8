"""bf16 storage noise floor of the synthetic net, measured with the ORACLE alone.

The north star stores activations and weights in bf16 (MFMA operands); the reference is fp32.  How far an
exact implementation with bf16 storage lands from the fp32 reference is a property of the net, not of
the kernels: this script runs the oracle twice -- fp32 (== the reference, pinned by net_*.npz) and with
bf16 storage emulated (O.net_forward_feat_bf16) -- over the same frame sequences as the golden fixtures
and records the agreement statistics in tests/golden/bf16_floor.json.  The -m gpu end-to-end tests then
require the GPU path to be no further from the reference than 1.5x this floor.

    python tests/golden/make_bf16_floor.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd import synth  # noqa: E402
from oracle import diga_oracle as O  # noqa: E402

CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)


def sequence(p, B, H, W, n, bf16):
    st = O.SamState()
    hist = torch.zeros(19, 19)
    out = []
    for t in range(n):
        x, y = synth.synth_frame(t, B, H, W)
        info = {}
        with torch.no_grad():
            probs = O.forward_and_adapt(p, st, torch.from_numpy(x), torch.from_numpy(y), CFG, info, bf16=bf16)
            up = O.upsample_probs(probs, y.shape[-2:])
            hist = O.metric_update(hist, up, torch.from_numpy(y))
        out.append(dict(logits=info["logits"], pseudo=info["pseudo"], pred=up.argmax(1), probs=probs))
    return out, [float(O.miou(hist, k)) for k in ("default", "16", "13")]


def main():
    p = O.to_torch(synth.synth_params(0))
    rec = {}
    for tag, (B, H, W, n) in dict(s1=(1, 65, 129, 3), s2=(2, 65, 129, 3), s3=(3, 65, 129, 3), m1=(1, 129, 257, 3),
                                  full=(1, 512, 1024, 2)).items():
        a, ma = sequence(p, B, H, W, n, False)
        b, mb = sequence(p, B, H, W, n, True)
        fr = []
        for ra, rb in zip(a, b):
            rng = float(ra["logits"].max() - ra["logits"].min())
            fr.append(dict(logit_err=float((ra["logits"] - rb["logits"]).abs().max()) / rng,
                           pseudo_agree=float((ra["pseudo"] == rb["pseudo"]).float().mean()),
                           pred_agree=float((ra["pred"] == rb["pred"]).float().mean()),
                           probs_err=float(np.nanmax(np.abs(ra["probs"].numpy() - rb["probs"].numpy())))))
        rec[tag] = dict(frames=fr, miou_fp32=ma, miou_bf16=mb)
        print(tag, fr, ma, mb)
    with open(os.path.join(ROOT, "tests", "golden", "bf16_floor.json"), "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
