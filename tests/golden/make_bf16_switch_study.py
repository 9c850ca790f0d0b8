"""What sets the bf16 floor?  (CPU only, the ORACLE alone.)

The product path stores ~130 activation tensors per frame in bf16; end to end that puts it at 92-99.6 % final arg-max
agreement with the fp32 reference and at per-stream mIoU deltas of +-0.1 .. 0.2 over 30 frames.  This script measures which
CLASS of stored tensors that distance comes from: the oracle's bf16-storage emulation (O.net_forward_feat_bf16) is run over
the five 30-frame 1024x512 batch-1 streams of the bench (frame seeds 0 .. 4) with one class at a time kept in fp32 --
the residual stream (block inputs / outputs), the raw conv outputs, the normalised activations inside a block, only the
layer4 output that feeds head + SAM, the weights -- against the fp32 oracle (== the reference, pinned by the fixtures) on the
same frames.  Per configuration and stream: final arg-max agreement, pseudo-label agreement, logit error, mIoU-19/16/13 delta;
the extra HBM traffic of each switch is stated from the algorithmic byte counts of DESIGN.md 3 (fp32 instead of bf16 for that
class).  Result: tests/golden/bf16_switch_study.json, table in DESIGN.md 1.

    python tests/golden/make_bf16_switch_study.py [--frames 30] [--seeds 0,1,2,3,4]
"""
import argparse
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
CONFIGS = [("all bf16 (the product path's storage)", ()), ("residual stream fp32", ("residual",)), ("raw conv outputs fp32", ("raw",)),
           ("act1 / act2 fp32", ("act",)), ("head / SAM input fp32", ("head_in",)), ("weights fp32", ("weights",))]
CACHE = os.environ.get("BF16_STUDY_CACHE", "/tmp/gold/study_cache")        # per-stream results: the study is an hour of CPU, resumable


def stream(p, seed, n, bf16, keep, H=512, W=1024):
    os.makedirs(CACHE, exist_ok=True)
    tag = os.path.join(CACHE, f"s{seed}_n{n}_{'fp32' if not bf16 else 'bf16_' + '+'.join(keep)}.npz")
    if os.path.exists(tag):
        with np.load(tag, allow_pickle=False) as z:
            return ([torch.from_numpy(a) for a in z["preds"]], [torch.from_numpy(a) for a in z["pseudos"]],
                    [torch.from_numpy(a) for a in z["logits"]], [float(v) for v in z["miou"]])
    st = O.SamState()
    hist = torch.zeros(19, 19)
    preds, pseudos, logit_rng = [], [], []
    for t in range(n):
        x, y = synth.synth_frame(t, 1, H, W, seed=seed)
        info = {}
        with torch.no_grad():
            probs = O.forward_and_adapt(p, st, torch.from_numpy(x), torch.from_numpy(y), CFG, info, bf16=bf16, keep_fp32=keep)
            up = O.upsample_probs(probs, y.shape[-2:])
            hist = O.metric_update(hist, up, torch.from_numpy(y))
        preds.append(up.argmax(1).to(torch.uint8))
        pseudos.append(info["pseudo"].to(torch.uint8))
        logit_rng.append(info["logits"][:, :, ::2, ::2].clone())
    miou = [float(O.miou(hist, k)) for k in ("default", "16", "13")]
    np.savez_compressed(tag + ".tmp.npz", preds=np.stack([a.numpy() for a in preds]), pseudos=np.stack([a.numpy() for a in pseudos]),
                        logits=np.stack([a.numpy() for a in logit_rng]), miou=np.array(miou))
    os.replace(tag + ".tmp.npz", tag)
    return preds, pseudos, logit_rng, miou


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--seeds", default="0,1,2,3,4")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    seeds = [int(s) for s in a.seeds.split(",")]
    p = O.to_torch(synth.synth_params(0))
    out_path = os.path.join(ROOT, "tests", "golden", "bf16_switch_study.json")
    rec = {"frames": a.frames, "seeds": seeds, "configs": {}}
    ref = {}
    for sd in seeds:
        ref[sd] = stream(p, sd, a.frames, False, ())
        print("fp32 reference stream", sd, ref[sd][3], flush=True)
    rec["reference_miou"] = {str(sd): ref[sd][3] for sd in seeds}
    for name, keep in CONFIGS:
        rows = []
        for sd in seeds:
            preds, pseudos, logits, miou = stream(p, sd, a.frames, True, keep)
            rp, rq, rl, rm = ref[sd]
            agree = float(np.mean([float((a_ == b_).float().mean()) for a_, b_ in zip(preds, rp)]))
            pagree = float(np.mean([float((a_ == b_).float().mean()) for a_, b_ in zip(pseudos, rq)]))
            lerr = float(max(float((a_ - b_).abs().max() / (b_.max() - b_.min())) for a_, b_ in zip(logits, rl)))
            rows.append(dict(seed=sd, argmax_agree=agree, pseudo_agree=pagree, logit_err=lerr, miou_delta=[x - y for x, y in zip(miou, rm)]))
            print(name, rows[-1], flush=True)
        d = np.array([r["miou_delta"] for r in rows])
        rec["configs"][name] = dict(keep_fp32=list(keep), streams=rows,
                                    argmax_disagree_mean=float(1 - np.mean([r["argmax_agree"] for r in rows])),
                                    pseudo_disagree_mean=float(1 - np.mean([r["pseudo_agree"] for r in rows])),
                                    logit_err_max=float(max(r["logit_err"] for r in rows)),
                                    miou_delta_mean=d.mean(0).tolist(), miou_delta_absmax=np.abs(d).max(0).tolist(),
                                    miou_delta_std=d.std(0).tolist())
        with open(out_path, "w") as f:
            json.dump(rec, f, indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
