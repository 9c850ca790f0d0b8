"""Calibrate the synthetic net's BN *source* statistics and commit them as data.

Runs the CPU oracle once over a full-resolution synthetic calibration frame; just before each of the
104 BN layers runs, its running_mean / running_var are set to the statistics of its actual input,
perturbed by a seeded factor (mean += N(0, 0.15 std), var *= U(0.8, 1.25)) that stands in for the
source->target domain shift DAM is designed to absorb.  Output: diga_amd/data/synth_bn_stats_seed0.npz
(421 KB), loaded by diga_amd.synth.synth_params().

    python tests/golden/make_synth_bn_stats.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd import synth  # noqa: E402
from oracle import diga_oracle as O  # noqa: E402


def main(seed=0, H=512, W=1024):
    torch.manual_seed(0)
    p = O.to_torch(synth.synth_params(seed, bn_stats=None))
    x, _ = synth.synth_frame(100000, 1, H, W, seed)
    stats = {}

    def hook(prefix, t, params):
        m = t.mean([0, 2, 3])
        v = t.var([0, 2, 3], unbiased=False)
        r = synth._rng(seed, "calib/" + prefix)
        m = m + torch.from_numpy((r.standard_normal(m.numel()) * 0.15).astype(np.float32)) * v.sqrt()
        v = v * torch.from_numpy(r.uniform(0.8, 1.25, m.numel()).astype(np.float32))
        params[prefix + ".running_mean"] = m
        params[prefix + ".running_var"] = v
        stats[prefix + ".running_mean"] = m.numpy().copy()
        stats[prefix + ".running_var"] = v.numpy().copy()

    O.BN_PRE_HOOK = hook
    with torch.no_grad():
        feat, logits = O.net_forward_feat(p, torch.from_numpy(x), 0.8)
    O.BN_PRE_HOOK = None
    out = os.path.join(ROOT, "diga_amd", "data", f"synth_bn_stats_seed{seed}.npz")
    np.savez_compressed(out, **stats)
    prob = logits.softmax(1)
    pm, am = prob.max(1)
    print("feat mean/std/max", float(feat.mean()), float(feat.std()), float(feat.max()))
    print("logits std", float(logits.std()), "confident frac", float((pm >= 0.9).float().mean()),
          "classes winning", int(am.unique().numel()))
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
