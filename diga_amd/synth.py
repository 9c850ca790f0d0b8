"""Seeded synthetic weights and frames (there is no network for checkpoints / Cityscapes).

Everything here is numpy PCG64 + exact integer / basic float arithmetic, so it regenerates
bit-identically on the build container and on the GPU box.  State-dict key names and tensor shapes
are the reference's (AdaptSegNet ``ResNetMulti(Bottleneck,[3,4,23,3])``, adaptseg.py:129-172,
SURVEY.md section 8b "State").

The reference's own init (``normal_(0, 0.01)`` on every conv, adaptseg.py:147-153) collapses the
activations and drives every frame into the all-NaN branch, so the
benchmark / parity weights are a build-defined init: He-scaled convs, BN affine drawn from narrow
ranges and BN source statistics *calibrated* on synthetic frames and then perturbed (a stand-in for
source-vs-target domain shift).  The calibrated statistics are committed as data
(``diga_amd/data/synth_bn_stats_seed0.npz``, written by ``tests/golden/make_synth_bn_stats.py``).
"""
from __future__ import annotations

import os
import zlib

import numpy as np

NUM_CLASSES = 19
LAYERS = (3, 4, 23, 3)
PLANES = (64, 128, 256, 512)
HEAD_DILATIONS = (6, 12, 18, 24)

_BN_STATS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "synth_bn_stats_seed0.npz")


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def conv_shapes(num_classes: int = NUM_CLASSES):
    """(key-prefix, weight shape, has_bias) for every conv in state-dict order."""
    out = [("conv1", (64, 3, 7, 7), False)]
    inpl = 64
    for li, (nb, pl) in enumerate(zip(LAYERS, PLANES), start=1):
        for b in range(nb):
            pre = f"layer{li}.{b}"
            out.append((pre + ".conv1", (pl, inpl, 1, 1), False))
            out.append((pre + ".conv2", (pl, pl, 3, 3), False))
            out.append((pre + ".conv3", (pl * 4, pl, 1, 1), False))
            if b == 0:
                out.append((pre + ".downsample.0", (pl * 4, inpl, 1, 1), False))
            inpl = pl * 4
    for k in range(4):
        out.append((f"layer5.conv2d_list.{k}", (num_classes, 1024, 3, 3), True))
    for k in range(4):
        out.append((f"layer6.conv2d_list.{k}", (num_classes, 2048, 3, 3), True))
    return out


def bn_names():
    """(key-prefix, channels) for the 104 BN layers in forward order."""
    out = [("bn1", 64)]
    for li, (nb, pl) in enumerate(zip(LAYERS, PLANES), start=1):
        for b in range(nb):
            pre = f"layer{li}.{b}"
            out += [(pre + ".bn1", pl), (pre + ".bn2", pl), (pre + ".bn3", pl * 4)]
            if b == 0:
                out.append((pre + ".downsample.1", pl * 4))
    return out


HEAD_STD = 0.05   # head weight std: tuned so that ~30-60 % of pixels clear the 0.9 confidence bar


def synth_params(seed: int = 0, num_classes: int = NUM_CLASSES, bn_stats: dict | str | None = "auto") -> dict:
    """Full AdaptSegNet state dict (fp32 numpy arrays, reference key names).

    bn_stats: dict {prefix+'.running_mean'/'.running_var': array}, a path to an .npz with such keys,
    "auto" (use the committed calibration file if present, else mean 0 / var 1) or None (mean 0 / var 1).
    """
    p = {}
    for name, shape, has_bias in conv_shapes(num_classes):
        r = _rng(seed, name + ".weight")
        fan_in = shape[1] * shape[2] * shape[3]
        if name.startswith("layer5") or name.startswith("layer6"):
            std = HEAD_STD
        else:
            std = np.sqrt(2.0 / fan_in)
        p[name + ".weight"] = (r.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)
        if has_bias:
            p[name + ".bias"] = (_rng(seed, name + ".bias").standard_normal(shape[0], dtype=np.float32)
                                 * np.float32(0.5)).astype(np.float32)
    if bn_stats == "auto":
        bn_stats = _BN_STATS_FILE if (seed == 0 and os.path.exists(_BN_STATS_FILE)) else None
    if isinstance(bn_stats, str):
        with np.load(bn_stats) as z:
            bn_stats = {k: z[k] for k in z.files}
    for name, c in bn_names():
        r = _rng(seed, name)
        # A random BN+ReLU stack is chaotic at initialisation (mean-centred ReLUs decorrelate their inputs
        # ~1.5x per layer), which would amplify bf16 storage rounding ~300x over the 33 blocks and make
        # end-to-end comparisons meaningless.  Trained nets are not like that; keep the synthetic one tame
        # the way trained ResNets are: the residual stream is dominated by the (linear) downsample path
        # (gamma ~ 1) and every block adds a small correction (bn3 gamma ~ 0.05-0.1).
        if name.endswith("bn3"):
            p[name + ".weight"] = r.uniform(0.05, 0.10, c).astype(np.float32)
            p[name + ".bias"] = (r.standard_normal(c) * 0.02).astype(np.float32)
        else:
            p[name + ".weight"] = r.uniform(0.8, 1.2, c).astype(np.float32)
            p[name + ".bias"] = (r.standard_normal(c) * 0.1).astype(np.float32)
        if bn_stats is not None:
            p[name + ".running_mean"] = np.asarray(bn_stats[name + ".running_mean"], np.float32)
            p[name + ".running_var"] = np.asarray(bn_stats[name + ".running_var"], np.float32)
        else:
            p[name + ".running_mean"] = np.zeros(c, np.float32)
            p[name + ".running_var"] = np.ones(c, np.float32)
        p[name + ".num_batches_tracked"] = np.zeros((), np.int64)
    return p


# --------------------------------------------------------------------------------------
# Frames
# --------------------------------------------------------------------------------------
# skewed class frequencies (road / building / vegetation dominant, Cityscapes-like)
_CLASS_W = np.array([30, 6, 22, 1, 1, 1.5, 0.3, 0.6, 16, 1.2, 4, 1.5, 0.3, 7, 0.4, 0.3, 0.3, 0.2, 0.5])


def _class_colors(seed: int = 0) -> np.ndarray:
    """Per-class mean BGR offset (mean-subtracted domain), fixed for a seed."""
    return _rng(seed, "class_colors").uniform(-90.0, 110.0, (NUM_CLASSES, 3)).astype(np.float32)


def _bilinear_up(a: np.ndarray, H: int, W: int) -> np.ndarray:
    """Separable bilinear resize of a [..., h, w] float32 array to [..., H, W] (align_corners=True style)."""
    h, w = a.shape[-2:]
    ys = np.linspace(0.0, h - 1.0, H, dtype=np.float64)
    xs = np.linspace(0.0, w - 1.0, W, dtype=np.float64)
    y0 = np.minimum(ys.astype(np.int64), h - 2)
    x0 = np.minimum(xs.astype(np.int64), w - 2)
    fy = (ys - y0).astype(np.float32)[:, None]
    fx = (xs - x0).astype(np.float32)
    rows = a[..., y0, :] * (1 - fy) + a[..., y0 + 1, :] * fy
    return rows[..., :, x0] * (1 - fx) + rows[..., :, x0 + 1] * fx


def synth_labels(t: int, B: int, H: int, W: int, seed: int = 0) -> np.ndarray:
    """[B,H,W] int64 blocky Voronoi label map, ~5 % pixels = 255, >= 3 classes absent per frame."""
    out = np.empty((B, H, W), np.int64)
    cell = 4
    hc, wc = (H + cell - 1) // cell, (W + cell - 1) // cell
    yy, xx = np.meshgrid(np.arange(hc, dtype=np.int64), np.arange(wc, dtype=np.int64), indexing="ij")
    for b in range(B):
        r = _rng(seed, f"labels/{t}/{b}")
        absent = r.choice(NUM_CLASSES, size=int(r.integers(3, 7)), replace=False)
        w = np.sqrt(_CLASS_W)
        w[absent] = 0
        w /= w.sum()
        k = 96
        cls = r.choice(NUM_CLASSES, size=k, p=w)
        sy = r.integers(0, hc, k)
        sx = r.integers(0, wc, k)
        # anisotropic integer metric -> wide, flat regions (road-like)
        d = (yy[..., None] - sy) ** 2 * 4 + (xx[..., None] - sx) ** 2
        lab = cls[np.argmin(d, axis=-1)]
        # ~5 % ignore: random rectangles of cells
        for _ in range(int(r.integers(4, 9))):
            y0, x0 = int(r.integers(0, hc)), int(r.integers(0, wc))
            hh, ww = int(r.integers(2, max(3, hc // 8))), int(r.integers(2, max(3, wc // 6)))
            lab[y0:y0 + hh, x0:x0 + ww] = 255
        full = np.repeat(np.repeat(lab, cell, axis=0), cell, axis=1)[:H, :W]
        out[b] = full
    return out


def synth_frame(t: int, B: int = 1, H: int = 512, W: int = 1024, seed: int = 0):
    """Frame t of the synthetic stream: (x fp32 [B,3,H,W] BGR mean-subtracted, y int64 [B,H,W]).

    Image = per-class colour of the label map + low-frequency field + pixel noise, clipped to the
    range of a mean-subtracted 8-bit image."""
    y = synth_labels(t, B, H, W, seed)
    r = _rng(seed, f"frame/{t}")
    colors = _class_colors(seed)
    ylut = np.where(y == 255, 0, y)
    base = colors[ylut]                                   # [B,H,W,3]
    base[y == 255] = 0.0
    x = np.ascontiguousarray(np.transpose(base, (0, 3, 1, 2)))
    coarse = r.standard_normal((B, 3, H // 32 + 2, W // 32 + 2), dtype=np.float32)
    x += np.float32(30.0) * _bilinear_up(coarse, H, W).astype(np.float32)
    x += np.float32(12.0) * r.standard_normal((B, 3, H, W), dtype=np.float32)
    np.clip(x, -123.0, 151.0, out=x)
    return x.astype(np.float32), y
