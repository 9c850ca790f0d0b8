"""Input pipeline contract: the arithmetic the reference's dataset classes apply to a decoded
frame, on the GPU.  Decoding and PIL resizing (Image.BICUBIC for images, Image.NEAREST for labels,
base_dataset.py:54-67) stay on the host exactly as in the reference; dataset file listing is out of scope.

    x = preprocess_image(rgb_uint8_hwc)                    # BaseDataset.preprocess, base_dataset.py:45-52
    y = map_labels(label_ids)                              # CityscapesDataSet.map_labels, cityscapes.py:50-51
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import DigaError, call

# ttda_datamodule.py:53 -- the BGR mean every dataset of the reference's test list is built with
IMG_MEAN = (104.00698793, 116.66876762, 122.67891434)
# src/datamodules/seg/cityscapes_list/info.json "label2train" (the standard Cityscapes id -> train id table; data)
CITYSCAPES_LABEL2TRAIN = [[0, 255], [1, 255], [2, 255], [3, 255], [4, 255], [5, 255], [6, 255], [7, 0], [8, 1], [9, 255], [10, 255],
                          [11, 2], [12, 3], [13, 4], [14, 255], [15, 255], [16, 255], [17, 5], [18, 255], [19, 6], [20, 7], [21, 8],
                          [22, 9], [23, 10], [24, 11], [25, 12], [26, 13], [27, 14], [28, 15], [29, 255], [30, 255], [31, 16],
                          [32, 17], [33, 18], [-1, 255]]


def map_vector(label2train=CITYSCAPES_LABEL2TRAIN) -> np.ndarray:
    """cityscapes.py:38-41: a vector of len(mapping) entries; the [-1, 255] row lands on the last entry."""
    mapping = np.array(label2train, dtype=np.int64)
    vec = np.zeros((mapping.shape[0],), dtype=np.int64)
    for source_label, target_label in mapping:
        vec[source_label] = target_label
    return vec


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def preprocess_image(img: torch.Tensor, mean=IMG_MEAN, swap_rb: bool = True) -> torch.Tensor:
    """img: [H,W,3] or [B,H,W,3] RGB, uint8 or float32, on the device -> fp32 [B,3,H,W] (BGR - mean when swap_rb, the
    BDD100K/Mapillary quirk -- mean subtracted from unswapped RGB -- when not).  A zero mean selects the reference's
    per-image-mean branch (base_dataset.py:49-51), which is not built."""
    if not img.is_cuda:
        raise DigaError("preprocess_image needs a device tensor: there is no CPU fallback")
    if not np.any(np.asarray(mean, dtype=np.float64)):
        raise DigaError("mean == 0 selects the reference's per-image-mean branch (base_dataset.py:49-51): not supported")
    if img.dim() == 3:
        img = img[None]
    if img.dtype not in (torch.uint8, torch.float32) or img.shape[-1] != 3:
        raise DigaError(f"preprocess_image: expected uint8/float32 [...,H,W,3], got {img.dtype} {tuple(img.shape)}")
    img = img.contiguous()
    B, H, W, _ = img.shape
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=img.device)
    call("diga_preprocess_image", _p(img), int(img.dtype == torch.uint8), _p(out), B, H, W, float(mean[0]), float(mean[1]),
         float(mean[2]), int(swap_rb), _stream())
    return out


_LUT_CACHE = {}


def map_labels(ids: torch.Tensor, label2train=CITYSCAPES_LABEL2TRAIN, out_dtype=torch.int64) -> torch.Tensor:
    """ids: label-id map (uint8 / int32 / float32, any shape) on the device -> train ids (255 = ignore)."""
    if not ids.is_cuda:
        raise DigaError("map_labels needs a device tensor: there is no CPU fallback")
    kind = {torch.uint8: 0, torch.int32: 1, torch.float32: 2}.get(ids.dtype)
    if kind is None or out_dtype not in (torch.int64, torch.int32):
        raise DigaError(f"map_labels: unsupported dtypes {ids.dtype} -> {out_dtype}")
    key = (id(label2train), ids.device)
    if key not in _LUT_CACHE:
        _LUT_CACHE[key] = torch.from_numpy(map_vector(label2train).astype(np.int32)).to(ids.device)
    lut = _LUT_CACHE[key]
    ids = ids.contiguous()
    out = torch.empty(ids.shape, dtype=out_dtype, device=ids.device)
    call("diga_map_labels", _p(ids), kind, _p(lut), int(lut.numel()), _p(out), int(out_dtype == torch.int64), int(ids.numel()),
         _stream())
    return out
