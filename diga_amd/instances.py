"""Optional SAM extension (BASELINE.json north_star (2)): connected components of the per-class
masks of a label map and instance-masked prototype pooling, on the GPU.

The reference (Waybaba/DIGA) has no connected-component step: nothing in the default DIGA path calls this module, and
with respect to the reference its results are "parity unpinned".  The specification is the build's own -- 4- or
8-connectivity per-class masks, components numbered from 1 inside each class in raster order of their first pixel,
background 0 -- and is pinned against scipy.ndimage.label (tests/golden/fn_ccl.npz).

    inst, ncomp = ccl_label(labels, num_classes=19, connectivity=8)          # int32 [B,h,w], int32 [B,C]
    table = instance_prototypes(feat_nhwc_bf16, labels, inst, ncomp, ...)    # per-component mean features
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch

from ._lib import DigaError, call


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _labels_i32(labels: torch.Tensor) -> torch.Tensor:
    if not labels.is_cuda:
        raise DigaError("instances: labels must be a device tensor: there is no CPU fallback")
    if labels.dim() == 2:
        labels = labels[None]
    if labels.dim() != 3 or labels.dtype not in (torch.int32, torch.int64, torch.uint8):
        raise DigaError(f"instances: expected integer labels [B,h,w], got {labels.dtype} {tuple(labels.shape)}")
    return labels.to(torch.int32).contiguous()


def ccl_label(labels: torch.Tensor, num_classes: int = 19, connectivity: int = 8):
    """labels: integer [B,h,w] (or [h,w]) on the device, values outside [0, num_classes) are background.
    -> (inst int32 [B,h,w], ncomp int32 [B,num_classes])."""
    lab = _labels_i32(labels)
    B, h, w = lab.shape
    nbytes = call("diga_ccl_work_bytes", B, h, w, num_classes)
    work = torch.empty(nbytes, dtype=torch.uint8, device=lab.device)
    inst = torch.empty(B, h, w, dtype=torch.int32, device=lab.device)
    ncomp = torch.empty(B, num_classes, dtype=torch.int32, device=lab.device)
    call("diga_ccl_label", _p(lab), 0, B, h, w, num_classes, connectivity, _p(inst), _p(ncomp), _p(work), _stream())
    return inst, ncomp


class InstanceTable(NamedTuple):
    protos: torch.Tensor     # fp32 [N, CH] mean feature of each component
    counts: torch.Tensor     # int32 [N] pixels
    classes: torch.Tensor    # int32 [N]
    samples: torch.Tensor    # int32 [N]
    total: int               # components found (> N when max_instances cut the table)


def instance_prototypes(feat_nhwc: torch.Tensor, labels: torch.Tensor, inst: torch.Tensor, ncomp: torch.Tensor,
                        num_classes: int = 19, max_instances: int = 4096) -> InstanceTable:
    """feat_nhwc: bf16 [B*h*w, CH] (the trunk's feature map as the kernels keep it); labels / inst / ncomp as given to /
    returned by ccl_label.  Rows are ordered (sample, class, component number).  Reads `total` back (one sync)."""
    lab = _labels_i32(labels)
    B, h, w = lab.shape
    if not feat_nhwc.is_cuda or feat_nhwc.dtype != torch.bfloat16 or feat_nhwc.numel() % (B * h * w):
        raise DigaError("instance_prototypes: feat must be a device bf16 tensor of B*h*w rows")
    CH = feat_nhwc.numel() // (B * h * w)
    dev = lab.device
    nbytes = call("diga_instance_pool_work_bytes", B, h * w, num_classes, CH, max_instances)
    work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    protos = torch.empty(max_instances, CH, dtype=torch.float32, device=dev)
    meta = torch.empty(3, max_instances, dtype=torch.int32, device=dev)
    n_total = torch.empty(1, dtype=torch.int32, device=dev)
    call("diga_instance_pool", _p(feat_nhwc.contiguous()), _p(lab), 0, _p(inst.contiguous()), _p(ncomp.contiguous()), B, h * w,
         num_classes, CH, max_instances, _p(protos), _p(meta[0]), _p(meta[1]), _p(meta[2]), _p(n_total), _p(work), _stream())
    total = int(n_total.item())
    n = min(total, max_instances)
    return InstanceTable(protos[:n], meta[0, :n], meta[1, :n], meta[2, :n], total)
