"""Helpers for the -m gpu parity tests (test infrastructure; torch ops here are only used to move and
re-layout test data, the code under test is always libdiga_hip.so through the C-ABI)."""
import ctypes as C

import numpy as np
import torch

from diga_amd import _lib
from diga_amd._lib import ConvDesc, call

DEV = "cuda:0"


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """fp32 tensor -> fp32 tensor holding bf16-representable values (RNE)."""
    return t.to(torch.bfloat16).to(torch.float32)


def to_nhwc_bf16(x_nchw: torch.Tensor) -> torch.Tensor:
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def from_nhwc_bf16(t: torch.Tensor, shape_nhwc) -> torch.Tensor:
    """device bf16 flat/NHWC buffer -> CPU fp32 NCHW."""
    B, H, W, Cc = shape_nhwc
    return t[:B * H * W * Cc].reshape(B, H, W, Cc).to(torch.float32).permute(0, 3, 1, 2).contiguous().cpu()


def conv_desc(B, H, W, Cin, Cout, stride=1, k3=False, dil=1, tile=0, cout_pad=None):
    d = ConvDesc()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, Cin, Ho, Wo, Cout
    d.stride = stride
    d.ntaps = 9 if k3 else 1
    if k3:
        for t in range(9):
            d.dy[t], d.dx[t] = (t // 3 - 1) * dil, (t % 3 - 1) * dil
    d.cout_pad = cout_pad or Cout
    d.splitk, d.out_f32, d.tile = 1, 0, tile
    return d


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
