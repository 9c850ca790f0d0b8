"""-m gpu: diga_proto_classify_fuse_noalloc -- the entry point for callers that cannot allocate device memory (its
workspace is a file-scope device array inside the library) -- against diga_proto_classify_fuse with a caller-supplied
workspace.  Same kernels, so the bar is bit-identity (NaN pattern included).  Two calls in a row, the second with a smaller
input (the array keeps the first call's contents), then the size limits.  Collected last (file name) so that a failure
here cannot hide another test under `pytest -x`; passes alone in a fresh process and after the other tests."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, P, call, stream, to_nhwc_bf16
from diga_amd import _lib

pytestmark = pytest.mark.gpu
NC = 19
MAX_PIXELS, MAX_CH = 16384, 2048      # DIGA_NOALLOC_MAX_PIXELS / DIGA_NOALLOC_MAX_CH (include/diga_hip.h)


def _case(seed, B, CH, h, w, nan_class=False):
    g = torch.Generator().manual_seed(seed)
    feat = (torch.randn(B, CH, h, w, generator=g).relu() * 0.5)
    logits = torch.randn(B, NC, h, w, generator=g) * 4
    ip = torch.randn(NC, CH, generator=g).relu() * 0.5
    rp = torch.randn(NC, CH, generator=g).relu() * 0.5
    ifl = (torch.rand(NC, generator=g) < 0.7).float()
    rfl = (torch.rand(NC, generator=g) < 0.8).float()
    if nan_class:
        ifl.zero_()                    # no instance prototype exists: 0/0 -> NaN in the instance classifier (ttda:1732-1733)
    return feat, logits, ip, ifl, rp, rfl


def _both(case):
    feat, logits, ip, ifl, rp, rfl = case
    B, CH, h, w = feat.shape
    f32 = lambda v: float(np.float32(v))
    keep = [to_nhwc_bf16(feat)] + [t.to(DEV).contiguous() for t in (logits, ip, ifl, rp, rfl)]
    coef = (1.0, f32(0.8), f32(1 - 0.8), f32(0.8), f32(1 - 0.8))
    ref = torch.full((B, NC, h, w), -7.0, dtype=torch.float32, device=DEV)
    got = torch.full((B, NC, h, w), -9.0, dtype=torch.float32, device=DEV)
    work = torch.empty(call("diga_proto_classify_work_bytes", B * h * w, CH), dtype=torch.uint8, device=DEV)
    call("diga_proto_classify_fuse", *(P(t) for t in keep), P(ref), P(work), B, NC, h * w, CH, *coef, stream())
    call("diga_proto_classify_fuse_noalloc", *(P(t) for t in keep), P(got), B, NC, h * w, CH, *coef, stream())
    torch.cuda.synchronize()
    return got.cpu(), ref.cpu()


def _same(got, ref):
    assert torch.equal(torch.isnan(got), torch.isnan(ref)), "NaN pattern differs"
    assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(ref, nan=-1.0))


def test_noalloc_matches_workspace_entry_point_twice_second_smaller():
    # first call: a full 1024x512 frame's feature grid (65x129) at the real feature width
    got, ref = _both(_case(1, 1, 2048, 65, 129))
    assert torch.isfinite(ref).all()
    _same(got, ref)
    # second call, smaller in every dimension (batch 2, ragged pixel count, fewer channels, a NaN branch): whatever the
    # first call left in the array beyond this call's extent must not matter
    got, ref = _both(_case(2, 2, 256, 9, 17, nan_class=True))
    assert torch.isnan(ref).any()
    _same(got, ref)
    # and a third, back up to the pixel limit exactly (batch 2 x 8192)
    got, ref = _both(_case(3, 2, 64, 64, 128))
    _same(got, ref)


def test_noalloc_rejects_sizes_beyond_the_static_scratch():
    d = torch.zeros(64, dtype=torch.float32, device=DEV)
    args = [P(d)] * 7
    for (B, hw, CH) in [(1, MAX_PIXELS + 1, 64), (2, MAX_PIXELS // 2 + 1, 64), (1, 64, MAX_CH + 64)]:
        with pytest.raises(_lib.DigaError, match="static scratch"):
            call("diga_proto_classify_fuse_noalloc", *args, B, NC, hw, CH, 1.0, 0.8, 0.2, 0.8, 0.2, stream())
    torch.cuda.synchronize()
