"""fp32-exact verification mode of the DIGA forward path.

The reference is fp32 end to end (ttda_module.py:1634-1645); the product path stores ~130 activation tensors in bf16,
which puts its end-to-end distance to the reference at the bf16 floor (~1 % of the logit range).  This engine runs the
SAME kernels with fp32-grade operands so that the north-star numbers (logits within 1e-3 relative, integer outputs
bit-exact, mIoU within 0.01) can be demonstrated end to end, not only stage by stage:

* convolutions: ``diga_conv_igemm_bf16`` with ``desc.nplanes`` = 2 or 3 -- the same MFMA kernel, every operand given as
  bf16 planes whose sum is the fp32 value, the plane products accumulated as extra K ranges ("bf16x3": ~2^-17 per
  product, "bf16x6": fp32-grade), fp32 output + DAM statistics from the fp32 accumulators;
* DAM apply / max-pool / stem: the fp32 forms (``diga_bn_apply_f32`` ...), which also write the next conv's operand planes;
* SAM: ``diga_proto_pool_f32`` / ``diga_proto_classify_fuse_f32`` on the fp32 features; head, labels, update, fusion and
  the tail are the product kernels unchanged (they are fp32 already).

Batch semantics (one batch of B samples per step, the reference's) or, with stream=True, B consecutive batch-1 frames of
a stream in flight (per-sample DAM statistics, SAM frame by frame in stream order: every frame bit-identical to its own
batch-1 step, as Engine(stream=True)); no exchange.  Not tuned for speed (3x / 6x the MFMA work, element-per-thread apply
kernels, per-frame SAM launches); bench.py reports what it runs at next to the product path ("exact_mode").
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, call
from .engine import DILATIONS, FEAT_CH, HEAD_DILS, LAYERS, PLANES, STRIDES, Engine, NetWeights, _p, _stream, pack_taps


def split_planes(t: torch.Tensor, nplanes: int) -> torch.Tensor:
    """fp32 device tensor -> bf16 planes [nplanes, *t.shape] whose sum is t (through diga_split_planes_f32)."""
    t = t.contiguous()
    out = torch.empty((nplanes,) + tuple(t.shape), dtype=torch.bfloat16, device=t.device)
    call("diga_split_planes_f32", _p(t), _p(out), nplanes, t.numel(), _stream())
    return out


class ExactWeights(NetWeights):
    """NetWeights with every conv weight additionally split into operand planes ([nplanes][cout_pad][K] bf16)."""

    def __init__(self, sd: dict, device, num_classes: int = 19, nplanes: int = 3):
        self.nplanes = nplanes
        super().__init__(sd, device, num_classes)

        def dev(v):
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(np.ascontiguousarray(v))
            return v.detach().to(device=device, dtype=torch.float32).contiguous()

        self.conv_planes = {}
        for name in self.conv:
            w = dev(sd[name + ".weight"])
            self.conv_planes[name] = split_planes(pack_taps([w], None, dtype=torch.float32), nplanes)

        def head_planes(prefix):
            w0, w1 = dev(sd[prefix + ".conv2d_list.0.weight"]), dev(sd[prefix + ".conv2d_list.1.weight"])
            rows = torch.cat([w.permute(2, 3, 0, 1).reshape(9 * num_classes, w.shape[1]) for w in (w0, w1)], 0)
            return split_planes(pack_taps([rows[:, :, None, None]], self.head_cpad, dtype=torch.float32), nplanes)

        self.head_planes = head_planes("layer6")
        torch.cuda.synchronize()


class ExactEngine(Engine):
    """Engine whose trunk, head and SAM run at fp32-grade precision (see the module docstring)."""

    def __init__(self, weights: ExactWeights, B: int, H: int, W: int, bn_lambda: float = 0.8, confidence_threshold: float = 0.9,
                 conv_tile: int = 0, stream: bool = False):
        super().__init__(weights, B, H, W, bn_lambda=bn_lambda, confidence_threshold=confidence_threshold, conv_tile=conv_tile,
                         stream=stream, fuse_apply=False)
        self.np_ = weights.nplanes
        dev, f32, bf = self.dev, torch.float32, torch.bfloat16
        n1, n2, n3 = B * self.h1 * self.w1, B * self.h2 * self.w2, B * self.h3 * self.w3
        big = max(n2 * 256, n3 * 2048)
        mid = max(n2 * 64, n2 * 128, n3 * 512)

        def buf(n, dt):
            return torch.empty(int(n), dtype=dt, device=dev)

        self.f_stem = buf(n1 * 64, f32)
        self.f_raw1, self.f_raw2 = buf(mid, f32), buf(mid, f32)
        self.f_raw3, self.f_rawd = buf(big, f32), buf(big, f32)
        self.f_xa, self.f_xb = buf(big, f32), buf(big, f32)                       # fp32 block inputs / outputs (residuals, features)
        self.p_xa, self.p_xb = buf(self.np_ * big, bf), buf(self.np_ * big, bf)   # their operand planes
        self.p_act1, self.p_act2 = buf(self.np_ * mid, bf), buf(self.np_ * mid, bf)
        self.feat_f32 = None

    # ------------------------------------------------------------------
    def _conv_x(self, name, planes_in, out_f32, Hin, Win, Cin, Ho, Wo, Cout, stride, dil, k3, bn, wgt=None, stats=True, cout_pad=None):
        d = ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = self.B, Hin, Win, Cin, Ho, Wo, Cout
        d.stride = stride
        if k3:
            d.ntaps = 9
            for t in range(9):
                d.dy[t], d.dx[t] = (t // 3 - 1) * dil, (t % 3 - 1) * dil
        else:
            d.ntaps = 1
        d.cout_pad = cout_pad or Cout
        d.splitk, d.out_f32, d.tile, d.nplanes = 1, 1, self.conv_tile, self.np_
        w = self.w.conv_planes[name] if wgt is None else wgt
        if stats:
            call("diga_conv_igemm_bf16", C.byref(d), _p(planes_in), _p(w), _p(out_f32), _p(self.slab), _p(bn.running_mean), _stream())
        else:
            call("diga_conv_igemm_bf16", C.byref(d), _p(planes_in), _p(w), _p(out_f32), None, None, _stream())
        return (Ho * Wo + 31) // 32

    def _apply_x(self, raw, sc, sh, res, rsc, rsh, out_f32, out_planes, M, Cc, relu=1):
        call("diga_bn_apply_f32", _p(raw), _p(sc), _p(sh), _p(res), _p(rsc), _p(rsh), _p(out_f32), _p(out_planes), self.np_, M, Cc, relu,
             self.B if self.stream else 0, _stream())

    def _tapf(self, name, t, shape):
        if self.taps is not None:
            self.taps[name] = (t[:int(np.prod(shape))].clone(), shape)

    def forward_feat(self, x: torch.Tensor, want_aux: bool = False):
        """net(x, feat=True) (adaptseg.py:177-188) at fp32-grade precision: leaves the fp32 features in self.feat_f32 ([P,2048]),
        the logits in self.logits and the pseudo labels in self.labels[:P]."""
        assert x.shape == (self.B, 3, self.H, self.W) and x.dtype == torch.float32 and x.is_contiguous()
        assert not want_aux, "the verification mode runs the feat=True path only"
        w, B = self.w, self.B
        bn = w.bn["bn1"]
        call("diga_stem_conv7x7s2_f32", _p(x), _p(w.stem_w), _p(self.f_stem), _p(self.slab), _p(bn.running_mean), B, self.H, self.W,
             _stream())
        self._tapf("conv1", self.f_stem, (B, self.h1, self.w1, 64))
        sc, sh = self._finalize(bn, self.stem_units // B, self.h1 * self.w1)
        fcur, fnxt, pcur, pnxt = self.f_xa, self.f_xb, self.p_xa, self.p_xb
        call("diga_bn_relu_maxpool_f32", _p(self.f_stem), _p(sc), _p(sh), _p(fcur), _p(pcur), self.np_, B, self.h1, self.w1, 64,
             self.h2, self.w2, int(self.stream), _stream())
        self._tapf("maxpool", fcur, (B, self.h2, self.w2, 64))
        Hc, Wc, Cc = self.h2, self.w2, 64
        for li, (nb, pl, st, dil) in enumerate(zip(LAYERS, PLANES, STRIDES, DILATIONS), start=1):
            for b in range(nb):
                pre = f"layer{li}.{b}"
                s = st if b == 0 else 1
                Ho, Wo = (Hc - 1) // s + 1, (Wc - 1) // s + 1
                M, rows = B * Ho * Wo, Ho * Wo
                bn1, bn2, bn3 = w.bn[pre + ".bn1"], w.bn[pre + ".bn2"], w.bn[pre + ".bn3"]
                nt = self._conv_x(pre + ".conv1", pcur, self.f_raw1, Hc, Wc, Cc, Ho, Wo, pl, s, 1, False, bn1)
                self._tapf(pre + ".conv1", self.f_raw1, (B, Ho, Wo, pl))
                sc, sh = self._finalize(bn1, nt, rows)
                self._apply_x(self.f_raw1, sc, sh, None, None, None, None, self.p_act1, M, pl)
                nt = self._conv_x(pre + ".conv2", self.p_act1, self.f_raw2, Ho, Wo, pl, Ho, Wo, pl, 1, dil, True, bn2)
                self._tapf(pre + ".conv2", self.f_raw2, (B, Ho, Wo, pl))
                sc, sh = self._finalize(bn2, nt, rows)
                self._apply_x(self.f_raw2, sc, sh, None, None, None, None, self.p_act2, M, pl)
                nt = self._conv_x(pre + ".conv3", self.p_act2, self.f_raw3, Ho, Wo, pl, Ho, Wo, pl * 4, 1, 1, False, bn3)
                self._tapf(pre + ".conv3", self.f_raw3, (B, Ho, Wo, pl * 4))
                sc3, sh3 = self._finalize(bn3, nt, rows)
                if b == 0:
                    bnd = w.bn[pre + ".downsample.1"]
                    nt = self._conv_x(pre + ".downsample.0", pcur, self.f_rawd, Hc, Wc, Cc, Ho, Wo, pl * 4, s, 1, False, bnd)
                    scd, shd = self._finalize(bnd, nt, rows)
                    self._apply_x(self.f_raw3, sc3, sh3, self.f_rawd, scd, shd, fnxt, pnxt, M, pl * 4)
                else:
                    self._apply_x(self.f_raw3, sc3, sh3, fcur, None, None, fnxt, pnxt, M, pl * 4)
                fcur, fnxt, pcur, pnxt = fnxt, fcur, pnxt, pcur
                Hc, Wc, Cc = Ho, Wo, pl * 4
                self._tapf(pre, fcur, (B, Ho, Wo, Cc))
        assert (Hc, Wc, Cc) == (self.h, self.wd, FEAT_CH)
        self.feat_f32 = fcur[:self.P * FEAT_CH]
        self.feat = None
        # head: the same dense 1x1 GEMM over the 18 (tap, class) rows + tap gather, operands as planes
        self._conv_x("head", pcur, self.head_part, self.h, self.wd, FEAT_CH, self.h, self.wd, w.head_cout, 1, 1, False, None,
                     wgt=w.head_planes, stats=False, cout_pad=w.head_cpad)
        call("diga_head_tapsum_pseudo_label", _p(self.head_part), w.head_cpad, 18, self.head_dy, self.head_dx, _p(w.head_bias),
             _p(self.logits), _p(self.labels), B, self.C, self.h, self.wd, self.threshold, 2 * self.hw if self.stream else 0, _stream())
        return self.feat_f32, self.logits

    def sam_fuse(self, y: torch.Tensor, cfg):
        """multi_proto_label + fusion (ttda:1678-1710, 1644) on the fp32 features."""
        B, Cn = self.B, self.C
        assert y.shape[0] == B and y.dtype in (torch.int64, torch.int32) and y.is_contiguous()
        rho = float(cfg["proto_rho"])
        pl, fl = float(cfg["proto_lambda"]), float(cfg["fusion_lambda"])
        coef = (float(np.float32(pl)), float(np.float32(1 - pl)), float(np.float32(fl)), float(np.float32(1 - fl)))
        hw = self.hw
        if self.stream:
            # labels [frame][pseudo | gt][hw]; every frame takes the batch-1 launches in stream order (the running prototypes of
            # frame t feed frame t + 1, ttda:1702-1705)
            call("diga_nearest_down_labels", _p(y), int(y.dtype == torch.int64), _p(self.labels[hw:]), B, y.shape[1], y.shape[2], self.h,
                 self.wd, 2 * hw, _stream())
            for f in range(B):
                feat = self.feat_f32[f * hw * FEAT_CH:(f + 1) * hw * FEAT_CH]
                call("diga_proto_pool_f32", _p(feat), _p(self.labels[f * 2 * hw:]), 2, hw, Cn, FEAT_CH, _p(self.pool_work), _p(self.pool_sums),
                     _p(self.pool_counts), _stream())
                call("diga_proto_update", _p(self.pool_sums), _p(self.pool_counts), Cn, FEAT_CH, _p(self.inst_proto), _p(self.inst_flag),
                     _p(self.run_proto), _p(self.run_flag), _p(self.state_flag), float(np.float32(1 - rho)), float(np.float32(rho)), _stream())
                call("diga_proto_classify_fuse_f32", _p(feat), _p(self.logits[f]), _p(self.inst_proto), _p(self.inst_flag), _p(self.run_proto),
                     _p(self.run_flag), _p(self.probs[f]), _p(self.pcf_work), 1, Cn, hw, FEAT_CH, 1.0, *coef, _stream())
            return self.probs
        call("diga_nearest_down_labels", _p(y), int(y.dtype == torch.int64), _p(self.labels[self.P:]), B, y.shape[1], y.shape[2], self.h,
             self.wd, 0, _stream())
        call("diga_proto_pool_f32", _p(self.feat_f32), _p(self.labels), 2, self.P, Cn, FEAT_CH, _p(self.pool_work), _p(self.pool_sums),
             _p(self.pool_counts), _stream())
        call("diga_proto_update", _p(self.pool_sums), _p(self.pool_counts), Cn, FEAT_CH, _p(self.inst_proto), _p(self.inst_flag),
             _p(self.run_proto), _p(self.run_flag), _p(self.state_flag), float(np.float32(1 - rho)), float(np.float32(rho)), _stream())
        call("diga_proto_classify_fuse_f32", _p(self.feat_f32), _p(self.logits), _p(self.inst_proto), _p(self.inst_flag), _p(self.run_proto),
             _p(self.run_flag), _p(self.probs), _p(self.pcf_work), B, Cn, self.hw, FEAT_CH, 1.0, *coef, _stream())
        return self.probs

    def feat_nchw(self):
        out = torch.empty(self.B, FEAT_CH, self.h, self.wd, dtype=torch.float32, device=self.dev)
        call("diga_nhwc_f32_to_nchw_f32", _p(self.feat_f32), _p(out), self.B, self.hw, FEAT_CH, _stream())
        return out
