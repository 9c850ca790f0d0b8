"""Host-side execution plan for the DIGA forward path on one MI355X.

PyTorch is used for device memory, streams, HIP-graph capture and (optionally) torch.distributed;
every arithmetic step is a call into libdiga_hip.so (include/diga_hip.h).  The plan mirrors the
reference's call stack:

    net(x, feat=True)            adaptseg.py:177-188     -> Engine.forward_feat
    multi_proto_label + fusion   ttda_module.py:1678-1710, 1644   -> Engine.sam_fuse
    step(): upsample + metric    ttda_module.py:1610, 158-173     -> Engine.tail

Data layout in HBM: activations NHWC bf16, weights [Cout][tap][Cin] bf16, statistics / prototypes /
logits / probabilities fp32, labels int32.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, call
from .dist import group_plan

LAYERS = (3, 4, 23, 3)
PLANES = (64, 128, 256, 512)
STRIDES = (1, 2, 1, 1)
DILATIONS = (1, 1, 2, 4)
HEAD_DILS = (6, 12)          # only the first two head branches ever run (adaptseg.py:119-122)
FEAT_CH = 2048


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pool_out(L):
    """MaxPool2d(3, 2, 1, ceil_mode=True) output size (adaptseg.py:139)."""
    o = L // 2 + 1
    if (o - 1) * 2 >= L + 1:
        o -= 1
    return o


def feature_grid(H, W):
    h1, w1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1          # conv1 7x7 s2 p3
    h2, w2 = pool_out(h1), pool_out(w1)                   # maxpool
    h3, w3 = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1         # layer2 stride-2 1x1 (no padding)
    return (h1, w1), (h2, w2), (h3, w3)


def pack_conv_weight(w: torch.Tensor, cout_pad: int | None = None) -> torch.Tensor:
    """[Cout,Cin,kh,kw] fp32 -> [cout_pad][Cin/64][kh*kw][64] bf16 (K order of diga_conv_igemm_bf16: 64-channel chunk,
    tap, channel), zero-padded rows."""
    return pack_taps([w], cout_pad)


def pack_taps(ws, cout_pad: int | None = None, dtype=torch.bfloat16) -> torch.Tensor:
    """Several [Cout,Cin,kh,kw] kernels over the same input concatenated along the tap axis (the fused head).
    dtype=torch.float32 keeps the packed values unrounded (the verification mode splits them into operand planes)."""
    co, ci = ws[0].shape[:2]
    assert ci % 64 == 0 or ci < 64, ci
    cw = min(ci, 64)
    taps = torch.cat([w.permute(0, 2, 3, 1).reshape(co, -1, ci // cw, cw) for w in ws], 1)     # [co][tap][chunk][cw]
    out = taps.permute(0, 2, 1, 3).reshape(co, -1)
    if cout_pad is not None and cout_pad > co:
        out = torch.cat([out, out.new_zeros(cout_pad - co, out.shape[1])], 0)
    return out.to(dtype).contiguous()


@dataclass
class BNP:
    """Device-resident parameters of one DAM BN layer."""
    running_mean: torch.Tensor
    running_var: torch.Tensor
    weight: torch.Tensor
    bias: torch.Tensor
    C: int
    off: int = 0        # offset of this layer's scale/shift in the shared buffers
    lam: float | None = None   # this layer's own lambda_ (ttda:335-336: every module mixes with ITS parameter); None -> Engine.lam


class NetWeights:
    """Packs a reference-layout AdaptSegNet state dict for the kernels (one-off, at load time)."""

    def __init__(self, sd: dict, device, num_classes: int = 19):
        def dev(v, dt=torch.float32):
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(np.ascontiguousarray(v))
            return v.detach().to(device=device, dtype=dt).contiguous()

        self.device = device
        self.num_classes = num_classes
        self.stem_w = dev(sd["conv1.weight"]).reshape(64, 147).contiguous()
        self.conv = {}
        self.bn = {}
        off = 0

        def add_bn(name):
            nonlocal off
            b = BNP(dev(sd[name + ".running_mean"]), dev(sd[name + ".running_var"]), dev(sd[name + ".weight"]),
                    dev(sd[name + ".bias"]), int(sd[name + ".weight"].shape[0]), off)
            if name + ".lambda_" in sd:        # SIFABatchNorm2dTrainable's parameter (ttda:369), fp32 as the reference holds it
                b.lam = float(np.float32(float(sd[name + ".lambda_"])))
            off += b.C
            self.bn[name] = b

        add_bn("bn1")
        for li, nb in enumerate(LAYERS, start=1):
            for b in range(nb):
                pre = f"layer{li}.{b}"
                for k in ("conv1", "conv2", "conv3"):
                    self.conv[f"{pre}.{k}"] = pack_conv_weight(dev(sd[f"{pre}.{k}.weight"]))
                for k in ("bn1", "bn2", "bn3"):
                    add_bn(f"{pre}.{k}")
                if b == 0:
                    self.conv[f"{pre}.downsample.0"] = pack_conv_weight(dev(sd[f"{pre}.downsample.0.weight"]))
                    add_bn(f"{pre}.downsample.1")
        self.bn_channels = off
        # heads: the two executed branches (dil 6 + dil 12, adaptseg.py:119-122) fused into one 18-tap conv,
        # Cout padded 19 -> 32.  layer6 = main head (on layer4), layer5 = aux head (on layer3, only for net(x)).
        # As a GEMM: z[m][t*C + c] = feat[m] . W_t[c] for the 18 taps t -> one 1x1 conv with Cout = 18*C (padded to a
        # multiple of 128), then a tap gather (diga_head_tapsum_pseudo_label): the feature map is read once, not 18 times.
        self.head_cout = 18 * num_classes
        self.head_cpad = -(-self.head_cout // 128) * 128

        def pack_head(prefix):
            w0, w1 = dev(sd[prefix + ".conv2d_list.0.weight"]), dev(sd[prefix + ".conv2d_list.1.weight"])
            # rows (tap, class): [br][kh][kw][c][ci]
            rows = torch.cat([w.permute(2, 3, 0, 1).reshape(9 * num_classes, w.shape[1]) for w in (w0, w1)], 0)
            hw = pack_conv_weight(rows[:, :, None, None], self.head_cpad)
            return hw, (dev(sd[prefix + ".conv2d_list.0.bias"]) + dev(sd[prefix + ".conv2d_list.1.bias"])).contiguous()

        self.head_w, self.head_bias = pack_head("layer6")
        self.aux_w, self.aux_bias = pack_head("layer5") if "layer5.conv2d_list.0.weight" in sd else (None, None)


class Engine:
    """Static plan + buffers for one input shape (B, H, W) on the current device."""

    def __init__(self, weights: NetWeights, B: int, H: int, W: int, bn_lambda: float = 0.8, group=None,
                 confidence_threshold: float = 0.9, stream: bool = False, conv_tile: int = 0,
                 autotune: bool = False, fuse_conv3: bool = False, force_exchange: bool = False, fold_finalize: bool = False,
                 fuse_apply="auto", oneshot=False, fuse_finapply=False, exchange_timeout_ms: int | None = None):
        """stream=False: x is ONE batch of B samples -> the reference's batch semantics (half_first DAM statistics over
        the batch, prototypes pooled over the batch).  stream=True: x holds B CONSECUTIVE batch-1 frames of a stream
        ("frames in flight"): the trunk runs once over all of them with per-sample DAM statistics (DAM carries no
        state between frames), SAM/fusion then run frame by frame in order, so every frame's result is bit-identical
        to B separate batch-1 steps (tests/test_gpu_net.py::test_stream_mode_is_bit_identical_to_batch1_steps)."""
        _lib.load()
        self.stream = bool(stream)
        self.conv_tile = conv_tile
        # autotune: the first (eager) pass times the tile candidates of every distinct conv shape on the real buffers and
        # keeps the fastest.  Outputs and statistics are bit-identical for every tile shape
        # (tests/test_gpu_kernels.py::test_conv_results_do_not_depend_on_tile_or_batch), so this never changes results.
        self.autotune = bool(autotune) and conv_tile == 0
        self.tile_cache = {}
        # fuse_conv3: "conv3 -> bn3 -> += residual -> relu" as two passes over the 1x1 GEMM (statistics only, then
        # recompute + fused apply): the widest tensor of a bottleneck is never written raw and read back.  Blocks with a
        # downsample branch (4 of 33) keep the three-kernel form.  Measured (15 frames in flight, same-box A/B): 631.6 vs
        # 629.1 frames/s -- the statistics pass runs at 770 TFLOP/s but the residual read in the fused epilogue is not
        # overlapped with the K loop (168 us against 133 + 131 us for conv3 + bn_apply), so it is off by default.
        self.fuse_conv3 = bool(fuse_conv3)
        # fuse_apply: the DAM apply + ReLU of bn1 / bn2 runs on the operand path of conv2 / conv3 (act1 / act2 are never
        # materialised), and "bn3 + residual + ReLU" of a block runs on the operand path of the NEXT block's conv1, which also
        # writes the block input the next residual needs (diga_conv_igemm_bf16_fused_in).  Same arithmetic, same single bf16
        # rounding: every tensor is bit-identical to the unfused path (tests/test_gpu_net.py).  Blocks with a downsample branch and
        # the last block of a layer keep diga_bn_apply.
        # fuse_apply = "conv1": only the residual form; "all": conv2 / conv3 too; False: every apply is its own launch; "auto"
        # (default): "conv1" when the pass is large enough to be bandwidth-bound (>= 4 frames' worth of 1024x512 feature rows),
        # else False.  Measured (same-box A/B, 15 frames in flight, one window, HIP-graph replay): False 22.17 ms, "conv1"
        # 21.23 ms (+4.4 % frames/s: the fused conv1 costs +83 us and replaces a 131 us bn_apply), "all" 22.55 ms (conv2 / conv3
        # slow down by 45 us each -- the transform runs 9x / 4x redundantly over taps / N tiles -- for a 25 us bn_apply each);
        # one frame per step: 3.25 / 3.43 / 3.83 ms, the operand-path transform is exposed in a latency-bound K loop.
        if fuse_apply == "auto":
            fuse_apply = "conv1" if B * ((H + 7) // 8) * ((W + 7) // 8) >= 4 * 8385 else False
        self.fuse_apply = False if self.fuse_conv3 else ("all" if fuse_apply is True else fuse_apply)
        assert self.fuse_apply in (False, "conv1", "all"), fuse_apply
        # fuse_finapply: diga_bn_finalize + diga_bn_apply as ONE launch (diga_bn_finalize_apply; bit-identical; not with the
        # cross-rank exchange, where the sums travel in between).  Built for the one-frame-per-step plan and measured there:
        # 3.22 vs 3.21-3.24 ms per step (the 9.7 us fused launch replaces 5.1 + 7.0 us of kernel time, but a graph replay does not
        # get shorter by it), 6.40 vs 5.8-6.1 ms at three frames -- off by default (profiles/experiments/ab_finapply.py).
        self.fuse_finapply = bool(fuse_finapply)
        # stream + group: the B local samples are this rank's frames of B consecutive GLOBAL batches (one frame per rank
        # per batch): statistics are exchanged per batch, so every global batch keeps the reference's batch semantics
        self.w = weights
        self.dev = weights.device
        self.B, self.H, self.W = B, H, W
        self.C = weights.num_classes
        self.lam = float(np.float32(bn_lambda))
        self.group = group
        if group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        # the statistics / prototype exchange runs whenever there is more than one rank; force_exchange takes the same
        # code path with a single-rank group (how the RCCL calls are exercised on a one-GPU box, tests/test_gpu_dist.py)
        self.exchange = self.world > 1 or (bool(force_exchange) and group is not None)
        # oneshot: the exchanges go through the one-launch direct-peer-write transport (diga_amd/xgmi.py) instead of
        # torch.distributed (RCCL): per BN layer ONE launch (diga_bn_exchange_finalize: slab reduce + peer writes + bounded wait +
        # rank-order sum + scale / shift), two launches for the prototype sums / counts -- no host round trip, capturable in a HIP
        # graph.  True: required (raises when it cannot connect); "auto": used when it connects and passes its self-test on every
        # rank, else the reason is logged and the exchanges go through torch.distributed; False: torch.distributed.
        # Slot size in 8-BYTE elements: the fp32 prototype sums (pairs) and, for the three-launch fallback of a BN layer the fused
        # form does not cover (more than 16 sets in flight), the fp64 group sums.
        self.oneshot = None
        if oneshot and self.exchange:
            from . import xgmi
            need = max(1 << 14, (B if stream else 1) * 4 * 2048, -(-(B if stream else 1) * 2 * weights.num_classes * FEAT_CH // 2))
            need = min(need, xgmi.MAX_ELEMS_LIMIT)       # larger tensors travel over torch.distributed (OneShotComm.supports)
            if oneshot == "auto":
                self.oneshot = xgmi.connect_or_none(group, need, timeout_ms=exchange_timeout_ms)
            else:
                self.oneshot = xgmi.OneShotComm(group, max_elems=need, timeout_ms=exchange_timeout_ms)
        # fold_finalize: where every sample is normalised with its own statistics (batch 1, or a stream's frames in flight) the
        # DAM finalize can run inside the conv launch (diga_conv_igemm_bf16_dam: the last tile of each (sample, N tile) group
        # reduces the slab) -- bit-identical scale / shift, 103 launches fewer per pass.  Measured (same-box A/B):
        # 708 vs 715 frames/s at 2 x 15 frames in flight and 4.09 vs 3.30 ms at one frame per step -- the last tile's reduction
        # reads the slab with L2-bypassing loads (the XCDs' L2s are not coherent inside a launch) and that serial tail costs more
        # than the launch it saves -- so it is OFF by default.  Batches (half_first spans samples) and the sharded path (the
        # sums are exchanged first) always use the separate diga_bn_finalize / reduce + finalize_sums launches.
        self.fold_finalize = bool(fold_finalize) and not self.exchange and (self.stream or B == 1)
        (self.h1, self.w1), (self.h2, self.w2), (self.h3, self.w3) = feature_grid(H, W)
        self.h, self.wd = self.h3, self.w3
        self.hw = self.h * self.wd
        self.P = B * self.hw
        self.threshold = float(np.float32(confidence_threshold))
        dev = self.dev
        bf, f32, i32 = torch.bfloat16, torch.float32, torch.int32

        def buf(n, dt):
            return torch.empty(int(n), dtype=dt, device=dev)

        n1 = B * self.h1 * self.w1
        n2 = B * self.h2 * self.w2
        n3 = B * self.h3 * self.w3
        self.stem_raw = buf(n1 * 64, bf)
        big = max(n2 * 256, n3 * 2048)            # largest block input / output
        self.xa, self.xb = buf(big, bf), buf(big, bf)
        mid = max(n2 * 64, n2 * 128, n3 * 512)    # conv1 / conv2 outputs (layer2.0.conv1 is already strided)
        self.raw1, self.act1, self.raw2, self.act2 = buf(mid, bf), buf(mid, bf), buf(mid, bf), buf(mid, bf)
        self.raw3, self.rawd = buf(big, bf), buf(big, bf)
        self.na = B if self.stream else 1          # affines per BN layer: one per sample in stream mode
        self.scale = buf(weights.bn_channels * self.na, f32)
        self.shift = buf(weights.bn_channels * self.na, f32)
        # statistics slab shared by all layers (each is consumed by its finalize before the next conv runs):
        # [unit][2][C], a unit = 32 output rows of one sample (igemm) or one stem block
        self.stem_units = call("diga_stem_conv7x7s2", None, None, None, None, None, B, H, W, None)

        def units(rows):
            return B * ((rows + 31) // 32)
        self.slab = buf(max(self.stem_units * 2 * 64, units(self.h2 * self.w2) * 2 * 256, units(self.h3 * self.w3) * 2 * 2048), f32)
        self.sums = buf(4 * 2048 * self.na, torch.float64)
        self.ticket = torch.zeros(B * 64, dtype=torch.int32, device=dev)      # (sample, N tile) counters of the fused finalize
        # head / SAM
        self.head_part = buf(self.P * weights.head_cpad, f32)        # z of the head GEMM
        self.head_dy = (C.c_int16 * 18)(*[(t % 9 // 3 - 1) * HEAD_DILS[t // 9] for t in range(18)])
        self.head_dx = (C.c_int16 * 18)(*[(t % 3 - 1) * HEAD_DILS[t // 9] for t in range(18)])
        self.logits = torch.empty(B, self.C, self.h, self.wd, dtype=f32, device=dev)
        # labels: batch mode [2][P] = [pseudo | ground truth] over the batch; stream mode [B][2][hw] (per frame)
        self.labels = buf(2 * self.P, i32)
        self.Ps = self.hw if self.stream else self.P             # pixels one SAM pass pools over
        self.pool_work = torch.empty(int(_lib.load().diga_proto_pool_frames_work_bytes(self.na, 2, self.Ps, self.C, FEAT_CH)),
                                     dtype=torch.uint8, device=dev)
        self.pcf_work = torch.empty(int(_lib.load().diga_proto_classify_frames_work_bytes(self.na, self.P, FEAT_CH)), dtype=torch.uint8,
                                    device=dev)
        self.pool_sums = torch.zeros(self.na * 2 * self.C * FEAT_CH, dtype=f32, device=dev)      # per frame in stream mode
        self.pool_counts = torch.zeros(self.na * 2 * self.C, dtype=i32, device=dev)
        # stream mode keeps every frame's instance prototypes and the running prototypes after that frame's update
        self.inst_proto = torch.zeros(self.na, self.C, FEAT_CH, dtype=f32, device=dev)
        self.inst_flag = torch.zeros(self.na, self.C, dtype=f32, device=dev)
        self.run_protos = torch.zeros(self.na, self.C, FEAT_CH, dtype=f32, device=dev) if self.stream else None
        self.run_flags = torch.zeros(self.na, self.C, dtype=f32, device=dev) if self.stream else None
        self.run_proto = torch.zeros(self.C, FEAT_CH, dtype=f32, device=dev)
        self.run_flag = torch.zeros(self.C, dtype=f32, device=dev)
        self.state_flag = torch.zeros(1, dtype=i32, device=dev)
        self.probs = torch.empty(B, self.C, self.h, self.wd, dtype=f32, device=dev)
        self.pred = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
        self.hist = torch.zeros(self.C * self.C, dtype=torch.int64, device=dev)
        self.feat = None          # bf16 NHWC view of the layer4 output, set by forward_feat
        self.conv_flops = 0       # 2*MAC of the executed convs (filled by the first forward_feat)
        self.exchange_launches = 0
        self.conv_events = None   # optional list: (name, flops, start_event, end_event) per igemm launch (bench.py)
        self.aux_events = None    # optional list: (kernel, algorithmic bytes, start_event, end_event) per HBM-bound launch (bench.py)
        self.taps = None          # optional dict: name -> (tensor, shape) of intermediate activations (tests)

    # ------------------------------------------------------------------ helpers
    def _groups(self, rows_per_sample):
        """(split_row, n_a, n_b) of the reference's half_first rule (ttda:313-320) for the *global* batch."""
        from .dist import group_plan
        return group_plan(self.rank, self.world, self.B, rows_per_sample)

    def _all_reduce(self, t):
        """Sum over ranks, in place.  RCCL ("nccl") reduces the device tensor on the stream; with the gloo backend
        (single-GPU rehearsal of the protocol in tests) the tensor takes a host round trip."""
        import torch.distributed as dist
        self.exchange_launches += 1
        if self.oneshot is not None and self.oneshot.supports(t):
            self.oneshot.all_reduce(t)
            return
        if dist.get_backend(self.group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)

    def _affine(self, bn: BNP):
        n = bn.C * self.na
        return self.scale[bn.off * self.na:bn.off * self.na + n], self.shift[bn.off * self.na:bn.off * self.na + n]

    def _finalize(self, bn: BNP, ups, rows_per_sample):
        """slab (this rank's B samples x ups units) -> scale/shift of one BN layer."""
        sc, sh = self._affine(bn)
        lam = self.lam if bn.lam is None else bn.lam
        if self.stream and not self.exchange:
            call("diga_bn_finalize", _p(self.slab), ups, self.B, self.B, 1, bn.C, float(rows_per_sample), 0.0,
                 _p(bn.running_mean), _p(bn.running_var), _p(bn.weight), _p(bn.bias), lam, 1e-5, _p(sc), _p(sh), _stream())
            return sc, sh
        if self.stream:      # B global batches in flight, one frame of each on this rank: one exchange for all of them
            _, n_a, n_b = group_plan(self.rank, self.world, 1, rows_per_sample)
            if self.oneshot is not None and self.oneshot.supports_bn(bn.C, self.B):
                self.exchange_launches += 1
                self.oneshot.bn_exchange_finalize(self.slab, ups, self.B, 1 if self.rank == 0 else 0, 1, bn.C, n_a, n_b, bn.running_mean,
                                                  bn.running_var, bn.weight, bn.bias, lam, 1e-5, sc, sh)
                return sc, sh
            sums = self.sums[:self.B * 4 * bn.C]
            call("diga_bn_reduce_slab", _p(self.slab), ups, self.B, 1 if self.rank == 0 else 0, 1, bn.C, _p(sums), _stream())
            self._all_reduce(sums)
            call("diga_bn_finalize_sums", _p(sums), self.B, bn.C, n_a, n_b, _p(bn.running_mean), _p(bn.running_var), _p(bn.weight),
                 _p(bn.bias), lam, 1e-5, _p(sc), _p(sh), _stream())
            return sc, sh
        split, n_a, n_b = self._groups(rows_per_sample)
        a_samples = split // rows_per_sample
        if self.exchange:
            if self.oneshot is not None and self.oneshot.supports_bn(bn.C, 1):
                self.exchange_launches += 1
                self.oneshot.bn_exchange_finalize(self.slab, ups, self.B, a_samples, 0, bn.C, n_a, n_b, bn.running_mean, bn.running_var,
                                                  bn.weight, bn.bias, lam, 1e-5, sc, sh)
                return sc, sh
            sums = self.sums[:4 * bn.C]
            call("diga_bn_reduce_slab", _p(self.slab), ups, self.B, a_samples, 0, bn.C, _p(sums), _stream())
            self._all_reduce(sums)
            call("diga_bn_finalize_sums", _p(sums), 1, bn.C, n_a, n_b, _p(bn.running_mean), _p(bn.running_var), _p(bn.weight),
                 _p(bn.bias), lam, 1e-5, _p(sc), _p(sh), _stream())
            return sc, sh
        call("diga_bn_finalize", _p(self.slab), ups, self.B, a_samples, 0, bn.C, n_a, n_b, _p(bn.running_mean), _p(bn.running_var),
             _p(bn.weight), _p(bn.bias), lam, 1e-5, _p(sc), _p(sh), _stream())
        return sc, sh

    def _finapply(self, bn: BNP, ups, rows_per_sample, raw, res, out, M, Cc):
        """_finalize + _apply (no residual / normalised residual) as one launch -> (scale, shift) of the BN layer."""
        sc, sh = self._affine(bn)
        lam = self.lam if bn.lam is None else bn.lam
        ev = self._aux_start()
        if self.stream:
            call("diga_bn_finalize_apply", _p(self.slab), ups, self.B, self.B, 1, bn.C, float(rows_per_sample), 0.0, _p(bn.running_mean),
                 _p(bn.running_var), _p(bn.weight), _p(bn.bias), lam, 1e-5, _p(sc), _p(sh), _p(raw), _p(res), _p(out), M, 1, _stream())
        else:
            split, n_a, n_b = self._groups(rows_per_sample)
            call("diga_bn_finalize_apply", _p(self.slab), ups, self.B, split // rows_per_sample, 0, bn.C, n_a, n_b, _p(bn.running_mean),
                 _p(bn.running_var), _p(bn.weight), _p(bn.bias), lam, 1e-5, _p(sc), _p(sh), _p(raw), _p(res), _p(out), M, 1, _stream())
        self._aux_end(ev, "bn_apply", M * Cc * 2 * (2 if res is None else 3))
        return sc, sh

    def _apply(self, raw, sc, sh, res, rsc, rsh, out, M, Cc):
        ev = self._aux_start()
        call("diga_bn_apply", _p(raw), _p(sc), _p(sh), _p(res), _p(rsc), _p(rsh), _p(out), M, Cc, 1,
             self.B if self.stream else 0, _stream())
        self._aux_end(ev, "bn_apply", M * Cc * 2 * (2 if res is None else 3))     # read raw (+ residual), write the activation

    def _conv_bn(self, name, x, out, Hin, Win, Cin, Ho, Wo, Cout, stride, dil, k3, bn: BNP, fused_in=None):
        """One trunk conv + its DAM finalize -> (scale, shift) of the BN layer that follows it."""
        if self.fold_finalize and fused_in is None:
            self._conv(name, x, out, Hin, Win, Cin, Ho, Wo, Cout, stride, dil, k3, bn, fold=True)
            return self._affine(bn)
        nt = self._conv(name, x, out, Hin, Win, Cin, Ho, Wo, Cout, stride, dil, k3, bn, fused_in=fused_in)
        return self._finalize(bn, nt, Ho * Wo)

    def _conv(self, name, x, out, Hin, Win, Cin, Ho, Wo, Cout, stride, dil, k3, bn: BNP | None, apply=None, fold=False, fused_in=None):
        """One trunk conv.  out=None: statistics only.  apply=(scale, shift, residual): pass 2 of a fused conv3
        (diga_conv_igemm_bf16_apply; no statistics).  fold: the DAM finalize runs inside the launch (per-sample statistics).
        fused_in=(scale, shift, residual | None, act_out | None): x is the producer's RAW output and the DAM apply + ReLU
        (+ residual) runs on the operand path (diga_conv_igemm_bf16_fused_in).  Returns the statistics units per sample."""
        d = ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = self.B, Hin, Win, Cin, Ho, Wo, Cout
        d.stride = stride
        if k3:
            d.ntaps = 9
            for t in range(9):
                d.dy[t], d.dx[t] = (t // 3 - 1) * dil, (t % 3 - 1) * dil
        else:
            d.ntaps = 1
        d.cout_pad = Cout
        d.splitk, d.out_f32, d.tile = 1, 0, self.conv_tile
        wgt = self.w.conv[name]
        if fused_in is not None:
            fsc, fsh, fres, fact = fused_in

            def launch():
                call("diga_conv_igemm_bf16_fused_in", C.byref(d), _p(x), _p(fsc), _p(fsh), _p(fres), _p(fact), self.B if self.stream else 0,
                     _p(wgt), _p(out), _p(self.slab), _p(bn.running_mean), _stream())
            mode = "fin2" if fres is not None else "fin1"
        elif apply is None and fold:
            sc, sh = self._affine(bn)
            lam = self.lam if bn.lam is None else bn.lam

            def launch():
                call("diga_conv_igemm_bf16_dam", C.byref(d), _p(x), _p(wgt), _p(out), _p(self.slab), _p(bn.running_mean),
                     _p(bn.running_var), _p(bn.weight), _p(bn.bias), lam, 1e-5, _p(sc), _p(sh), _p(self.ticket), _stream())
            mode = "stats+fin" if out is None else "conv+fin"
        elif apply is None:
            def launch():
                call("diga_conv_igemm_bf16", C.byref(d), _p(x), _p(wgt), _p(out), _p(self.slab), _p(bn.running_mean), _stream())
            mode = "stats" if out is None else "conv"
        else:
            sc, sh, res = apply

            def launch():
                call("diga_conv_igemm_bf16_apply", C.byref(d), _p(x), _p(wgt), _p(out), _p(sc), _p(sh), _p(res), 1,
                     self.B if self.stream else 0, _stream())
            mode = "apply"
        # algorithmic FLOPs are counted once per conv: the recomputation in pass 2 of a fused conv3 is overhead
        flops = 0 if mode == "apply" else 2 * self.B * Ho * Wo * Cout * Cin * d.ntaps
        if self.autotune:
            key = (Hin, Win, Cin, Ho, Wo, Cout, stride, dil, k3, mode)
            if key not in self.tile_cache and not torch.cuda.is_current_stream_capturing():
                self.tile_cache[key] = self._tune(d, launch)
            d.tile = self.tile_cache.get(key, 0)
        ev = self._ev_start()
        launch()
        self._ev_end(ev, name + (":apply" if mode == "apply" else ":fused_residual_in" if mode == "fin2" else ""), flops)
        self.conv_flops += flops
        return (Ho * Wo + 31) // 32          # statistics units per sample

    TILE_CANDIDATES = {76: 128, 77: 256, 23: 64, 13: 64, 22: 64, 12: 64, 24: 128, 14: 128, 21: 128, 26: 128, 16: 128, 27: 256, 17: 256, 28: 256,
                       33: 64, 32: 64, 34: 128, 31: 128, 36: 128, 37: 256, 38: 256,
                       46: 128, 47: 256, 48: 256, 56: 128, 57: 256, 58: 256, 61: 128, 64: 128, 66: 128, 67: 256,
                       91: 128}   # id -> N tile (91: the weights-stationary 1x1 kernel; refused, and skipped, for other convs.  Its K = 512
    # form, id 92, is built and bit-identical but loses to the tiled kernels on every K = 512 shape of the net -- 454 vs 366 us on
    # layer4.conv3, 237 vs 197 us on layer3's downsample at 15 frames, profiles/r3_tile92_sweep.log -- so it is not a candidate)

    def _tune(self, d, launch, reps=3):
        best, best_t = 0, float("inf")
        for tile, bn_tile in self.TILE_CANDIDATES.items():
            if d.cout_pad % bn_tile:
                continue
            d.tile = tile
            evs = []
            try:
                for r in range(reps + 1):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    launch()
                    e1.record()
                    evs.append((e0, e1))
            except _lib.DigaError:          # this tile does not support the epilogue (e.g. LDS of the fused apply)
                continue
            torch.cuda.synchronize()
            t = min(a.elapsed_time(b) for a, b in evs[1:])
            if t < best_t:
                best, best_t = tile, t
        return best

    def _ev_start(self):
        if self.conv_events is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _aux_start(self):
        if self.aux_events is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _aux_end(self, start, kernel, nbytes):
        if start is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.aux_events.append((kernel, nbytes, start, e))

    def _ev_end(self, start, name, flops):
        if start is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.conv_events.append((name, flops, start, e))

    def _tap(self, name, t, shape):
        if self.taps is not None:
            self.taps[name] = (t[:int(np.prod(shape))].clone(), shape)

    # ------------------------------------------------------------------ net(x, feat=True)
    def forward_feat(self, x: torch.Tensor, want_aux: bool = False):
        """x: fp32 NCHW [B,3,H,W] on the device.  Leaves the layer4 features in self.feat (bf16 [P,2048]),
        the logits in self.logits (fp32 NCHW) and the pseudo labels in self.labels[:P].  adaptseg.py:177-188.
        want_aux also runs the aux head (layer5 on the layer3 output, adaptseg.py:190) into self.aux_logits."""
        assert x.shape == (self.B, 3, self.H, self.W) and x.dtype == torch.float32 and x.is_contiguous()
        w, B = self.w, self.B
        self.conv_flops = 2 * B * self.h1 * self.w1 * 64 * 147
        self.exchange_launches = 0     # cross-rank exchanges of this step (sharded path: 104 BN layers + 2 for the prototypes)
        bn = w.bn["bn1"]
        call("diga_stem_conv7x7s2", _p(x), _p(w.stem_w), _p(self.stem_raw), _p(self.slab), _p(bn.running_mean), B, self.H,
             self.W, _stream())
        self._tap("conv1", self.stem_raw, (B, self.h1, self.w1, 64))
        sc, sh = self._finalize(bn, self.stem_units // B, self.h1 * self.w1)
        cur, nxt = self.xa, self.xb
        call("diga_bn_relu_maxpool", _p(self.stem_raw), _p(sc), _p(sh), _p(cur), B, self.h1, self.w1, 64, self.h2, self.w2,
             int(self.stream), _stream())
        self._tap("maxpool", cur, (B, self.h2, self.w2, 64))
        Hc, Wc, Cc = self.h2, self.w2, 64
        pending = None          # (scale3, shift3, block name) of a block whose closing apply is deferred into the next conv1
        for li, (nb, pl, st, dil) in enumerate(zip(LAYERS, PLANES, STRIDES, DILATIONS), start=1):
            for b in range(nb):
                pre = f"layer{li}.{b}"
                s = st if b == 0 else 1
                Ho, Wo = (Hc - 1) // s + 1, (Wc - 1) // s + 1
                M = B * Ho * Wo
                rows = Ho * Wo
                fuse = self.fuse_apply == "all"
                fa = self.fuse_finapply and not fuse and not self.exchange and not self.fold_finalize and not self.fuse_conv3
                # conv1 (1x1, carries the stride) -> bn1 -> relu
                bn1 = w.bn[pre + ".bn1"]
                pending_was = pending
                if pending is not None:
                    # the previous block's "bn3 + residual + relu" on this conv's operand path; it writes the block input into nxt
                    psc, psh, ppre = pending
                    sc, sh = self._conv_bn(pre + ".conv1", self.raw3, self.raw1, Hc, Wc, Cc, Ho, Wo, pl, s, 1, False, bn1,
                                           fused_in=(psc, psh, cur, nxt))
                    cur, nxt = nxt, cur
                    self._tap(ppre, cur, (B, Hc, Wc, Cc))
                    pending = None
                elif fa:
                    nt1 = self._conv(pre + ".conv1", cur, self.raw1, Hc, Wc, Cc, Ho, Wo, pl, s, 1, False, bn1)
                else:
                    sc, sh = self._conv_bn(pre + ".conv1", cur, self.raw1, Hc, Wc, Cc, Ho, Wo, pl, s, 1, False, bn1)
                self._tap(pre + ".conv1", self.raw1, (B, Ho, Wo, pl))
                # conv2 (3x3 dilated, pad = dilation) -> bn2 -> relu
                bn2 = w.bn[pre + ".bn2"]
                if fuse:
                    sc, sh = self._conv_bn(pre + ".conv2", self.raw1, self.raw2, Ho, Wo, pl, Ho, Wo, pl, 1, dil, True, bn2,
                                           fused_in=(sc, sh, None, None))
                else:
                    if fa and pending_was is None:
                        self._finapply(bn1, nt1, rows, self.raw1, None, self.act1, M, pl)
                    else:
                        self._apply(self.raw1, sc, sh, None, None, None, self.act1, M, pl)
                    self._tap(pre + ".act1", self.act1, (B, Ho, Wo, pl))
                    if fa:
                        nt2 = self._conv(pre + ".conv2", self.act1, self.raw2, Ho, Wo, pl, Ho, Wo, pl, 1, dil, True, bn2)
                    else:
                        sc, sh = self._conv_bn(pre + ".conv2", self.act1, self.raw2, Ho, Wo, pl, Ho, Wo, pl, 1, dil, True, bn2)
                self._tap(pre + ".conv2", self.raw2, (B, Ho, Wo, pl))
                # conv3 (1x1) -> bn3 (+ residual, relu)
                bn3 = w.bn[pre + ".bn3"]
                if self.fuse_conv3 and b != 0:
                    self._apply(self.raw2, sc, sh, None, None, None, self.act2, M, pl)
                    self._tap(pre + ".act2", self.act2, (B, Ho, Wo, pl))
                    sc3, sh3 = self._conv_bn(pre + ".conv3", self.act2, None, Ho, Wo, pl, Ho, Wo, pl * 4, 1, 1, False, bn3)      # statistics
                    self._conv(pre + ".conv3", self.act2, nxt, Ho, Wo, pl, Ho, Wo, pl * 4, 1, 1, False, None, apply=(sc3, sh3, cur))
                else:
                    if fuse:
                        sc3, sh3 = self._conv_bn(pre + ".conv3", self.raw2, self.raw3, Ho, Wo, pl, Ho, Wo, pl * 4, 1, 1, False, bn3,
                                                 fused_in=(sc, sh, None, None))
                    else:
                        if fa:
                            self._finapply(bn2, nt2, rows, self.raw2, None, self.act2, M, pl)
                        else:
                            self._apply(self.raw2, sc, sh, None, None, None, self.act2, M, pl)
                        self._tap(pre + ".act2", self.act2, (B, Ho, Wo, pl))
                        fa3 = fa and b != 0 and not (self.fuse_apply and b < nb - 1)     # plain "bn3 + residual + relu" follows
                        if fa3:
                            nt3 = self._conv(pre + ".conv3", self.act2, self.raw3, Ho, Wo, pl, Ho, Wo, pl * 4, 1, 1, False, bn3)
                        else:
                            sc3, sh3 = self._conv_bn(pre + ".conv3", self.act2, self.raw3, Ho, Wo, pl, Ho, Wo, pl * 4, 1, 1, False, bn3)
                    self._tap(pre + ".conv3", self.raw3, (B, Ho, Wo, pl * 4))
                    if b == 0:    # downsample branch: 1x1 (stride) + bn on the block input
                        bnd = w.bn[pre + ".downsample.1"]
                        scd, shd = self._conv_bn(pre + ".downsample.0", cur, self.rawd, Hc, Wc, Cc, Ho, Wo, pl * 4, s, 1, False, bnd)
                        self._tap(pre + ".downsample.0", self.rawd, (B, Ho, Wo, pl * 4))
                        self._apply(self.raw3, sc3, sh3, self.rawd, scd, shd, nxt, M, pl * 4)
                    elif self.fuse_apply and b < nb - 1:
                        # deferred: the next block's conv1 applies bn3 + residual + relu while it loads its operand
                        pending = (sc3, sh3, pre)
                        Hc, Wc, Cc = Ho, Wo, pl * 4
                        continue
                    elif fa and b != 0:
                        self._finapply(bn3, nt3, rows, self.raw3, cur, nxt, M, pl * 4)
                    else:
                        self._apply(self.raw3, sc3, sh3, cur, None, None, nxt, M, pl * 4)
                cur, nxt = nxt, cur
                Hc, Wc, Cc = Ho, Wo, pl * 4
                self._tap(pre, cur, (B, Ho, Wo, Cc))
            if li == 3 and want_aux:
                self.aux_logits = torch.empty_like(self.logits)
                self._head(cur[:self.P * 1024], 1024, self.w.aux_w, self.w.aux_bias, self.aux_logits, self.labels[self.P:], 0)
        assert (Hc, Wc, Cc) == (self.h, self.wd, FEAT_CH)
        self.feat = cur[:self.P * FEAT_CH]
        self._head(self.feat, FEAT_CH, self.w.head_w, self.w.head_bias, self.logits, self.labels, 2 * self.hw if self.stream else 0)
        return self.feat, self.logits

    def _head(self, feat, cin, hw, hbias, logits, labels, lstride):
        d = ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = self.B, self.h, self.wd, cin, self.h, self.wd, self.w.head_cout
        d.stride, d.ntaps = 1, 1
        d.cout_pad = self.w.head_cpad
        d.splitk, d.out_f32, d.tile = 1, 1, self.conv_tile
        if self.autotune:
            key = ("head", cin)
            if key not in self.tile_cache and not torch.cuda.is_current_stream_capturing():
                self.tile_cache[key] = self._tune(d, lambda: call("diga_conv_igemm_bf16", C.byref(d), _p(feat), _p(hw), _p(self.head_part),
                                                                  None, None, _stream()))
            d.tile = self.tile_cache.get(key, 0)
        flops = 2 * self.P * self.C * cin * 18
        ev = self._ev_start()
        call("diga_conv_igemm_bf16", C.byref(d), _p(feat), _p(hw), _p(self.head_part), None, None, _stream())
        self._ev_end(ev, "head", flops)
        self.conv_flops += flops
        # tap gather + bias -> logits (NCHW fp32) and the confidence-thresholded pseudo labels (ttda:1662-1676)
        call("diga_head_tapsum_pseudo_label", _p(self.head_part), self.w.head_cpad, 18, self.head_dy, self.head_dx, _p(hbias),
             _p(logits), _p(labels), self.B, self.C, self.h, self.wd, self.threshold, lstride, _stream())

    # ------------------------------------------------------------------ SAM + fusion
    def sam_fuse(self, y: torch.Tensor, cfg):
        """multi_proto_label + fusion (ttda:1678-1710, 1644) on the current features/logits.
        y: [B,H,W] int64/int32 ground truth on the device (255 = ignore)."""
        B, Cn = self.B, self.C
        assert y.shape[0] == B and y.dtype in (torch.int64, torch.int32) and y.is_contiguous()
        assert float(np.float32(cfg["confidence_threshold"])) == self.threshold
        rho = float(cfg["proto_rho"])
        pl, fl = float(cfg["proto_lambda"]), float(cfg["fusion_lambda"])
        hw, Ps = self.hw, self.Ps
        call("diga_nearest_down_labels", _p(y), int(y.dtype == torch.int64), _p(self.labels[hw:] if self.stream else self.labels[self.P:]),
             B, y.shape[1], y.shape[2], self.h, self.wd, 2 * hw if self.stream else 0, _stream())
        # batch mode: one pass over the whole batch; stream mode: one pass per frame, in stream order (the running
        # prototypes of frame t feed frame t+1, ttda:1702-1705).  Pooling does not depend on the running state, so all
        # frames are pooled first and (sharded) exchanged in one collective.
        nset = B if self.stream else 1
        # (stream mode: labels are [frame][pseudo | gt][hw], features [frame][hw][CH], sums [frame][2][C][CH] -- one set of
        # launches pools every frame, each bit-identical to its own call)
        ev = self._aux_start()
        call("diga_proto_pool_frames", _p(self.feat), _p(self.labels), nset, 2, Ps, Cn, FEAT_CH, _p(self.pool_work), _p(self.pool_sums),
             _p(self.pool_counts), _stream())
        self._aux_end(ev, "proto_pool", nset * Ps * (FEAT_CH * 2 + 2 * 4))      # one feature read serves both label sets
        if self.exchange:
            self._all_reduce(self.pool_sums)
            self._all_reduce(self.pool_counts)
        coef = (float(np.float32(pl)), float(np.float32(1 - pl)), float(np.float32(fl)), float(np.float32(1 - fl)))
        if self.stream:
            # the frames' seed / EMA chain in stream order inside one launch pair (every frame's prototypes are kept), then
            # ONE classifier pass in which frame f is measured against the prototypes as they stood after ITS update
            call("diga_proto_update_frames", _p(self.pool_sums), _p(self.pool_counts), nset, Cn, FEAT_CH, _p(self.inst_proto),
                 _p(self.inst_flag), _p(self.run_protos), _p(self.run_flags), _p(self.run_proto), _p(self.run_flag), _p(self.state_flag),
                 float(np.float32(1 - rho)), float(np.float32(rho)), _stream())
            call("diga_proto_classify_fuse_frames", _p(self.feat), _p(self.logits), _p(self.inst_proto), _p(self.inst_flag),
                 _p(self.run_protos), _p(self.run_flags), _p(self.probs), _p(self.pcf_work), nset, Cn, hw, FEAT_CH, 1.0, *coef, _stream())
            return self.probs
        call("diga_proto_update", _p(self.pool_sums), _p(self.pool_counts), Cn, FEAT_CH, _p(self.inst_proto), _p(self.inst_flag),
             _p(self.run_proto), _p(self.run_flag), _p(self.state_flag), float(np.float32(1 - rho)), float(np.float32(rho)), _stream())
        call("diga_proto_classify_fuse", _p(self.feat), _p(self.logits), _p(self.inst_proto), _p(self.inst_flag), _p(self.run_proto),
             _p(self.run_flag), _p(self.probs), _p(self.pcf_work), B, Cn, hw, FEAT_CH, 1.0, *coef, _stream())
        return self.probs

    def forward_and_adapt(self, x, y, cfg):
        self.forward_feat(x)
        return self.sam_fuse(y, cfg)

    # ------------------------------------------------------------------ tail
    def tail(self, y: torch.Tensor, probs_up: torch.Tensor | None = None):
        """Upsample + argmax + confusion matrix (ttda:1610, 158-173). Accumulates into self.hist."""
        H, W = y.shape[1], y.shape[2]
        if self.pred.shape != y.shape:
            self.pred = torch.empty(y.shape, dtype=torch.uint8, device=self.dev)
        if not self.stream:
            call("diga_upsample_argmax_hist", _p(self.probs), _p(y), int(y.dtype == torch.int64), _p(self.pred), _p(self.hist),
                 _p(probs_up), self.B, self.C, self.h, self.wd, H, W, _stream())
            return self.pred
        # stream mode: the reference's metric sees batch-1 updates (its confusion matrix is scaled by the batch size,
        # ttda:158-173): one launch that counts every frame once
        call("diga_upsample_argmax_hist_frames", _p(self.probs), _p(y), int(y.dtype == torch.int64), _p(self.pred), _p(self.hist),
             _p(probs_up), self.B, self.C, self.h, self.wd, H, W, _stream())
        return self.pred

    def check_exchange(self):
        """Call at a point where the host may synchronise (never under graph capture): raises DigaError when a wait of the one-shot
        transport has expired since the last call -- the affected statistics / prototypes were poisoned with NaN on this rank."""
        if self.oneshot is not None:
            self.oneshot.check()

    def close(self):
        """Unmap and free the one-shot transport's buffers (set-up memory; the exchanges themselves never allocate)."""
        if self.oneshot is not None:
            self.oneshot.close()
            self.oneshot = None

    def reset_state(self):
        self.state_flag.zero_()
        self.run_proto.zero_()
        self.run_flag.zero_()
        self.hist.zero_()

    # ------------------------------------------------------------------ layout helpers (API / tests)
    def nhwc_to_nchw(self, t: torch.Tensor, shape):
        B, H, W, Cc = shape
        out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=self.dev)
        call("diga_nhwc_bf16_to_nchw_f32", _p(t), _p(out), B, H * W, Cc, _stream())
        return out

    def feat_nchw(self):
        return self.nhwc_to_nchw(self.feat, (self.B, self.h, self.wd, FEAT_CH))
