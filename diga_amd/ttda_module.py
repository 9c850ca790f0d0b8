"""Host-side mirror of the reference's module API for the DIGA hot path.

Same class names, constructor signatures, method names / argument meaning / return values, attribute
names, state-dict keys and error behaviour as ``src/models/ttda_module.py`` of Waybaba/DIGA
("ttda" below) and ``src/models/components/adaptseg.py`` ("adaptseg"), so that Hydra ``_target_``
strings and user code keep working -- but every arithmetic step runs in libdiga_hip.so on the MI355X.
The ``nn.Module`` tree below only *holds* parameters (so ``state_dict()`` / ``load_state_dict()`` /
checkpoints keep the reference's key layout); it never computes with torch ops.  There is no CPU
fallback: without the HIP library or a GPU the calls raise.
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import Any, List

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import call
from .engine import FEAT_CH, Engine, NetWeights, _p, _stream

logger = logging.getLogger("diga_amd")
affine_par = True

# The reference's DIGA is a pytorch_lightning.LightningModule handed to trainer.test (ttda:1562, train_task.py:73).  Lightning is
# not a dependency of the kernels: when it is importable DIGA subclasses LightningModule (so a Trainer accepts it), otherwise
# it is a plain nn.Module with the same hooks, driven by src/eval.py's loop.
try:
    import pytorch_lightning as _pl
    _ModuleBase = _pl.LightningModule
except ImportError:
    _pl = None
    _ModuleBase = nn.Module


class AttrDict(dict):
    """dict with attribute access: the reference reads hparams both ways (ttda:1573 vs ttda:1640)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _attr(v):
    if isinstance(v, dict) and not isinstance(v, AttrDict):
        return AttrDict({k: _attr(x) for k, x in v.items()})
    if hasattr(v, "items") and not isinstance(v, dict):      # OmegaConf DictConfig and friends
        try:
            return AttrDict({k: _attr(x) for k, x in v.items()})
        except Exception:
            return v
    return v


def _f32(v) -> float:
    return float(np.float32(v))


# ------------------------------------------------------------------------------------------------
# metric (ttda:147-220)
# ------------------------------------------------------------------------------------------------
class SegmentationMetric(nn.Module):
    """Confusion-matrix mIoU with the reference's semantics (torchmetrics.Metric is not needed).

    ``update(preds, target)`` accepts what the reference accepts (probabilities [B,C,H,W] or labels
    [B,H,W]); on device tensors the argmax + bincount run in ``diga_upsample_argmax_hist``."""

    IDX16 = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 15, 17]   # ttda:203
    IDX13 = [0, 1, 2, 6, 7, 8, 10, 11, 12, 13, 15, 17]            # ttda:207

    def __init__(self, num_classes: int, ignore_index: int = 255):
        super().__init__()
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self.register_buffer("hist", torch.zeros((num_classes, num_classes)))
        self.register_buffer("last_hist", torch.zeros((num_classes, num_classes)))

    def add_counts(self, counts: torch.Tensor):
        """counts: integer [C,C] confusion counts of one update (already multiplied by batch_size, ttda:165)."""
        self.last_hist = counts.to(self.hist.dtype).to(self.hist.device)
        self.hist += self.last_hist

    def update(self, preds: torch.Tensor, target: torch.Tensor):
        assert preds.dim() in (3, 4)
        B = target.shape[0]
        Cn = self.num_classes
        dev = torch.device("cuda", torch.cuda.current_device())
        tgt = target.to(dev)
        tgt = tgt.contiguous() if tgt.dtype in (torch.int32, torch.int64) else tgt.long().contiguous()
        H, W = tgt.shape[-2:]
        if preds.dim() == 3:   # labels: express them as one-hot "probabilities" of the same size (exact arg-max)
            probs = torch.zeros(B, Cn, H, W, dtype=torch.float32, device=dev)
            probs.scatter_(1, preds.to(dev).long().unsqueeze(1), 1.0)
        else:
            probs = preds.to(dev, torch.float32).contiguous()
        assert probs.shape[-2:] == tgt.shape[-2:]
        hist = torch.zeros(Cn * Cn, dtype=torch.int64, device=dev)
        call("diga_argmax_hist", _p(probs), _p(tgt), int(tgt.dtype == torch.int64), None, _p(hist), B, Cn, H, W, _stream())
        self.add_counts(hist.reshape(Cn, Cn))

    def forward(self, preds, target):
        self.update(preds, target)
        return self.compute()

    def _per_class_iou(self, hist):
        d = torch.diag(hist)
        return d / (hist.sum(dim=1) + hist.sum(dim=0) - d)

    def compute(self):
        return torch.nanmean(self._per_class_iou(self.hist)) * 100.0

    def compute_iou(self, type="default"):
        ious = self._per_class_iou(self.hist)
        if type == "16":
            ious = ious[self.IDX16]
        elif type == "13":
            ious = ious[self.IDX13]
        elif type != "default":
            raise NotImplementedError
        return torch.nanmean(ious) * 100.0

    def compute_class_iou(self):
        return self._per_class_iou(self.hist) * 100.0

    def compute_confusion_matrix(self):
        return self.hist

    def reset(self):
        self.hist.zero_()
        self.last_hist.zero_()


# ------------------------------------------------------------------------------------------------
# DAM batch norm (ttda:298-370)
# ------------------------------------------------------------------------------------------------
class SIFABatchNorm2dTrainable(nn.BatchNorm2d):
    """Parameter holder + stand-alone DAM BN.  Inside the network the statistics come fused from the
    conv epilogue (engine.py); calling the module on its own (fp32 NCHW tensor, eval mode) runs
    diga_bn_stats_nchw_f32 -> diga_bn_finalize -> diga_bn_apply_nchw_f32."""

    def __init__(self, num_features=0, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)

    def from_bn(self, bn):
        self.__init__(bn.num_features, bn.eps, bn.momentum, bn.affine, bn.track_running_stats)
        self.load_state_dict(bn.state_dict())
        self.lambda_ = nn.Parameter(torch.tensor(0.5), requires_grad=True)    # ttda:369
        return self

    def forward(self, input):
        self._check_input_dim(input)
        if self.training:
            raise NotImplementedError("diga_amd implements the backward-free test-time path only (eval mode)")
        x = input.to(torch.float32).contiguous()
        if not x.is_cuda:
            raise _lib.DigaError("SIFABatchNorm2dTrainable.forward needs a device tensor: there is no CPU fallback")
        B, Cc, H, W = x.shape
        dev = x.device
        prm = [t.detach().to(dev, torch.float32).contiguous() for t in (self.running_mean, self.running_var, self.weight, self.bias)]
        slab = torch.empty(4 * Cc, dtype=torch.float32, device=dev)
        sc, sh = torch.empty(Cc, dtype=torch.float32, device=dev), torch.empty(Cc, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        split_b = 1 if B > 1 else B                       # half_first (ttda:313-320)
        call("diga_bn_stats_nchw_f32", _p(x), B, Cc, H * W, split_b, _p(prm[0]), _p(slab), _stream())
        call("diga_bn_finalize", _p(slab), 1, 2, 1, 0, Cc, float(split_b * H * W), float((B - split_b) * H * W), _p(prm[0]), _p(prm[1]),
             _p(prm[2]), _p(prm[3]), _f32(float(self.lambda_)), _f32(self.eps), _p(sc), _p(sh), _stream())
        call("diga_bn_apply_nchw_f32", _p(x), _p(sc), _p(sh), _p(y), B, Cc, H * W, _stream())
        return y


SIFABatchNorm2d = SIFABatchNorm2dTrainable


# ------------------------------------------------------------------------------------------------
# AdaptSegNet parameter tree (adaptseg:58-194) -- holders only
# ------------------------------------------------------------------------------------------------
class _Conv(nn.Module):
    def __init__(self, cin, cout, k, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k).normal_(0, 0.01), requires_grad=False)   # adaptseg:150
        if bias:
            bound = 1.0 / (cin * k * k) ** 0.5
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound), requires_grad=False)

    def forward(self, *a, **k):
        raise RuntimeError("parameter holder: convolutions run inside libdiga_hip.so (diga_amd.engine)")


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = _Conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes, affine=affine_par)
        self.conv2 = _Conv(planes, planes, 3)
        self.bn2 = nn.BatchNorm2d(planes, affine=affine_par)
        self.conv3 = _Conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4, affine=affine_par)
        self.downsample = downsample
        self.stride, self.dilation = stride, dilation


class Classifier_Module(nn.Module):
    def __init__(self, inplanes, dilation_series, padding_series, num_classes):
        super().__init__()
        self.conv2d_list = nn.ModuleList(_Conv(inplanes, num_classes, 3, bias=True) for _ in dilation_series)


class ResNetMulti(nn.Module):
    def __init__(self, block, layers, num_classes):
        super().__init__()
        self.inplanes = 64
        self.conv1 = _Conv(3, 64, 7)
        self.bn1 = nn.BatchNorm2d(64, affine=affine_par)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=4)
        self.layer5 = Classifier_Module(1024, [6, 12, 18, 24], [6, 12, 18, 24], num_classes)
        self.layer6 = Classifier_Module(2048, [6, 12, 18, 24], [6, 12, 18, 24], num_classes)
        for p in self.parameters():
            p.requires_grad = False

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        downsample = nn.Sequential(_Conv(self.inplanes, planes * block.expansion, 1),
                                   nn.BatchNorm2d(planes * block.expansion, affine=affine_par))
        layers = [block(self.inplanes, planes, stride, dilation=dilation, downsample=downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=dilation))
        return nn.Sequential(*layers)


class DeepLabv2AdaptSeg(ResNetMulti):
    """ttda:940-1016.  ``net(x, feat=True) -> (feat [B,2048,h,w] fp32, logits [B,C,h,w] fp32)``."""

    def __init__(self, num_classes=19, multi_level=True, output_size=(321, 321), restore_from=None):
        super().__init__(Bottleneck, [3, 4, 23, 3], num_classes)
        self.num_classes = num_classes
        self._engines = {}
        self._weights = None
        self._bn_lambda = 0.8
        self._threshold = 0.9
        self._group = None
        self._sam_state = None
        self._precision = "bf16"      # "fp32x2" / "fp32x3": the fp32-exact verification mode (diga_amd/verify.py)
        if restore_from is not None:
            self.restore_from = restore_from
            self.restore()

    # ---- checkpoint loading (ttda:948-1016)
    def restore(self):
        if not hasattr(self, "restore_from"):
            return
        name = self.restore_from.split("/")[-1]
        if self.restore_from[:4] == "http":
            raise ValueError("diga_amd runs offline: download the checkpoint and pass a local path")
        saved = torch.load(self.restore_from, map_location="cpu", weights_only=True)
        if "imagenet" in name or "resnet" in name:
            self.load_imagenet_pretrained(saved)
        elif "Cityscapes_source_class13" in name:
            self.load_13_pretrained(saved["state_dict"])
        elif "cityscapesbest" in name or "models/MaxSquareLoss" in self.restore_from:
            self.load_imagenet_pretrained(saved["state_dict"])
        elif "source" in name:
            self.load_source_pretrained(saved)
        elif "my" in name:
            self.load_imagenet_pretrained(saved["state_dict"])
        elif "baseline" in name or "dl" in name:
            self.load_baseline_pretrained(saved)
        else:
            raise ValueError("Can not find pretrained model type.")

    def load_imagenet_pretrained(self, saved):
        new = self.state_dict().copy()
        for k in saved:
            parts = k.split(".")
            if not self.num_classes == 19 or not parts[1] == "layer5":
                new[".".join(parts[1:])] = saved[k]
        self.load_state_dict(new)

    def load_13_pretrained(self, saved):
        new = self.state_dict().copy()
        for k in saved:
            parts = k.split(".")
            if parts[1] == "layer6":
                new[".".join(parts[1:])] = self._param_13_to_19(saved[k])
            elif not self.num_classes == 19 or not parts[1] == "layer5":
                new[".".join(parts[1:])] = saved[k]
        self.load_state_dict(new)

    def load_source_pretrained(self, saved):
        own = self.state_dict()
        self.load_state_dict({k: v for k, v in saved.items() if k in own})

    def load_baseline_pretrained(self, saved):
        """File key ``layer5.*`` -> model ``layer6.*``; other ``layer*`` keys verbatim (ttda:1000-1016)."""
        new = self.state_dict().copy()
        for k in saved:
            parts = k.split(".")
            if parts[0] == "layer5":
                new[".".join(["layer6"] + parts[1:])] = saved[k]
            elif "layer" in parts[0]:
                new[k] = saved[k]
            elif k in new:
                new[k] = saved[k]
            else:
                raise ValueError("Can not find pretrained model type.")
        self.load_state_dict(new)

    @staticmethod
    def _param_13_to_19(param):
        res = torch.zeros(19, *param.shape[1:]).to(param.device)
        for dst, src in zip((0, 1, 2, 6, 7, 8, 10, 11, 12, 13, 15, 17, 18), range(13)):
            res[dst] = param[src]
        return res

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.invalidate()
        return out

    # ---- execution
    def invalidate(self):
        """Forget packed weights / plans (call after changing parameters in place)."""
        self._weights = None
        self._engines = {}

    def engine(self, B, H, W, stream: bool = False) -> Engine:
        """Plan + buffers for one input shape.  stream=True: the B samples are B consecutive batch-1 frames of a stream
        (per-sample DAM statistics, SAM frame by frame) instead of one batch of B."""
        dev = torch.device("cuda", torch.cuda.current_device())
        exact = self._precision != "bf16"
        if exact and (stream or self._group is not None):
            raise _lib.DigaError("the fp32 verification mode runs single-process batches only (no frames in flight, no process group)")
        if self._weights is None:
            sd = {k: v for k, v in self.state_dict().items()}
            if exact:
                from .verify import ExactWeights
                self._weights = ExactWeights(sd, dev, self.num_classes, {"fp32x2": 2, "fp32x3": 3}[self._precision])
            else:
                self._weights = NetWeights(sd, dev, self.num_classes)
        key = (B, H, W, bool(stream))
        if key not in self._engines:
            if exact:
                from .verify import ExactEngine
                e = ExactEngine(self._weights, B, H, W, bn_lambda=self._bn_lambda, confidence_threshold=self._threshold)
            else:
                e = Engine(self._weights, B, H, W, bn_lambda=self._bn_lambda, group=None if stream else self._group,
                           confidence_threshold=self._threshold, stream=stream, autotune=True)
            if self._sam_state is not None:      # all shapes share the one cross-frame state (ttda:1702-1705)
                e.run_proto, e.run_flag, e.state_flag = self._sam_state
            self._engines[key] = e
        return self._engines[key]

    def forward(self, x, feat=False):
        if not x.is_cuda:
            raise _lib.DigaError("DeepLabv2AdaptSeg.forward needs a device tensor: there is no CPU fallback")
        x = x.to(torch.float32).contiguous()
        e = self.engine(*[int(x.shape[0]), int(x.shape[2]), int(x.shape[3])])
        e.forward_feat(x, want_aux=not feat)
        if feat:
            return e.feat_nchw(), e.logits.clone()
        return e.aux_logits.clone(), e.logits.clone()      # (x1, x2), adaptseg:190-194


# ------------------------------------------------------------------------------------------------
# DIGA (ttda:1562-1841)
# ------------------------------------------------------------------------------------------------
class DIGA(_ModuleBase):
    """Backward-free TTA module: DAM (BN statistic mixing) + SAM (prototype classifier) + fusion.
    A pytorch_lightning.LightningModule when Lightning is importable (ttda:1562), else an nn.Module with the same hooks."""

    def __init__(self, net, dataset_info: dict = {}, cfg: object = None, process_group=None, precision: str = "bf16"):
        """precision: "bf16" (the product path) or "fp32x2" / "fp32x3" -- the fp32-exact verification mode: the same kernels
        with split bf16 operand planes and fp32 activation storage (diga_amd/verify.py), for end-to-end parity runs."""
        super().__init__()
        if precision not in ("bf16", "fp32x2", "fp32x3"):
            raise ValueError(f"precision {precision!r}: expected 'bf16', 'fp32x2' or 'fp32x3'")
        if _pl is not None:
            # ttda:1569: hparams.net / dataset_info / cfg as the reference keeps them (the build's own arguments stay out of it)
            self.save_hyperparameters(logger=False, ignore=["process_group", "precision"])
            for k in ("dataset_info", "cfg"):                 # item AND attribute access on both (ttda:1573 vs ttda:1640)
                self.hparams[k] = _attr(self.hparams[k])
        else:
            self.hparams = AttrDict(net=net, dataset_info=_attr(dataset_info), cfg=_attr(cfg))
            self.loggers = []
        self.net = net(num_classes=self.hparams.dataset_info["num_classes"],
                       output_size=self.hparams.dataset_info["image_size"])
        self.test_acc = nn.ModuleList([SegmentationMetric(num_classes=self.hparams.dataset_info["num_classes"], ignore_index=255)
                                       for _ in self.hparams.dataset_info["test_list"]])
        self.test_step_global = 0
        self.logged = {}
        self._epoch_result = None      # set by the first epoch-end hook of an epoch, cleared by the next test step
        self._replace_bn()
        self._configure_bn_running_stats()
        Cn = self.hparams.dataset_info["num_classes"]
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if dev is not None:
            self._run_proto = torch.zeros(Cn, FEAT_CH, dtype=torch.float32, device=dev)
            self._run_flag = torch.zeros(Cn, dtype=torch.float32, device=dev)
            self._state_flag = torch.zeros(1, dtype=torch.int32, device=dev)
            self.net._sam_state = (self._run_proto, self._run_flag, self._state_flag)
        self.net._bn_lambda = float(self.hparams.cfg.bn_lambda)
        self.net._threshold = float(self.hparams.cfg.confidence_threshold)
        self.net._group = process_group
        self.net._precision = precision

    # ---- cross-frame state with the reference's lazy-attribute semantics (ttda:1702-1705)
    if _pl is None:
        @property
        def device(self):          # LightningModule's own property otherwise (the Trainer moves the module)
            return torch.device("cuda", torch.cuda.current_device())

    def _cuda(self):
        """The device every kernel call runs on: the current HIP device (there is no CPU path)."""
        return torch.device("cuda", torch.cuda.current_device())

    def __getattr__(self, name):
        if name in ("classifier_running_proto", "classifier_running_proto_exists_flag"):
            d = self.__dict__
            if "_state_flag" in d and int(d["_state_flag"].item()) != 0:
                return d["_run_proto"] if name == "classifier_running_proto" else d["_run_flag"]
            raise AttributeError(name)
        return super().__getattr__(name)

    def __delattr__(self, name):
        if name in ("classifier_running_proto", "classifier_running_proto_exists_flag"):
            self._state_flag.zero_()
            self._run_proto.zero_()
            self._run_flag.zero_()
            return
        super().__delattr__(name)

    def _in_trainer(self) -> bool:
        """Attached to a Lightning Trainer?  The private attribute is `_trainer` in Lightning 1.6+ / 2.x (the public `trainer`
        property RAISES when detached in 2.x) and the plain attribute `trainer` before that."""
        if _pl is None:
            return False
        d = self.__dict__
        return d.get("_trainer") is not None or d.get("trainer") is not None

    def log(self, name, value, **kw):
        self.logged[name] = float(value) if torch.is_tensor(value) and value.numel() == 1 else value
        if self._in_trainer():         # inside trainer.test: Lightning's own logging too
            super().log(name, value, **kw)

    def on_test_start(self):
        pass

    def on_test_epoch_end(self):
        """Lightning >= 2.0 spelling of the epoch-end hook (test_epoch_end(outputs) is the 1.x one the reference uses, ttda:1647).
        Lightning 1.x fires BOTH hooks: whichever comes second in an epoch returns the first one's result and logs nothing (the
        metrics have been reset by then; computing them again would overwrite the logged mIoU with NaN)."""
        return self.test_epoch_end(None)

    # ---- reference API
    def forward(self, x: torch.Tensor):
        outputs = self.net(x)
        if isinstance(outputs, tuple):
            outputs = outputs[-1]
        return outputs

    def step(self, batch: Any):
        x, y, shape_, name_ = batch
        outputs, info = self.forward_and_adapt((x, y))
        loss = torch.tensor(0.0)
        eng = self._last_engine
        y_dev = self._labels_on_device(y)
        up = torch.empty(y.shape[0], eng.C, y.shape[-2], y.shape[-1], dtype=torch.float32, device=eng.dev)
        eng.hist.zero_()
        eng.tail(y_dev, probs_up=up)
        info = dict(info, hist=eng.hist.reshape(eng.C, eng.C).clone(), pred=eng.pred)
        return loss, up.cpu(), y.cpu(), info

    def test_step(self, batch: Any, batch_idx: int, dataloader_idx: int = 0, fast: bool = True):
        """fast=True keeps the [B,C,H,W] probabilities on the device (the metric only needs the fused
        argmax + confusion counts); fast=False reproduces step()'s host round trip (ttda:1605-1611)."""
        x, y, shape_, name_ = batch
        if fast:
            self.forward_and_adapt((x, y), to_cpu=False)
            eng = self._last_engine
            eng.hist.zero_()
            eng.tail(self._labels_on_device(y))
            self.test_acc[dataloader_idx].add_counts(eng.hist.reshape(eng.C, eng.C))
            acc = self.test_acc[dataloader_idx].compute()
            loss = torch.tensor(0.0)
        else:
            loss, preds, targets, info = self.step(batch)
            self.test_acc[dataloader_idx].add_counts(info["hist"])
            acc = self.test_acc[dataloader_idx].compute()
        self._epoch_result = None
        self.log("test/loss", loss)
        self.log("test/acc", acc)
        self.test_step_global += 1
        return {"loss": loss}

    def test_step_stream(self, batches, batch_idx: int, dataloader_idx: int = 0):
        """K consecutive batch-1 test batches of the stream in ONE pass (frames in flight): the trunk runs once over the
        K frames with per-sample DAM statistics, SAM / fusion / metric then run frame by frame in stream order.  No
        counterpart in the reference (it has no look-ahead); every frame's outputs, the running prototypes and the metric
        are bit-identical to K test_step calls (tests/test_gpu_net.py).  Returns the K per-frame probability maps (device)."""
        xs, ys = [b[0] for b in batches], [b[1] for b in batches]
        if any(int(x.shape[0]) != 1 for x in xs):
            raise ValueError("test_step_stream pipelines batch-1 frames; use test_step for real batches")
        x = torch.cat([t.to(self._cuda(), torch.float32) for t in xs]).contiguous()
        y = torch.cat([self._labels_on_device(t) for t in ys]).contiguous()
        eng = self.net.engine(int(x.shape[0]), int(x.shape[2]), int(x.shape[3]), stream=True)
        self._last_engine = eng
        probs = eng.forward_and_adapt(x, y, self.hparams.cfg)
        eng.hist.zero_()
        eng.tail(y)
        self.test_acc[dataloader_idx].add_counts(eng.hist.reshape(eng.C, eng.C))
        self._epoch_result = None
        self.log("test/loss", torch.tensor(0.0))
        self.log("test/acc", self.test_acc[dataloader_idx].compute())
        self.test_step_global += len(batches)
        return probs

    def _labels_on_device(self, y):
        y = y.to(self._cuda())
        if y.dtype not in (torch.int32, torch.int64):
            y = y.long()
        return y.contiguous()

    def forward_and_adapt(self, pair, to_cpu: bool = True):
        """ttda:1623-1645.  Returns (probs [B,C,h,w], {}); probs on the CPU as the reference does unless
        to_cpu=False."""
        x, y = pair
        x = x.to(self._cuda(), torch.float32).contiguous()
        eng = self.net.engine(int(x.shape[0]), int(x.shape[2]), int(x.shape[3]))
        self._last_engine = eng
        probs = eng.forward_and_adapt(x, self._labels_on_device(y), self.hparams.cfg)
        return (probs.cpu() if to_cpu else probs), {}

    def test_epoch_end(self, outputs: List[Any] = None):
        if self._epoch_result is not None:          # the other spelling of the hook has already closed this epoch
            return self._epoch_result
        accs = [m.compute() for m in self.test_acc]
        self.log("test/acc/mean", sum(accs) / len(accs))
        res = {}
        for i, m in enumerate(self.test_acc):
            res[i] = dict(miou=float(accs[i]), miou16=float(m.compute_iou("16")), miou13=float(m.compute_iou("13")),
                          class_iou=m.compute_class_iou().tolist())
            self.log(f"test/acc/dataloaderr_idx_{i}", res[i]["miou"])
            self.log(f"test/acc/dataloaderr16_idx_{i}", res[i]["miou16"])
            self.log(f"test/acc/dataloaderr13_idx_{i}", res[i]["miou13"])
        for m in self.test_acc:
            m.reset()
        self._epoch_result = res
        return res

    # ---- SAM helpers with the reference's signatures, each one a kernel call (NCHW fp32 tensors in/out).
    # The caller's features are arbitrary fp32 values and the reference computes these functions in fp32 (ttda:1712-1805), so
    # the helpers run the fp32-feature kernels (diga_proto_pool_f32 / diga_proto_classify_fuse_f32): nothing is rounded to bf16.
    def _feat_nhwc(self, feature):
        """NCHW fp32 -> NHWC fp32 [P][CH] on the device."""
        feature = feature.to(self._cuda(), torch.float32).contiguous()
        B, CH, h, w = feature.shape
        out = torch.empty(B * h * w * CH, dtype=torch.float32, device=self._cuda())
        call("diga_nchw_f32_to_nhwc_f32", _p(feature), _p(out), B, h * w, CH, _stream())
        return out

    def _feat_nhwc_bf16(self, feature):
        feature = feature.to(self._cuda(), torch.float32).contiguous()
        B, CH, h, w = feature.shape
        out = torch.empty(B * h * w * CH, dtype=torch.bfloat16, device=self._cuda())
        call("diga_nchw_f32_to_nhwc_bf16", _p(feature), _p(out), B, h * w, CH, _stream())
        return out

    def high_confident_proto_label(self, outputs, threshold):
        """ttda:1662-1676.  outputs: logits [B,C,H,W] -> int64 [B,H,W] in {0..C-1, 255}."""
        lg = outputs.to(self._cuda(), torch.float32).contiguous()
        B, Cn, h, w = lg.shape
        lab = torch.empty(B * h * w, dtype=torch.int32, device=self._cuda())
        call("diga_head_finish_pseudo_label", None, 1, Cn, None, _p(lg), None, _p(lab), B, Cn, h * w, _f32(threshold), 0, _stream())
        return lab.reshape(B, h, w).long()

    def cal_proto(self, feature, y):
        """ttda:1788-1805 -> (proto [C,CH], with_flag [C])."""
        Cn = self.hparams.dataset_info["num_classes"]
        B, CH, h, w = feature.shape
        f = self._feat_nhwc(feature)
        labs = y.to(self._cuda()).reshape(-1).to(torch.int32).contiguous()
        Pn = B * h * w
        work = torch.empty(call("diga_proto_pool_work_bytes", 1, Pn, Cn, CH), dtype=torch.uint8, device=self._cuda())
        sums = torch.empty(Cn * CH, dtype=torch.float32, device=self._cuda())
        counts = torch.empty(Cn, dtype=torch.int32, device=self._cuda())
        call("diga_proto_pool_f32", _p(f), _p(labs), 1, Pn, Cn, CH, _p(work), _p(sums), _p(counts), _stream())
        # mean = sum / n via the update kernel (state_flag = 0 -> running := instance)
        inst = torch.empty(Cn, CH, dtype=torch.float32, device=self._cuda())
        iflag = torch.empty(Cn, dtype=torch.float32, device=self._cuda())
        run, rflag = torch.zeros_like(inst), torch.zeros_like(iflag)
        cnt2 = torch.cat([counts, torch.zeros_like(counts)])
        sums2 = torch.cat([sums, torch.zeros_like(sums)])
        call("diga_proto_update", _p(sums2), _p(cnt2), Cn, CH, _p(inst), _p(iflag), _p(run), _p(rflag),
             _p(torch.zeros(1, dtype=torch.int32, device=self._cuda())), 1.0, 0.0, _stream())
        return inst, iflag

    # ---- optional extension, NOT part of the reference's DIGA and not called by forward_and_adapt
    def instance_labels(self, y, connectivity: int = 8):
        """Connected components of the per-class masks of a label map y [B,h,w] (e.g. the pseudo-labels of
        high_confident_proto_label) -> (inst int64 [B,h,w]: 0 = background / ignore, else the component's number inside
        its class, from 1 in raster order; ncomp int64 [B,C])."""
        from . import instances
        inst, ncomp = instances.ccl_label(y.to(self._cuda()), self.hparams.dataset_info["num_classes"], connectivity)
        return inst.long(), ncomp.long()

    def cal_instance_proto(self, feature, y, connectivity: int = 8, max_instances: int = 4096):
        """cal_proto per connected component instead of per class: feature [B,CH,h,w], y [B,h,w] ->
        instances.InstanceTable (protos [N,CH], counts, classes, samples, total), rows in (sample, class, number) order.
        Unlike the reference-mirroring helpers this optional extension pools the features as the network stores them, in
        bf16: fp32 input is rounded to bf16 first (exact for features that came out of net(x, feat=True))."""
        from . import instances
        Cn = self.hparams.dataset_info["num_classes"]
        B, CH, h, w = feature.shape
        labs = y.to(self._cuda()).reshape(B, h, w)
        inst, ncomp = instances.ccl_label(labs, Cn, connectivity)
        return instances.instance_prototypes(self._feat_nhwc_bf16(feature).reshape(B * h * w, CH), labs, inst, ncomp, Cn, max_instances)

    def cal_pred_of_proto(self, feature, proto, exists_flag, tau=1.0):
        """ttda:1712-1734 -> weight [B,C,H,W] (NaN where no class exists / all underflow)."""
        B, CH, h, w = feature.shape
        Cn = proto.shape[0]
        f = self._feat_nhwc(feature)
        pr = proto.to(self._cuda(), torch.float32).contiguous()
        fl = exists_flag.to(self._cuda(), torch.float32).contiguous()
        zeros = torch.zeros(B, Cn, h, w, dtype=torch.float32, device=self._cuda())
        out = torch.empty_like(zeros)
        work = torch.empty(call("diga_proto_classify_work_bytes", B * h * w, CH), dtype=torch.uint8, device=self._cuda())
        call("diga_proto_classify_fuse_f32", _p(f), _p(zeros), _p(pr), _p(fl), _p(pr), _p(fl), _p(out), _p(work), B, Cn, h * w, CH,
             float(tau), 1.0, 0.0, 1.0, 0.0, _stream())
        return out

    def _update_classifier_proto(self, feature, y, classifier_proto, classifier_proto_exists_flag, rho=None):
        """ttda:1736-1765 (operates on clones, returns the new (proto, flag))."""
        Cn = self.hparams.dataset_info["num_classes"]
        B, CH, h, w = feature.shape
        f = self._feat_nhwc(feature)
        Pn = B * h * w
        labs = torch.cat([torch.full((Pn,), 255, dtype=torch.int32, device=self._cuda()),
                          y.to(self._cuda()).reshape(-1).to(torch.int32)]).contiguous()
        work = torch.empty(call("diga_proto_pool_work_bytes", 2, Pn, Cn, CH), dtype=torch.uint8, device=self._cuda())
        sums = torch.empty(2 * Cn * CH, dtype=torch.float32, device=self._cuda())
        counts = torch.empty(2 * Cn, dtype=torch.int32, device=self._cuda())
        call("diga_proto_pool_f32", _p(f), _p(labs), 2, Pn, Cn, CH, _p(work), _p(sums), _p(counts), _stream())
        run = classifier_proto.to(self._cuda(), torch.float32).clone().contiguous()
        rflag = classifier_proto_exists_flag.to(self._cuda(), torch.float32).clone().contiguous()
        inst = torch.empty(Cn, CH, dtype=torch.float32, device=self._cuda())
        iflag = torch.empty(Cn, dtype=torch.float32, device=self._cuda())
        rho = 0.0 if rho is None else float(rho)
        call("diga_proto_update", _p(sums), _p(counts), Cn, CH, _p(inst), _p(iflag), _p(run), _p(rflag),
             _p(torch.ones(1, dtype=torch.int32, device=self._cuda())), _f32(1 - rho), _f32(rho), _stream())
        return run, rflag

    def multi_proto_label(self, feature, outputs, y, class_names, cfg):
        """ttda:1678-1710 on caller-supplied feature / logits tensors (stateful). Returns (multi_pred, {})."""
        cfg = _attr(cfg)
        B, CH, h, w = feature.shape
        Cn = outputs.shape[1]
        dev = self._cuda()
        f = self._feat_nhwc(feature)
        lg = outputs.to(dev, torch.float32).contiguous()
        yd = self._labels_on_device(y)
        Pn = B * h * w
        labels = torch.empty(2 * Pn, dtype=torch.int32, device=dev)
        call("diga_head_finish_pseudo_label", None, 1, Cn, None, _p(lg), None, _p(labels), B, Cn, h * w,
             _f32(cfg.confidence_threshold), 0, _stream())
        call("diga_nearest_down_labels", _p(yd), int(yd.dtype == torch.int64), _p(labels[Pn:]), B, yd.shape[1], yd.shape[2], h, w,
             0, _stream())
        work = torch.empty(call("diga_proto_pool_work_bytes", 2, Pn, Cn, CH), dtype=torch.uint8, device=dev)
        sums = torch.empty(2 * Cn * CH, dtype=torch.float32, device=dev)
        counts = torch.empty(2 * Cn, dtype=torch.int32, device=dev)
        call("diga_proto_pool_f32", _p(f), _p(labels), 2, Pn, Cn, CH, _p(work), _p(sums), _p(counts), _stream())
        if self._run_proto.shape[1] != CH:          # helper used with a non-2048 feature width (tests)
            self._run_proto = torch.zeros(Cn, CH, dtype=torch.float32, device=dev)
            self._run_flag.zero_()
            self._state_flag.zero_()
        inst = torch.empty(Cn, CH, dtype=torch.float32, device=dev)
        iflag = torch.empty(Cn, dtype=torch.float32, device=dev)
        rho = float(cfg.proto_rho)
        call("diga_proto_update", _p(sums), _p(counts), Cn, CH, _p(inst), _p(iflag), _p(self._run_proto), _p(self._run_flag),
             _p(self._state_flag), _f32(1 - rho), _f32(rho), _stream())
        zeros = torch.zeros(B, Cn, h, w, dtype=torch.float32, device=dev)
        out = torch.empty_like(zeros)
        pl = float(cfg.proto_lambda)
        cwork = torch.empty(call("diga_proto_classify_work_bytes", Pn, CH), dtype=torch.uint8, device=dev)
        call("diga_proto_classify_fuse_f32", _p(f), _p(zeros), _p(inst), _p(iflag), _p(self._run_proto), _p(self._run_flag), _p(out),
             _p(cwork), B, Cn, h * w, CH, 1.0, _f32(pl), _f32(1 - pl), 1.0, 0.0, _stream())
        return out, {}

    # ---- BN replacement (ttda:1808-1833)
    def _configure_bn_running_stats(self):
        for m in self.net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.lambda_.data = torch.tensor(self.hparams.cfg.bn_lambda)
        self.net._bn_lambda = float(self.hparams.cfg.bn_lambda)
        self.net.invalidate()          # packed per-layer lambdas / cached engines are stale now

    def _replace_bn(self):
        for n, module in list(self.net.named_modules()):
            if isinstance(module, nn.BatchNorm2d) and not isinstance(module, SIFABatchNorm2dTrainable):
                parent = self.net
                *path, leaf = n.split(".")
                for a in path:
                    parent = getattr(parent, a)
                setattr(parent, leaf, SIFABatchNorm2dTrainable().from_bn(module))
        self.net.invalidate()

    @property
    def wandb(self):
        for lg in self.loggers:
            if "wandb" in lg.__module__:
                return lg
        raise ValueError("No wandb logger found")
