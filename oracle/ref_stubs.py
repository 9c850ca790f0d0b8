"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Stand-in modules that let the *reference's own source* (``/root/reference/src/models/ttda_module.py``)
be imported in the build container, where pytorch_lightning / torchmetrics / torchvision / wandb are
not installed.  The stubs contain no DIGA logic at all: they only provide the base classes and names
that the reference file mentions at import time (ttda_module.py:8-10,17,22-23; adaptseg.py:6).

Used by ``oracle/gen_golden.py`` (golden-vector generation) and by nothing else.  The reference does
not exist on the GPU box, so nothing under ``tests/ -m gpu``, ``smoke()`` or ``bench.py`` touches this.
"""
import inspect
import sys
import types

import torch
import torch.nn as nn


class AttrDict(dict):
    """dict with attribute access (the reference uses both d["k"], ttda_module.py:1573, and d.k, :1640)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, AttrDict):
        return AttrDict({k: _wrap(x) for k, x in v.items()})
    return v


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.__dict__["hparams"] = AttrDict()
        self.loggers = []

    def save_hyperparameters(self, *args, logger=True, ignore=()):
        frame = inspect.currentframe().f_back
        names, _, _, values = inspect.getargvalues(frame)
        for n in names:
            if n == "self" or n in ignore:
                continue
            self.hparams[n] = _wrap(values[n])

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def log(self, *a, **k):
        return None


class _Dummy:
    def __init__(self, *a, **k):
        pass


class Metric(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def add_state(self, name, default, dist_reduce_fx=None):
        self.register_buffer(name, default)

    def forward(self, *a, **k):
        self.update(*a, **k)
        return self.compute()

    def reset(self):
        pass


class MaxMetric(Metric):
    def update(self, *a, **k):
        pass

    def compute(self):
        return torch.tensor(0.0)


def install():
    """Register the stub modules in sys.modules (idempotent)."""
    if "pytorch_lightning" in sys.modules and getattr(sys.modules["pytorch_lightning"], "_diga_stub", False):
        return
    pl = types.ModuleType("pytorch_lightning")
    pl._diga_stub = True
    pl.LightningModule = LightningModule
    pl.LightningDataModule = _Dummy
    pl.Trainer = _Dummy
    pl.Callback = _Dummy
    sys.modules["pytorch_lightning"] = pl

    tm = types.ModuleType("torchmetrics")
    tm.Metric = Metric
    tm.MaxMetric = MaxMetric
    tm.JaccardIndex = MaxMetric
    sys.modules["torchmetrics"] = tm
    tmc = types.ModuleType("torchmetrics.classification")
    tmca = types.ModuleType("torchmetrics.classification.accuracy")
    tmca.Accuracy = MaxMetric
    tmc.accuracy = tmca
    tm.classification = tmc
    sys.modules["torchmetrics.classification"] = tmc
    sys.modules["torchmetrics.classification.accuracy"] = tmca

    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tvu.draw_segmentation_masks = lambda *a, **k: None
    tvt = types.ModuleType("torchvision.transforms")
    tvm = types.ModuleType("torchvision.models")
    tv.utils, tv.transforms, tv.models = tvu, tvt, tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.utils"] = tvu
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.models"] = tvm

    sys.modules["wandb"] = types.ModuleType("wandb")


def import_reference(ref_root="/root/reference"):
    """Import the reference's ttda_module by path and return it."""
    install()
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    from src.models import ttda_module as T  # noqa: the *reference's* src package
    return T
