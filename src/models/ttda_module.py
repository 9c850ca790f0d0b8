"""Drop-in module path of the reference (`_target_: src.models.ttda_module.DIGA`, configs/model/diga.yaml:3):
re-exports the MI355X implementation."""
from diga_amd.ttda_module import (DIGA, DeepLabv2AdaptSeg, SIFABatchNorm2d, SIFABatchNorm2dTrainable,  # noqa: F401
                                  SegmentationMetric)
