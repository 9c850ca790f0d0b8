"""Synthetic checkpoint files for every branch of DeepLabv2AdaptSeg.restore() (ttda:948-998).  Reference-free: shared by
oracle/gen_golden.py (which loads the files with the REFERENCE's loader and records hashes in tests/golden/loader.json) and by
tests/test_cpu_host.py (which loads them with diga_amd's loader and compares)."""
import os

import torch

LOADER_KEYS = ["conv1.weight", "bn1.running_var", "layer2.1.bn2.bias", "layer3.5.conv2.weight", "layer4.0.downsample.1.running_mean",
               "layer6.conv2d_list.0.weight", "layer6.conv2d_list.1.bias", "layer6.conv2d_list.3.weight"]


def loader_branch_files(p, tmp):
    """The synthetic checkpoint files of the other restore() branches (ttda:948-998), built from synth_params: shared by this
    generator (which loads them with the REFERENCE) and tests/test_cpu_host.py (which loads them with diga_amd and compares hashes).
    Returns {case: (path, description)}."""
    full = {k: torch.from_numpy(v) for k, v in p.items()}
    files = {}
    # 'source' in the file name -> load_source_pretrained: plain state dict, keys the model does not have are dropped, strict load
    sd = dict(full)
    sd["not_in_model.weight"] = torch.ones(3)
    files["source"] = (os.path.join(tmp, "synth_SYNTHIA_source.pth"), sd)
    # a source checkpoint that lacks a model key -> the strict load raises
    sd2 = dict(full)
    del sd2["layer3.5.conv2.weight"]
    files["source_missing"] = (os.path.join(tmp, "synth_short_source.pth"), sd2)
    # 'imagenet' / 'resnet' in the name -> load_imagenet_pretrained: keys carry ONE prefix component; layer5.* skipped for 19 classes
    files["imagenet"] = (os.path.join(tmp, "synth_resnet101_imagenet.pth"), {"Scale." + k: v for k, v in full.items()})
    # path contains models/MaxSquareLoss (configs/model/net/gta5_synthia_source.yaml) -> load_imagenet_pretrained(saved["state_dict"])
    os.makedirs(os.path.join(tmp, "models", "MaxSquareLoss"), exist_ok=True)
    files["maxsquare"] = (os.path.join(tmp, "models", "MaxSquareLoss", "gta5+synthiabest.pth"), {"state_dict": {"module." + k: v for k, v in full.items()}})
    # 'my' in the name -> the same through saved["state_dict"]
    files["my"] = (os.path.join(tmp, "my_run.pth"), {"state_dict": {"module." + k: v for k, v in full.items()}})
    # 'Cityscapes_source_class13' -> load_13_pretrained: a 13-class layer6 head scattered into 19 rows (_param_13_to_19)
    idx13 = [0, 1, 2, 6, 7, 8, 10, 11, 12, 13, 15, 17, 18]
    sd13 = {}
    for k, v in full.items():
        sd13["module." + k] = v[idx13].clone() if k.startswith("layer6.") else v
    files["class13"] = (os.path.join(tmp, "Cityscapes_source_class13.pth"), {"state_dict": sd13})
    for path, obj in files.values():
        torch.save(obj, path)
    return {k: v[0] for k, v in files.items()}
