"""Entry point mirroring the reference's `src/eval.py` (Hydra is not a dependency here: the two yaml groups the hot
path needs are read with PyYAML and the same keys).  Unlike the reference's evaluate() (eval_task.py:33) it passes
`dataset_info` to the model, without which DIGA.__init__ cannot run.

    python src/eval.py [datamodule.frames=20] [datamodule.batch_size=1] [datamodule.frames_in_flight=4] [model.cfg.bn_lambda=0.8]
                       [model/net=synthia_source] [precision=fp32x3]

`group/option=name` picks a yaml of a config group as Hydra's override syntax does (model/net=synthia_source reads
configs/model/net/synthia_source.yaml); `a.b.c=value` sets one key; `precision=fp32x2|fp32x3` runs the fp32-exact verification mode.
"""
import functools
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLASS_NAMES = ["road", "sidewalk", "building", "wall", "fence", "pole", "light", "sign", "vegetation", "terrain", "sky",
               "person", "rider", "car", "truck", "bus", "train", "motocycle", "bicycle"]


def _yaml(*parts):
    path = os.path.join(ROOT, "configs", *parts) + ".yaml"
    if not os.path.exists(path):
        raise FileNotFoundError(f"no config {path}")
    with open(path) as f:
        return yaml.safe_load(f)


def load_cfg(argv):
    cfg = _yaml("eval")
    groups = {k: v for k, v in (a.split("=", 1) for a in argv if "/" in a.split("=", 1)[0] or a.split("=", 1)[0] == "model")}
    model = _yaml("model", groups.get("model", cfg["model"]))
    net = _yaml("model", "net", groups.get("model/net", model["defaults"][0]["net"]))
    cfg["model"] = dict(model, net=net)
    cfg.setdefault("precision", "bf16")
    for a in argv:                       # key.path=value overrides
        k, v = a.split("=", 1)
        if k in groups:
            continue
        d = cfg
        *path, leaf = k.split(".")
        for p in path:
            d = d[p]
        d[leaf] = yaml.safe_load(v)
    return cfg


def main(argv):
    from diga_amd import synth
    from src.models import ttda_module as T
    cfg = load_cfg(argv)
    dm = cfg["datamodule"]
    data_dir = os.environ.get("UDATADIR", "./data")
    restore = cfg["model"]["net"]["restore_from"].replace("${paths.data_dir}", data_dir)
    have_ckpt = os.path.exists(restore)
    net = functools.partial(T.DeepLabv2AdaptSeg, multi_level=cfg["model"]["net"]["multi_level"],
                            restore_from=restore if have_ckpt else None)
    info = dict(num_classes=19, image_size=tuple(dm["image_size"]), test_list=["synthetic"], class_names=CLASS_NAMES)
    m = T.DIGA(net=net, dataset_info=info, cfg=cfg["model"]["cfg"], precision=cfg["precision"])
    if not have_ckpt:
        print(f"[eval] {restore} not found: using the seeded synthetic weights")
        m.net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(0).items()}, strict=False)
        m._configure_bn_running_stats()
    H, W = dm["image_size"]
    bs, nfr, fif = int(dm["batch_size"]), int(dm["frames"]), int(dm.get("frames_in_flight", 1))
    t = 0
    while t < nfr:
        k = min(fif, nfr - t) if bs == 1 else 1
        batches = []
        for j in range(k):
            x, y = synth.synth_frame(t + j, bs, H, W)
            batches.append((torch.from_numpy(x), torch.from_numpy(y), None, f"synthetic/{t + j}"))
        if k > 1:                          # batch-1 stream, k frames in flight: same results as k test_step calls
            m.test_step_stream(batches, t)
        else:
            m.test_step(batches[0], t)
        t += k
    res = m.test_epoch_end()
    for i, r in res.items():
        print(f"test/acc/dataloaderr_idx_{i} {r['miou']:.5f}  dataloaderr16_idx_{i} {r['miou16']:.5f}  dataloaderr13_idx_{i} {r['miou13']:.5f}")
    return res


if __name__ == "__main__":
    main(sys.argv[1:])
