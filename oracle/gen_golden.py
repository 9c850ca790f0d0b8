"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by running the REFERENCE's own source.

Runs only in the build container (needs /root/reference).  The reference's modules are imported by
path with the stand-in modules of ``oracle/ref_stubs.py``; its classes/methods are then *called* on
seeded inputs and the inputs + outputs are written as small fixtures.  No reference source text is
copied anywhere.  The GPU box never runs this (the reference does not travel); it only reads the
fixtures.

    python oracle/gen_golden.py            # writes tests/golden/{fn_bn,fn_sam,fn_tail,net_small,net_full}.npz, loader.json
"""
import functools
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_stubs  # noqa: E402

T = ref_stubs.import_reference()          # the reference's src.models.ttda_module
sys.path.append(ROOT)
from diga_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CLASS_NAMES = ["road", "sidewalk", "building", "wall", "fence", "pole", "light", "sign", "vegetation", "terrain",
               "sky", "person", "rider", "car", "truck", "bus", "train", "motocycle", "bicycle"]
CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_diga(load_synth=True, cfg=None):
    cfg = ref_stubs.AttrDict(cfg or CFG)
    m = T.DIGA(net=functools.partial(T.DeepLabv2AdaptSeg, multi_level=True, restore_from=None),
               dataset_info=dict(num_classes=19, image_size=(321, 321), test_list=[0], class_names=CLASS_NAMES),
               cfg=cfg).eval()
    if load_synth:
        sd = {k: torch.from_numpy(v) for k, v in synth.synth_params(0).items()}
        missing, unexpected = m.net.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all(k.endswith("lambda_") for k in missing), missing
        m._configure_bn_running_stats()
    return m


def np_(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------
def gen_fn_bn():
    """SIFABatchNorm2dTrainable.forward in eval mode, B = 1, 2, 3 (half_first rule)."""
    out = {}
    g = torch.Generator().manual_seed(11)
    for B in (1, 2, 3):
        C, H, W = 8, 5, 7
        bn = torch.nn.BatchNorm2d(C)
        bn.running_mean = torch.randn(C, generator=g)
        bn.running_var = torch.rand(C, generator=g) + 0.5
        bn.weight.data = torch.rand(C, generator=g) + 0.5
        bn.bias.data = torch.randn(C, generator=g) * 0.1
        sb = T.SIFABatchNorm2dTrainable().from_bn(bn).eval()
        sb.lambda_.data = torch.tensor(0.8)
        x = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
        with torch.no_grad():
            y = sb(x)
        for k, v in dict(x=x, y=y, running_mean=bn.running_mean, running_var=bn.running_var,
                         weight=bn.weight, bias=bn.bias).items():
            out[f"B{B}_{k}"] = np_(v)
    np.savez_compressed(os.path.join(GOLD, "fn_bn.npz"), **out)


def gen_fn_sam(m):
    out = {}
    g = torch.Generator().manual_seed(22)
    C, CH, h, w = 19, 64, 9, 11
    # -- pseudo labels: random logits + engineered rows straddling the 0.9 bar and exact ties
    logits = torch.randn(2, C, h, w, generator=g) * 4
    # two-class rows: p_max = sigmoid(d); d = ln 9 gives exactly-at-threshold in exact arithmetic
    d0 = float(np.log(9.0))
    for i, d in enumerate([d0 - 1e-3, d0 - 1e-6, d0, d0 + 1e-6, d0 + 1e-3]):
        logits[0, :, 0, i] = -1e4
        logits[0, 3, 0, i] = d
        logits[0, 7, 0, i] = 0.0
    logits[0, :, 1, 0] = 0.0                       # 19-way tie -> argmax 0, p = 1/19 -> 255
    logits[0, :, 1, 1] = -50.0
    logits[0, 5, 1, 1] = 20.0
    logits[0, 11, 1, 1] = 20.0                      # exact 2-way tie at 0.5 -> 255
    logits[0, :, 1, 2] = float("nan")              # NaN row
    with torch.no_grad():
        pl = m.high_confident_proto_label(logits.clone(), 0.9)
    out["pl_logits"], out["pl_labels"] = np_(logits), np_(pl)
    # -- cal_proto: empty classes + 255
    for B in (1, 2):
        feat = torch.randn(B, CH, h, w, generator=g).relu().bfloat16().float()   # bf16-representable: the kernels store features as bf16
        lab = torch.randint(0, 12, (B, h, w), generator=g)         # classes 12..18 empty
        lab[torch.rand(B, h, w, generator=g) < 0.2] = 255
        lab[lab == 5] = 4                                           # class 5 empty too
        with torch.no_grad():
            proto, flag = m.cal_proto(feat, lab)
        out[f"cp{B}_feat"], out[f"cp{B}_lab"], out[f"cp{B}_proto"], out[f"cp{B}_flag"] = map(np_, (feat, lab, proto, flag))
    # -- _update_classifier_proto: first-seen / EMA / rho = 0
    feat = torch.randn(1, CH, h, w, generator=g).relu().bfloat16().float()
    y = torch.randint(0, 10, (1, h, w), generator=g)
    run = torch.randn(C, CH, generator=g)
    rflag = torch.zeros(C)
    rflag[:5] = 1
    rflag[15] = 1
    with torch.no_grad():
        r1, f1 = m._update_classifier_proto(feat, y, run, rflag, 0.1)
        r0, f0 = m._update_classifier_proto(feat, y, run, rflag, 0)
    for k, v in dict(up_feat=feat, up_y=y, up_run=run, up_flag=rflag, up_run_rho01=r1, up_flag_rho01=f1,
                     up_run_rho0=r0, up_flag_rho0=f0).items():
        out[k] = np_(v)
    # -- cal_pred_of_proto: normal, all-flags-zero -> NaN, far prototypes -> underflow NaN / partial underflow
    feat = (torch.randn(2, CH, h, w, generator=g).relu() * 0.5).bfloat16().float()
    proto = torch.randn(C, CH, generator=g).relu() * 0.5
    flag = (torch.rand(C, generator=g) < 0.7).float()
    far = proto.clone()
    far[:10] += 20.0                              # dist ~160 -> exp underflows for those classes
    with torch.no_grad():
        w_norm = m.cal_pred_of_proto(feat, proto, flag)
        w_zero = m.cal_pred_of_proto(feat, proto, torch.zeros(C))
        w_far = m.cal_pred_of_proto(feat, far, flag)
        w_allfar = m.cal_pred_of_proto(feat, proto + 30.0, flag)
    for k, v in dict(pp_feat=feat, pp_proto=proto, pp_flag=flag, pp_far=far, pp_w=w_norm, pp_w_zero=w_zero,
                     pp_w_far=w_far, pp_w_allfar=w_allfar).items():
        out[k] = np_(v)
    # -- multi_proto_label over 3 frames (state carry), B = 1 and B = 2, plus fusion line
    for B in (1, 2):
        if hasattr(m, "classifier_running_proto"):
            del m.classifier_running_proto
        for t in range(3):
            feat = (torch.randn(B, CH, h, w, generator=g).relu() * 0.5).bfloat16().float()
            lg = torch.randn(B, C, h, w, generator=g) * 5
            y = torch.randint(0, 14, (B, 37, 45), generator=g)
            y[torch.rand(B, 37, 45, generator=g) < 0.1] = 255
            with torch.no_grad():
                mp, _ = m.multi_proto_label(feat, lg, y, CLASS_NAMES, m.hparams.cfg)
                fused = mp * 0.8 + lg.softmax(1) * 0.2
            for k, v in dict(feat=feat, logits=lg, y=y, multi=mp, fused=fused, run=m.classifier_running_proto,
                             rflag=m.classifier_running_proto_exists_flag).items():
                out[f"mpl{B}_{t}_{k}"] = np_(v)
    del m.classifier_running_proto
    np.savez_compressed(os.path.join(GOLD, "fn_sam.npz"), **out)


def gen_fn_tail(m):
    out = {}
    g = torch.Generator().manual_seed(33)
    # nearest down-sampling of y (ttda:1695)
    for name, (H, W, h, w) in dict(a=(512, 1024, 65, 129), b=(720, 1280, 91, 161), c=(37, 45, 9, 11)).items():
        y = torch.randint(0, 19, (1, H, W), generator=g)
        y[torch.rand(1, H, W, generator=g) < 0.05] = 255
        ys = torch.nn.functional.interpolate(y.unsqueeze(1).float(), size=(h, w), mode="nearest").squeeze(1).long()
        if name == "c":
            out["nd_c_y"] = np_(y)
        else:
            out[f"nd_{name}_seed"] = np.array([H, W])
            out[f"nd_{name}_y_sha"] = np.array(sha(np_(y)))
        out[f"nd_{name}_ys"] = np_(ys)
        out[f"nd_{name}_shape"] = np.array([H, W, h, w])
        if name != "c":
            out[f"nd_{name}_y"] = np_(y).astype(np.uint8)
    # step(): bilinear upsample (align_corners=False) + metric
    probs = torch.rand(2, 19, 9, 17, generator=g)
    probs = probs / probs.sum(1, keepdim=True)
    probs[0, :, 2, 3] = float("nan")
    tgt = torch.randint(0, 19, (2, 65, 129), generator=g)
    tgt[torch.rand(2, 65, 129, generator=g) < 0.1] = 255
    up = torch.nn.Upsample(size=(65, 129), mode="bilinear")(probs)
    metric = T.SegmentationMetric(num_classes=19, ignore_index=255)
    metric.update(up, tgt)
    h1 = metric.hist.clone()
    metric.update(up.flip(0), tgt)
    out.update(tail_probs=np_(probs), tail_tgt=np_(tgt), tail_up=np_(up), tail_argmax=np_(up.argmax(1)),
               tail_hist1=np_(h1), tail_hist2=np_(metric.hist),
               tail_miou=np.array([float(metric.compute()), float(metric.compute_iou("16")),
                                   float(metric.compute_iou("13"))]),
               tail_class_iou=np_(metric.compute_class_iou()))
    np.savez_compressed(os.path.join(GOLD, "fn_tail.npz"), **out)


def classifier_mass64(feat, proto, flag):
    """The normaliser of cal_pred_of_proto (ttda:1729-1733), sum_c exp(-||proto_c - f||) * flag_c, evaluated in float64 from the
    reference's own features and prototypes.  Where it is below ~1e-33 the reference's fp32 weights are denormals (a few bits)
    or flushed to 0 (the 0/0 -> NaN branch): its probabilities there are quantisation noise, and the fixtures mark those pixels."""
    f = feat.double()
    mass = torch.zeros(f.shape[0], f.shape[2], f.shape[3], dtype=torch.float64)
    for c in range(proto.shape[0]):
        if float(flag[c]) == 0:
            continue
        d = (f - proto[c].double()[None, :, None, None]).pow(2).sum(1).sqrt()
        mass += torch.exp(-d)
    return mass


def run_sequence(m, frames, taps_first=None):
    """Run forward_and_adapt + step() tail over a list of (x, y) frames with fresh SAM state."""
    if hasattr(m, "classifier_running_proto"):
        del m.classifier_running_proto
    metric = T.SegmentationMetric(num_classes=19, ignore_index=255)
    res = []
    for t, (x, y) in enumerate(frames):
        x, y = torch.from_numpy(x), torch.from_numpy(y)
        with torch.no_grad():
            feat, logits = m.net(x, feat=True)
            pl = m.high_confident_proto_label(logits, 0.9)
            probs, _ = m.forward_and_adapt((x, y))
            up = torch.nn.Upsample(size=y.shape[-2:], mode="bilinear")(probs)
            metric.update(up, y)
            sm = logits.softmax(1).topk(2, dim=1)[0]
            top2 = up.topk(2, dim=1)[0]
            inst_proto, inst_flag = m.cal_proto(feat, pl)
            lowmass = (classifier_mass64(feat, inst_proto, inst_flag) < 1e-33) | \
                      (classifier_mass64(feat, m.classifier_running_proto, m.classifier_running_proto_exists_flag) < 1e-33)
        res.append(dict(feat=feat, logits=logits, pseudo=pl, probs=probs, pred=up.argmax(1),
                        run=m.classifier_running_proto.clone(), rflag=m.classifier_running_proto_exists_flag.clone(),
                        hist=metric.hist.clone(),
                        # how far every integer decision is from flipping: the confidence rule's distance to the bar and the
                        # top-2 gaps of the pseudo-label soft-max and of the up-sampled fused probabilities
                        pl_pmax=sm[:, 0].clone(), pl_gap=(sm[:, 0] - sm[:, 1]).clone(), pred_gap=(top2[:, 0] - top2[:, 1]).clone(),
                        pred_nan=torch.isnan(up).any(1), lowmass=lowmass))
    return res, metric


def gen_net_small(m):
    """Whole net + full TTA sequence on small inputs: 65x129 (B=1,2,3) and 129x257 (B=1)."""
    out = {}
    hooks, taps = [], {}
    want = ["conv1", "bn1", "maxpool", "layer1.0.conv1", "layer1.0.bn1", "layer1.0.conv2", "layer1.0", "layer2.0.conv1",
            "layer2.0.downsample.1", "layer2.3", "layer3.0.conv2", "layer3.22.bn3", "layer3.22", "layer4.0.conv2",
            "layer4.2.bn3"]
    for n, mod in m.net.named_modules():
        if n in want:
            hooks.append(mod.register_forward_hook(lambda _m, _i, o, n=n: taps.__setitem__(n, o.detach().clone())))
    for tag, (B, H, W) in dict(s1=(1, 65, 129), s2=(2, 65, 129), s3=(3, 65, 129), m1=(1, 129, 257)).items():
        frames = [synth.synth_frame(t, B, H, W) for t in range(3)]
        taps.clear()
        with torch.no_grad():
            m.net(torch.from_numpy(frames[0][0]), feat=True)
        if tag == "s1":
            for k, v in taps.items():
                out[f"{tag}_tap_{k}"] = np_(v if v.shape[1] < 1024 else v[:, ::4])
        if tag == "s2":      # B = 2: the half_first BN rule, channel-subsampled
            for k in ("bn1", "layer1.0.bn1", "layer2.0.downsample.1", "layer3.22.bn3"):
                out[f"{tag}_tap_{k}"] = np_(taps[k][:, ::4])
        res, metric = run_sequence(m, frames)
        out[f"{tag}_shape"] = np.array([B, H, W])
        for t, r in enumerate(res):
            out[f"{tag}_{t}_x_sha"] = np.array(sha(frames[t][0]))
            for k in ("logits", "pseudo", "probs", "run", "rflag", "hist"):
                out[f"{tag}_{t}_{k}"] = np_(r[k])
            out[f"{tag}_{t}_pred"] = np_(r["pred"]).astype(np.uint8)
            if t == 0 and tag == "s1":
                out[f"{tag}_{t}_feat"] = np_(r["feat"])
            out[f"{tag}_{t}_feat_sub"] = np_(r["feat"][:, ::16])
        out[f"{tag}_miou"] = np.array([float(metric.compute()), float(metric.compute_iou("16")),
                                       float(metric.compute_iou("13"))])
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(GOLD, "net_small.npz"), **out)


def gen_net_full(m):
    """Full 1024x512 frames, B=1, two frames (seed branch + EMA branch)."""
    out = {}
    frames = [synth.synth_frame(t, 1, 512, 1024) for t in range(2)]
    res, metric = run_sequence(m, frames)
    for t, r in enumerate(res):
        out[f"{t}_x_sha"] = np.array(sha(frames[t][0]))
        out[f"{t}_y_sha"] = np.array(sha(frames[t][1]))
        out[f"{t}_feat_sub"] = np_(r["feat"][:, ::16, ::4, ::4])
        out[f"{t}_feat_mean_ch"] = np_(r["feat"].mean([0, 2, 3]))
        for k in ("logits", "pseudo", "probs", "run", "rflag", "hist"):
            out[f"{t}_{k}"] = np_(r[k])
        out[f"{t}_pred"] = np_(r["pred"]).astype(np.uint8)
        out[f"{t}_pl_pmax"] = np_(r["pl_pmax"])
        out[f"{t}_pl_gap_bucket"] = gap_bucket(r["pl_gap"])
        out[f"{t}_pred_gap_bucket"] = gap_bucket(r["pred_gap"])
        out[f"{t}_lowmass"] = np_(r["lowmass"])
    out["miou"] = np.array([float(metric.compute()), float(metric.compute_iou("16")), float(metric.compute_iou("13"))])
    np.savez_compressed(os.path.join(GOLD, "net_full.npz"), **out)


MULTIRES = [(720, 1280), (512, 1024), (540, 960)]          # BASELINE configs[4]: BDD, Cityscapes-size and IDD-size frames


def gen_net_multires(m):
    """One stream whose frame size changes (BASELINE configs[4]) without resetting the prototypes: 3 frames, B = 1.
    logits / probs are stored as float16 (compared at the bf16 floor, ~1e-2), integers exactly."""
    out = {}
    frames = [synth.synth_frame(t, 1, H, W) for t, (H, W) in enumerate(MULTIRES)]
    res, metric = run_sequence(m, frames)
    for t, r in enumerate(res):
        out[f"mr_{t}_x_sha"] = np.array(sha(frames[t][0]))
        for k in ("logits", "probs"):
            out[f"mr_{t}_{k}"] = np_(r[k]).astype(np.float16)
        for k in ("pseudo", "run", "rflag", "hist"):
            out[f"mr_{t}_{k}"] = np_(r[k])
        out[f"mr_{t}_pred"] = np_(r["pred"]).astype(np.uint8)
    out["mr_shapes"] = np.array(MULTIRES)
    out["mr_miou"] = np.array([float(metric.compute()), float(metric.compute_iou("16")), float(metric.compute_iou("13"))])
    np.savez_compressed(os.path.join(GOLD, "net_multires.npz"), **out)


STREAM_FRAMES = 30


def gap_bucket(gap):
    """top-2 gap -> uint8 bucket floor(-log10(gap)) clipped to [0, 9]; NaN gaps (0/0 pixels) -> 255.  A decision whose
    bucket is b can flip under a perturbation of about 10^-b: the tests state their bit-exactness band in these units."""
    g = np_(gap).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        b = np.floor(-np.log10(g))
    b = np.where(np.isnan(g), 255, np.clip(np.where(np.isinf(b), 9, b), 0, 9))
    return b.astype(np.uint8)


def gen_net_stream(m):
    """BASELINE configs[1] as the bench runs it: one batch-1 stream of 1024x512 frames (frame t = synth_frame(t)), fresh
    prototypes, the reference's forward_and_adapt + Upsample + SegmentationMetric.  Stored per frame: the cumulative
    confusion matrix, the running-prototype flags and the prediction's class histogram; and the stream's mIoU-19/16/13 --
    the "mIoU delta vs ref" half of BASELINE.json's metric (bench.py reports it, tests/test_gpu_net.py bounds it)."""
    out = {}
    frames = [synth.synth_frame(t, 1, 512, 1024) for t in range(STREAM_FRAMES)]
    res, metric = run_sequence(m, frames)
    for t, r in enumerate(res):
        out[f"st_{t}_hist"] = np_(r["hist"]).astype(np.int64)
        out[f"st_{t}_rflag"] = np_(r["rflag"])
        out[f"st_{t}_pred_count"] = np.bincount(np_(r["pred"]).ravel(), minlength=19).astype(np.int64)
        out[f"st_{t}_nan_pixels"] = np.int64(torch.isnan(r["probs"]).any(1).sum())
        out[f"st_{t}_logits_sub"] = np_(r["logits"][:, :, ::4, ::4])
        out[f"st_{t}_pseudo"] = np_(r["pseudo"]).astype(np.uint8)
        out[f"st_{t}_pl_pmax"] = np_(r["pl_pmax"])
        out[f"st_{t}_pl_gap_bucket"] = gap_bucket(r["pl_gap"])
        out[f"st_{t}_pred"] = np_(r["pred"]).astype(np.uint8)
        out[f"st_{t}_pred_gap_bucket"] = gap_bucket(r["pred_gap"])
        out[f"st_{t}_run_norm"] = np_(r["run"].norm(dim=1))
        out[f"st_{t}_lowmass"] = np_(r["lowmass"])
    out["st_run_final"] = np_(res[-1]["run"])
    out["st_miou"] = np.array([float(metric.compute()), float(metric.compute_iou("16")), float(metric.compute_iou("13"))])
    out["st_frames"] = np.int64(STREAM_FRAMES)
    np.savez_compressed(os.path.join(GOLD, "net_stream.npz"), **out)


MS_SEEDS = (1, 2, 3, 4)
MS_FRAMES = 30


def gen_net_streams_ms(m):
    """Four more 30-frame 1024x512 batch-1 streams (frame seeds 1..4, same weights): the reference's mIoU-19/16/13 and
    final confusion matrix of each.  The bf16 product path is gated on the MEAN deviation over these streams and
    net_stream.npz (seed 0): one stream's deviation is a single draw of the bf16 rounding noise."""
    out = {}
    for sd in MS_SEEDS:
        frames = [synth.synth_frame(t, 1, 512, 1024, seed=sd) for t in range(MS_FRAMES)]
        res, metric = run_sequence(m, frames)
        out[f"ms{sd}_miou"] = np.array([float(metric.compute()), float(metric.compute_iou("16")), float(metric.compute_iou("13"))])
        out[f"ms{sd}_hist"] = np_(res[-1]["hist"]).astype(np.int64)
        out[f"ms{sd}_rflag"] = np_(res[-1]["rflag"])
        out[f"ms{sd}_x0_sha"] = np.array(sha(frames[0][0]))
        print("stream seed", sd, out[f"ms{sd}_miou"], flush=True)
    out["ms_seeds"] = np.array(MS_SEEDS)
    out["ms_frames"] = np.int64(MS_FRAMES)
    np.savez_compressed(os.path.join(GOLD, "net_streams_ms.npz"), **out)


LONG_FRAMES = 102


def gen_net_long(m):
    """BASELINE configs[4] as a LONG stream: 102 batch-1 frames cycling 1280x720 / 1024x512 / 960x540 without resetting
    the prototypes (ttda:1702-1705): ~100 EMA steps (ttda:1764).  Stored per frame: the running prototypes' norms and
    flags, the cumulative confusion matrix, NaN-pixel count and the prediction's class histogram; the final prototypes."""
    out = {}
    if hasattr(m, "classifier_running_proto"):
        del m.classifier_running_proto
    metric = T.SegmentationMetric(num_classes=19, ignore_index=255)
    for t in range(LONG_FRAMES):
        H, W = MULTIRES[t % 3]
        x, y = synth.synth_frame(t, 1, H, W)
        x, y = torch.from_numpy(x), torch.from_numpy(y)
        with torch.no_grad():
            probs, _ = m.forward_and_adapt((x, y))
            up = torch.nn.Upsample(size=y.shape[-2:], mode="bilinear")(probs)
            metric.update(up, y)
        run = m.classifier_running_proto
        out[f"lg_{t}_run_norm"] = np_(run.norm(dim=1))
        out[f"lg_{t}_run_sub"] = np_(run[:, ::64])
        out[f"lg_{t}_rflag"] = np_(m.classifier_running_proto_exists_flag)
        out[f"lg_{t}_hist"] = np_(metric.hist).astype(np.int64)
        out[f"lg_{t}_nan_pixels"] = np.int64(torch.isnan(probs).any(1).sum())
        out[f"lg_{t}_pred_count"] = np.bincount(np_(up.argmax(1)).ravel(), minlength=19).astype(np.int64)
        if t % 10 == 0:
            print("long stream frame", t, float(metric.compute()), flush=True)
    out["lg_run_final"] = np_(m.classifier_running_proto)
    out["lg_frames"] = np.int64(LONG_FRAMES)
    out["lg_shapes"] = np.array(MULTIRES)
    out["lg_miou"] = np.array([float(metric.compute()), float(metric.compute_iou("16")), float(metric.compute_iou("13"))])
    np.savez_compressed(os.path.join(GOLD, "net_long.npz"), **out)


def gen_net_hires(m):
    """BASELINE configs[3]'s resolution: 2048x1024 frames, B = 1, two frames (seed branch + EMA branch) -> the 129x257 feature
    grid of adaptseg.py:177-188.  logits / probs sub-sampled, integer decisions and their flip margins in full (B = 4 at this
    size needs > 40 GB of cal_pred_of_proto temporaries, SURVEY 8c: the batch rule itself is pinned at small sizes)."""
    out = {}
    frames = [synth.synth_frame(t, 1, 1024, 2048) for t in range(2)]
    res, metric = run_sequence(m, frames)
    for t, r in enumerate(res):
        out[f"hr_{t}_x_sha"] = np.array(sha(frames[t][0]))
        out[f"hr_{t}_y_sha"] = np.array(sha(frames[t][1]))
        out[f"hr_{t}_feat_sub"] = np_(r["feat"][:, ::32, ::4, ::4])
        out[f"hr_{t}_feat_mean_ch"] = np_(r["feat"].mean([0, 2, 3]))
        out[f"hr_{t}_logits_sub"] = np_(r["logits"][:, :, ::2, ::2])
        out[f"hr_{t}_probs_sub"] = np_(r["probs"][:, :, ::2, ::2])
        for k in ("run", "rflag", "hist"):
            out[f"hr_{t}_{k}"] = np_(r[k])
        out[f"hr_{t}_pseudo"] = np_(r["pseudo"]).astype(np.uint8)
        out[f"hr_{t}_pred"] = np_(r["pred"]).astype(np.uint8)
        out[f"hr_{t}_pl_pmax"] = np_(r["pl_pmax"])
        out[f"hr_{t}_pl_gap_bucket"] = gap_bucket(r["pl_gap"])
        out[f"hr_{t}_pred_gap_bucket"] = gap_bucket(r["pred_gap"])
        out[f"hr_{t}_lowmass"] = np.packbits(np_(r["lowmass"]))
        out[f"hr_{t}_pred_nan"] = np.packbits(np_(r["pred_nan"]))
        print("hires frame", t, float(metric.compute()), flush=True)
    out["hr_shape"] = np.array([1, 1024, 2048])
    out["hr_miou"] = np.array([float(metric.compute()), float(metric.compute_iou("16")), float(metric.compute_iou("13"))])
    np.savez_compressed(os.path.join(GOLD, "net_hires.npz"), **out)


EVAL_FRAMES = 500
EVAL_EVERY = 50


def gen_net_eval500(m):
    """One evaluation-length batch-1 stream: 500 frames of 1024x512 (GTA5 -> Cityscapes val has 500 images, README.md:241-243;
    frame t = synth_frame(t): the first 30 frames are net_stream.npz's), fresh prototypes, forward_and_adapt + Upsample +
    SegmentationMetric (ttda:1605-1660).  Stored: the cumulative confusion matrix, mIoU-19/16/13, the running prototypes' norms and
    flags every 50 frames, and the final prototypes."""
    out = {}
    if hasattr(m, "classifier_running_proto"):
        del m.classifier_running_proto
    metric = T.SegmentationMetric(num_classes=19, ignore_index=255)
    nan_px, t_start = 0, 0
    ckpt = os.environ.get("GOLDEN_EVAL_CKPT", "/tmp/gold/eval500_state.npz")     # resumable: an hour of CPU
    if os.path.exists(ckpt):
        with np.load(ckpt, allow_pickle=False) as z:
            out = {k: z[k] for k in z.files if k.startswith("ev_")}
            t_start, nan_px = int(z["t_next"]), int(z["nan_px"])
            metric.hist.copy_(torch.from_numpy(z["hist"]))
            m.classifier_running_proto = torch.from_numpy(z["run"]).clone()
            m.classifier_running_proto_exists_flag = torch.from_numpy(z["rflag"]).clone()
        print("resuming the eval stream at frame", t_start, flush=True)
    for t in range(t_start, EVAL_FRAMES):
        x, y = synth.synth_frame(t, 1, 512, 1024)
        x, y = torch.from_numpy(x), torch.from_numpy(y)
        with torch.no_grad():
            probs, _ = m.forward_and_adapt((x, y))
            up = torch.nn.Upsample(size=y.shape[-2:], mode="bilinear")(probs)
            metric.update(up, y)
        nan_px += int(torch.isnan(probs).any(1).sum())
        if (t + 1) % EVAL_EVERY == 0:
            out[f"ev_{t + 1}_hist"] = np_(metric.hist).astype(np.int64)
            out[f"ev_{t + 1}_miou"] = np.array([float(metric.compute()), float(metric.compute_iou("16")), float(metric.compute_iou("13"))])
            out[f"ev_{t + 1}_run_norm"] = np_(m.classifier_running_proto.norm(dim=1))
            out[f"ev_{t + 1}_rflag"] = np_(m.classifier_running_proto_exists_flag)
            out[f"ev_{t + 1}_nan_pixels"] = np.int64(nan_px)
            print("eval stream frame", t + 1, out[f"ev_{t + 1}_miou"], flush=True)
        if (t + 1) % 10 == 0:
            os.makedirs(os.path.dirname(ckpt), exist_ok=True)
            np.savez(ckpt + ".tmp.npz", t_next=np.int64(t + 1), nan_px=np.int64(nan_px), hist=np_(metric.hist),
                     run=np_(m.classifier_running_proto), rflag=np_(m.classifier_running_proto_exists_flag), **out)
            os.replace(ckpt + ".tmp.npz", ckpt)
    out["ev_run_final"] = np_(m.classifier_running_proto)
    out["ev_frames"] = np.int64(EVAL_FRAMES)
    out["ev_every"] = np.int64(EVAL_EVERY)
    np.savez_compressed(os.path.join(GOLD, "net_eval500.npz"), **out)


def gen_net_b8(m):
    """BASELINE configs[2]'s batch: ONE batch of 8 frames (half_first = sample 0 vs samples 1..7, ttda:313-320), the
    reference run with batch_size 8.  b8s: 65x129 inputs, 3 frames (everything stored); b8f: 1024x512, 2 frames (seed +
    EMA branch; logits sub-sampled, integers in full)."""
    out = {}
    hooks, taps = [], {}
    for n, mod in m.net.named_modules():
        if n in ("bn1", "layer1.0.bn1", "layer2.0.downsample.1", "layer3.22.bn3", "layer4.2.bn3"):
            hooks.append(mod.register_forward_hook(lambda _m, _i, o, n=n: taps.__setitem__(n, o.detach().clone())))
    for tag, (H, W, nfr) in dict(b8s=(65, 129, 3), b8f=(512, 1024, 2)).items():
        frames = [synth.synth_frame(200 + t, 8, H, W) for t in range(nfr)]
        taps.clear()
        if tag == "b8s":
            with torch.no_grad():
                m.net(torch.from_numpy(frames[0][0]), feat=True)
            for k, v in taps.items():
                out[f"{tag}_tap_{k}"] = np_(v[:, ::16])
        res, metric = run_sequence(m, frames)
        out[f"{tag}_shape"] = np.array([8, H, W])
        for t, r in enumerate(res):
            out[f"{tag}_{t}_x_sha"] = np.array(sha(frames[t][0]))
            small = tag == "b8s"
            out[f"{tag}_{t}_logits"] = np_(r["logits"] if small else r["logits"][:, :, ::4, ::4])
            out[f"{tag}_{t}_probs"] = np_(r["probs"] if small else r["probs"][:, :, ::4, ::4])
            out[f"{tag}_{t}_feat_sub"] = np_(r["feat"][:, ::64] if small else r["feat"][:, ::64, ::4, ::4])
            for k in ("pseudo", "run", "rflag", "hist"):
                out[f"{tag}_{t}_{k}"] = np_(r[k])
            out[f"{tag}_{t}_pred"] = np_(r["pred"]).astype(np.uint8)
            out[f"{tag}_{t}_pl_pmax"] = np_(r["pl_pmax"])
            out[f"{tag}_{t}_pl_gap_bucket"] = gap_bucket(r["pl_gap"])
            out[f"{tag}_{t}_pred_gap_bucket"] = gap_bucket(r["pred_gap"])
            out[f"{tag}_{t}_lowmass"] = np_(r["lowmass"])
        out[f"{tag}_miou"] = np.array([float(metric.compute()), float(metric.compute_iou("16")), float(metric.compute_iou("13"))])
        print(tag, out[f"{tag}_miou"], flush=True)
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(GOLD, "net_b8.npz"), **out)


def gen_loader():
    """restore(): file named *baseline* with layer5.* head keys -> model layer6.* (ttda:1000-1016)."""
    p = synth.synth_params(0)
    saved = {}
    for k, v in p.items():
        if k.startswith("layer5."):
            continue
        if k.startswith("layer6."):
            saved["layer5." + k.split(".", 1)[1]] = torch.from_numpy(v)     # as GTA5_baseline.pth stores the head
        else:
            saved[k] = torch.from_numpy(v)
    path = "/tmp/synth_GTA5_baseline.pth"
    torch.save(saved, path)
    net = T.DeepLabv2AdaptSeg(num_classes=19, multi_level=True, restore_from=path)
    sd = net.state_dict()
    rec = dict(file_keys=len(saved), model_keys=len(sd), checks={})
    for k in ["conv1.weight", "bn1.running_var", "layer3.5.conv2.weight", "layer6.conv2d_list.1.weight",
              "layer6.conv2d_list.3.bias", "layer4.0.downsample.1.running_mean"]:
        rec["checks"][k] = sha(np_(sd[k]))
        assert rec["checks"][k] == sha(p[k]), k
    # model layer5 keeps its own (random) init: not equal to the synthetic tensors
    rec["layer5_untouched"] = bool(sha(np_(sd["layer5.conv2d_list.0.weight"])) != sha(p["layer5.conv2d_list.0.weight"]))
    bad = dict(saved)
    bad["mystery.weight"] = torch.zeros(1)
    torch.save(bad, "/tmp/synth_bad_baseline.pth")
    try:
        T.DeepLabv2AdaptSeg(num_classes=19, restore_from="/tmp/synth_bad_baseline.pth")
        rec["bad_key_error"] = None
    except ValueError as e:
        rec["bad_key_error"] = str(e)
    torch.save(saved, "/tmp/synth_unknown_name.pth")
    try:
        T.DeepLabv2AdaptSeg(num_classes=19, restore_from="/tmp/synth_unknown_name.pth")
        rec["unknown_name_error"] = None
    except ValueError as e:
        rec["unknown_name_error"] = str(e)
    for f in ("/tmp/synth_GTA5_baseline.pth", "/tmp/synth_bad_baseline.pth", "/tmp/synth_unknown_name.pth"):
        os.remove(f)
    rec["branches"] = gen_loader_branches(p)
    with open(os.path.join(GOLD, "loader.json"), "w") as f:
        json.dump(rec, f, indent=1)


sys.path.append(os.path.join(ROOT, "tests"))
from loader_cases import LOADER_KEYS, loader_branch_files  # noqa: E402  (reference-free: shared with tests/test_cpu_host.py)


def gen_loader_branches(p):
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        paths = loader_branch_files(p, tmp)
        for case, path in paths.items():
            try:
                net = T.DeepLabv2AdaptSeg(num_classes=19, multi_level=True, restore_from=path)
            except Exception as e:          # noqa: BLE001 -- the error class IS the recorded behaviour
                out[case] = {"error": type(e).__name__}
                continue
            sd = net.state_dict()
            out[case] = {"checks": {k: sha(np_(sd[k])) for k in LOADER_KEYS},
                         "layer5_from_file": bool(sha(np_(sd["layer5.conv2d_list.0.weight"])) == sha(p["layer5.conv2d_list.0.weight"]))}
    return out


def gen_fn_input():
    """Input pipeline: BaseDataset.preprocess and CityscapesDataSet.map_labels called on the reference's own classes
    (instances made with object.__new__: their __init__ reads dataset files), info.json's label2train table."""
    import importlib
    import types
    base = importlib.import_module("src.datamodules.seg.base_dataset")
    city = importlib.import_module("src.datamodules.seg.cityscapes")
    rng = np.random.default_rng(5)
    out = {}
    mean = np.array((104.00698793, 116.66876762, 122.67891434))          # ttda_datamodule.py:53
    ds = object.__new__(base.BaseDataset)
    ds.mean = mean
    for tag, shape in (("a", (9, 13, 3)), ("b", (32, 64, 3))):
        img = rng.integers(0, 256, shape).astype(np.float32)              # what _load_img returns (base_dataset.py:67)
        out[f"img_{tag}"] = img.astype(np.uint8)
        out[f"pre_{tag}"] = base.BaseDataset.preprocess(ds, img.copy()).copy()
        out[f"bdd_{tag}"] = torch.from_numpy(np.array(img.astype(np.uint8)) - mean).float().permute(2, 0, 1).numpy()  # bdd100k.py:290,296
    info = json.load(open("/root/reference/src/datamodules/seg/cityscapes_list/info.json"))
    cs = object.__new__(city.CityscapesDataSet)
    mapping = np.array(info["label2train"], dtype=np.int64)                # cityscapes.py:38-41 (np.int is gone in numpy 2)
    cs.map_vector = np.zeros((mapping.shape[0],), dtype=np.int64)
    for source_label, target_label in mapping:
        cs.map_vector[source_label] = target_label
    ids = rng.integers(-1, 34, (17, 23)).astype(np.float32)
    out["label2train"] = mapping
    out["ids"] = ids
    out["train_ids"] = city.CityscapesDataSet.map_labels(cs, ids)
    np.savez_compressed(os.path.join(GOLD, "fn_input.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
    which = sys.argv[1:] or ["bn", "sam", "tail", "small", "full", "multires", "stream", "loader", "input"]
    m = make_diga() if set(which) & {"sam", "tail", "small", "full", "multires", "stream", "streams_ms", "long", "b8", "hires", "eval500"} else None
    if "bn" in which:
        gen_fn_bn()
    if "sam" in which:
        gen_fn_sam(m)
    if "tail" in which:
        gen_fn_tail(m)
    if "small" in which:
        gen_net_small(m)
    if "full" in which:
        gen_net_full(m)
    if "multires" in which:
        gen_net_multires(m)
    if "stream" in which:
        gen_net_stream(m)
    if "b8" in which:
        gen_net_b8(m)
    if "streams_ms" in which:
        gen_net_streams_ms(m)
    if "long" in which:
        gen_net_long(m)
    if "hires" in which:
        gen_net_hires(m)
    if "eval500" in which:
        gen_net_eval500(m)
    if "loader" in which:
        gen_loader()
    if "input" in which:
        gen_fn_input()
    print("done", which)
