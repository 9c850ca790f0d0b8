"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of DIGA's backward-free TTA forward path.

This file restates, in plain fp32 torch-CPU / numpy operations, the algorithm of the reference
(Waybaba/DIGA) for the hot path of SURVEY.md section 8.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``diga_amd``) never imports it and fails loudly when the HIP library is missing.

Pinning: ``oracle/gen_golden.py`` imports the reference's own source in the build container and
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function below against
those vectors (<=1e-6 relative for floats, bit-exact for integers).  The reference has no tests or
fixtures of its own, so with respect to the authors' torch-1.12/cuDNN numerical
environment parity is unpinned; with respect to the reference source executed by this container's
torch-2.10 CPU it is pinned by those fixtures.

Reference citations: "ttda" = /root/reference/src/models/ttda_module.py,
"adaptseg" = /root/reference/src/models/components/adaptseg.py.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

NUM_CLASSES = 19
LAYERS = (3, 4, 23, 3)           # adaptseg:242 / ttda:942
PLANES = (64, 128, 256, 512)     # adaptseg:140-143
STRIDES = (1, 2, 1, 1)
DILATIONS = (1, 1, 2, 4)


# --------------------------------------------------------------------------------------
# DAM batch norm  (ttda:299-345, eval branch; installed by DIGA._replace_bn ttda:1814-1833)
# --------------------------------------------------------------------------------------
def dam_current_stats(x: torch.Tensor):
    """Current-batch per-channel statistics with the reference's ``half_first`` rule (ttda:313-320).

    B == 1: plain mean / biased var over (N,H,W).  B > 1: the *average of two group statistics*,
    group A = sample 0, group B = samples 1..B-1 (NOT the global-batch statistic).
    """
    if x.shape[0] > 1:
        a, b = x[:1], x[1:]
        mean = (a.mean([0, 2, 3]) + b.mean([0, 2, 3])) / 2
        var = (a.var([0, 2, 3], unbiased=False) + b.var([0, 2, 3], unbiased=False)) / 2
    else:
        mean = x.mean([0, 2, 3])
        var = x.var([0, 2, 3], unbiased=False)
    return mean, var


def dam_bn(x, running_mean, running_var, weight, bias, lam, eps=1e-5):
    """y = (x - mean)/sqrt(var+eps)*weight + bias with mean/var = lam*source + (1-lam)*current
    (ttda:335-343).  No state is written in eval mode."""
    mean_cur, var_cur = dam_current_stats(x)
    # lambda_ is an fp32 0-dim tensor in the reference (ttda:369,1812), so (1 - lambda_) is the *fp32*
    # subtraction 1 - fp32(0.8) = 0.19999999, not fp32(0.2): one ulp that compounds over 104 layers.
    lam = torch.as_tensor(lam, dtype=torch.float32)
    mean = lam * running_mean + (1 - lam) * mean_cur
    var = lam * running_var + (1 - lam) * var_cur
    y = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
    return y * weight[None, :, None, None] + bias[None, :, None, None]


# --------------------------------------------------------------------------------------
# AdaptSegNet DeepLabV2-ResNet101 trunk + 2-branch head  (adaptseg:58-194)
# --------------------------------------------------------------------------------------
BN_PRE_HOOK = None   # test/calibration aid: callable(prefix, x, params) run before each BN


def _bn(p, prefix, x, lam):
    if BN_PRE_HOOK is not None:
        BN_PRE_HOOK(prefix, x, p)
    return dam_bn(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                  p[prefix + ".weight"], p[prefix + ".bias"], lam)


def bottleneck(p, prefix, x, stride, dilation, has_ds, lam, taps=None):
    """adaptseg:82-102.  Stride sits on the first 1x1 (adaptseg:63); 3x3 has pad = dilation (adaptseg:68-70)."""
    out = F.conv2d(x, p[prefix + ".conv1.weight"], stride=stride)
    if taps is not None:
        taps[prefix + ".conv1"] = out
    out = torch.relu(_bn(p, prefix + ".bn1", out, lam))
    out = F.conv2d(out, p[prefix + ".conv2.weight"], padding=dilation, dilation=dilation)
    if taps is not None:
        taps[prefix + ".conv2"] = out
    out = torch.relu(_bn(p, prefix + ".bn2", out, lam))
    out = F.conv2d(out, p[prefix + ".conv3.weight"])
    if taps is not None:
        taps[prefix + ".conv3"] = out
    out = _bn(p, prefix + ".bn3", out, lam)
    if has_ds:
        res = F.conv2d(x, p[prefix + ".downsample.0.weight"], stride=stride)
        if taps is not None:
            taps[prefix + ".downsample.0"] = res
        res = _bn(p, prefix + ".downsample.1", res, lam)
    else:
        res = x
    return torch.relu(out + res)


def net_forward_feat(p: dict, x: torch.Tensor, lam: float = 0.8, taps: dict | None = None):
    """``net(x, feat=True)`` (adaptseg:177-188): returns (layer4 features, logits).

    Head (adaptseg:116-122): the early ``return`` inside the loop means logits =
    conv_dil6(x) + conv_dil12(x) only; dilations 18/24 are never executed.
    ``taps`` (optional dict) receives raw conv outputs keyed by the conv's state-dict prefix.
    """
    x = F.conv2d(x, p["conv1.weight"], stride=2, padding=3)
    if taps is not None:
        taps["conv1"] = x
    x = torch.relu(_bn(p, "bn1", x, lam))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1, ceil_mode=True)  # adaptseg:139
    if taps is not None:
        taps["maxpool"] = x
    for li, (nb, st, dil) in enumerate(zip(LAYERS, STRIDES, DILATIONS), start=1):
        for b in range(nb):
            x = bottleneck(p, f"layer{li}.{b}", x, st if b == 0 else 1, dil, b == 0, lam, taps)
            if taps is not None:
                taps[f"layer{li}.{b}"] = x
    feat = x
    logits = head_logits(p, feat)
    return feat, logits


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> nearest-even bf16 -> fp32 (the storage rounding of the GPU pipeline)."""
    return t.to(torch.bfloat16).to(torch.float32)


def dam_affine(x_raw, running_mean, running_var, weight, bias, lam, eps=1e-5):
    """Per-channel (scale, shift) with y = x*scale + shift == dam_bn(x) (the form the GPU applies)."""
    mean_cur, var_cur = dam_current_stats(x_raw)
    lam = torch.as_tensor(lam, dtype=torch.float32)
    mean = lam * running_mean + (1 - lam) * mean_cur
    var = lam * running_var + (1 - lam) * var_cur
    sc = weight / torch.sqrt(var + eps)
    return sc, bias - mean * sc


def net_forward_feat_bf16(p: dict, x: torch.Tensor, lam: float = 0.8, taps: dict | None = None, keep_fp32=()):
    """The same network with the GPU pipeline's *storage* precision emulated: conv weights (except the
    fp32 stem) and every stored activation are rounded to bf16, all arithmetic stays fp32, and the DAM
    statistics are taken from the un-rounded fp32 conv outputs (as the conv epilogue does).  This is
    the oracle for end-to-end parity of the bf16 path: it differs from ``net_forward_feat`` only by
    rounding, and the GPU differs from it only by fp32 summation order.

    keep_fp32: which classes of stored tensors are NOT rounded (the study of what sets the bf16 floor,
    tests/golden/make_bf16_switch_study.py): "residual" = the block input / output stream, "raw" = raw
    conv outputs, "act" = the normalised activations between the convs of a block (and the pooled stem
    output), "head_in" = only the layer4 output that feeds the head and SAM, "weights" = conv weights."""
    keep = set(keep_fp32)
    assert keep <= {"residual", "raw", "act", "head_in", "weights"}, keep
    ident = lambda t: t      # noqa: E731
    r_res = ident if "residual" in keep else bf16_round
    r_raw = ident if "raw" in keep else bf16_round
    r_act = ident if "act" in keep else bf16_round
    r_w = ident if "weights" in keep else bf16_round

    def aff(prefix, raw):
        return dam_affine(raw, p[prefix + ".running_mean"], p[prefix + ".running_var"], p[prefix + ".weight"],
                          p[prefix + ".bias"], lam)

    def app(raw_q, sc, sh):
        return raw_q * sc[None, :, None, None] + sh[None, :, None, None]

    raw = F.conv2d(x, p["conv1.weight"], stride=2, padding=3)
    sc, sh = aff("bn1", raw)
    if taps is not None:
        taps["conv1"] = r_raw(raw)
    cur = r_res(F.max_pool2d(torch.relu(app(r_raw(raw), sc, sh)), 3, 2, 1, ceil_mode=True))
    if taps is not None:
        taps["maxpool"] = cur
    nblocks = sum(LAYERS)
    done = 0
    for li, (nb, st, dil) in enumerate(zip(LAYERS, STRIDES, DILATIONS), start=1):
        for b in range(nb):
            pre = f"layer{li}.{b}"
            s = st if b == 0 else 1
            raw1 = F.conv2d(cur, r_w(p[pre + ".conv1.weight"]), stride=s)
            sc, sh = aff(pre + ".bn1", raw1)
            a1 = r_act(torch.relu(app(r_raw(raw1), sc, sh)))
            raw2 = F.conv2d(a1, r_w(p[pre + ".conv2.weight"]), padding=dil, dilation=dil)
            sc, sh = aff(pre + ".bn2", raw2)
            a2 = r_act(torch.relu(app(r_raw(raw2), sc, sh)))
            raw3 = F.conv2d(a2, r_w(p[pre + ".conv3.weight"]))
            sc3, sh3 = aff(pre + ".bn3", raw3)
            if b == 0:
                rawd = F.conv2d(cur, r_w(p[pre + ".downsample.0.weight"]), stride=s)
                scd, shd = aff(pre + ".downsample.1", rawd)
                res = app(r_raw(rawd), scd, shd)
            else:
                res = cur
            done += 1
            last = done == nblocks
            out = torch.relu(app(r_raw(raw3), sc3, sh3) + res)
            cur = out if (last and "head_in" in keep) else r_res(out)
            if taps is not None:
                taps[pre] = cur
    hp = {k: (r_w(v) if k.endswith("weight") else v) for k, v in p.items() if k.startswith("layer6.")}
    return cur, head_logits(hp, cur)


def head_logits(p, feat):
    return (F.conv2d(feat, p["layer6.conv2d_list.0.weight"], p["layer6.conv2d_list.0.bias"], padding=6, dilation=6)
            + F.conv2d(feat, p["layer6.conv2d_list.1.weight"], p["layer6.conv2d_list.1.bias"], padding=12, dilation=12))


# --------------------------------------------------------------------------------------
# SAM  (ttda:1662-1805)
# --------------------------------------------------------------------------------------
def nearest_down_labels(y: torch.Tensor, size) -> torch.Tensor:
    """ttda:1695 -- F.interpolate(y.float(), mode='nearest') then .long(): src = floor(dst*in/out)."""
    return F.interpolate(y.unsqueeze(1).float(), size=tuple(size), mode="nearest").squeeze(1).long()


def pseudo_labels(logits: torch.Tensor, threshold: float) -> torch.Tensor:
    """ttda:1662-1676 -- argmax of softmax, 255 where max prob < threshold (strict)."""
    prob = logits.softmax(dim=1)
    lab = prob.argmax(dim=1)
    lab[prob.max(dim=1)[0] < threshold] = 255
    return lab


def class_prototypes(feature: torch.Tensor, labels: torch.Tensor, num_classes: int = NUM_CLASSES):
    """ttda:1788-1805 -- per-class mean of the feature vectors of pixels with that label.
    Returns (proto [C,CH], flag [C] in {0,1})."""
    ch = feature.shape[1]
    rows = feature.permute(0, 2, 3, 1).reshape(-1, ch)   # same row order as permute/flatten at ttda:1804
    lab = labels.reshape(-1)
    proto = torch.zeros(num_classes, ch)
    flag = torch.zeros(num_classes)
    for c in range(num_classes):
        m = lab == c
        if int(m.sum()) == 0:
            continue
        flag[c] = 1
        proto[c] = rows[m].mean(0)
    return proto, flag


def proto_probs(feature, proto, flag, tau: float = 1.0):
    """ttda:1712-1734 -- w[b,c,i,j] = exp(-||proto[c]-feature[b,:,i,j]||_2 / tau)*flag[c], normalised
    over c.  0/0 -> NaN when no class exists or every class underflows (reproduced, not fixed)."""
    b, ch, h, w = feature.shape
    c = proto.shape[0]
    dist = torch.empty(b, c, h, w)
    for k in range(c):
        dist[:, k] = torch.norm(proto[k][None, :, None, None] - feature, dim=1)
    wgt = torch.exp(-dist / tau) * flag[None, :, None, None]
    return wgt / wgt.sum(dim=1, keepdim=True)


def update_running_proto(feature, y_small, run_proto, run_flag, rho, num_classes: int = NUM_CLASSES):
    """ttda:1736-1765 -- EMA of class prototypes computed from the (down-sampled) *ground-truth*
    labels; first-seen classes are copied; rho == 0 freezes."""
    run_proto, run_flag = run_proto.clone(), run_flag.clone()
    new, has = class_prototypes(feature, y_small, num_classes)
    for c in range(num_classes):
        if not has[c] or rho == 0:
            continue
        if run_flag[c] == 0:
            run_proto[c] = new[c]
            run_flag[c] = 1
        else:
            run_proto[c] = (1 - rho) * run_proto[c] + rho * new[c]
    return run_proto, run_flag


class SamState:
    """The only cross-frame state (ttda:1702-1705): created lazily on the first frame."""

    def __init__(self):
        self.run_proto = None
        self.run_flag = None


def multi_proto_label(state: SamState, feature, logits, y, cfg, info: dict | None = None):
    """ttda:1678-1710."""
    y_small = nearest_down_labels(y, logits.shape[2:])
    pl = pseudo_labels(logits, cfg["confidence_threshold"])
    proto, flag = class_prototypes(feature, pl)
    if state.run_proto is None:
        state.run_proto, state.run_flag = proto, flag
    else:
        state.run_proto, state.run_flag = update_running_proto(
            feature, y_small, state.run_proto, state.run_flag, cfg["proto_rho"])
    inst = proto_probs(feature, proto, flag)
    mean = proto_probs(feature, state.run_proto, state.run_flag)
    if info is not None:
        info.update(y_small=y_small, pseudo=pl, inst_proto=proto, inst_flag=flag,
                    inst_pred=inst, mean_pred=mean)
    return cfg["proto_lambda"] * mean + (1 - cfg["proto_lambda"]) * inst


def fuse(multi_pred, logits, fusion_lambda):
    """ttda:1644."""
    return multi_pred * fusion_lambda + logits.softmax(1) * (1 - fusion_lambda)


def forward_and_adapt(p, state: SamState, x, y, cfg, info: dict | None = None, bf16: bool = False, keep_fp32=()):
    """ttda:1623-1645 (without the .cpu()).  bf16=True uses the storage-precision emulation of the GPU path."""
    if bf16:
        feat, logits = net_forward_feat_bf16(p, x, cfg["bn_lambda"], keep_fp32=keep_fp32)
    else:
        feat, logits = net_forward_feat(p, x, cfg["bn_lambda"])
    if info is not None:
        info.update(feature=feat, logits=logits)
    return fuse(multi_proto_label(state, feat, logits, y, cfg, info), logits, cfg["fusion_lambda"])


# --------------------------------------------------------------------------------------
# Tail: upsample + metric  (ttda:1610, 147-211)
# --------------------------------------------------------------------------------------
def upsample_probs(probs, size):
    """ttda:1610 -- nn.Upsample(size, mode='bilinear') == align_corners=False."""
    return F.interpolate(probs, size=tuple(size), mode="bilinear", align_corners=False)


def fast_hist(pred: torch.Tensor, target: torch.Tensor, num_classes: int = NUM_CLASSES):
    """ttda:168-173 -- rows = target, cols = prediction; targets outside [0,C) ignored."""
    pred, target = pred.reshape(-1), target.reshape(-1)
    k = (target >= 0) & (target < num_classes)
    return torch.bincount(num_classes * target[k].int() + pred[k], minlength=num_classes ** 2).reshape(
        num_classes, num_classes)


def metric_update(hist: torch.Tensor, probs_up: torch.Tensor, target: torch.Tensor):
    """ttda:158-166 -- hist += fast_hist(argmax) * batch_size."""
    pred = probs_up.argmax(dim=1)
    return hist + fast_hist(pred, target) * target.shape[0]


IDX16 = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 15, 17]   # ttda:203 (class 18 omitted as written)
IDX13 = [0, 1, 2, 6, 7, 8, 10, 11, 12, 13, 15, 17]            # ttda:207


def miou(hist: torch.Tensor, kind: str = "default"):
    """ttda:184-211."""
    hist = hist.float()
    d = torch.diag(hist)
    ious = d / (hist.sum(1) + hist.sum(0) - d)
    if kind == "16":
        ious = ious[IDX16]
    elif kind == "13":
        ious = ious[IDX13]
    elif kind != "default":
        raise NotImplementedError(kind)
    return torch.nanmean(ious) * 100.0


# --------------------------------------------------------------------------------------
# Checkpoint key remap  (ttda:1000-1016)
# --------------------------------------------------------------------------------------
def remap_baseline_keys(saved: dict, model_keys) -> dict:
    """``load_baseline_pretrained``: file ``layer5.*`` -> model ``layer6.*``; other ``layer*`` keys
    verbatim; remaining keys copied if the model has them, else ValueError."""
    out = {}
    for k, v in saved.items():
        head = k.split(".")[0]
        if head == "layer5":
            out["layer6." + k.split(".", 1)[1]] = v
        elif "layer" in head:
            out[k] = v
        elif k in model_keys:
            out[k] = v
        else:
            raise ValueError("Can not find pretrained model type.")
    return out


def to_torch(params: dict) -> dict:
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in params.items()}


# ------------------------------------------------------------------------------------------------
# Input pipeline.  numpy, as the reference's datasets are.
# ------------------------------------------------------------------------------------------------
def preprocess_image(image_hwc_rgb, mean, swap_rb=True):
    """BaseDataset.preprocess (src/datamodules/seg/base_dataset.py:45-52) for swap_rb, the BDD100K / Mapillary form
    `np.array(img) - self.mean` then `.float()` (src/datamodules/seg/bdd100k.py:290,296) otherwise.
    float32 image, float64 mean: numpy evaluates `image -= mean` in float64 and casts the result back to float32."""
    import numpy as np
    image = np.asarray(image_hwc_rgb)
    mean = np.asarray(mean, dtype=np.float64)
    if not mean.any():
        raise NotImplementedError("per-image-mean branch (base_dataset.py:49-51)")
    if swap_rb:
        image = image[:, :, ::-1]
    out = (image.astype(np.float64) - mean).astype(np.float32)
    return np.ascontiguousarray(out.transpose((2, 0, 1)))


def map_labels(label_ids, label2train):
    """CityscapesDataSet.__init__ / map_labels (src/datamodules/seg/cityscapes.py:38-41, 50-51)."""
    import numpy as np
    mapping = np.array(label2train, dtype=np.int64)
    map_vector = np.zeros((mapping.shape[0],), dtype=np.int64)
    for source_label, target_label in mapping:
        map_vector[source_label] = target_label
    return map_vector[np.asarray(label_ids).astype(np.int64, copy=False)]


# ------------------------------------------------------------------------------------------------
# Optional component: connected components of the per-class masks + instance-masked pooling.
# The reference has NO such code: with respect to Waybaba/DIGA this is "parity unpinned".  The
# specification is the build's own (BASELINE.json north_star (2)): per-class binary masks, 4- or 8-connectivity,
# components numbered from 1 inside each class in raster order of their first pixel, background 0.  Pinned
# against scipy.ndimage.label (scipy 1.15.3) outputs committed in tests/golden/fn_ccl.npz (oracle/gen_golden_ccl.py).
# ------------------------------------------------------------------------------------------------
def ccl_label(labels, num_classes: int = NUM_CLASSES, connectivity: int = 8):
    """labels int [B,h,w] -> (inst int32 [B,h,w], ncomp int32 [B,C]).  Run-based two-pass union-find: the runs of a
    row are united with the overlapping same-class runs of the row above; the root of a set is its lowest-numbered
    run (runs are numbered in raster order), so roots appear in raster order of the components' first pixels."""
    assert connectivity in (4, 8)
    lab_all = np.asarray(labels)
    B, h, w = lab_all.shape
    inst = np.zeros((B, h, w), dtype=np.int32)
    ncomp = np.zeros((B, num_classes), dtype=np.int32)
    reach = 1 if connectivity == 8 else 0
    for b in range(B):
        lab = lab_all[b]
        parent, runs = [], []          # runs[i] = (y, x0, x1, class)

        def find(i):
            while parent[i] != i:
                parent[i] = parent[parent[i]]
                i = parent[i]
            return i

        prev = []
        for y in range(h):
            row = lab[y]
            cut = np.flatnonzero(row[1:] != row[:-1]) + 1
            starts = np.concatenate(([0], cut))
            ends = np.concatenate((cut, [w]))
            cur = []
            k = 0                       # first run of the previous row that can still touch
            for x0, x1 in zip(starts.tolist(), ends.tolist()):
                c = int(row[x0])
                if not 0 <= c < num_classes:
                    continue
                i = len(parent)
                parent.append(i)
                runs.append((y, x0, x1, c))
                cur.append(i)
                while k < len(prev) and runs[prev[k]][2] + reach <= x0:      # ends left of this run: never touches later ones
                    k += 1
                j = k
                while j < len(prev) and runs[prev[j]][1] < x1 + reach:
                    if runs[prev[j]][3] == c and runs[prev[j]][2] + reach > x0:
                        a, r = find(prev[j]), find(i)
                        if a != r:
                            parent[max(a, r)] = min(a, r)
                    j += 1
            prev = cur
        number = {}
        for i, (y, x0, x1, c) in enumerate(runs):
            r = find(i)
            if r == i:
                ncomp[b, c] += 1
                number[i] = ncomp[b, c]
            inst[b, y, x0:x1] = number[r]
    return inst, ncomp


def instance_prototypes(feature, labels, inst, ncomp, num_classes: int = NUM_CLASSES):
    """Instance-masked pooling: the mean feature of every component, rows ordered (sample, class, component number).
    feature [B,CH,h,w] float; -> (protos float64 [N,CH], counts int64 [N], cls int32 [N], sample int32 [N])."""
    f = np.asarray(feature, dtype=np.float64)
    B, CH, h, w = f.shape
    f = f.transpose(0, 2, 3, 1).reshape(B, h * w, CH)
    lab, ins = np.asarray(labels).reshape(B, -1), np.asarray(inst).reshape(B, -1)
    protos, counts, cls, smp = [], [], [], []
    for b in range(B):
        for c in range(num_classes):
            for i in range(1, int(ncomp[b, c]) + 1):
                m = (lab[b] == c) & (ins[b] == i)
                protos.append(f[b][m].mean(0))
                counts.append(int(m.sum()))
                cls.append(c)
                smp.append(b)
    if not protos:
        return np.zeros((0, CH)), np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32)
    return np.stack(protos), np.array(counts, np.int64), np.array(cls, np.int32), np.array(smp, np.int32)
