"""-m gpu: BASELINE configs[4] as a LONG stream -- 102 batch-1 frames cycling 1280x720 / 1024x512 / 960x540 without
resetting the prototypes (ttda:1702-1705), ~100 EMA steps (ttda:1764) -- against the reference run of the same stream
(tests/golden/net_long.npz, oracle/gen_golden.py long).  Drift is bounded PER FRAME (running-prototype norms, flags,
cumulative confusion matrix), not only through the final mIoU.

 * fp32-exact mode: prototypes within 1e-4 relative at every frame, flags identical, cumulative confusion matrix within
   0.1 % of the counted pixels at every frame (the reference's own denormal-weight pixels and in-band decisions, see
   tests/test_gpu_exact.py), mIoU within 0.01 at the end;
 * bf16 product path: the same quantities at the bf16-storage floor (bounds stated below, measured values printed).
"""
import functools

import numpy as np
import pytest
import torch

from conftest import load_golden
from diga_amd import synth
from diga_amd import ttda_module as T

pytestmark = pytest.mark.gpu
CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)


def _diga(synth_params_np, precision):
    m = T.DIGA(net=functools.partial(T.DeepLabv2AdaptSeg, multi_level=True, restore_from=None),
               dataset_info=dict(num_classes=19, image_size=(321, 321), test_list=[0], class_names=[str(i) for i in range(19)]), cfg=CFG,
               precision=precision)
    m.net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params_np.items()}, strict=False)
    m._configure_bn_running_stats()
    return m


def _miou(hist):
    met = T.SegmentationMetric(19)
    met.add_counts(torch.from_numpy(np.asarray(hist)))
    return np.array([float(met.compute()), float(met.compute_iou("16")), float(met.compute_iou("13"))])


def _run(m, g):
    n = int(g["lg_frames"])
    shapes = [tuple(int(v) for v in hw) for hw in g["lg_shapes"]]
    rows = []
    for t in range(n):
        H, W = shapes[t % 3]
        x, y = synth.synth_frame(t, 1, H, W)
        m.test_step((torch.from_numpy(x), torch.from_numpy(y), None, None), t)
        torch.cuda.synchronize()
        run = m.classifier_running_proto
        hist = m.test_acc[0].hist.cpu().numpy()
        ref_norm, ref_sub = g[f"lg_{t}_run_norm"], g[f"lg_{t}_run_sub"]
        rows.append(dict(
            norm_err=float(np.abs(run.norm(dim=1).cpu().numpy() - ref_norm).max() / ref_norm.max()),
            sub_err=float(np.abs(run[:, ::64].cpu().numpy() - ref_sub).max() / np.abs(ref_sub).max()),
            flags_equal=bool(np.array_equal(m.classifier_running_proto_exists_flag.cpu().numpy(), g[f"lg_{t}_rflag"])),
            hist_moved=float(np.abs(hist - g[f"lg_{t}_hist"]).sum() / 2 / max(1.0, g[f"lg_{t}_hist"].sum())),
            miou_delta=float(np.abs(_miou(hist) - _miou(g[f"lg_{t}_hist"])).max()),
            nan_px=(int(torch.isnan(m._last_engine.probs).any(1).sum()), int(g[f"lg_{t}_nan_pixels"]))))
    return rows


def test_long_multires_stream_exact_mode(synth_params_np):
    g = load_golden("net_long.npz")
    m = _diga(synth_params_np, "fp32x3")
    rows = _run(m, g)
    worst = {k: max(r[k] for r in rows) for k in ("norm_err", "sub_err", "hist_moved", "miou_delta")}
    print("exact mode, worst over", len(rows), "frames:", worst, "last:", rows[-1])
    assert all(r["flags_equal"] for r in rows)
    assert worst["norm_err"] <= 1e-4 and worst["sub_err"] <= 1e-4, worst          # ~100 EMA steps: no drift beyond fp32 rounding
    # pixels that change class: the reference's own denormal-weight pixels (tests/test_gpu_exact.py) and in-band decisions;
    # measured 4e-4 of the counted pixels at worst (early frames), 1e-4 at the end
    assert worst["hist_moved"] <= 1e-3 and rows[-1]["hist_moved"] <= 3e-4, (worst, rows[-1])
    assert rows[-1]["miou_delta"] <= 0.01 and worst["miou_delta"] <= 0.02, (worst, rows[-1])
    assert all(abs(a - b) <= max(3, 0.2 * b) for a, b in (r["nan_px"] for r in rows)), [r["nan_px"] for r in rows]
    ref_final = g["lg_run_final"]
    assert float(np.abs(m.classifier_running_proto.cpu().numpy() - ref_final).max() / np.abs(ref_final).max()) <= 1e-4


def test_long_multires_stream_bf16_path(synth_params_np):
    g = load_golden("net_long.npz")
    m = _diga(synth_params_np, "bf16")
    rows = _run(m, g)
    worst = {k: max(r[k] for r in rows) for k in ("norm_err", "sub_err", "hist_moved", "miou_delta")}
    print("bf16 path, worst over", len(rows), "frames:", worst, "last:", rows[-1], "flag mismatches at frames",
          [t for t, r in enumerate(rows) if not r["flags_equal"]])
    # the running prototypes are means over thousands of bf16-stored feature rows: their error stays at the storage floor and
    # does NOT grow over the ~100 EMA steps
    late = rows[len(rows) // 2:]
    assert max(r["norm_err"] for r in late) <= 1.5 * max(r["norm_err"] for r in rows[:len(rows) // 2]) + 1e-3
    # measured: prototype norms within 1.2 %, single prototype entries within 7 % (classes seeded from a handful of pixels in
    # the first frames), 0.6 % once the EMA has averaged ~50 frames; 3.3 % of the counted pixels change class at worst, 0.3 % at
    # the end; mIoU within 0.41 at worst (early, few pixels per class), 0.08 after 102 frames.  Bounds = 1.5-2x those.
    assert worst["norm_err"] <= 0.025 and worst["sub_err"] <= 0.15 and max(r["sub_err"] for r in late) <= 0.03, worst
    # flags come from ground-truth labels from frame 1 on (ttda:1705): identical once every class has been seen
    assert all(r["flags_equal"] for r in rows[5:])
    assert worst["hist_moved"] <= 0.06 and rows[-1]["hist_moved"] <= 0.01, (worst, rows[-1])
    assert rows[-1]["miou_delta"] <= 0.25 and max(r["miou_delta"] for r in late) <= 0.5, (worst, rows[-1])


# ------------------------------------------------------------------------------------------------------------------------
# One evaluation-length stream: 500 batch-1 frames of 1024x512 (GTA5 -> Cityscapes val has 500 images, README.md:241-243),
# tests/golden/net_eval500.npz: the reference's cumulative confusion matrix / mIoU / prototype norms every 50 frames.
# ------------------------------------------------------------------------------------------------------------------------
def _eval500_checkpoints(step_fn, g, per_step):
    """Drive `step_fn(list of (x, y) frames)` over the stream in steps of `per_step` frames (a divisor of the 50-frame checkpoint
    spacing); after every 50 frames yield (frames so far, the checkpoint's reference entries)."""
    n, every = int(g["ev_frames"]), int(g["ev_every"])
    assert every % per_step == 0
    for t0 in range(0, n, per_step):
        step_fn([synth.synth_frame(t, 1, 512, 1024) for t in range(t0, t0 + per_step)])
        done = t0 + per_step
        if done % every == 0:
            yield done, {k: g[f"ev_{done}_{k}"] for k in ("hist", "miou", "run_norm", "rflag", "nan_pixels")}


@pytest.fixture(scope="module")
def weights(synth_params_np):
    from diga_amd.engine import NetWeights
    return NetWeights(synth_params_np, torch.device("cuda:0"))


def test_eval_length_stream_500_frames_bf16_path(weights):
    """The bf16 PRODUCT path on ONE stream of realistic length, through the bench's configuration (two windows of frames in
    flight, results bit-identical to one-frame steps): mIoU-19/16/13 within the north star's +-0.1 of the reference at 500
    frames -- a single stream, no averaging over draws: over 500 frames the storage noise of the individual frames averages out."""
    from diga_amd.pipeline import StreamWindows
    g = load_golden("net_eval500.npz")
    K = 25
    sw = StreamWindows(weights, K, 2, 512, 1024, bn_lambda=CFG["bn_lambda"], autotune=False)

    def step(frames):
        wins = []
        for w in range(2):
            fr = frames[w * K:(w + 1) * K]
            wins.append((torch.from_numpy(np.concatenate([a for a, _ in fr])).cuda(), torch.from_numpy(np.concatenate([b for _, b in fr])).cuda()))
        sw.step(wins, CFG)
        torch.cuda.synchronize()

    rows = []
    for done, ref in _eval500_checkpoints(step, g, 2 * K):
        e = sw.engines[0]
        hist = e.hist.reshape(19, 19).cpu().numpy()
        rows.append(dict(frames=done, delta=(_miou(hist) - ref["miou"]).tolist(),
                         moved=float(np.abs(hist - ref["hist"]).sum() / 2 / ref["hist"].sum()),
                         norm_err=float(np.abs(e.run_proto.norm(dim=1).cpu().numpy() - ref["run_norm"]).max() / ref["run_norm"].max()),
                         flags_equal=bool(np.array_equal(e.run_flag.cpu().numpy(), ref["rflag"]))))
        print(rows[-1], flush=True)
    last = rows[-1]
    assert last["frames"] == 500 and all(r["flags_equal"] for r in rows)
    assert np.abs(np.array(last["delta"])).max() <= 0.1, last                 # the north star's bar, single stream, 500 frames
    assert all(np.abs(np.array(r["delta"])).max() <= 0.2 for r in rows[1:]), rows     # from 100 frames on
    assert last["moved"] <= 0.02 and max(r["norm_err"] for r in rows) <= 0.02, rows


def test_eval_length_stream_500_frames_exact_mode(synth_params_np):
    """The fp32-exact mode on the same 500-frame stream (10 frames in flight: three operand planes of 25 frames would exceed the
    conv kernel's 2 GiB tensor range): mIoU within 0.01 of the reference at EVERY 50-frame checkpoint, prototype norms to fp32
    rounding, flags identical."""
    from diga_amd.verify import ExactEngine, ExactWeights
    g = load_golden("net_eval500.npz")
    K = 10
    eng = ExactEngine(ExactWeights(synth_params_np, torch.device("cuda:0"), nplanes=3), K, 512, 1024, stream=True)

    def step(frames):
        x = torch.from_numpy(np.concatenate([a for a, _ in frames])).cuda()
        y = torch.from_numpy(np.concatenate([b for _, b in frames])).cuda()
        eng.forward_and_adapt(x, y, CFG)
        eng.tail(y)
        torch.cuda.synchronize()

    rows = []
    for done, ref in _eval500_checkpoints(step, g, K):
        hist = eng.hist.reshape(19, 19).cpu().numpy()
        rows.append(dict(frames=done, delta=float(np.abs(_miou(hist) - ref["miou"]).max()),
                         moved=float(np.abs(hist - ref["hist"]).sum() / 2 / ref["hist"].sum()),
                         norm_err=float(np.abs(eng.run_proto.norm(dim=1).cpu().numpy() - ref["run_norm"]).max() / ref["run_norm"].max()),
                         flags_equal=bool(np.array_equal(eng.run_flag.cpu().numpy(), ref["rflag"]))))
        print(rows[-1], flush=True)
    assert rows[-1]["frames"] == 500 and all(r["flags_equal"] for r in rows)
    assert max(r["delta"] for r in rows) <= 0.01, rows
    assert max(r["norm_err"] for r in rows) <= 1e-4 and max(r["moved"] for r in rows) <= 1e-3, rows
    ref_final = g["ev_run_final"]
    assert float(np.abs(eng.run_proto.cpu().numpy() - ref_final).max() / np.abs(ref_final).max()) <= 1e-4


def test_bench_configuration_is_deterministic_under_two_streams(weights):
    """The measured configuration (two windows of 15 frames on two HIP streams, autotuned tiles incl. the hand-counted-wait kernels):
    two independent runs of 6 steps (180 frames) end with the SAME BITS of confusion matrix, running prototypes, flags, and of the
    last step's logits and features -- a wait that is too lenient, or a race between the two streams, would show here as a
    difference between runs."""
    from diga_amd.pipeline import StreamWindows
    K, NW, steps = 15, 2, 6
    frames = [synth.synth_frame(t, 1, 512, 1024) for t in range(K * NW)]
    wins = []
    for w in range(NW):
        fr = frames[w * K:(w + 1) * K]
        wins.append((torch.from_numpy(np.concatenate([a for a, _ in fr])).cuda(), torch.from_numpy(np.concatenate([b for _, b in fr])).cuda()))
    outs = []
    for run in range(2):
        sw = StreamWindows(weights, K, NW, 512, 1024, bn_lambda=CFG["bn_lambda"], autotune=True)
        for _ in range(steps):
            sw.step(wins, CFG)
        torch.cuda.synchronize()
        e0, e1 = sw.engines
        outs.append([e0.hist.clone(), e0.run_proto.clone(), e0.run_flag.clone(), e0.logits.clone(), e1.logits.clone(),
                     e0.feat.view(torch.int16).clone(), e1.feat.view(torch.int16).clone(), e1.pred.clone()])
        del sw
    for i, (a, b) in enumerate(zip(*outs)):
        a = torch.nan_to_num(a.float(), nan=-1.0) if a.is_floating_point() else a
        b = torch.nan_to_num(b.float(), nan=-1.0) if b.is_floating_point() else b
        assert torch.equal(a, b), i
