"""-m gpu: the optional connected-component / instance-pooling kernels through the C-ABI.
No counterpart in the reference ("parity unpinned" w.r.t. Waybaba/DIGA): bit-exact against the scipy.ndimage.label
fixture and against the oracle's union-find restatement; pooled means within 1e-3 relative of the fp64 oracle."""
import numpy as np
import pytest
import torch

from diga_amd import instances
from diga_amd._lib import DigaError
from oracle import diga_oracle as O

pytestmark = pytest.mark.gpu

CASES = ("tiny", "feat", "noisy", "wide", "shapes")


def test_ccl_matches_scipy_fixture(gold_ccl):
    g = gold_ccl
    for tag in CASES:
        lab = torch.from_numpy(g[f"{tag}_labels"]).cuda()
        for conn in (4, 8):
            inst, ncomp = instances.ccl_label(lab, int(g[f"{tag}_C"]), conn)
            assert np.array_equal(ncomp.cpu().numpy(), g[f"{tag}_ncomp{conn}"]), (tag, conn)
            assert np.array_equal(inst.cpu().numpy(), g[f"{tag}_inst{conn}"]), (tag, conn)


def _labels(rng, B, h, w, C, cell, noise, ignore):
    coarse = rng.integers(0, C, (B, -(-h // cell), -(-w // cell)))
    lab = np.repeat(np.repeat(coarse, cell, 1), cell, 2)[:, :h, :w].copy()
    flip = rng.random((B, h, w)) < noise
    lab[flip] = rng.integers(0, C, int(flip.sum()))
    lab[rng.random((B, h, w)) < ignore] = 255
    return lab.astype(np.int32)


@pytest.mark.parametrize("shape", [(3, 129, 257, 19, 16, 0.05), (1, 512, 1024, 19, 40, 0.01), (2, 1, 700, 5, 7, 0.1), (2, 300, 1, 5, 7, 0.1),
                                   (1, 192, 192, 2, 1, 0.0)])
def test_ccl_matches_oracle_at_frame_sizes(shape):
    """Feature grids, a full-resolution label map, single-row / single-column frames and pure noise (cell 1)."""
    B, h, w, Cn, cell, noise = shape
    lab = _labels(np.random.default_rng(h * 7 + w), B, h, w, Cn, cell, noise, 0.04)
    for conn in (4, 8):
        want_inst, want_n = O.ccl_label(lab, Cn, conn)
        inst, ncomp = instances.ccl_label(torch.from_numpy(lab).cuda(), Cn, conn)
        assert np.array_equal(ncomp.cpu().numpy(), want_n)
        assert np.array_equal(inst.cpu().numpy(), want_inst)


def test_ccl_edge_cases():
    dev = "cuda"
    # nothing but background; one class everywhere; int64 / uint8 label dtypes; a serpentine that merges only at the end
    inst, ncomp = instances.ccl_label(torch.full((2, 9, 11), 255, dtype=torch.int64, device=dev), 19, 8)
    assert int(inst.abs().sum()) == 0 and int(ncomp.sum()) == 0
    inst, ncomp = instances.ccl_label(torch.zeros(1, 65, 129, dtype=torch.uint8, device=dev), 19, 4)
    assert int((inst != 1).sum()) == 0 and ncomp[0].tolist() == [1] + [0] * 18
    h, w = 64, 63
    lab = np.full((1, h, w), 255, np.int32)
    lab[0, ::2, :] = 3
    lab[0, 1::4, w - 1] = 3
    lab[0, 3::4, 0] = 3                       # one snake through every even row
    inst, ncomp = instances.ccl_label(torch.from_numpy(lab).cuda(), 19, 4)
    assert int(ncomp[0, 3]) == 1 and np.array_equal(inst.cpu().numpy()[0] > 0, lab[0] == 3)
    with pytest.raises(DigaError):
        instances.ccl_label(torch.zeros(1, 4, 4, dtype=torch.int32), 19, 8)            # CPU tensor: no fallback
    with pytest.raises(DigaError):
        instances.ccl_label(torch.zeros(1, 4, 4, dtype=torch.int32, device=dev), 19, 6)  # connectivity


@pytest.mark.parametrize("case", [(2, 33, 41, 64, 5, 6, 0.1), (1, 65, 129, 2048, 19, 9, 0.06), (3, 65, 129, 256, 19, 20, 0.02)])
def test_instance_pool_matches_oracle(case):
    B, h, w, CH, Cn, cell, noise = case
    rng = np.random.default_rng(CH + h)
    lab = _labels(rng, B, h, w, Cn, cell, noise, 0.05)
    feat = torch.from_numpy(rng.standard_normal((B, CH, h, w)).astype(np.float32) * 2 + 0.5).to(torch.bfloat16)
    f_nhwc = feat.permute(0, 2, 3, 1).contiguous().cuda().reshape(B * h * w, CH)
    labd = torch.from_numpy(lab).cuda()
    inst, ncomp = instances.ccl_label(labd, Cn, 8)
    want_inst, want_n = O.ccl_label(lab, Cn, 8)
    assert np.array_equal(inst.cpu().numpy(), want_inst)
    t = instances.instance_prototypes(f_nhwc, labd, inst, ncomp, Cn, max_instances=8192)
    protos, counts, cls, smp = O.instance_prototypes(feat.float().numpy(), lab, want_inst, want_n, Cn)
    assert t.total == len(counts) == int(want_n.sum())
    assert np.array_equal(t.counts.cpu().numpy(), counts)          # integers: bit-exact
    assert np.array_equal(t.classes.cpu().numpy(), cls)
    assert np.array_equal(t.samples.cpu().numpy(), smp)
    got = t.protos.cpu().numpy().astype(np.float64)
    # tolerance 1e-3 relative to the prototype's scale (north_star: "fp32 logits/prototypes within 1e-3 relative")
    scale = np.maximum(np.abs(protos).max(axis=1, keepdims=True), 1e-6)
    assert float((np.abs(got - protos) / scale).max()) < 1e-3
    # deterministic: a second call is bit-identical
    t2 = instances.instance_prototypes(f_nhwc, labd, inst, ncomp, Cn, max_instances=8192)
    assert torch.equal(t.protos, t2.protos)
    # a table that is too small keeps its first rows and reports the true total
    small = instances.instance_prototypes(f_nhwc, labd, inst, ncomp, Cn, max_instances=7)
    assert small.total == t.total and small.protos.shape[0] == 7 and torch.equal(small.protos, t.protos[:7])
    assert torch.equal(small.counts, t.counts[:7])


def test_module_level_api():
    """DIGA.instance_labels / DIGA.cal_instance_proto (optional extension of the module mirror) on NCHW fp32 tensors."""
    import functools

    from diga_amd.ttda_module import DIGA, DeepLabv2AdaptSeg
    names = [f"c{i}" for i in range(19)]
    m = DIGA(net=functools.partial(DeepLabv2AdaptSeg, multi_level=True, restore_from=None),
             dataset_info=dict(num_classes=19, image_size=(64, 64), test_list=[0], class_names=names),
             cfg=dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)).cuda().eval()
    rng = np.random.default_rng(4)
    lab = _labels(rng, 2, 17, 33, 19, 5, 0.1, 0.05)
    feat = torch.from_numpy(rng.standard_normal((2, 64, 17, 33)).astype(np.float32))
    inst, ncomp = m.instance_labels(torch.from_numpy(lab).long())
    want_inst, want_n = O.ccl_label(lab, 19, 8)
    assert inst.dtype == torch.int64 and np.array_equal(inst.cpu().numpy(), want_inst) and np.array_equal(ncomp.cpu().numpy(), want_n)
    t = m.cal_instance_proto(feat, torch.from_numpy(lab).long())
    protos, counts, cls, smp = O.instance_prototypes(feat.to(torch.bfloat16).float().numpy(), lab, want_inst, want_n, 19)
    assert t.total == len(counts) and np.array_equal(t.counts.cpu().numpy(), counts) and np.array_equal(t.classes.cpu().numpy(), cls)
    scale = np.maximum(np.abs(protos).max(axis=1, keepdims=True), 1e-6)
    assert float((np.abs(t.protos.cpu().numpy() - protos) / scale).max()) < 1e-3
