"""-m gpu: kernel-level parity of libdiga_hip.so (through the C-ABI) against the CPU oracle.

Stage-isolated contract (SURVEY.md section 7 hard part 1): every kernel is fed inputs that are exactly
representable in its storage type, the oracle is fed the same values in fp32, so the only differences
are fp32 summation order and the final bf16 rounding of stored activations:
  * bf16 activations: |gpu - bf16(oracle)| <= 1 bf16 ulp  (2^-8 relative to the tensor scale here)
  * fp32 outputs (statistics -> scale/shift, logits): <= 1e-4 relative (north-star bar: 1e-3)
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, P, bf16_round, call, conv_desc, from_nhwc_bf16, rel_err, stream, to_nhwc_bf16
from diga_amd.engine import pack_conv_weight, pack_taps
from oracle import diga_oracle as O

pytestmark = pytest.mark.gpu

BF16_TOL = 2.0 ** -7   # one bf16 spacing at the tensor scale (RNE error is half of it, a flipped rounding a full one)


def _bn_params(Cc, g):
    rm = torch.randn(Cc, generator=g) * 0.3
    rv = torch.rand(Cc, generator=g) + 0.5
    w = torch.rand(Cc, generator=g) + 0.5
    b = torch.randn(Cc, generator=g) * 0.1
    return rm, rv, w, b


def _oracle_scale_shift(ref_out, rm, rv, w, b, lam=0.8, eps=1e-5):
    mean_cur, var_cur = O.dam_current_stats(ref_out)
    lam_t = torch.tensor(lam, dtype=torch.float32)
    mean = lam_t * rm + (1 - lam_t) * mean_cur
    var = lam_t * rv + (1 - lam_t) * var_cur
    sc = w / torch.sqrt(var + eps)
    return sc, b - mean * sc


CONV_CASES = [
    # B, H,  W,  Cin, Cout, stride, k3,   dil, tile
    (1, 9, 17, 64, 64, 1, False, 1, 3),
    (1, 17, 33, 256, 128, 2, False, 1, 2),
    (2, 9, 17, 128, 128, 1, True, 2, 1),
    (1, 13, 21, 64, 256, 1, True, 4, 4),
    (3, 9, 17, 64, 64, 1, True, 1, 0),
    (1, 33, 65, 64, 256, 1, False, 1, 0),
    (2, 17, 33, 128, 64, 1, True, 1, 3),
    (1, 23, 31, 512, 512, 1, True, 4, 0),
    # 8-wave tiles, 64- and 32-channel K steps
    (2, 23, 31, 128, 256, 1, True, 2, 48),
    (1, 33, 65, 64, 256, 1, False, 1, 66),
    (3, 9, 17, 256, 512, 1, False, 1, 57),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,k3,dil,tile", CONV_CASES)
def test_conv_igemm(B, H, W, Cin, Cout, stride, k3, dil, tile):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin + Cout + dil)
    x = bf16_round(torch.randn(B, Cin, H, W, generator=g).relu() + 0.1 * torch.randn(B, Cin, H, W, generator=g))
    ks = 3 if k3 else 1
    wt = bf16_round(torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5)
    rm, rv, bw, bb = _bn_params(Cout, g)
    ref = F.conv2d(x, wt, stride=stride, padding=dil if k3 else 0, dilation=dil if k3 else 1)
    Ho, Wo = ref.shape[2:]
    d = conv_desc(B, H, W, Cin, Cout, stride, k3, dil, tile=tile)
    assert (d.Ho, d.Wo) == (Ho, Wo)
    xin = to_nhwc_bf16(x)
    wp = pack_conv_weight(wt).to(DEV)
    out = torch.zeros(B * Ho * Wo * Cout, dtype=torch.bfloat16, device=DEV)
    nu = call("diga_conv_stat_units", C.byref(d))
    ups = nu // B
    assert nu == B * ((Ho * Wo + 31) // 32)
    slab = torch.full((nu * 2 * Cout,), float("nan"), dtype=torch.float32, device=DEV)
    dv = [t.to(DEV) for t in (rm, rv, bw, bb)]
    call("diga_conv_igemm_bf16", C.byref(d), P(xin), P(wp), P(out), P(slab), P(dv[0]), stream())
    sc = torch.empty(Cout, dtype=torch.float32, device=DEV)
    sh = torch.empty(Cout, dtype=torch.float32, device=DEV)
    n_a = float(Ho * Wo)
    n_b = float((B - 1) * Ho * Wo)
    call("diga_bn_finalize", P(slab), ups, B, 1 if B > 1 else B, 0, Cout, n_a, n_b, P(dv[0]), P(dv[1]), P(dv[2]), P(dv[3]), 0.8, 1e-5,
         P(sc), P(sh), stream())
    # per-sample statistics (a stream of batch-1 frames riding in one launch): [B][Cout] affines
    scp = torch.empty(B * Cout, dtype=torch.float32, device=DEV)
    shp = torch.empty(B * Cout, dtype=torch.float32, device=DEV)
    call("diga_bn_finalize", P(slab), ups, B, B, 1, Cout, n_a, 0.0, P(dv[0]), P(dv[1]), P(dv[2]), P(dv[3]), 0.8, 1e-5,
         P(scp), P(shp), stream())
    torch.cuda.synchronize()
    assert not torch.isnan(slab).any()          # every unit of every sample was written
    got = from_nhwc_bf16(out, (B, Ho, Wo, Cout))
    e = rel_err(got, bf16_round(ref))
    assert e <= BF16_TOL, f"raw conv output rel err {e}"
    rsc, rsh = _oracle_scale_shift(ref, rm, rv, bw, bb)
    assert rel_err(sc.cpu(), rsc) < 1e-4 and rel_err(sh.cpu(), rsh) < 1e-4, (rel_err(sc.cpu(), rsc), rel_err(sh.cpu(), rsh))
    for b in range(B):
        rsc, rsh = _oracle_scale_shift(ref[b:b + 1], rm, rv, bw, bb)
        assert rel_err(scp.cpu()[b * Cout:(b + 1) * Cout], rsc) < 1e-4 and rel_err(shp.cpu()[b * Cout:(b + 1) * Cout], rsh) < 1e-4


def test_conv_results_do_not_depend_on_tile_or_batch():
    """The statistics units (32 rows of one sample) and the outputs are bit-identical for every tile shape, and a
    sample's rows/units are bit-identical whether it is computed alone or as part of a batch: this is what makes
    stream mode (k batch-1 frames per trunk pass) exact."""
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout = 3, 13, 21, 128, 256
    x = bf16_round(torch.randn(B, Cin, H, W, generator=g).relu())
    wt = bf16_round(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03)
    wp = pack_conv_weight(wt).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)

    def run(xs, tile):
        b = xs.shape[0]
        d = conv_desc(b, H, W, Cin, Cout, 1, True, 2, tile=tile)
        nu = call("diga_conv_stat_units", C.byref(d))
        xin = to_nhwc_bf16(xs)
        out = torch.zeros(b * H * W * Cout, dtype=torch.bfloat16, device=DEV)
        slab = torch.zeros(nu * 2 * Cout, dtype=torch.float32, device=DEV)
        call("diga_conv_igemm_bf16", C.byref(d), P(xin), P(wp), P(out), P(slab), P(shift), stream())
        torch.cuda.synchronize()
        return out.view(torch.int16).cpu().reshape(b, -1), slab.cpu().reshape(b, -1)

    o0, s0 = run(x, 23)
    for tile in (1, 12, 24, 3, 21, 26, 16, 27, 17, 28, 31, 32, 33, 34, 36, 37, 38, 46, 47, 48, 56, 57, 58, 61, 64, 66, 67):
        o, sl = run(x, tile)
        assert torch.equal(o, o0) and torch.equal(sl.view(torch.int32), s0.view(torch.int32)), tile
    for b in range(B):
        o, sl = run(x[b:b + 1], 24)
        assert torch.equal(o[0], o0[b]) and torch.equal(sl[0].view(torch.int32), s0[b].view(torch.int32)), b


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 13, 21, 256, 256), (2, 9, 17, 128, 512), (1, 33, 65, 64, 256), (4, 7, 5, 256, 1024),
                                            (6, 65, 129, 256, 1024), (2, 129, 257, 64, 256), (5, 65, 129, 128, 512),
                                            (3, 13, 21, 512, 2048), (4, 65, 129, 512, 1024), (1, 7, 5, 512, 64)])
def test_weights_stationary_1x1_is_bit_identical(B, H, W, Cin, Cout):
    """Tile id 91 (conv1x1_ws_kernel: weights resident in LDS, one wave per 32-row statistics unit) against the tiled kernel:
    raw output and statistics slab bit-identical, ragged last unit of every sample included; shapes it does not cover are
    refused with an error."""
    g = torch.Generator().manual_seed(B * 100 + Cin + Cout)
    x = bf16_round(torch.randn(B, Cin, H, W, generator=g).relu() + 0.1 * torch.randn(B, Cin, H, W, generator=g))
    wt = bf16_round(torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5)
    wp = pack_conv_weight(wt).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    xin = to_nhwc_bf16(x)

    def run(tile):
        d = conv_desc(B, H, W, Cin, Cout, 1, False, 1, tile=tile)
        nu = call("diga_conv_stat_units", C.byref(d))
        out = torch.full((B * H * W * Cout,), float("nan"), dtype=torch.bfloat16, device=DEV)
        slab = torch.full((nu * 2 * Cout,), float("nan"), dtype=torch.float32, device=DEV)
        call("diga_conv_igemm_bf16", C.byref(d), P(xin), P(wp), P(out), P(slab), P(shift), stream())
        torch.cuda.synchronize()
        return out.view(torch.int16).cpu(), slab.cpu()

    ws = 92 if Cin == 512 else 91   # 92: the K = 512 form with 64-column slices
    o0, s0 = run(23)
    for tile in ((ws,) if Cin == 512 else (91, 94, 95, 96)):  # 94: two 64-column passes per unit, 12 waves; 95: pipelined halves
        for rep in range(3):        # the kernel counts its own memory operations: screen for a wait that is too lenient
            o1, s1 = run(tile)
            assert not torch.isnan(s1).any()
            assert torch.equal(o0, o1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32)), (tile, rep)
    ref = F.conv2d(x, wt)
    assert rel_err(from_nhwc_bf16(o1.view(torch.bfloat16).to(DEV), (B, H, W, Cout)), bf16_round(ref)) <= BF16_TOL
    # a 3x3 conv / Cin = 512 has no weights-stationary kernel: refused, not silently mis-run
    d = conv_desc(B, H, W, Cin, Cout, 1, True, 1, tile=ws)
    w9 = torch.zeros(Cout * 9 * Cin, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(B * H * W * Cout, dtype=torch.bfloat16, device=DEV)
    slab = torch.zeros(call("diga_conv_stat_units", C.byref(d)) * 2 * Cout, dtype=torch.float32, device=DEV)
    with pytest.raises(Exception, match=f"tile {ws}"):
        call("diga_conv_igemm_bf16", C.byref(d), P(xin), P(w9), P(out), P(slab), P(shift), stream())


@pytest.mark.parametrize("B,rows,Cc,per_sample,res", [(1, 8385, 256, 0, False), (1, 1000, 64, 0, True), (3, 545, 128, 1, False),
                                                      (2, 2145, 1024, 1, True), (4, 300, 512, 0, True), (1, 33153, 64, 0, False)])
def test_bn_finalize_apply_is_bit_identical(B, rows, Cc, per_sample, res):
    """diga_bn_finalize_apply == diga_bn_finalize followed by diga_bn_apply: scale, shift and the activation bit for bit
    (both statistics modes: per-sample, and the half_first groups of a batch)."""
    g = torch.Generator().manual_seed(rows + Cc + B)
    ups = (rows + 31) // 32
    M = B * rows
    slab = (torch.randn(B * ups * 2 * Cc, generator=g) * 3).abs_().to(DEV)
    rm, rv, bw, bb = [t.to(DEV) for t in _bn_params(Cc, g)]
    raw = torch.randn(M * Cc, generator=g).to(torch.bfloat16).to(DEV)
    r = torch.randn(M * Cc, generator=g).relu().to(torch.bfloat16).to(DEV) if res else None
    ns = B if per_sample else 1
    a_samples = B if per_sample else max(1, B // 2)
    n_a = float(rows if per_sample else a_samples * rows)
    n_b = 0.0 if per_sample else float((B - a_samples) * rows)
    sc0, sh0 = torch.empty(ns * Cc, device=DEV), torch.empty(ns * Cc, device=DEV)
    out0 = torch.empty(M * Cc, dtype=torch.bfloat16, device=DEV)
    call("diga_bn_finalize", P(slab), ups, B, a_samples, per_sample, Cc, n_a, n_b, P(rm), P(rv), P(bw), P(bb), 0.8, 1e-5, P(sc0), P(sh0), stream())
    call("diga_bn_apply", P(raw), P(sc0), P(sh0), P(r) if res else None, None, None, P(out0), M, Cc, 1, B if per_sample else 0, stream())
    sc1, sh1 = torch.full_like(sc0, float("nan")), torch.full_like(sh0, float("nan"))
    out1 = torch.full((M * Cc,), float("nan"), dtype=torch.bfloat16, device=DEV)
    call("diga_bn_finalize_apply", P(slab), ups, B, a_samples, per_sample, Cc, n_a, n_b, P(rm), P(rv), P(bw), P(bb), 0.8, 1e-5, P(sc1), P(sh1),
         P(raw), P(r) if res else None, P(out1), M, 1, stream())
    torch.cuda.synchronize()
    assert torch.equal(sc0, sc1) and torch.equal(sh0, sh1)
    assert torch.equal(out0.view(torch.int16), out1.view(torch.int16))


@pytest.mark.parametrize("B,per_sample,tile", [(1, 0, 0), (3, 1, 48), (2, 0, 24), (2, 1, 67)])
def test_conv3_two_pass_fused_apply(B, per_sample, tile):
    """conv3 -> bn3 -> += residual -> relu as two passes over the GEMM: pass 1 (out = NULL) leaves exactly the slab of the
    storing kernel, pass 2 (diga_conv_igemm_bf16_apply) writes relu(acc*scale + shift + residual) from the fp32 accumulators."""
    g = torch.Generator().manual_seed(31 + B)
    H, W, Cin, Cout = 13, 21, 128, 512
    x = bf16_round(torch.randn(B, Cin, H, W, generator=g).relu())
    wt = bf16_round(torch.randn(Cout, Cin, 1, 1, generator=g) * 0.08)
    res = bf16_round(torch.randn(B, Cout, H, W, generator=g).relu())
    rm, rv, bw, bb = _bn_params(Cout, g)
    d = conv_desc(B, H, W, Cin, Cout, tile=tile)
    nu = call("diga_conv_stat_units", C.byref(d))
    xin, wp, rin = to_nhwc_bf16(x), pack_conv_weight(wt).to(DEV), to_nhwc_bf16(res)
    dv = [t.to(DEV) for t in (rm, rv, bw, bb)]
    out = torch.zeros(B * H * W * Cout, dtype=torch.bfloat16, device=DEV)
    slab_a = torch.full((nu * 2 * Cout,), float("nan"), dtype=torch.float32, device=DEV)
    slab_b = torch.full((nu * 2 * Cout,), float("nan"), dtype=torch.float32, device=DEV)
    call("diga_conv_igemm_bf16", C.byref(d), P(xin), P(wp), P(out), P(slab_a), P(dv[0]), stream())
    call("diga_conv_igemm_bf16", C.byref(d), P(xin), P(wp), None, P(slab_b), P(dv[0]), stream())
    torch.cuda.synchronize()
    assert torch.equal(slab_a.view(torch.int32), slab_b.view(torch.int32))
    ns = B if per_sample else 1
    sc = torch.empty(ns * Cout, dtype=torch.float32, device=DEV)
    sh = torch.empty(ns * Cout, dtype=torch.float32, device=DEV)
    if per_sample:
        call("diga_bn_finalize", P(slab_b), nu // B, B, B, 1, Cout, float(H * W), 0.0, P(dv[0]), P(dv[1]), P(dv[2]), P(dv[3]), 0.8,
             1e-5, P(sc), P(sh), stream())
    else:
        call("diga_bn_finalize", P(slab_b), nu // B, B, 1 if B > 1 else B, 0, Cout, float(H * W), float((B - 1) * H * W), P(dv[0]),
             P(dv[1]), P(dv[2]), P(dv[3]), 0.8, 1e-5, P(sc), P(sh), stream())
    y = torch.zeros(B * H * W * Cout, dtype=torch.bfloat16, device=DEV)
    call("diga_conv_igemm_bf16_apply", C.byref(d), P(xin), P(wp), P(y), P(sc), P(sh), P(rin), 1, B if per_sample else 0, stream())
    torch.cuda.synchronize()
    conv = F.conv2d(x, wt)
    scc, shc = sc.cpu().reshape(ns, Cout, 1, 1), sh.cpu().reshape(ns, Cout, 1, 1)
    ref = torch.relu(conv * scc + shc + res)
    assert rel_err(from_nhwc_bf16(y, (B, H, W, Cout)), bf16_round(ref)) <= BF16_TOL
    for b in range(B if per_sample else 0):      # the affine itself, per sample, against the oracle
        rsc, rsh = _oracle_scale_shift(conv[b:b + 1], rm, rv, bw, bb)
        assert rel_err(sc.cpu()[b * Cout:(b + 1) * Cout], rsc) < 1e-4 and rel_err(sh.cpu()[b * Cout:(b + 1) * Cout], rsh) < 1e-4


def test_conv_rejects_bad_shapes():
    from diga_amd._lib import DigaError
    d = conv_desc(1, 9, 17, 48, 64)       # Cin not a multiple of 64
    with pytest.raises(DigaError, match="Cin"):
        call("diga_conv_stat_units", C.byref(d))


@pytest.mark.parametrize("B,h,w,splitk", [(1, 9, 17, 4), (2, 13, 21, 3), (1, 33, 65, 1)])
def test_head_conv_and_pseudo_labels(B, h, w, splitk):
    """Fused 2-branch head (dil 6 + dil 12, adaptseg:119-121) as one 18-tap split-K igemm + finish."""
    g = torch.Generator().manual_seed(7 + h)
    CH, NC = 2048, 19
    feat = bf16_round(torch.randn(B, CH, h, w, generator=g).relu() * 0.3)
    w0 = bf16_round(torch.randn(NC, CH, 3, 3, generator=g) * 0.05)
    w1 = bf16_round(torch.randn(NC, CH, 3, 3, generator=g) * 0.05)
    b0, b1 = torch.randn(NC, generator=g) * 0.5, torch.randn(NC, generator=g) * 0.5
    p = {"layer6.conv2d_list.0.weight": w0, "layer6.conv2d_list.0.bias": b0, "layer6.conv2d_list.1.weight": w1,
         "layer6.conv2d_list.1.bias": b1}
    ref = O.head_logits(p, feat)
    d = conv_desc(B, h, w, CH, NC, cout_pad=32, tile=5)
    d.ntaps = 18
    for br, dil in enumerate((6, 12)):
        for t in range(9):
            d.dy[br * 9 + t], d.dx[br * 9 + t] = (t // 3 - 1) * dil, (t % 3 - 1) * dil
    d.splitk, d.out_f32 = splitk, 1
    hw = pack_taps([w0, w1], 32).to(DEV)
    Pn = B * h * w
    part = torch.full((splitk * Pn * 32,), float("nan"), dtype=torch.float32, device=DEV)
    fd = to_nhwc_bf16(feat)
    call("diga_conv_igemm_bf16", C.byref(d), P(fd), P(hw), P(part), None, None, stream())
    logits = torch.empty(B, NC, h, w, dtype=torch.float32, device=DEV)
    labels = torch.empty(Pn, dtype=torch.int32, device=DEV)
    bias = (b0 + b1).to(DEV)
    call("diga_head_finish_pseudo_label", P(part), splitk, 32, P(bias), None, P(logits), P(labels), B, NC, h * w, float(np.float32(0.9)),
         0, stream())
    torch.cuda.synchronize()
    assert rel_err(logits.cpu(), ref) < 1e-4
    # pseudo labels from the GPU's own logits must match the oracle applied to those logits, except within 1e-6 of the bar
    lg = logits.cpu()
    want = O.pseudo_labels(lg, 0.9).reshape(-1).numpy()
    pm = lg.softmax(1).max(1)[0].reshape(-1).numpy()
    safe = np.abs(pm - 0.9) > 1e-6
    got = labels.cpu().numpy()
    assert np.array_equal(got[safe], want[safe])
    assert (~safe).sum() <= 2


@pytest.mark.parametrize("B,h,w", [(1, 9, 17), (2, 13, 21), (1, 33, 65)])
def test_head_as_gemm_plus_tap_gather(B, h, w):
    """The product path's head: one dense 1x1 igemm (Cout = 18 taps x 19 classes) + diga_head_tapsum_pseudo_label."""
    g = torch.Generator().manual_seed(17 + h)
    CH, NC = 2048, 19
    feat = bf16_round(torch.randn(B, CH, h, w, generator=g).relu() * 0.3)
    w0 = bf16_round(torch.randn(NC, CH, 3, 3, generator=g) * 0.05)
    w1 = bf16_round(torch.randn(NC, CH, 3, 3, generator=g) * 0.05)
    b0, b1 = torch.randn(NC, generator=g) * 0.5, torch.randn(NC, generator=g) * 0.5
    ref = O.head_logits({"layer6.conv2d_list.0.weight": w0, "layer6.conv2d_list.0.bias": b0, "layer6.conv2d_list.1.weight": w1,
                         "layer6.conv2d_list.1.bias": b1}, feat)
    rows = torch.cat([x.permute(2, 3, 0, 1).reshape(9 * NC, CH) for x in (w0, w1)], 0)
    cpad = 384
    wp = pack_conv_weight(rows[:, :, None, None], cpad).to(DEV)
    d = conv_desc(B, h, w, CH, 18 * NC, cout_pad=cpad)
    d.out_f32 = 1
    Pn = B * h * w
    z = torch.full((Pn * cpad,), float("nan"), dtype=torch.float32, device=DEV)
    fd = to_nhwc_bf16(feat)
    call("diga_conv_igemm_bf16", C.byref(d), P(fd), P(wp), P(z), None, None, stream())
    dy = (C.c_int16 * 18)(*[(t % 9 // 3 - 1) * (6, 12)[t // 9] for t in range(18)])
    dx = (C.c_int16 * 18)(*[(t % 3 - 1) * (6, 12)[t // 9] for t in range(18)])
    logits = torch.empty(B, NC, h, w, dtype=torch.float32, device=DEV)
    labels = torch.empty(Pn, dtype=torch.int32, device=DEV)
    bias = (b0 + b1).to(DEV)
    call("diga_head_tapsum_pseudo_label", P(z), cpad, 18, dy, dx, P(bias), P(logits), P(labels), B, NC, h, w, float(np.float32(0.9)), 0,
         stream())
    torch.cuda.synchronize()
    assert rel_err(logits.cpu(), ref) < 1e-4
    lg = logits.cpu()
    want = O.pseudo_labels(lg, 0.9).reshape(-1).numpy()
    safe = np.abs(lg.softmax(1).max(1)[0].reshape(-1).numpy() - 0.9) > 1e-6
    assert np.array_equal(labels.cpu().numpy()[safe], want[safe]) and (~safe).sum() <= 2


@pytest.mark.parametrize("B,H,W", [(1, 65, 129), (2, 47, 35), (1, 129, 257), (3, 260, 1030)])   # the last: 891 strips for 512 persistent blocks
def test_stem_bn_maxpool(B, H, W):
    g = torch.Generator().manual_seed(H)
    x = (torch.randn(B, 3, H, W, generator=g) * 50).clamp(-123, 151)
    wt = torch.randn(64, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5
    rm, rv, bw, bb = _bn_params(64, g)
    ref = F.conv2d(x, wt, stride=2, padding=3)
    Ho, Wo = ref.shape[2:]
    nu = call("diga_stem_conv7x7s2", None, None, None, None, None, B, H, W, None)
    out = torch.zeros(B * Ho * Wo * 64, dtype=torch.bfloat16, device=DEV)
    slab = torch.full((nu * 2 * 64,), float("nan"), dtype=torch.float32, device=DEV)
    dv = [t.to(DEV) for t in (rm, rv, bw, bb)]
    xd, wd = x.to(DEV), wt.to(DEV)      # keep device copies alive until the kernels have run
    call("diga_stem_conv7x7s2", P(xd), P(wd), P(out), P(slab), P(dv[0]), B, H, W, stream())
    sc = torch.empty(64, dtype=torch.float32, device=DEV)
    sh = torch.empty(64, dtype=torch.float32, device=DEV)
    call("diga_bn_finalize", P(slab), nu // B, B, 1 if B > 1 else B, 0, 64, float(Ho * Wo), float((B - 1) * Ho * Wo), P(dv[0]),
         P(dv[1]), P(dv[2]), P(dv[3]), 0.8, 1e-5, P(sc), P(sh), stream())
    torch.cuda.synchronize()
    got = from_nhwc_bf16(out, (B, Ho, Wo, 64))
    assert rel_err(got, bf16_round(ref)) <= BF16_TOL
    rsc, rsh = _oracle_scale_shift(ref, rm, rv, bw, bb)
    assert rel_err(sc.cpu(), rsc) < 1e-4 and rel_err(sh.cpu(), rsh) < 1e-4
    # bn1 + relu + ceil-mode maxpool on the GPU's own raw output / affine
    from diga_amd.engine import pool_out
    Hp, Wp = pool_out(Ho), pool_out(Wo)
    pooled = torch.zeros(B * Hp * Wp * 64, dtype=torch.bfloat16, device=DEV)
    call("diga_bn_relu_maxpool", P(out), P(sc), P(sh), P(pooled), B, Ho, Wo, 64, Hp, Wp, 0, stream())
    torch.cuda.synchronize()
    y = torch.relu(got * sc.cpu()[None, :, None, None] + sh.cpu()[None, :, None, None])
    refp = F.max_pool2d(y, 3, 2, 1, ceil_mode=True)
    assert refp.shape[2:] == (Hp, Wp)
    assert rel_err(from_nhwc_bf16(pooled, (B, Hp, Wp, 64)), bf16_round(refp)) <= BF16_TOL
    # per-sample affines (stream mode)
    scp = torch.empty(B * 64, dtype=torch.float32, device=DEV)
    shp = torch.empty(B * 64, dtype=torch.float32, device=DEV)
    call("diga_bn_finalize", P(slab), nu // B, B, B, 1, 64, float(Ho * Wo), 0.0, P(dv[0]), P(dv[1]), P(dv[2]), P(dv[3]), 0.8, 1e-5,
         P(scp), P(shp), stream())
    call("diga_bn_relu_maxpool", P(out), P(scp), P(shp), P(pooled), B, Ho, Wo, 64, Hp, Wp, 1, stream())
    torch.cuda.synchronize()
    for b in range(B):
        rsc, rsh = _oracle_scale_shift(ref[b:b + 1], rm, rv, bw, bb)
        assert rel_err(scp.cpu()[b * 64:(b + 1) * 64], rsc) < 1e-4 and rel_err(shp.cpu()[b * 64:(b + 1) * 64], rsh) < 1e-4
    scc, shc = scp.cpu().reshape(B, 64, 1, 1), shp.cpu().reshape(B, 64, 1, 1)
    refp = F.max_pool2d(torch.relu(got * scc + shc), 3, 2, 1, ceil_mode=True)
    assert rel_err(from_nhwc_bf16(pooled, (B, Hp, Wp, 64)), bf16_round(refp)) <= BF16_TOL


@pytest.mark.parametrize("Cc,mode", [(64, 0), (256, 1), (2048, 2), (512, 2), (1024, 1)])
def test_bn_apply(Cc, mode):
    g = torch.Generator().manual_seed(Cc + mode)
    M = 1237
    raw = bf16_round(torch.randn(M, Cc, generator=g) * 2)
    res = bf16_round(torch.randn(M, Cc, generator=g))
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    rsc, rsh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    ref = raw * sc + sh
    if mode == 1:
        ref = ref + res
    if mode == 2:
        ref = ref + (res * rsc + rsh)
    ref = torch.relu(ref)
    out = torch.zeros(M * Cc, dtype=torch.bfloat16, device=DEV)
    rawd, resd, scd, shd, rscd, rshd = (t.to(DEV) for t in (raw.to(torch.bfloat16), res.to(torch.bfloat16), sc, sh, rsc, rsh))
    call("diga_bn_apply", P(rawd), P(scd), P(shd), P(resd) if mode else None, P(rscd) if mode == 2 else None,
         P(rshd) if mode == 2 else None, P(out), M, Cc, 1, 0, stream())
    torch.cuda.synchronize()
    assert rel_err(out.float().cpu().reshape(M, Cc), bf16_round(ref)) <= BF16_TOL


@pytest.mark.parametrize("Cc,mode,ns", [(64, 0, 3), (512, 2, 4), (1024, 1, 2)])
def test_bn_apply_per_sample_affine(Cc, mode, ns):
    g = torch.Generator().manual_seed(Cc + mode + ns)
    Ms = 419
    raw = bf16_round(torch.randn(ns, Ms, Cc, generator=g) * 2)
    res = bf16_round(torch.randn(ns, Ms, Cc, generator=g))
    sc, sh = torch.rand(ns, 1, Cc, generator=g) + 0.5, torch.randn(ns, 1, Cc, generator=g)
    rsc, rsh = torch.rand(ns, 1, Cc, generator=g) + 0.5, torch.randn(ns, 1, Cc, generator=g)
    ref = raw * sc + sh
    if mode == 1:
        ref = ref + res
    if mode == 2:
        ref = ref + (res * rsc + rsh)
    ref = torch.relu(ref)
    out = torch.zeros(ns * Ms * Cc, dtype=torch.bfloat16, device=DEV)
    rawd, resd, scd, shd, rscd, rshd = (t.contiguous().to(DEV) for t in (raw.to(torch.bfloat16), res.to(torch.bfloat16), sc, sh, rsc, rsh))
    call("diga_bn_apply", P(rawd), P(scd), P(shd), P(resd) if mode else None, P(rscd) if mode == 2 else None,
         P(rshd) if mode == 2 else None, P(out), ns * Ms, Cc, 1, ns, stream())
    torch.cuda.synchronize()
    assert rel_err(out.float().cpu().reshape(ns, Ms, Cc), bf16_round(ref)) <= BF16_TOL


@pytest.mark.parametrize("B", [1, 2, 3])
def test_standalone_dam_bn_matches_reference_fixture(gold_bn, B):
    """SIFABatchNorm2dTrainable.forward (fp32 NCHW) against the reference's own output (fn_bn.npz)."""
    gd = gold_bn
    x = torch.from_numpy(gd[f"B{B}_x"]).to(DEV)
    Bn, Cc, H, W = x.shape
    prm = [torch.from_numpy(gd[f"B{B}_{k}"]).to(DEV) for k in ("running_mean", "running_var", "weight", "bias")]
    slab = torch.empty(4 * Cc, dtype=torch.float32, device=DEV)
    sc = torch.empty(Cc, dtype=torch.float32, device=DEV)
    sh = torch.empty(Cc, dtype=torch.float32, device=DEV)
    y = torch.empty_like(x)
    split_b = 1 if B > 1 else B
    call("diga_bn_stats_nchw_f32", P(x), Bn, Cc, H * W, split_b, P(prm[0]), P(slab), stream())
    call("diga_bn_finalize", P(slab), 1, 2, 1, 0, Cc, float(split_b * H * W), float((B - split_b) * H * W), P(prm[0]), P(prm[1]), P(prm[2]),
         P(prm[3]), 0.8, 1e-5, P(sc), P(sh), stream())
    call("diga_bn_apply_nchw_f32", P(x), P(sc), P(sh), P(y), Bn, Cc, H * W, stream())
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), gd[f"B{B}_y"]) < 2e-6


def test_layout_roundtrip():
    g = torch.Generator().manual_seed(5)
    x = bf16_round(torch.randn(2, 70, 9, 13, generator=g))
    nhwc = torch.zeros(2 * 9 * 13 * 70, dtype=torch.bfloat16, device=DEV)
    xd = x.to(DEV)
    call("diga_nchw_f32_to_nhwc_bf16", P(xd), P(nhwc), 2, 9 * 13, 70, stream())
    back = torch.zeros(2, 70, 9, 13, dtype=torch.float32, device=DEV)
    call("diga_nhwc_bf16_to_nchw_f32", P(nhwc), P(back), 2, 9 * 13, 70, stream())
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), x)
    assert torch.equal(from_nhwc_bf16(nhwc, (2, 9, 13, 70)), x)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k3,dil,tile", [(1, 33, 65, 64, 256, False, 1, 0), (3, 23, 31, 128, 128, True, 2, 23),
                                                        (2, 65, 129, 256, 1024, False, 1, 67), (4, 40, 50, 64, 256, False, 1, 48),
                                                        (3, 65, 129, 256, 256, True, 2, 58), (2, 17, 19, 64, 64, True, 1, 24)])
def test_conv_with_fused_dam_finalize_is_bit_identical(B, H, W, Cin, Cout, k3, dil, tile):
    """diga_conv_igemm_bf16_dam (finalize folded into the conv launch: the last tile of each (sample, N tile) group reduces
    the slab) == diga_conv_igemm_bf16 + diga_bn_finalize(per_sample = 1): output, slab, scale and shift bit for bit, on
    repeated launches (the ticket words return to zero), for several tile shapes."""
    if tile and Cout % {23: 64, 24: 128, 67: 256, 48: 256, 58: 256}[tile]:
        pytest.skip("tile does not divide Cout")
    g = torch.Generator().manual_seed(H * 7 + Cout + tile)
    x = bf16_round(torch.randn(B, Cin, H, W, generator=g).relu() + 0.1 * torch.randn(B, Cin, H, W, generator=g))
    ks = 3 if k3 else 1
    wt = bf16_round(torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5)
    rm, rv, w, b = (t.to(DEV) for t in _bn_params(Cout, g))
    d = conv_desc(B, H, W, Cin, Cout, k3=k3, dil=dil, tile=tile)
    xin, wp = to_nhwc_bf16(x), pack_conv_weight(wt).to(DEV)
    M = B * d.Ho * d.Wo
    units = call("diga_conv_stat_units", d)
    ups = units // B
    lam, eps = float(np.float32(0.8)), 1e-5

    def buffers():
        return (torch.zeros(M * Cout, dtype=torch.bfloat16, device=DEV), torch.zeros(units * 2 * Cout, dtype=torch.float32, device=DEV),
                torch.zeros(B * Cout, dtype=torch.float32, device=DEV), torch.zeros(B * Cout, dtype=torch.float32, device=DEV))
    out0, slab0, sc0, sh0 = buffers()
    call("diga_conv_igemm_bf16", d, P(xin), P(wp), P(out0), P(slab0), P(rm), stream())
    call("diga_bn_finalize", P(slab0), ups, B, B, 1, Cout, float(d.Ho * d.Wo), 0.0, P(rm), P(rv), P(w), P(b), lam, eps, P(sc0), P(sh0), stream())
    ticket = torch.zeros(call("diga_conv_dam_ticket_words", d), dtype=torch.int32, device=DEV)
    for rep in range(3):
        out1, slab1, sc1, sh1 = buffers()
        sc1.fill_(float("nan")), sh1.fill_(float("nan"))
        call("diga_conv_igemm_bf16_dam", d, P(xin), P(wp), P(out1), P(slab1), P(rm), P(rv), P(w), P(b), lam, eps, P(sc1), P(sh1),
             P(ticket), stream())
        torch.cuda.synchronize()
        assert torch.equal(out1.view(torch.int16), out0.view(torch.int16)) and torch.equal(slab1, slab0), rep
        assert torch.equal(sc1, sc0) and torch.equal(sh1, sh0), (rep, float((sc1 - sc0).abs().max()))
        assert int(ticket.abs().sum()) == 0, rep


@pytest.mark.parametrize("B,H,W,Cin,Cout,k3,dil,res,per_sample,tile", [
    (1, 33, 65, 64, 256, False, 1, False, 0, 0), (2, 23, 31, 128, 128, True, 2, False, 2, 21), (3, 20, 27, 256, 64, False, 1, True, 3, 23),
    (2, 65, 129, 256, 1024, False, 1, False, 2, 77), (2, 65, 129, 1024, 256, False, 1, True, 2, 76), (1, 65, 129, 512, 128, False, 1, True, 0, 61),
    (2, 40, 50, 256, 256, True, 2, False, 2, 28), (1, 19, 23, 64, 128, True, 4, False, 0, 64), (2, 31, 45, 512, 256, False, 1, True, 0, 27),
    # the residual form with two K steps of operand registers in flight (M tiles of 64 / 128 rows, 1x1): ragged samples, two N
    # tiles (only the first writes the block input back), per-sample affines
    (3, 65, 129, 1024, 256, False, 1, True, 3, 27), (2, 33, 41, 512, 512, False, 1, True, 2, 27), (2, 29, 37, 256, 256, False, 1, True, 0, 24),
    (5, 65, 129, 2048, 512, False, 1, True, 5, 27),
    # tile 93: the A-stationary conv3 (transform once per element, weights streamed past the resident unit): ragged last units,
    # fewer units than waves, several unit groups per block (more than 256 groups), per-sample affines
    (1, 33, 65, 64, 256, False, 1, False, 0, 93), (3, 13, 21, 128, 512, False, 1, False, 3, 93), (2, 65, 129, 256, 1024, False, 1, False, 2, 93),
    (2, 7, 5, 256, 64, False, 1, False, 0, 93), (5, 65, 129, 256, 1024, False, 1, False, 5, 93), (3, 129, 257, 64, 256, False, 1, False, 3, 93)])
def test_conv_with_fused_input_apply_is_bit_identical(B, H, W, Cin, Cout, k3, dil, res, per_sample, tile):
    """diga_conv_igemm_bf16_fused_in (DAM apply + ReLU [+ residual] on the operand path) == diga_bn_apply followed by
    diga_conv_igemm_bf16: raw output and statistics slab bit for bit; in the residual form the activation written on the side
    (the next block's input) equals diga_bn_apply's output bit for bit.  3x3 halos stay zero padding."""
    g = torch.Generator().manual_seed(H * 3 + Cin + Cout + tile)
    raw = bf16_round(torch.randn(B, Cin, H, W, generator=g) * 1.5)
    resid = bf16_round(torch.randn(B, Cin, H, W, generator=g).relu())
    ks = 3 if k3 else 1
    wt = bf16_round(torch.randn(Cout, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5)
    ns = per_sample if per_sample else 1
    sc = (torch.rand(ns, Cin, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(ns, Cin, generator=g) * 0.3).to(DEV)
    shift = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    d = conv_desc(B, H, W, Cin, Cout, k3=k3, dil=dil, tile=tile)
    rin, rres, wp = to_nhwc_bf16(raw), to_nhwc_bf16(resid), pack_conv_weight(wt).to(DEV)
    M = B * H * W
    units = call("diga_conv_stat_units", d)
    act = torch.zeros(M * Cin, dtype=torch.bfloat16, device=DEV)
    call("diga_bn_apply", P(rin), P(sc), P(sh), P(rres) if res else None, None, None, P(act), M, Cin, 1, per_sample, stream())
    d0 = conv_desc(B, H, W, Cin, Cout, k3=k3, dil=dil, tile=0)
    out0 = torch.zeros(M * Cout, dtype=torch.bfloat16, device=DEV)
    slab0 = torch.zeros(units * 2 * Cout, dtype=torch.float32, device=DEV)
    call("diga_conv_igemm_bf16", d0, P(act), P(wp), P(out0), P(slab0), P(shift), stream())
    for rep in range(3):       # (the kernel counts its own memory operations: repeated runs screen for a wait that is too lenient)
        out1 = torch.full((M * Cout,), float("nan"), dtype=torch.bfloat16, device=DEV)
        slab1 = torch.full((units * 2 * Cout,), float("nan"), dtype=torch.float32, device=DEV)
        act1 = torch.full((M * Cin,), float("nan"), dtype=torch.bfloat16, device=DEV) if res else None
        call("diga_conv_igemm_bf16_fused_in", d, P(rin), P(sc), P(sh), P(rres) if res else None, P(act1), per_sample, P(wp), P(out1), P(slab1),
             P(shift), stream())
        torch.cuda.synchronize()
        assert torch.equal(out1.view(torch.int16), out0.view(torch.int16)), (rep, float((out1.float() - out0.float()).abs().max()))
        assert torch.equal(slab1, slab0), rep
        if res:
            assert torch.equal(act1.view(torch.int16), act.view(torch.int16)), rep
