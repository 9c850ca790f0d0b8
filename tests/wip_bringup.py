"""Bring-up diagnostics (not collected by pytest): prints per-stage errors of the HIP path vs the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, torch.nn.functional as F
from conftest import load_golden
from gpu_util import from_nhwc_bf16, bf16_round, rel_err
from diga_amd import synth
from diga_amd.engine import Engine, NetWeights, LAYERS, STRIDES, DILATIONS
from oracle import diga_oracle as O

CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)
dev = torch.device("cuda:0")
pn = synth.synth_params(0)
W = NetWeights(pn, dev)
pt = O.to_torch(pn)
g = load_golden("net_small.npz")

def stage_isolated(B, H, Wd):
    eng = Engine(W, B, H, Wd)
    eng.taps = {}
    x, y = synth.synth_frame(0, B, H, Wd)
    eng.forward_feat(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    taps = {k: from_nhwc_bf16(t, shp) for k, (t, shp) in eng.taps.items()}
    worst = 0
    def chk(name, got, ref):
        nonlocal worst
        e = rel_err(got, bf16_round(ref)); worst = max(worst, e)
        if e > 2**-7: print("  !!", name, e)
    ref = F.conv2d(torch.from_numpy(x), pt["conv1.weight"], stride=2, padding=3)
    chk("conv1", taps["conv1"], ref)
    cur = taps["maxpool"]
    for li, (nb, st, dil) in enumerate(zip(LAYERS, STRIDES, DILATIONS), start=1):
        for b in range(nb):
            pre = f"layer{li}.{b}"; s = st if b == 0 else 1
            wq = lambda k: bf16_round(pt[pre + k])
            chk(pre + ".conv1", taps[pre + ".conv1"], F.conv2d(cur, wq(".conv1.weight"), stride=s))
            chk(pre + ".conv2", taps[pre + ".conv2"], F.conv2d(taps[pre + ".act1"], wq(".conv2.weight"), padding=dil, dilation=dil))
            chk(pre + ".conv3", taps[pre + ".conv3"], F.conv2d(taps[pre + ".act2"], wq(".conv3.weight")))
            # BN stages, fed the GPU's raw outputs
            a1 = torch.relu(O._bn(pt, pre + ".bn1", taps[pre + ".conv1"], 0.8)); chk(pre + ".act1", taps[pre + ".act1"], a1)
            a2 = torch.relu(O._bn(pt, pre + ".bn2", taps[pre + ".conv2"], 0.8)); chk(pre + ".act2", taps[pre + ".act2"], a2)
            o3 = O._bn(pt, pre + ".bn3", taps[pre + ".conv3"], 0.8)
            if b == 0:
                chk(pre + ".ds", taps[pre + ".downsample.0"], F.conv2d(cur, wq(".downsample.0.weight"), stride=s))
                res = O._bn(pt, pre + ".downsample.1", taps[pre + ".downsample.0"], 0.8)
            else:
                res = cur
            chk(pre, taps[pre], torch.relu(o3 + res))
            cur = taps[pre]
    lg = O.head_logits({k: (bf16_round(v) if k.endswith("weight") else v) for k, v in pt.items() if k.startswith("layer6")}, cur)
    print(f"stage-isolated B={B} {H}x{Wd}: worst bf16-stage rel err {worst:.3e} (1 ulp = {2**-7:.3e}); logits rel err {rel_err(eng.logits.cpu(), lg):.3e}")
    return eng

def end_to_end(tag):
    B, H, Wd = (int(v) for v in g[f"{tag}_shape"])
    eng = Engine(W, B, H, Wd)
    st = O.SamState()
    for t in range(3):
        x, y = synth.synth_frame(t, B, H, Wd)
        info = {}
        with torch.no_grad():
            op = O.forward_and_adapt(pt, st, torch.from_numpy(x), torch.from_numpy(y), CFG, info, bf16=True)
            opred = O.upsample_probs(op, y.shape[-2:]).argmax(1).numpy()
        yd = torch.from_numpy(y).to(dev)
        probs = eng.forward_and_adapt(torch.from_numpy(x).to(dev), yd, CFG)
        eng.tail(yd)
        torch.cuda.synchronize()
        lg = eng.logits.cpu().numpy(); P = eng.P
        pl = eng.labels[:P].cpu().numpy().reshape(B, eng.h, eng.wd)
        print(f"  {tag} t={t} vs bf16-oracle: logits relerr {rel_err(lg, info['logits']):.3e} feat relerr {rel_err(eng.feat_nchw().cpu(), info['feature']):.3e} pseudo agree {(pl == info['pseudo'].numpy()).mean():.4f}"
              f" probs maxabs {np.nanmax(np.abs(probs.cpu().numpy() - op.numpy())):.3e} pred agree {(eng.pred.cpu().numpy() == opred).mean():.5f} run relerr {rel_err(eng.run_proto.cpu(), st.run_proto):.3e}")
        print(f"  {tag} t={t} vs fp32 ref  : logits relerr {rel_err(lg, g[f'{tag}_{t}_logits']):.3e}  pseudo agree {(pl == g[f'{tag}_{t}_pseudo']).mean():.4f}"
              f"  probs maxabs {np.nanmax(np.abs(probs.cpu().numpy() - g[f'{tag}_{t}_probs'])):.3e}  pred agree {(eng.pred.cpu().numpy() == g[f'{tag}_{t}_pred']).mean():.4f}"
              f"  run relerr {rel_err(eng.run_proto.cpu(), g[f'{tag}_{t}_run']):.3e} rflag eq {np.array_equal(eng.run_flag.cpu().numpy(), g[f'{tag}_{t}_rflag'])}"
              f"  hist L1 {np.abs(eng.hist.cpu().numpy().reshape(19,19) - g[f'{tag}_{t}_hist']).sum():.0f}/{g[f'{tag}_{t}_hist'].sum():.0f}")
    h = eng.hist.cpu().reshape(19, 19)
    print(f"  {tag} mIoU gpu {[round(float(O.miou(h, k)), 4) for k in ('default', '16', '13')]} ref {g[f'{tag}_miou'].round(4).tolist()}")

if __name__ == "__main__":
    t0 = time.time()
    stage_isolated(1, 65, 129)
    stage_isolated(2, 65, 129)
    for tag in ("s1", "s2", "s3", "m1"):
        end_to_end(tag)
    print("elapsed", time.time() - t0)
