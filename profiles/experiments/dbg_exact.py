import sys, functools
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from diga_amd import synth, ttda_module as T
from conftest import load_golden
CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)
g = load_golden("net_full.npz")
m = T.DIGA(net=functools.partial(T.DeepLabv2AdaptSeg, multi_level=True, restore_from=None),
           dataset_info=dict(num_classes=19, image_size=(321, 321), test_list=[0], class_names=[str(i) for i in range(19)]), cfg=CFG, precision=sys.argv[1])
m.net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(0).items()}, strict=False)
m._configure_bn_running_stats()
for t in range(2):
    x, y = synth.synth_frame(t, 1, 512, 1024)
    m.test_step((torch.from_numpy(x), torch.from_numpy(y), None, None), t)
    e = m._last_engine; torch.cuda.synchronize()
    lg, rl = e.logits.cpu().numpy(), g[f"{t}_logits"]
    print("frame", t, "logit err", np.abs(lg-rl).max()/(rl.max()-rl.min()))
    pr, rp = e.probs.cpu().numpy(), g[f"{t}_probs"]
    d = np.abs(pr-rp); print(" probs abs err max", np.nanmax(d), "p99.9", np.nanpercentile(d, 99.9), "nan mismatch", (np.isnan(pr)!=np.isnan(rp)).sum())
    rel = d/np.maximum(rp, 1e-30); big = rp > 1e-3
    print(" probs rel err (p>1e-3) max", np.nanmax(rel[big]))
    mism = e.pred.cpu().numpy() != g[f"{t}_pred"]
    b = g[f"{t}_pred_gap_bucket"]
    print(" pred mismatches by ref gap bucket:", {int(k): int((mism & (b == k)).sum()) for k in np.unique(b[mism])}, "pixels per bucket", {int(k): int((b == k).sum()) for k in np.unique(b)})
    lm = torch.from_numpy(g[f"{t}_lowmass"].astype(np.float32))[:, None]
    lmf = (torch.nn.functional.interpolate(lm, size=(512, 1024), mode="bilinear", align_corners=False)[:, 0] > 0).numpy()
    out = mism & ~lmf & (b < 4)
    ys, xs = np.nonzero(out[0])
    print(" lowmass px", int(g[f"{t}_lowmass"].sum()), "mismatches outside band:", [(int(y), int(x), int(b[0, y, x]), int(e.pred[0, y, x]), int(g[f"{t}_pred"][0, y, x])) for y, x in zip(ys, xs)])
    for y, x in list(zip(ys, xs))[:4]:
        fy, fx = y // 8, x // 8
        print("   feat px", fy, fx, "lowmass nbhd", g[f"{t}_lowmass"][0, max(0, fy-2):fy+3, max(0, fx-2):fx+3].astype(int).tolist())
        print("   probs gpu", np.round(pr[0, :, fy, fx], 5).tolist()); print("   probs ref", np.round(rp[0, :, fy, fx], 5).tolist())
    pl = e.labels[:e.P].cpu().numpy().reshape(g[f"{t}_pseudo"].shape); rpl = g[f"{t}_pseudo"]
    mm = np.nonzero(pl != rpl)
    print(" pseudo mismatches:", [(int(a), int(b_), int(pl[0, a, b_]), int(rpl[0, a, b_]), float(g[f"{t}_pl_pmax"][0, a, b_])) for a, b_ in zip(mm[1], mm[2])],
          "class-3 confident px ref", int((rpl == 3).sum()), "gpu", int((pl == 3).sum()))
    # the oracle's own sensitivity: recompute the reference's probabilities from ITS features in float64 -> how far is the fp32 reference from exact arithmetic?
    rr, rg = m.classifier_running_proto.cpu().numpy(), g[f"{t}_run"]
    print(" run proto rel err", np.abs(rr-rg).max()/np.abs(rg).max())
