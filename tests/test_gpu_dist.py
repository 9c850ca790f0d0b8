"""-m gpu: the frame-sharded path on ONE GPU: two ranks share cuda:0 and exchange the BN sums and
prototype sums over gloo (RCCL refuses two ranks on one device; the collectives are the same torch.distributed
calls).  The sharded global batch of 2 must reproduce the single-process B = 2 result, i.e. the reference's
half_first batch semantics (ttda:313-320), and the SAM state must be identical on both ranks."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CFG = dict(bn_lambda=0.8, proto_lambda=0.8, fusion_lambda=0.8, confidence_threshold=0.9, proto_rho=0.1)
H, W, NF = 65, 129, 2


def _worker(rank, world, port, q):
    import traceback
    os.makedirs("gpurun_out", exist_ok=True)
    log = open(f"gpurun_out/dist_rank{rank}.log", "w")
    try:
        _worker_body(rank, world, port, q, log)
    except Exception:
        traceback.print_exc(file=log)
        log.flush()
        q.put((rank, None))
        raise


def _worker_body(rank, world, port, q, log):
    import torch.distributed as dist
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 1, H, W, group=dist.group.WORLD)
    print("engine built", file=log, flush=True)
    out = []
    for t in range(NF):
        x, y = synth.synth_frame(t, world, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x[rank:rank + 1]).cuda(), torch.from_numpy(y[rank:rank + 1]).cuda(), CFG)
        torch.cuda.synchronize()
        print("frame", t, "done", file=log, flush=True)
        out.append(dict(probs=probs.cpu().numpy(), logits=eng.logits.cpu().numpy(), scale=eng.scale.cpu().numpy(),
                        run=eng.run_proto.cpu().numpy(), rflag=eng.run_flag.cpu().numpy()))
    dist.barrier()
    q.put((rank, out))
    try:
        dist.destroy_process_group()
    except Exception:
        pass


def test_two_ranks_reproduce_batch2_semantics():
    import torch.multiprocessing as mp
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in procs)
    assert all(v is not None for v in res.values()), "a rank failed: see gpurun_out/dist_rank*.log"
    for p in procs:
        p.join(timeout=60)
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 2, H, W)
    for t in range(NF):
        x, y = synth.synth_frame(t, 2, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), CFG)
        torch.cuda.synchronize()
        ref_logits, ref_probs = eng.logits.cpu().numpy(), probs.cpu().numpy()
        nbn1 = 64        # the first BN layer's scale is untouched by any upstream rounding flip: tight check of the protocol
        for r in range(2):
            got = res[r][t]
            assert np.abs(got["scale"][:nbn1] - eng.scale.cpu().numpy()[:nbn1]).max() <= 1e-6 * np.abs(eng.scale.cpu().numpy()[:nbn1]).max()
            rng = ref_logits.max() - ref_logits.min()
            assert np.abs(got["logits"][0] - ref_logits[r]).max() / rng < 0.02     # same bf16 floor as the e2e tests
            # probabilities: single pixels can move a lot when one pseudo label flips (bf16_floor.json shows max-abs up to
            # 0.8 for the oracle's own emulation), so bound the mean and the arg-max agreement instead of the max
            assert np.nanmean(np.abs(got["probs"][0] - ref_probs[r])) < 5e-3
            assert (got["probs"][0].argmax(0) == ref_probs[r].argmax(0)).mean() > 0.9
            assert np.array_equal(got["rflag"], eng.run_flag.cpu().numpy())
        # replicated SAM state is bitwise identical across ranks
        assert np.array_equal(res[0][t]["run"], res[1][t]["run"])
        # vs the single-process batch: rows of classes with only a few (pseudo-labelled) pixels move a lot when one label
        # flips through bf16 rounding, so require 80 % of the existing class rows to agree within 5 %
        ref_run = eng.run_proto.cpu().numpy()
        rows = np.nonzero(eng.run_flag.cpu().numpy())[0]
        rel = np.array([np.abs(res[0][t]["run"][c] - ref_run[c]).max() / max(np.abs(ref_run[c]).max(), 1e-6) for c in rows])
        assert (rel < 0.05).mean() >= 0.8, rel
