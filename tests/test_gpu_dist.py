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


def _worker(rank, world, port, q, k):
    import traceback
    os.makedirs("gpurun_out", exist_ok=True)
    log = open(f"gpurun_out/dist_rank{rank}.log", "w")
    try:
        (_worker_oneshot if k == 100 else _worker_four_ranks if k == 104 else _worker_oneshot_stream if k == 101 else _worker_late_peer if k == 102 else
         _worker_missing_peer if k == 103 else _worker_b8 if k == 8 else _worker_windows if k < 0 else _worker_stream if k > 1 else _worker_body)(rank, world, port, q, log, k)
    except Exception:
        traceback.print_exc(file=log)
        log.flush()
        q.put((rank, None))
        raise


def _worker_body(rank, world, port, q, log, k):
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


def _worker_b8(rank, world, port, q, log, k):
    """BASELINE configs[2]'s global batch of 8 on 2 ranks: 4 frames per rank (group A = global sample 0, on rank 0)."""
    import torch.distributed as dist
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    per = 8 // world
    eng = Engine(w, per, H, W, group=dist.group.WORLD)
    out = []
    for t in range(2):
        x, y = synth.synth_frame(200 + t, 8, H, W)
        sl = slice(rank * per, (rank + 1) * per)
        probs = eng.forward_and_adapt(torch.from_numpy(x[sl]).cuda(), torch.from_numpy(y[sl]).cuda(), CFG)
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


def _worker_oneshot(rank, world, port, q, log, k):
    """The one-launch direct-peer-write all-reduce (csrc/xgmi.hip) between two processes that share cuda:0 (their buffers are
    mapped into each other with hipIpc exactly as two GPUs' would be), then the sharded engine on top of it."""
    import torch.distributed as dist
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    from diga_amd.xgmi import OneShotComm
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = OneShotComm(dist.group.WORLD, max_elems=1 << 17)
    print("communicator connected", file=log, flush=True)
    ok = True
    g = torch.Generator().manual_seed(123)          # the same stream on both ranks: each knows the other's contribution
    for rep in range(40):                            # back to back: parities alternate, epochs advance per block
        for dt, n in ((torch.float64, 4 * 64), (torch.float64, 4 * 2048), (torch.float32, 2 * 19 * 2048), (torch.int32, 38), (torch.float64, 100003)):
            if dt == torch.int32:
                parts = [torch.randint(0, 10000, (n,), generator=g, dtype=dt) for _ in range(world)]
            else:
                parts = [torch.randn(n, generator=g, dtype=dt) for _ in range(world)]
            t = parts[rank].cuda()
            comm.all_reduce(t)
            want = parts[0].clone()
            for p in parts[1:]:
                want += p                            # rank order: bit-exact
            good = torch.equal(t.cpu(), want)
            if not good:
                print("MISMATCH", rep, dt, n, float((t.cpu().double() - want.double()).abs().max()), file=log, flush=True)
            ok = ok and good
    comm.check()
    print("raw all-reduces", "ok" if ok else "FAILED", file=log, flush=True)
    # the sharded engine through the one-shot transport
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 1, H, W, group=dist.group.WORLD, oneshot=True)
    out = []
    for t in range(NF):
        x, y = synth.synth_frame(t, world, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x[rank:rank + 1]).cuda(), torch.from_numpy(y[rank:rank + 1]).cuda(), CFG)
        torch.cuda.synchronize()
        out.append(dict(probs=probs.cpu().numpy(), logits=eng.logits.cpu().numpy(), scale=eng.scale.cpu().numpy(),
                        run=eng.run_proto.cpu().numpy(), rflag=eng.run_flag.cpu().numpy()))
    # and one step replayed from a HIP graph (the exchange is an ordinary kernel launch: no host round trip)
    x, y = synth.synth_frame(NF, world, H, W)
    xd, yd = torch.from_numpy(x[rank:rank + 1]).cuda(), torch.from_numpy(y[rank:rank + 1]).cuda()
    state = [t_.clone() for t_ in (eng.run_proto, eng.run_flag, eng.state_flag)]
    eager = eng.forward_and_adapt(xd, yd, CFG).clone()
    torch.cuda.synchronize()
    for dst, src in zip((eng.run_proto, eng.run_flag, eng.state_flag), state):
        dst.copy_(src)
    dist.barrier()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        eng.forward_and_adapt(xd, yd, CFG)
    for dst, src in zip((eng.run_proto, eng.run_flag, eng.state_flag), state):
        dst.copy_(src)
    dist.barrier()
    gr.replay()
    torch.cuda.synchronize()
    graph_ok = bool(torch.equal(torch.nan_to_num(eng.probs, nan=-1.0), torch.nan_to_num(eager, nan=-1.0)))
    eng.oneshot.check()
    print("graph replay", "identical" if graph_ok else "DIFFERENT", file=log, flush=True)
    dist.barrier()
    q.put((rank, dict(ok=ok and graph_ok, frames=out)))
    try:
        dist.destroy_process_group()
    except Exception:
        pass


def _worker_oneshot_stream(rank, world, port, q, log, k):
    """3 global batches in flight through the FUSED exchange (diga_bn_exchange_finalize, per_sample sets), then the same window
    of 15 batches in flight at the bench's default size class (constructs the B = 15 communicator: ADVICE r2)."""
    import torch.distributed as dist
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    K = 3
    eng = Engine(w, K, H, W, group=dist.group.WORLD, stream=True, oneshot=True)
    xy = [synth.synth_frame(t, world, H, W) for t in range(K)]
    x = torch.from_numpy(np.concatenate([a[rank:rank + 1] for a, _ in xy])).cuda()
    y = torch.from_numpy(np.concatenate([b[rank:rank + 1] for _, b in xy])).cuda()
    probs = eng.forward_and_adapt(x, y, CFG)
    torch.cuda.synchronize()
    eng.check_exchange()
    launches = eng.exchange_launches
    print("window done,", launches, "exchange launches", file=log, flush=True)
    out = dict(probs=probs.cpu().numpy(), logits=eng.logits.cpu().numpy(), run=eng.run_proto.cpu().numpy(), rflag=eng.run_flag.cpu().numpy(),
               launches=launches)
    eng.close()
    # 15 batches in flight: the prototype sums (15 x 2 x 19 x 2048 fp32) must fit the communicator the engine sizes for itself
    big = Engine(w, 15, H, W, group=dist.group.WORLD, stream=True, oneshot=True)
    xb = torch.from_numpy(np.concatenate([synth.synth_frame(t, world, H, W)[0][rank:rank + 1] for t in range(15)])).cuda()
    yb = torch.from_numpy(np.concatenate([synth.synth_frame(t, world, H, W)[1][rank:rank + 1] for t in range(15)])).cuda()
    big.forward_and_adapt(xb, yb, CFG)
    torch.cuda.synchronize()
    big.check_exchange()
    out["big_ok"] = bool(big.oneshot.supports(big.pool_sums) and big.exchange_launches == 106 and torch.isfinite(big.scale).all())
    out["big_logits3"] = big.logits[:3].cpu().numpy()
    big.close()
    dist.barrier()
    q.put((rank, out))
    try:
        dist.destroy_process_group()
    except Exception:
        pass


def _worker_four_ranks(rank, world, port, q, log, k):
    """Global batch 4 on FOUR ranks through the fused exchange (rank-order sums over more than two contributions)."""
    import torch.distributed as dist
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 1, H, W, group=dist.group.WORLD, oneshot=True)
    out = []
    for t in range(NF):
        x, y = synth.synth_frame(t, world, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x[rank:rank + 1]).cuda(), torch.from_numpy(y[rank:rank + 1]).cuda(), CFG)
        torch.cuda.synchronize()
        eng.check_exchange()
        out.append(dict(probs=probs.cpu().numpy(), logits=eng.logits.cpu().numpy(), scale=eng.scale.cpu().numpy(), shift=eng.shift.cpu().numpy(),
                        run=eng.run_proto.cpu().numpy(), rflag=eng.run_flag.cpu().numpy(), launches=eng.exchange_launches))
    dist.barrier()
    eng.close()
    q.put((rank, out))
    try:
        dist.destroy_process_group()
    except Exception:
        pass


def _worker_late_peer(rank, world, port, q, log, k):
    """A peer that arrives 2 s late is waited for (the wait is bounded by wall-clock time, 10 s here), not an error."""
    import time

    import torch.distributed as dist
    from diga_amd.xgmi import OneShotComm
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = OneShotComm(dist.group.WORLD, max_elems=1 << 14, timeout_ms=10000)
    t = torch.full((4096,), float(rank + 1), dtype=torch.float64, device="cuda")
    if rank == 1:
        time.sleep(2.0)
    t0 = time.perf_counter()
    comm.all_reduce(t)
    torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    comm.check()
    ok = bool((t == 3.0).all())
    print("late peer: waited", round(waited, 2), "s, sum", "ok" if ok else "WRONG", file=log, flush=True)
    dist.barrier()
    comm.close()
    q.put((rank, dict(ok=ok, waited=waited)))
    try:
        dist.destroy_process_group()
    except Exception:
        pass


def _worker_missing_peer(rank, world, port, q, log, k):
    """A peer that never arrives: the launch ends after the time-out, the payload is poisoned (NaN / INT32_MIN / NaN scale and
    shift) and check() raises -- no hang, no plausible-looking un-reduced numbers; the exchanges AFTER the first expired wait give
    up at once (four launches cost one 2 s time-out, not four: 8 s)."""
    import time

    import torch.distributed as dist
    from diga_amd._lib import DigaError
    from diga_amd.xgmi import OneShotComm
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = OneShotComm(dist.group.WORLD, max_elems=1 << 14, timeout_ms=2000)
    res = dict(ok=True)
    if rank == 0:
        f64 = torch.ones(5000, dtype=torch.float64, device="cuda")
        f32 = torch.ones(64, dtype=torch.float32, device="cuda")
        i32 = torch.ones(38, dtype=torch.int32, device="cuda")
        t0 = time.perf_counter()
        comm.all_reduce(f64)
        comm.all_reduce(f32)
        comm.all_reduce(i32)
        Cc, ups = 128, 5
        slab = torch.randn(ups * 2 * Cc, device="cuda")
        one = torch.ones(Cc, device="cuda")
        sc, sh = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        comm.bn_exchange_finalize(slab, ups, 1, 1, 0, Cc, 160.0, 160.0, one, one, one, one, 0.8, 1e-5, sc, sh)
        torch.cuda.synchronize()
        took = time.perf_counter() - t0
        raised = False
        try:
            comm.check()
        except DigaError:
            raised = True
        res = dict(ok=bool(torch.isnan(f64).all() and torch.isnan(f32).all() and (i32 == -2 ** 31).all() and torch.isnan(sc).all()
                           and torch.isnan(sh).all() and raised and 1.6 < took < 6.5), took=took, raised=raised)
        print("missing peer:", res, file=log, flush=True)
    dist.barrier()          # rank 1 only ever comes here
    q.put((rank, res))
    try:
        dist.destroy_process_group()
    except Exception:
        pass


def _worker_stream(rank, world, port, q, log, k):
    """k global batches in flight: this rank holds its frame of each of them (Engine(stream=True, group=...))."""
    import torch.distributed as dist
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, k, H, W, group=dist.group.WORLD, stream=True)
    xy = [synth.synth_frame(t, world, H, W) for t in range(k)]
    x = torch.from_numpy(np.concatenate([a[rank:rank + 1] for a, _ in xy])).cuda()
    y = torch.from_numpy(np.concatenate([b[rank:rank + 1] for _, b in xy])).cuda()
    probs = eng.forward_and_adapt(x, y, CFG)
    torch.cuda.synchronize()
    print("window done", file=log, flush=True)
    out = dict(probs=probs.cpu().numpy(), logits=eng.logits.cpu().numpy(), run=eng.run_proto.cpu().numpy(), rflag=eng.run_flag.cpu().numpy())
    dist.barrier()
    q.put((rank, out))
    try:
        dist.destroy_process_group()
    except Exception:
        pass


def _worker_windows(rank, world, port, q, log, k):
    """Two windows (each on its own HIP stream and its own process group) of k global batches in flight."""
    import torch.distributed as dist
    from diga_amd import synth
    from diga_amd.engine import NetWeights
    from diga_amd.pipeline import StreamWindows
    k = -k
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    groups = [dist.group.WORLD, dist.new_group(backend="gloo")]
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    sw = StreamWindows(w, k, 2, H, W, groups=groups)
    wins = []
    for i in range(2):
        xy = [synth.synth_frame(i * k + t, world, H, W) for t in range(k)]
        wins.append((torch.from_numpy(np.concatenate([a[rank:rank + 1] for a, _ in xy])).cuda(),
                     torch.from_numpy(np.concatenate([b[rank:rank + 1] for _, b in xy])).cuda()))
    probs = sw.step(wins, CFG)
    torch.cuda.synchronize()
    print("step done", file=log, flush=True)
    out = dict(probs=np.concatenate([p.cpu().numpy() for p in probs]), logits=np.concatenate([e.logits.cpu().numpy() for e in sw.engines]),
               run=sw.engines[0].run_proto.cpu().numpy(), rflag=sw.engines[0].run_flag.cpu().numpy(), hist=sw.engines[0].hist.cpu().numpy())
    dist.barrier()
    q.put((rank, out))
    try:
        dist.destroy_process_group()
    except Exception:
        pass


def _run_ranks(k, world=2):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() + 7 * k) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, k)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in procs)
    assert all(v is not None for v in res.values()), "a rank failed: see gpurun_out/dist_rank*.log"
    for p in procs:
        p.join(timeout=60)
    return res


def test_two_ranks_stream_of_global_batches():
    """3 global batches of 2 in flight on 2 ranks (one BN exchange per layer for all of them, one prototype exchange) ==
    the single-process batch-2 engine run batch after batch: trunk bit-identical, SAM state up to fp32 rounding."""
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    K = 3
    res = _run_ranks(K)
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 2, H, W)
    for t in range(K):
        x, y = synth.synth_frame(t, 2, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), CFG)
        torch.cuda.synchronize()
        for r in range(2):
            assert np.array_equal(res[r]["logits"][t], eng.logits.cpu().numpy()[r]), (t, r)
            a, b = res[r]["probs"][t], probs.cpu().numpy()[r]
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b)) < 1e-4, (t, r)
    ref_run = eng.run_proto.cpu().numpy()
    assert np.array_equal(res[0]["run"], res[1]["run"]) and np.array_equal(res[0]["rflag"], eng.run_flag.cpu().numpy())
    assert np.abs(res[0]["run"] - ref_run).max() <= 1e-5 * np.abs(ref_run).max()


def test_two_ranks_two_windows_of_global_batches():
    """2 windows x 2 global batches of 2 on 2 ranks, each window with its own process group and HIP stream == the
    single-process batch-2 engine run batch after batch (trunk bit-identical, SAM state up to fp32 rounding)."""
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    K = 2
    res = _run_ranks(-K)
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 2, H, W)
    for t in range(2 * K):
        x, y = synth.synth_frame(t, 2, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), CFG)
        torch.cuda.synchronize()
        for r in range(2):
            assert np.array_equal(res[r]["logits"][t], eng.logits.cpu().numpy()[r]), (t, r)
            a, b = res[r]["probs"][t], probs.cpu().numpy()[r]
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b)) < 1e-4, (t, r)
    ref_run = eng.run_proto.cpu().numpy()
    assert np.array_equal(res[0]["run"], res[1]["run"]) and np.array_equal(res[0]["rflag"], eng.run_flag.cpu().numpy())
    assert np.abs(res[0]["run"] - ref_run).max() <= 1e-5 * np.abs(ref_run).max()


def test_two_ranks_reproduce_batch2_semantics():
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    res = _run_ranks(1)
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 2, H, W)
    for t in range(NF):
        x, y = synth.synth_frame(t, 2, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), CFG)
        torch.cuda.synchronize()
        ref_logits, ref_probs = eng.logits.cpu().numpy(), probs.cpu().numpy()
        sc_ref = eng.scale.cpu().numpy()
        for r in range(2):
            got = res[r][t]
            # BN exchange: each group's fp64 sums come from one rank's units in the same order as the single-process
            # finalize walks them, so scale/shift of all 104 layers -- and with them features and logits -- are bit-identical
            assert np.array_equal(got["scale"], sc_ref)
            assert np.array_equal(got["logits"][0], ref_logits[r])
            # prototype sums are added in a different order (per-rank fp32 sums + all-reduce): fp32 rounding only
            assert np.array_equal(np.isnan(got["probs"][0]), np.isnan(ref_probs[r]))
            assert np.nanmax(np.abs(got["probs"][0] - ref_probs[r])) < 1e-4
            assert np.array_equal(got["rflag"], eng.run_flag.cpu().numpy())
        # replicated SAM state is bitwise identical across ranks, and equal to the single-process batch up to fp32 rounding
        assert np.array_equal(res[0][t]["run"], res[1][t]["run"])
        ref_run = eng.run_proto.cpu().numpy()
        assert np.abs(res[0][t]["run"] - ref_run).max() <= 1e-5 * np.abs(ref_run).max()


def test_two_ranks_reproduce_batch8_and_the_reference_fixture():
    """Global batch 8 sharded 4 + 4 over two ranks == the single-process batch-8 engine (scale/shift of all 104 layers and the
    logits bit-identical; SAM state up to fp32 rounding), which tests/test_gpu_net.py::test_batch8_vs_reference_bf16_floor and
    tests/test_gpu_exact.py::test_exact_batch8_vs_reference tie to the reference's batch_size = 8 run (net_b8.npz, b8s)."""
    from conftest import load_golden
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    res = _run_ranks(8)
    g = load_golden("net_b8.npz")
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 8, H, W)
    for t in range(2):
        x, y = synth.synth_frame(200 + t, 8, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), CFG)
        torch.cuda.synchronize()
        ref_logits, ref_probs = eng.logits.cpu().numpy(), probs.cpu().numpy()
        for r in range(2):
            got = res[r][t]
            assert np.array_equal(got["scale"], eng.scale.cpu().numpy()), (t, r)
            assert np.array_equal(got["logits"], ref_logits[4 * r:4 * r + 4]), (t, r)
            assert np.array_equal(np.isnan(got["probs"]), np.isnan(ref_probs[4 * r:4 * r + 4]))
            assert np.nanmax(np.abs(got["probs"] - ref_probs[4 * r:4 * r + 4])) < 1e-4
            assert np.array_equal(got["rflag"], eng.run_flag.cpu().numpy())
        assert np.array_equal(res[0][t]["run"], res[1][t]["run"])
        # the sharded logits against the REFERENCE's batch-8 logits: the bf16-storage floor (1 % of the logit range)
        both = np.concatenate([res[0][t]["logits"], res[1][t]["logits"]])
        rl = g[f"b8s_{t}_logits"]
        assert np.abs(both - rl).max() / (rl.max() - rl.min()) < 0.03


def test_two_ranks_oneshot_allreduce_and_engine():
    """the one-launch all-reduce over direct peer writes, two processes on one GPU (hipIpc-mapped buffers): 200 raw
    all-reduces of fp64 / fp32 / int32 payloads back to back are bit-exact sums in rank order; the sharded engine on top of it
    reproduces the single-process batch-2 engine (scale / shift of all 104 layers and logits bit-identical); a step captured in
    a HIP graph replays to the same result."""
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    res = _run_ranks(100)
    assert res[0]["ok"] and res[1]["ok"], "see gpurun_out/dist_rank*.log"
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 2, H, W)
    for t in range(NF):
        x, y = synth.synth_frame(t, 2, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), CFG)
        torch.cuda.synchronize()
        for r in range(2):
            got = res[r]["frames"][t]
            assert np.array_equal(got["scale"], eng.scale.cpu().numpy()), (t, r)
            assert np.array_equal(got["logits"][0], eng.logits.cpu().numpy()[r]), (t, r)
            assert np.nanmax(np.abs(got["probs"][0] - probs.cpu().numpy()[r])) < 1e-4
            assert np.array_equal(got["rflag"], eng.run_flag.cpu().numpy())
        assert np.array_equal(res[0]["frames"][t]["run"], res[1]["frames"][t]["run"])


def test_two_ranks_fused_exchange_stream_of_global_batches():
    """SURVEY 8 f2 as specified: the exchange fused into the finalize, one launch per BN layer for all batches in flight
    (104 + 2 exchange launches per step).  3 global batches of 2 in flight on 2 ranks == the single-process batch-2 engine run
    batch after batch: logits bit-identical, SAM state up to fp32 rounding; and the 15-in-flight communicator constructs."""
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    K = 3
    res = _run_ranks(101)
    assert res[0]["launches"] == 106 and res[1]["launches"] == 106
    assert res[0]["big_ok"] and res[1]["big_ok"]
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 2, H, W)
    for t in range(K):
        x, y = synth.synth_frame(t, 2, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), CFG)
        torch.cuda.synchronize()
        for r in range(2):
            assert np.array_equal(res[r]["logits"][t], eng.logits.cpu().numpy()[r]), (t, r)
            assert np.array_equal(res[r]["big_logits3"][t], eng.logits.cpu().numpy()[r]), (t, r)
            a, b = res[r]["probs"][t], probs.cpu().numpy()[r]
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b)) < 1e-4, (t, r)
    ref_run = eng.run_proto.cpu().numpy()
    assert np.array_equal(res[0]["run"], res[1]["run"]) and np.array_equal(res[0]["rflag"], eng.run_flag.cpu().numpy())
    assert np.abs(res[0]["run"] - ref_run).max() <= 1e-5 * np.abs(ref_run).max()


def test_four_ranks_fused_exchange_reproduces_batch4():
    """World size 4 (four processes on this GPU): the DAM scale / shift of all 104 layers, the logits, the probabilities and the
    replicated SAM state are BIT-IDENTICAL on every rank (rank-order sums); against the single-process batch-4 engine the group
    sums are associated differently (per-rank fp64 partials vs one pass over all units), so scale / shift agree to fp32 rounding
    -- in practice bit for bit -- and the logits to the bf16 re-rounding that a last-bit change of an affine can cause."""
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    res = _run_ranks(104, world=4)
    w = NetWeights(synth.synth_params(0), torch.device("cuda:0"))
    eng = Engine(w, 4, H, W)
    for t in range(NF):
        x, y = synth.synth_frame(t, 4, H, W)
        probs = eng.forward_and_adapt(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), CFG)
        torch.cuda.synchronize()
        sc_ref, sh_ref = eng.scale.cpu().numpy(), eng.shift.cpu().numpy()
        lg_ref = eng.logits.cpu().numpy()
        for r in range(4):
            got = res[r][t]
            assert got["launches"] == 106
            assert np.array_equal(got["scale"], res[0][t]["scale"]) and np.array_equal(got["shift"], res[0][t]["shift"]), (t, r)
            assert np.array_equal(got["run"], res[0][t]["run"]) and np.array_equal(got["rflag"], eng.run_flag.cpu().numpy()), (t, r)
            assert np.abs(got["scale"] - sc_ref).max() <= 1e-6 * np.abs(sc_ref).max(), (t, r)
            assert np.abs(got["logits"][0] - lg_ref[r]).max() <= 0.02 * (lg_ref.max() - lg_ref.min()), (t, r)
            assert np.array_equal(np.isnan(got["probs"][0]), np.isnan(probs.cpu().numpy()[r]))
        print("frame", t, "scale bit-identical to the single-process batch:", bool(np.array_equal(res[0][t]["scale"], sc_ref)),
              "logits:", bool(all(np.array_equal(res[r][t]["logits"][0], lg_ref[r]) for r in range(4))))


def test_oneshot_waits_for_a_late_peer():
    res = _run_ranks(102)
    assert res[0]["ok"] and res[1]["ok"]
    assert res[0]["waited"] > 1.5, res[0]          # rank 0 really sat in the bounded wait


def test_oneshot_missing_peer_poisons_and_raises():
    res = _run_ranks(103)
    assert res[0]["ok"], res[0]


def test_bench_spawns_its_own_ranks_and_replays_a_graph():
    """`python bench.py --gpus 2` with no launcher: two fresh rank processes (sharing this box's GPU: a rehearsal), the one-shot
    transport connects, the sharded step is a HIP-graph replay with 104 + 2 exchange launches per window."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--height", "65",
                        "--width", "129", "--frames-in-flight", "3", "--windows", "2", "--no-autotune"], env=env, capture_output=True,
                       timeout=300, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["launch"] == "hipGraph replay", line["config"]
    assert line["config"]["exchange_launches_per_step"] == 2 * 106, line["config"]
    assert "one-shot" in line["config"]["exchange_transport"] and "rehearsal" in line
    assert line["config"]["exchange_verified"], line["config"]        # all ranks: the same bits of scale / shift / prototypes
    assert line["value"] > 0


def test_bench_falls_back_to_torch_distributed_when_the_transport_fails():
    """A wait that expired on ONE rank during the run (injected on the last rank): every rank learns of it, the step is measured
    again through torch.distributed, launched eagerly, and the line says why."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["DIGA_BENCH_INJECT_TRANSPORT_FAILURE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "65",
                        "--width", "129", "--frames-in-flight", "2", "--windows", "2", "--no-autotune"], env=env, capture_output=True,
                       timeout=300, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert "expired wait" in line["config"]["exchange_transport_fallback"], line["config"]     # rank 0 heard of it from its peer
    assert line["config"]["launch"] == "eager" and "torch.distributed" in line["config"]["exchange_transport"], line["config"]
    assert line["config"]["exchange_launches_per_step"] == 2 * 106 and line["value"] > 0       # exchanges per step (two windows)


def _rccl_worker(port, q):
    """One rank, backend "nccl" (= RCCL): the exchange code path of the sharded engine (fp64 BN sums, fp32 prototype sums,
    int32 counts all-reduced on the device, in stream order with the kernels) must reproduce the engine without a group."""
    import time
    import traceback

    import torch.distributed as dist
    from diga_amd import synth
    from diga_amd.engine import Engine, NetWeights
    os.makedirs("gpurun_out", exist_ok=True)
    log = open("gpurun_out/dist_rccl1.log", "w")
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        w = NetWeights(synth.synth_params(0), dev)
        ok = True
        for stream_mode in (False, True):
            a = Engine(w, 2, H, W, group=dist.group.WORLD, stream=stream_mode, force_exchange=True)
            b = Engine(w, 2, H, W, stream=stream_mode)
            assert a.exchange and not b.exchange
            for t in range(2):
                x, y = synth.synth_frame(t, 2, H, W)
                x, y = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
                pa, pb = a.forward_and_adapt(x, y, CFG), b.forward_and_adapt(x, y, CFG)
                a.tail(y), b.tail(y)
                torch.cuda.synchronize()
                same = (torch.equal(a.scale, b.scale) and torch.equal(a.logits, b.logits) and torch.equal(a.run_proto, b.run_proto)
                        and torch.equal(a.hist, b.hist) and torch.equal(torch.nan_to_num(pa, nan=-1.0), torch.nan_to_num(pb, nan=-1.0)))
                print("stream" if stream_mode else "batch", "frame", t, "bit-identical" if same else "DIFFERENT", file=log, flush=True)
                ok = ok and same
            t0 = time.perf_counter()
            for _ in range(3):
                a.forward_and_adapt(x, y, CFG)
            torch.cuda.synchronize()
            print(f"eager step with 104 + 2 RCCL all-reduces: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", file=log, flush=True)
        # two windows, each with its own communicator and HIP stream, against two windows without a group
        from diga_amd.pipeline import StreamWindows
        groups = [dist.group.WORLD, dist.new_group(backend="nccl")]
        sa = StreamWindows(w, 2, 2, H, W, groups=groups, force_exchange=True)
        sb = StreamWindows(w, 2, 2, H, W)
        for rep in range(2):
            wins = []
            for i in range(2):
                xy = [synth.synth_frame(4 * rep + 2 * i + t, 1, H, W) for t in range(2)]
                wins.append((torch.from_numpy(np.concatenate([a_ for a_, _ in xy])).cuda(), torch.from_numpy(np.concatenate([b_ for _, b_ in xy])).cuda()))
            pa, pb = sa.step(wins, CFG), sb.step(wins, CFG)
            torch.cuda.synchronize()
            same = all(torch.equal(torch.nan_to_num(u, nan=-1.0), torch.nan_to_num(v, nan=-1.0)) for u, v in zip(pa, pb)) and \
                torch.equal(sa.engines[0].run_proto, sb.engines[0].run_proto) and torch.equal(sa.engines[0].hist, sb.engines[0].hist)
            print("two windows, step", rep, "bit-identical" if same else "DIFFERENT", file=log, flush=True)
            ok = ok and same
        dist.barrier()
        dist.destroy_process_group()
        q.put(ok)
    except Exception:
        traceback.print_exc(file=log)
        log.flush()
        q.put(None)
        raise


def test_rccl_single_rank_group_takes_the_exchange_path():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29600 + (os.getpid() + 31) % 2000, q))
    p.start()
    ok = q.get(timeout=100)
    p.join(timeout=60)
    assert ok is True, "see gpurun_out/dist_rccl1.log"
