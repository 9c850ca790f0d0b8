"""Several windows of a batch-1 stream in flight on several HIP streams.

One `Engine(stream=True)` runs the trunk of k consecutive frames as one pass.  `StreamWindows` keeps `nwin` such engines,
each on its own HIP stream: the trunks of consecutive windows run concurrently (one window's memory-bound BN-apply and
SAM kernels fill the compute-bound convolutions' idle CUs and tile-quantisation tails of the other), while the SAM /
fusion / metric stages stay strictly in stream order -- window w+1's SAM waits for window w's, and all engines share the
one running-prototype state -- so every frame's result is still bit-identical to one-frame steps
(tests/test_gpu_net.py::test_two_windows_on_two_streams_are_bit_identical).  The whole step forks from and joins
the caller's current stream, so it can be captured into one HIP graph (bench.py does)."""
from __future__ import annotations

import torch

from .engine import Engine, NetWeights


class StreamWindows:
    def __init__(self, weights: NetWeights, k: int, nwin: int, H: int, W: int, groups=None, **engine_kw):
        """groups: None, or one torch.distributed process group PER WINDOW (the sharded form: every window holds this
        rank's frame of k consecutive global batches).  Each window needs its own group -- its own RCCL communicator and
        communicator stream -- because a communicator runs its collectives in issue order: with one shared group window
        w+1's first BN exchange would queue behind all 104 of window w's and the trunks could not overlap.  Every rank
        creates the groups and issues each group's collectives in the same order, so the usual ordering rule holds per
        communicator."""
        self.k, self.nwin = k, nwin
        assert groups is None or len(groups) == nwin
        self.engines = [Engine(weights, k, H, W, stream=k > 1 or nwin > 1, group=None if groups is None else groups[i], **engine_kw)
                        for i in range(nwin)]
        e0 = self.engines[0]
        for e in self.engines[1:]:               # one cross-frame state (ttda:1702-1705) and one metric for all windows
            e.run_proto, e.run_flag, e.state_flag, e.hist = e0.run_proto, e0.run_flag, e0.state_flag, e0.hist
            e.tile_cache = e0.tile_cache        # same shapes: the tile autotune runs once, in the first window
        self.streams = [torch.cuda.Stream() for _ in range(nwin)] if nwin > 1 else [None]

    def step(self, windows, cfg, tail: bool = True):
        """windows: nwin pairs (x [k,3,H,W] fp32, y [k,H,W] int) in stream order, on the device.  Returns the engines'
        probability buffers ([k,C,h,w] each), valid once the caller's stream has caught up."""
        assert len(windows) == self.nwin
        if self.nwin == 1:
            e = self.engines[0]
            x, y = windows[0]
            e.forward_and_adapt(x, y, cfg)
            if tail:
                e.tail(y)
            return [e.probs]
        cur = torch.cuda.current_stream()
        for e, (x, _), s in zip(self.engines, windows, self.streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e.forward_feat(x)
        prev = None
        for e, (_, y), s in zip(self.engines, windows, self.streams):
            with torch.cuda.stream(s):
                if prev is not None:
                    s.wait_event(prev)             # SAM in stream order
                e.sam_fuse(y, cfg)
                if tail:
                    e.tail(y)
                prev = torch.cuda.Event()
                prev.record(s)
        for s in self.streams:
            cur.wait_stream(s)
        return [e.probs for e in self.engines]

    def reset_state(self):
        self.engines[0].reset_state()
