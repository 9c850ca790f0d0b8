"""Several windows of a batch-1 stream in flight on several HIP streams.

One `Engine(stream=True)` runs the trunk of k consecutive frames as one pass.  `StreamWindows` keeps `nwin` such engines,
each on its own HIP stream: the trunks of consecutive windows run concurrently (one window's memory-bound BN-apply and
SAM kernels fill the compute-bound convolutions' idle CUs and tile-quantisation tails of the other), while the SAM /
fusion / metric stages stay strictly in stream order -- window w+1's SAM waits for window w's, and all engines share the
one running-prototype state -- so every frame's result is still bit-identical to one-frame steps
(tests/test_gpu_net.py::test_two_windows_on_two_streams_are_bit_identical).  The whole step forks from and joins
the caller's current stream, so it can be captured into one HIP graph (bench.py does).

`HostFeeder` is the caller's side of that step when the frames arrive in HOST memory (the reference's DataLoader hands
`DIGA.test_step` CPU tensors): two device input sets, an H2D copy stream, and the step replayed from one HIP graph per set, so
that step i + 1's frames cross PCIe while step i computes."""
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


class HostFeeder:
    """Double-buffered host -> device feeding of `StreamWindows.step`.

        feeder = HostFeeder(sw, cfg)
        feeder.feed(windows_0)                  # host tensors: nwin pairs (x [k,3,H,W] fp32, y [k,H,W] int64), ideally pinned
        for windows_next in stream:             # ... or None after the last one
            feeder.run(windows_next)            # computes the step fed LAST while `windows_next` travels; enqueue only
        feeder.run(None)
        torch.cuda.synchronize()                # results (sw.engines[i].probs / pred / hist ...) are those of the last step run

    Frames are copied on a side stream into the input set the running step does NOT read; events order the copy after the step
    that last read that set and the step after its copy.  Each set's step is captured into a HIP graph on first use (graph=True;
    shapes are fixed by `sw`), so a step is one replay.  Results are bit-identical to calling `sw.step` on device tensors
    (tests/test_gpu_net.py::test_host_feeder_is_bit_identical)."""

    def __init__(self, sw: StreamWindows, cfg, graph: bool = True, tail: bool = True):
        self.sw, self.cfg, self.graph, self.tail = sw, cfg, graph, tail
        self.side = torch.cuda.Stream()
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.done = [torch.cuda.Event() for _ in range(2)]
        self.dev = [None, None]                 # device input sets
        self.graphs = [None, None]
        self.fed = -1                           # number of feeds so far - 1: feed i goes to set i & 1
        self.pending = False

    def _alloc(self, s, windows):
        d = self.sw.engines[0].dev if hasattr(self.sw.engines[0], "dev") else torch.device("cuda", torch.cuda.current_device())
        self.dev[s] = [(torch.empty(x.shape, dtype=x.dtype, device=d), torch.empty(y.shape, dtype=y.dtype, device=d)) for x, y in windows]

    def feed(self, windows):
        """Enqueue the H2D copies of one step's frames (non-blocking when the host tensors are pinned)."""
        assert not self.pending, "run() the step fed last before feeding the next one"
        assert len(windows) == self.sw.nwin
        self.fed += 1
        s = self.fed & 1
        if self.dev[s] is None:
            self._alloc(s, windows)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.done[s])          # the step that read set s has finished (no-op before the first record)
            for (hx, hy), (dx, dy) in zip(windows, self.dev[s]):
                dx.copy_(hx, non_blocking=True)
                dy.copy_(hy, non_blocking=True)
            self.ready[s].record(self.side)
        self.pending = True

    def run(self, next_windows=None):
        """Compute the step fed last on the current stream; `next_windows` (host) starts travelling first, so that it crosses
        PCIe underneath.  Returns the engines' probability buffers (valid once the current stream has caught up)."""
        assert self.pending, "feed() first"
        s = self.fed & 1
        self.pending = False
        if next_windows is not None:
            self.feed(next_windows)                     # goes to the other set
            self.pending = True
        cur = torch.cuda.current_stream()
        cur.wait_event(self.ready[s])
        if self.graph:
            if self.graphs[s] is None:
                self.sw.step(self.dev[s], self.cfg, tail=self.tail)         # eager once (tile autotune, allocations) ...
                self._first_done(s)                                         # ... then captured for every later use of the set
                return [e.probs for e in self.sw.engines]
            self.graphs[s].replay()
        else:
            self.sw.step(self.dev[s], self.cfg, tail=self.tail)
        self.done[s].record(cur)
        return [e.probs for e in self.sw.engines]

    def _first_done(self, s):
        """The first step on a set ran eagerly (it may autotune and allocate); capture the same step for every later use.  The
        capture itself executes nothing, so the state (prototypes, metric) advances exactly once per run()."""
        cur = torch.cuda.current_stream()
        self.done[s].record(cur)
        cur.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.sw.step(self.dev[s], self.cfg, tail=self.tail)
        self.graphs[s] = g
