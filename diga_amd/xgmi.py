"""One-shot all-reduce over direct peer writes (csrc/xgmi.hip, include/diga_hip.h): the host side.

`OneShotComm(group, max_elems)` allocates this rank's exchange buffer, shares its hipIpc handle with the ranks of `group`
(one all_gather of 64 bytes at set-up) and maps the peers' buffers.  `all_reduce(t)` then sums a device tensor (fp64, fp32
or int32) over the ranks in ONE kernel launch on the current stream -- deterministic (rank order), identical bits on every
rank, no host involvement, capturable in a HIP graph.  The sharded Engine uses it for the 104 per-layer BN exchanges and
the prototype exchange when `Engine(oneshot=True)`; torch.distributed (RCCL) stays the default transport.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import call

_DT = {torch.float64: 0, torch.float32: 1, torch.int32: 2}


class OneShotComm:
    def __init__(self, group, max_elems: int = 1 << 18):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.max_elems = int(max_elems)
        _lib.load()
        self._comm = C.c_void_p()
        handle = (C.c_ubyte * 64)()
        call("diga_xgmi_create", self.world, self.rank, self.max_elems, C.byref(self._comm), C.cast(handle, C.c_void_p))
        mine = torch.tensor(list(handle), dtype=torch.uint8)
        if dist.get_backend(group) == "nccl":
            mine = mine.cuda()
        allh = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allh, mine, group=group)
        blob = bytes(torch.cat([h.cpu() for h in allh]).tolist())
        self._handles = C.create_string_buffer(blob, len(blob))
        call("diga_xgmi_connect", self._comm, C.cast(self._handles, C.c_void_p))
        dist.barrier(group=group)          # every rank has mapped every buffer before the first exchange

    def all_reduce(self, t: torch.Tensor):
        assert t.is_cuda and t.is_contiguous() and t.dtype in _DT, (t.dtype, t.device)
        call("diga_xgmi_allreduce", self._comm, C.c_void_p(t.data_ptr()), t.numel(), _DT[t.dtype],
             C.c_void_p(torch.cuda.current_stream().cuda_stream))

    def supports(self, t: torch.Tensor) -> bool:
        return t.dtype in _DT and t.numel() * t.element_size() <= self.max_elems * 8 and (t.dtype == torch.float64 or t.numel() % 2 == 0)

    def check(self):
        """Synchronises the device; raises DigaError if a peer's contribution never arrived."""
        call("diga_xgmi_check", self._comm)

    def close(self):
        if self._comm:
            call("diga_xgmi_destroy", self._comm)
            self._comm = C.c_void_p()
