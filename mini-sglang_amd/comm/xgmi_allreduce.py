"""Sum all-reduce across the tensor-parallel ranks of one node by direct peer reads (``msgl_ar_*`` in
``libmsgl_hip.so``), for the ``[tokens, hidden]`` bf16 partial sums the reference hands to
``torch.distributed.all_reduce`` after every row-parallel linear and in the vocab-parallel embedding
(mini_sglang/layers/linear.py:244-248, embed_head.py:45-53).

One process per GPU.  ``XgmiAllReduce(group)`` allocates this rank's staging memory, exchanges the IPC handles
through the process group it is given (one all_gather of 128 bytes), and maps the peers.  ``all_reduce(t)`` is one
kernel launch on the current stream and can be captured in a hipGraph.  Messages longer than the communicator's
``max_bytes`` (prefill-sized activations, bandwidth-bound) go to the process group's own all-reduce (RCCL).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.distributed as dist

from .. import _capi

_lib = _capi.lib
EXPORT_BYTES = 128
MAX_WORLD = 8


class XgmiAllReduce:
    def __init__(self, group=None, max_bytes: int = 8 << 20, device: Optional[torch.device] = None,
                 one_shot_bytes: int = 0):
        if not dist.is_initialized():
            raise RuntimeError("XgmiAllReduce needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if not 2 <= self.world <= MAX_WORLD:
            raise ValueError(f"XgmiAllReduce serves 2..{MAX_WORLD} ranks of one node, not {self.world}")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        unit = 16 * self.world
        self.max_bytes = -(-int(max_bytes) // unit) * unit
        self.one_shot_bytes = int(one_shot_bytes)
        self._h = ctypes.c_void_p()
        # Every rank takes part in both exchanges below whatever happened locally, so a failure on one rank
        # surfaces as the same exception on all of them instead of a hang in the next collective.
        payload, problem = b"", None
        with torch.cuda.device(self.device):
            try:
                _capi.check(_lib.msgl_ar_create(self.rank, self.world, self.max_bytes, ctypes.byref(self._h)))
                mine = ctypes.create_string_buffer(EXPORT_BYTES)
                _capi.check(_lib.msgl_ar_export(self._h, mine))
                payload = mine.raw
            except Exception as e:  # noqa: BLE001 - reported below, on every rank
                problem = f"rank {self.rank}: {e}"
            everyone = [None] * self.world
            dist.all_gather_object(everyone, (payload, problem), group=group)  # also: every rank has zeroed its flags
            if all(p == "" or p is None for _, p in everyone) and all(len(h) == EXPORT_BYTES for h, _ in everyone):
                try:
                    _capi.check(_lib.msgl_ar_open(self._h, b"".join(h for h, _ in everyone)))
                except Exception as e:  # noqa: BLE001
                    problem = f"rank {self.rank}: {e}"
            else:
                problem = problem or "a peer could not set up its buffers"
            verdicts = [None] * self.world
            dist.all_gather_object(verdicts, problem, group=group)
        failures = [v for v in verdicts if v]
        if failures:
            if self._h:
                _lib.msgl_ar_destroy(self._h)
                self._h = ctypes.c_void_p()
            raise RuntimeError("XgmiAllReduce setup failed: " + "; ".join(failures))

    def fits(self, t: torch.Tensor) -> bool:
        return (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() % 8 == 0
                and t.numel() * 2 <= self.max_bytes and t.data_ptr() % 16 == 0)

    def all_reduce(self, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """In place by default.  Sum in fp32 in rank order, one rounding: the same bits on every rank."""
        if not self.fits(t):
            raise ValueError("XgmiAllReduce.all_reduce: contiguous 16-byte aligned bf16 GPU tensor with numel % 8 == 0 and "
                             f"at most {self.max_bytes} bytes expected, got {tuple(t.shape)} {t.dtype}")
        out = t if out is None else out
        if out is not t and not (self.fits(out) and out.numel() == t.numel()):
            raise ValueError("XgmiAllReduce.all_reduce: out must match the input")
        _capi.check(_lib.msgl_ar_allreduce_bf16(self._h, t.data_ptr(), out.data_ptr(), t.numel(), self.one_shot_bytes,
                                                torch.cuda.current_stream().cuda_stream))
        return out

    def check(self) -> None:
        """Synchronise and raise if a kernel gave up waiting for a peer."""
        _capi.check(_lib.msgl_ar_check(self._h))

    def close(self) -> None:
        if self._h:
            torch.cuda.synchronize()
            dist.barrier(group=self.group)  # nobody is still reading this rank's memory
            _lib.msgl_ar_destroy(self._h)
            self._h = ctypes.c_void_p()


_active: Optional[XgmiAllReduce] = None


def enable_xgmi_allreduce(group=None, max_bytes: int = 8 << 20, **kw) -> XgmiAllReduce:
    """Route ``tp_all_reduce`` of messages up to ``max_bytes`` through direct peer reads."""
    global _active
    if _active is None:
        _active = XgmiAllReduce(group, max_bytes, **kw)
    return _active


def disable_xgmi_allreduce() -> None:
    global _active
    if _active is not None:
        _active.close()
        _active = None


def tp_all_reduce(t: torch.Tensor) -> torch.Tensor:
    """The all-reduce of the tensor-parallel linears (linear.py:246-247), in place."""
    if _active is not None and _active.fits(t):
        return _active.all_reduce(t)
    dist.all_reduce(t)
    return t
