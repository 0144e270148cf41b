"""Sum all-reduce across the tensor-parallel ranks of one node by direct peer reads (``msgl_ar_*`` in
``libmsgl_hip.so``), for the ``[tokens, hidden]`` bf16 partial sums the reference hands to
``torch.distributed.all_reduce`` after every row-parallel linear and in the vocab-parallel embedding
(mini_sglang/layers/linear.py:244-248, embed_head.py:45-53).

One process per GPU.  ``XgmiAllReduce(group)`` allocates this rank's staging memory, exchanges the IPC handles
through the process group it is given (one all_gather of 128 bytes), and maps the peers.  ``all_reduce(t)`` is one
kernel launch on the current stream and can be captured in a hipGraph.  Messages longer than the communicator's
``max_bytes`` (prefill-sized activations, bandwidth-bound) go to the process group's own all-reduce (RCCL).

The one-launch form has every rank's kernel wait for its peers' kernels, which is sound when each rank owns a GPU.
When several ranks share ONE device (a rehearsal on a single-GPU machine) nothing guarantees that kernels of two
processes run at the same time, so the communicator then issues every call step by step
(``msgl_ar_allreduce_bf16_step``) and crosses the protocol's synchronisation points on the host: same device code,
flags and results, not capturable.  ``stepped=None`` picks that mode exactly when two ranks report the same device.
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
                 one_shot_bytes: int = 0, stepped: Optional[bool] = None, timeout_ms: int = 0):
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
                if timeout_ms:
                    _capi.check(_lib.msgl_ar_set_timeout_ms(self._h, int(timeout_ms)))
                mine = ctypes.create_string_buffer(EXPORT_BYTES)
                _capi.check(_lib.msgl_ar_export(self._h, mine))
                payload = mine.raw
            except Exception as e:  # noqa: BLE001 - reported below, on every rank
                problem = f"rank {self.rank}: {e}"
            everyone = [None] * self.world
            where = _device_identity(self.device)
            dist.all_gather_object(everyone, (payload, problem, where), group=group)  # also: every rank has zeroed its flags
            if all(p == "" or p is None for _, p, _ in everyone) and all(len(h) == EXPORT_BYTES for h, _, _ in everyone):
                try:
                    _capi.check(_lib.msgl_ar_open(self._h, b"".join(h for h, _, _ in everyone)))
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
        places = [w for _, _, w in everyone]
        self.shared_device = len(set(places)) < len(places)
        self.stepped = self.shared_device if stepped is None else bool(stepped)

    def set_timeout_ms(self, milliseconds: int) -> None:
        _capi.check(_lib.msgl_ar_set_timeout_ms(self._h, int(milliseconds)))

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
        stream = torch.cuda.current_stream()
        if not self.stepped:
            _capi.check(_lib.msgl_ar_allreduce_bf16(self._h, t.data_ptr(), out.data_ptr(), t.numel(), self.one_shot_bytes,
                                                    stream.cuda_stream))
            return out
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("XgmiAllReduce: ranks that share a device run the call step by step with host barriers "
                               "in between, which a graph cannot capture")
        steps = int(_lib.msgl_ar_num_steps(self._h, t.numel(), self.one_shot_bytes))
        for step in range(steps):
            _capi.check(_lib.msgl_ar_allreduce_bf16_step(self._h, t.data_ptr(), out.data_ptr(), t.numel(), self.one_shot_bytes,
                                                         step, stream.cuda_stream))
            if step + 1 < steps:  # every rank's step has completed before anybody's next one looks at the flags
                stream.synchronize()
                dist.barrier(group=self.group)
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


def _device_identity(device: torch.device) -> str:
    """Something two ranks have in common exactly when they sit on the same physical GPU."""
    props = torch.cuda.get_device_properties(device)
    ident = getattr(props, "uuid", None)
    if ident is None:
        ident = f"{getattr(props, 'pci_bus_id', device.index)}:{getattr(props, 'pci_device_id', 0)}:{getattr(props, 'pci_domain_id', 0)}"
    import socket

    return f"{socket.gethostname()}/{ident}"


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


def _host_side_backend(group=None) -> bool:
    """True when the process group runs on the CPU (gloo): GPU tensors then cross it as host copies made HERE, with
    plain synchronous device<->host copies, instead of through the backend's own staging streams and events (two
    processes sharing one GPU in a rehearsal were seen to stall there)."""
    return dist.get_backend(group) != "nccl"


def tp_all_reduce(t: torch.Tensor) -> torch.Tensor:
    """The all-reduce of the tensor-parallel linears (linear.py:246-247), in place."""
    if _active is not None and _active.fits(t):
        return _active.all_reduce(t)
    if t.is_cuda and _host_side_backend():
        # (rehearsals on a CPU-side backend only) 16-bit floats are summed in fp32 and rounded once, as the direct-read
        # kernels do; every other dtype, integers included, is reduced as it is
        h = t.cpu()
        if h.dtype in (torch.bfloat16, torch.float16):
            h = h.float()
        dist.all_reduce(h)
        t.copy_(h)
        return t
    dist.all_reduce(t)
    return t


def tp_all_gather(t: torch.Tensor, group=None):
    """Every rank's `t` (equal shapes), in rank order (logits_processor.py:61-64).  A gather only copies, so the
    tensors travel as bytes, whatever the backend can reduce."""
    world = dist.get_world_size(group)
    if t.is_cuda and _host_side_backend(group):
        h = t.contiguous().view(torch.uint8).cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h, group=group)
        return [p.to(t.device).view(t.dtype).view(t.shape) for p in parts]
    t = t.contiguous()
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather([p.view(torch.uint8) for p in parts], t.view(torch.uint8), group=group)
    return parts


def agree(value: float, op: str = "max", device=None) -> float:
    """One number every rank ends up with: the max / min over the ranks' values (control decisions and timings)."""
    red = dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.MIN
    if device is None or _host_side_backend():
        t = torch.tensor([value], dtype=torch.float64)
        dist.all_reduce(t, op=red)
        return float(t[0])
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=red)
    return float(t[0])
