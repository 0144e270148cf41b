"""Prefill attention TFLOP/s (causal flops / HIP-event time) on the shapes the benchmark reports, with and without
the key-split workspace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_sglang_amd import ops
from mini_sglang_amd.runtime import synth, workload

D = 128
dev = torch.device("cuda")


def measure(Hq, Hkv, ps, Sq, prefix, steps=8, ws=False):
    S = prefix + Sq
    pages = -(-S // ps) + 8
    pt = workload.make_page_table([S], ps, pages, seed=5, scatter=True)
    kv = synth.normal(((pages + 1) * ps, 2, Hkv, D), 31, device=dev)
    q = synth.normal((Sq, Hq, D), 32, device=dev)
    out = torch.empty_like(q)
    r2t = pt.req_to_token.to(dev)
    idx = torch.zeros(1, dtype=torch.int32, device=dev)
    cu = torch.tensor([0, Sq], dtype=torch.int32, device=dev)
    pre = torch.tensor([prefix], dtype=torch.int32, device=dev)
    w = torch.empty(ops.extend_workspace_bytes(Sq, Hq, D), dtype=torch.uint8, device=dev) if ws else None
    f = lambda: ops.paged_extend(q, kv[:, 0], kv[:, 1], r2t, idx, cu, pre, Sq, D ** -0.5, ps, out=out, max_seqlen_k=S, workspace=w)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps)
    ms = sorted(ts)[1]
    tf = 4 * Hq * D * Sq * (prefix + (Sq + 1) / 2) / (ms * 1e-3) / 1e12
    print(f"Hq {Hq:2d} Hkv {Hkv} ps {ps:2d} Sq {Sq:6d} prefix {prefix:6d} workspace {int(ws)}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s ({tf/2500:.3f})", flush=True)


measure(32, 8, 16, 16384, 0)
measure(8, 1, 64, 16384, 0)
measure(8, 1, 64, 4096, 65536, steps=4)
measure(8, 1, 64, 4096, 65536, steps=4, ws=True)
measure(8, 1, 64, 1024, 65536, steps=4)
measure(8, 1, 64, 1024, 65536, steps=4, ws=True)
measure(32, 8, 16, 512, 32768, steps=4)
measure(32, 8, 16, 512, 32768, steps=4, ws=True)
