"""Extend (prefill) attention throughput: achieved TFLOP/s = 4*Hq*128*sum(Sq*(prefix+(Sq+1)/2)) / time."""
import json, os, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_sglang_amd import ops
from mini_sglang_amd.runtime import workload

dev = "cuda"
D = 128


PLAN = os.environ.get("BENCH_PLAN", "1") == "1"
HINT = os.environ.get("BENCH_HINT", "1") == "1"


def run(name, Hq, Hkv, ps, prefix, inp, steps=10):
    bs = len(inp)
    seq = [p + n for p, n in zip(prefix, inp)]
    pages = sum(-(-s // ps) for s in seq) + 8
    pt = workload.make_page_table(seq, ps, pages, 3)
    kv = torch.empty(((pages + 1) * ps, 2, Hkv, D), dtype=torch.bfloat16, device=dev).normal_()
    q = torch.randn((sum(inp), Hq, D), device=dev).bfloat16()
    cu = torch.nn.functional.pad(torch.cumsum(torch.tensor(inp, dtype=torch.int32), 0, dtype=torch.int32), (1, 0)).to(dev)
    r2t, idx, pre = pt.req_to_token.to(dev), torch.arange(bs, dtype=torch.int32, device=dev), torch.tensor(prefix, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    # as the backend calls it: longest request announced, launch order planned once per batch (not per layer)
    plan = ops.extend_plan(cu, pre, max(inp)) if (bs > 1 and PLAN) else None
    kmax = max(seq) if HINT else 0
    f = lambda: ops.paged_extend(q, kv[:, 0], kv[:, 1], r2t, idx, cu, pre, max(inp), D ** -0.5, ps, out=out, max_seqlen_k=kmax, plan=plan)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    flops = 4 * Hq * D * sum(n * (p + (n + 1) / 2) for p, n in zip(prefix, inp))
    print(json.dumps(dict(name=name, Hq=Hq, Hkv=Hkv, ps=ps, prefix=prefix if len(set(prefix)) > 1 else prefix[:1], inp=inp if len(set(inp)) > 1 else inp[:1], bs=bs, ms=round(ms, 3), TFLOPs=round(flops / ms / 1e9, 1))), flush=True)


run("8B TP1 Sq4096 x4", 32, 8, 16, [0] * 4, [4096] * 4)
run("8B TP1 Sq8192 x2", 32, 8, 16, [0] * 2, [8192] * 2)
run("8B TP1 Sq16384", 32, 8, 16, [0], [16384])
run("cfg5 32B TP8 Sq4096", 8, 1, 64, [0], [4096])
run("cfg5 32B TP8 Sq16384", 8, 1, 64, [0], [16384])
run("cfg5 32B TP8 Sq4096 prefix64k", 8, 1, 64, [65536], [4096])
run("cfg5 32B TP8 Sq16384 prefix64k", 8, 1, 64, [65536], [16384], steps=4)
run("0.6B Sq2048 x8", 16, 8, 1, [0] * 8, [2048] * 8)
