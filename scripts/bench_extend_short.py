"""Prefill attention on batches of short and ragged prompts: TFLOP/s of causal flops over HIP-event time, with the
device-built launch plan (BENCH_PLAN=0: without) and the longest-request hint (BENCH_HINT=0: without)."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_sglang_amd import ops
from mini_sglang_amd.runtime import synth, workload

D = 128
dev = torch.device("cuda")
PLAN = os.environ.get("BENCH_PLAN", "1") != "0"
HINT = os.environ.get("BENCH_HINT", "1") != "0"


def run(name, Hq, Hkv, ps, prefix, inputs, steps=8):
    seq = [p + n for p, n in zip(prefix, inputs)]
    pages = sum(-(-s // ps) for s in seq) + 8
    pt = workload.make_page_table(seq, ps, pages, seed=5, scatter=True)
    kv = synth.normal(((pages + 1) * ps, 2, Hkv, D), 31, device=dev)
    T = sum(inputs)
    q = synth.normal((T, Hq, D), 32, device=dev)
    out = torch.empty_like(q)
    r2t = pt.req_to_token.to(dev)
    idx = torch.arange(len(seq), dtype=torch.int32, device=dev)
    inp = torch.tensor(inputs, dtype=torch.int32)
    cu = torch.nn.functional.pad(torch.cumsum(inp, 0, dtype=torch.int32), (1, 0)).to(dev)
    pre = torch.tensor(prefix, dtype=torch.int32, device=dev)
    plan = ops.extend_plan(cu, pre, max(inputs)) if PLAN else None
    kmax = max(seq) if HINT else 0
    f = lambda: ops.paged_extend(q, kv[:, 0], kv[:, 1], r2t, idx, cu, pre, max(inputs), D ** -0.5, ps, out=out, max_seqlen_k=kmax, plan=plan)
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
    flops = sum(4 * Hq * D * n * (p + (n + 1) / 2) for p, n in zip(prefix, inputs))
    print(f"{name:32s} tokens {T:6d}: {ms * 1e3:9.1f} us  {flops / (ms * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)


run("8B 64x256", 32, 8, 16, [0] * 64, [256] * 64)
run("8B 32x512", 32, 8, 16, [0] * 32, [512] * 32)
run("8B 16x1024", 32, 8, 16, [0] * 16, [1024] * 16)
run("8B 8x2048", 32, 8, 16, [0] * 8, [2048] * 8)
run("8B 16x(1024 on 3072 prefix)", 32, 8, 16, [3072] * 16, [1024] * 16)
run("8B mixed", 32, 8, 16, [0, 1024, 0, 4096, 0, 0, 2048, 0], [100, 300, 2000, 50, 700, 4000, 128, 1500])
rng = random.Random(5)
inp = [rng.choice([30, 100, 300, 700, 1500, 3000]) for _ in range(32)]
pre = [rng.choice([0, 0, 0, 512, 2048, 8192]) for _ in range(32)]
run("8B ragged 32 requests", 32, 8, 16, pre, inp)
