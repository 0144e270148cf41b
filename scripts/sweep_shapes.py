"""Decode-attention bandwidth over the benchmark family's configurations (kernel level, one layer per launch)."""
import json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_sglang_amd import ops
from mini_sglang_amd.attn.hip_attn_backend import splits_for_batch
from mini_sglang_amd.runtime import synth, workload

dev = "cuda"
D = 128


def run(name, Hq, Hkv, ps, seq_lens, groups=None, prefix=None, layers=4, steps=20, chunk=0, max_splits=0, layout="interleaved"):
    bs = len(seq_lens)
    pages = sum(-(-s // ps) for s in seq_lens) + 8
    pt = workload.make_page_table(seq_lens, ps, pages, 3, groups=groups, group_prefix_lens=prefix)
    slots = (pages + 1) * ps
    per_layer = slots * 2 * Hkv * D * 2
    layers = max(1, min(layers, int(60e9 // per_layer)))
    pools = []
    for l in range(layers):
        kv = torch.empty((slots, 2, Hkv, D), dtype=torch.bfloat16, device=dev).normal_()
        pools.append((kv[:, 0], kv[:, 1]))
    q = torch.randn((bs, Hq, D), device=dev).bfloat16()
    r2t, sl, idx = pt.req_to_token.to(dev), pt.seq_lens.to(dev), torch.arange(bs, dtype=torch.int32, device=dev)
    ms_ = max_splits or splits_for_batch(bs)
    ch = chunk or ops.pick_split(max(seq_lens), sum(seq_lens), bs, ms_, Hkv)
    ws = torch.empty(max(16, ops.decode_workspace_bytes(bs, Hq, D, ms_)), dtype=torch.uint8, device=dev)
    out = torch.empty_like(q)
    reuse = torch.tensor([prefix[g] for g in groups], dtype=torch.int32, device=dev) if groups is not None else None
    stream = os.environ.get("KV_STREAM", "1") == "1"
    ms_ = max(1, min(ms_, -(-max(seq_lens) // ch)))
    plan = ops.decode_plan(sl, ms_, ch) if (os.environ.get("PLAN", "auto") == "1" or (os.environ.get("PLAN", "auto") == "auto" and max(seq_lens) > 1.25 * sum(seq_lens) / bs)) and ms_ > 1 else None
    cascade = None
    if groups is not None and os.environ.get("SHARED", "0") == "1":
        # shared prefixes once per group through the matrix-core kernel, the rest per request
        ng = max(groups) + 1
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
        first, count = [groups.index(g) for g in range(ng)], [groups.count(g) for g in range(ng)]
        kc_ = int(os.environ.get("KEY_CHUNK", "1024"))
        start = i32([prefix[g] for g in groups])
        extra = i32([-(-prefix[g] // kc_) for g in groups])
        rest = [s - prefix[g] for s, g in zip(seq_lens, groups)]
        ms_ = max_splits or splits_for_batch(bs)
        ch = chunk or ops.pick_split_planned(rest, ms_ - max(-(-p_ // kc_) for p_ in prefix), Hkv)
        ms_ = -(-max(rest) // ch) + max(-(-p_ // kc_) for p_ in prefix)
        ws = torch.empty(max(16, ops.decode_workspace_bytes(bs, Hq, D, ms_)), dtype=torch.uint8, device=dev)
        plan = ops.decode_plan(sl, ms_, ch, start_lens=start)
        cascade = (i32(first), i32(count), i32(prefix), max(count), max(prefix), kc_, start, extra)
    def f(i):
        kc, vc = pools[i % layers]
        if cascade is not None:
            gf, gc, gp, mm, mp, kc_, start, extra = cascade
            ops.shared_prefix_decode(q, kc, vc, r2t, idx, sl, start, gf, gc, gp, mm, mp, kc_, D ** -0.5, ws, ms_, ch)
            return ops.paged_decode(q, kc, vc, r2t, idx, sl, D ** -0.5, ps, out=out, workspace=ws, max_splits=ms_, chunk_tokens=ch,
                                    kv_stream=stream, plan=plan, start_lens=start, extra_splits=extra)
        return f0(i)
    f0 = lambda i: ops.paged_decode(q, pools[i % layers][0], pools[i % layers][1], r2t, idx, sl, D ** -0.5, ps, out=out, workspace=ws, max_splits=ms_, chunk_tokens=ch,
                                   kv_stream=stream, reuse_lens=reuse if os.environ.get("REUSE", "1") == "1" else None, plan=plan)
    for i in range(5): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): f(i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    b = 2 * Hkv * D * 2 * sum(seq_lens)
    ub = 2 * Hkv * D * 2 * pt.unique_tokens
    print(json.dumps(dict(name=name, bs=bs, Hq=Hq, Hkv=Hkv, ps=ps, tokens=sum(seq_lens), layers=layers, max_splits=ms_, chunk=ch, planned=plan is not None, cascade=cascade is not None, ms=round(ms, 4),
                          GBps=round(b / ms / 1e6, 1), unique_GBps=round(ub / ms / 1e6, 1))), flush=True)
    del pools


which = sys.argv[1:] or ["cfg2", "cfg3i", "cfg3ii", "tp8", "cfg5"]
if "cfg2" in which:
    run("cfg2 0.6B bs64 2k ps1", 16, 8, 1, [2048] * 64)
if "cfg3i" in which:
    run("cfg3i 8B bs256 8k ps16", 32, 8, 16, [8192] * 256)
if "cfg3ii" in which:
    lens = workload.mixed_lengths(256, 1024, 32768, 3)
    # groups of GROUP (default 16) requests sharing a page-aligned 1-4k prefix
    gsz = int(os.environ.get("GROUP", "16"))
    groups = [i // gsz for i in range(256)]
    prefix = [1024 * (1 + g % 4) for g in range(256 // gsz)]
    lens = [max(l, prefix[g] + 16) for l, g in zip(lens, groups)]
    tot = sum(lens)
    run("cfg3ii 8B bs256 mixed 1k-32k ps16 radix-shared", 32, 8, 16, lens, groups, prefix, layers=2)
    run("cfg3ii-unshared", 32, 8, 16, lens, layers=2)
if "tp8" in which:
    run("cfg4 8B TP8 shard bs256 8k ps16", 4, 1, 16, [8192] * 256, layers=8)
    run("cfg4 8B TP2 shard", 16, 4, 16, [8192] * 256)
    run("cfg4 8B TP4 shard", 8, 2, 16, [8192] * 256)
if "cfg5" in which:
    run("cfg5 32B TP8 shard bs1 128k ps64", 8, 1, 64, [131072], layers=8)
    run("cfg5 32B TP8 shard bs8 128k ps64", 8, 1, 64, [131072] * 8, layers=8)
    run("8B bs1 32k ps16", 32, 8, 16, [32768], layers=8)
    run("8B bs8 8k ps16", 32, 8, 16, [8192] * 8, layers=8)
    run("32B TP1 bs64 8k ps16", 64, 8, 16, [8192] * 64)
