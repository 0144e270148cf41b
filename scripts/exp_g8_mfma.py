"""Experiment: G = 8 decode through the matrix-core kernel (every request its own 'group' of one member) against
the gather kernel, same shapes as the long-context configuration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_sglang_amd import ops
from mini_sglang_amd.attn.hip_attn_backend import splits_for_batch
from mini_sglang_amd.runtime import workload

dev = torch.device("cuda")
D = 128


def run(name, Hq, Hkv, ps, lens, key_chunk, layers=4, steps=12):
    bs = len(lens)
    pages = sum(-(-n // ps) for n in lens) + 8
    pt = workload.make_page_table(lens, ps, pages, 3)
    slots = (pages + 1) * ps
    pools = [torch.empty((slots, 2, Hkv, D), dtype=torch.bfloat16, device=dev).normal_() for _ in range(layers)]
    q = torch.randn((bs, Hq, D), device=dev).bfloat16()
    r2t, sl = pt.req_to_token.to(dev), pt.seq_lens.to(dev)
    idx = torch.arange(bs, dtype=torch.int32, device=dev)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    b = 2 * Hkv * D * 2

    def timeit(f):
        for i in range(3):
            f(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(steps):
                f(i)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps * 1e-3

    # baseline: gather kernel
    ms = splits_for_batch(bs)
    chunk = ops.pick_split(max(lens), sum(lens), bs, ms, Hkv)
    ms = max(1, min(ms, -(-max(lens) // chunk)))
    ws = torch.empty(max(16, ops.decode_workspace_bytes(bs, Hq, D, ms)), dtype=torch.uint8, device=dev)
    o1 = torch.empty_like(q)
    t1 = timeit(lambda i: ops.paged_decode(q, pools[i % layers][:, 0], pools[i % layers][:, 1], r2t, idx, sl, D ** -0.5, ps, out=o1, workspace=ws,
                                           max_splits=ms, chunk_tokens=chunk))
    # matrix-core kernel: prefix = all but the last (S % 16 or 16) tokens
    pre = [(n - 1) // 16 * 16 for n in lens]
    extra = [-(-p // key_chunk) for p in pre]
    ms2 = max(extra) + 1
    ws2 = torch.empty(ops.decode_workspace_bytes(bs, Hq, D, ms2), dtype=torch.uint8, device=dev)
    start, ex = i32(pre), i32(extra)
    gf, gc, gp = i32(list(range(bs))), i32([1] * bs), i32(pre)
    o2 = torch.empty_like(q)

    def f2(i):
        kv = pools[i % layers]
        ops.shared_prefix_decode(q, kv[:, 0], kv[:, 1], r2t, idx, sl, start, gf, gc, gp, 1, max(pre), key_chunk, D ** -0.5, ws2, ms2, 16)
        ops.paged_decode(q, kv[:, 0], kv[:, 1], r2t, idx, sl, D ** -0.5, ps, out=o2, workspace=ws2, max_splits=ms2, chunk_tokens=16,
                         start_lens=start, extra_splits=ex)
    t2 = timeit(f2)
    f2(0); ops.paged_decode(q, pools[0][:, 0], pools[0][:, 1], r2t, idx, sl, D ** -0.5, ps, out=o1, workspace=ws, max_splits=ms, chunk_tokens=chunk)
    torch.cuda.synchronize()
    err = (o1.float() - o2.float()).abs().max().item()
    tot = b * sum(lens)
    print(f"{name} key_chunk {key_chunk}: gather {t1*1e6:.1f} us = {tot/t1/1e9:.0f} GB/s ({tot/t1/8e12:.3f}); mfma {t2*1e6:.1f} us = {tot/t2/1e9:.0f} GB/s ({tot/t2/8e12:.3f}); diff {err:.4f}", flush=True)


for kc in (1024, 2048, 4096, 8192):
    run("32b_tp8_bs8_128k", 8, 1, 64, [131072] * 8, kc)
for kc in (512, 1024, 2048):
    run("32b_tp8_bs1_128k", 8, 1, 64, [131072], kc, layers=8, steps=20)
for kc in (1024, 2048, 4096):
    run("32b_tp1_bs64_8k", 64, 8, 16, [8192] * 64, kc, layers=2)
for kc in (2048, 4096):
    run("32b_tp1_bs256_8k", 64, 8, 16, [8192] * 256, kc, layers=2, steps=6)
