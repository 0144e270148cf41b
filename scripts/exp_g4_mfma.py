"""G = 4 decode (Qwen3-8B heads and its TP shards): the vector gather kernel against the matrix-core kernel, A/B in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_sglang_amd import ops
from mini_sglang_amd.attn.hip_attn_backend import splits_for_batch
from mini_sglang_amd.runtime import workload

dev = torch.device("cuda")
D = 128


def build(Hq, Hkv, ps, lens, layers):
    bs = len(lens)
    pages = sum(-(-n // ps) for n in lens) + 8
    pt = workload.make_page_table(lens, ps, pages, 3)
    pools = [torch.empty(((pages + 1) * ps, 2, Hkv, D), dtype=torch.bfloat16, device=dev).normal_() for _ in range(layers)]
    q = torch.randn((bs, Hq, D), device=dev).bfloat16()
    return pools, q, pt.req_to_token.to(dev), pt.seq_lens.to(dev), torch.arange(bs, dtype=torch.int32, device=dev)


def run(name, Hq, Hkv, ps, lens, layers=4, steps=8):
    pools, q, r2t, sl, idx = build(Hq, Hkv, ps, lens, layers)
    bs = len(lens)
    ms = splits_for_batch(bs)
    ck = ops.pick_split(max(lens), sum(lens), bs, ms, Hkv)
    ms = max(1, min(ms, -(-max(lens) // ck)))
    ws = torch.empty(max(16, ops.decode_workspace_bytes(bs, Hq, D, ms)), dtype=torch.uint8, device=dev)
    o = torch.empty_like(q)
    f = lambda i: ops.paged_decode(q, pools[i % layers][:, 0], pools[i % layers][:, 1], r2t, idx, sl, D ** -0.5, ps, out=o, workspace=ws, max_splits=ms, chunk_tokens=ck)
    res = {}
    outs = {}
    for rnd in range(2):
        for mode in ("0", "4"):
            ops.set_variant("decode_mfma", int(mode))
            for i in range(2):
                f(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(steps):
                    f(i)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / steps * 1e-3)
            f(0); torch.cuda.synchronize(); outs[mode] = o.clone()
    tot = 2 * Hkv * D * 2 * sum(lens)
    d = (outs["0"].float() - outs["4"].float()).abs().max().item()
    print(f"{name:26s} chunk {ck:5d}: gather {min(res['0'])*1e6:8.1f} us ({tot/min(res['0'])/8e12:.3f})  mfma {min(res['4'])*1e6:8.1f} us ({tot/min(res['4'])/8e12:.3f})  diff {d:.4f}", flush=True)


run("8b_tp1_bs256_8k", 32, 8, 16, [8192] * 256, layers=4, steps=6)
run("8b_tp2_bs256_8k", 16, 4, 16, [8192] * 256)
run("8b_tp4_bs256_8k", 8, 2, 16, [8192] * 256)
run("8b_tp8_bs256_8k", 4, 1, 16, [8192] * 256)
run("8b_tp8_bs64_8k", 4, 1, 16, [8192] * 64)
run("8b_tp1_bs16_8k", 32, 8, 16, [8192] * 16)
run("8b_tp1_bs1_32k", 32, 8, 16, [32768], layers=8, steps=16)
