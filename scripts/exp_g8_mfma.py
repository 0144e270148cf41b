"""G = 8 decode: the matrix-core kernel against the vector gather kernel (msgl_set_variant), one process each."""
import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for mode in ("mfma", "gather"):
        subprocess.run([sys.executable, "-u", __file__, mode], check=False)
    sys.exit(0)
import torch
from mini_sglang_amd import ops
ops.set_variant("decode_mfma", 8 if sys.argv[1] == "mfma" else 0)
from mini_sglang_amd.attn.hip_attn_backend import splits_for_batch
from mini_sglang_amd.runtime import workload

dev = torch.device("cuda")
D = 128


def run(name, Hq, Hkv, ps, lens, layers=4, steps=12, chunk=None):
    bs = len(lens)
    pages = sum(-(-n // ps) for n in lens) + 8
    pt = workload.make_page_table(lens, ps, pages, 3)
    slots = (pages + 1) * ps
    pools = [torch.empty((slots, 2, Hkv, D), dtype=torch.bfloat16, device=dev).normal_() for _ in range(layers)]
    q = torch.randn((bs, Hq, D), device=dev).bfloat16()
    r2t, sl = pt.req_to_token.to(dev), pt.seq_lens.to(dev)
    idx = torch.arange(bs, dtype=torch.int32, device=dev)
    ms = splits_for_batch(bs)
    ck = chunk or ops.pick_split(max(lens), sum(lens), bs, ms, Hkv)
    ms = max(1, min(ms, -(-max(lens) // ck)))
    ws = torch.empty(max(16, ops.decode_workspace_bytes(bs, Hq, D, ms)), dtype=torch.uint8, device=dev)
    o1 = torch.empty_like(q)
    f = lambda i: ops.paged_decode(q, pools[i % layers][:, 0], pools[i % layers][:, 1], r2t, idx, sl, D ** -0.5, ps, out=o1, workspace=ws,
                                   max_splits=ms, chunk_tokens=ck)
    for i in range(3):
        f(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(steps):
            f(i)
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e-3)
    t = sorted(ts)[1]
    tot = 2 * Hkv * D * 2 * sum(lens)
    print(f"{sys.argv[1]:6s} {name:24s} chunk {ck:5d} splits {ms:3d}: {t*1e6:8.1f} us  {tot/t/1e9:6.0f} GB/s ({tot/t/8e12:.3f})  checksum {o1.float().sum().item():.3f}", flush=True)


run("32b_tp8_bs8_128k", 8, 1, 64, [131072] * 8)
for ck in (1024, 2048, 8192):
    run("32b_tp8_bs8_128k", 8, 1, 64, [131072] * 8, chunk=ck)
run("32b_tp8_bs1_128k", 8, 1, 64, [131072], layers=8, steps=20)
run("32b_tp8_bs1_128k", 8, 1, 64, [131072], layers=8, steps=20, chunk=256)
run("32b_tp1_bs64_8k", 64, 8, 16, [8192] * 64, layers=2)
run("32b_tp1_bs256_8k", 64, 8, 16, [8192] * 256, layers=2, steps=6)
run("32b_tp2_bs64_8k", 32, 4, 16, [8192] * 64, layers=2)
