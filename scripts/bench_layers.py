"""HBM bandwidth of the fused layer kernels at decode (T=256) and prefill (T=16384) sizes, Qwen3-8B shapes."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_sglang_amd import ops
from mini_sglang_amd.layers.rotary_embedding import build_cos_sin_cache

dev = "cuda"


def timeit(f, steps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def report(name, T, nbytes, ms):
    print(json.dumps(dict(op=name, T=T, us=round(ms * 1e3, 2), GBps=round(nbytes / ms / 1e6, 1))), flush=True)


hid, inter, Hq, Hkv, D = 4096, 12288, 32, 8, 128
cache = build_cos_sin_cache(D, 40960, 1e6).to(dev)
for T in (256, 16384):
    # rotate over several buffers so that the prefill-size runs are not served from the Infinity Cache
    nb = 1 if T <= 256 else 6
    xs = [torch.randn(T, hid, device=dev).bfloat16() for _ in range(nb)]
    rs = [torch.randn(T, hid, device=dev).bfloat16() for _ in range(nb)]
    w = torch.ones(hid, device=dev).bfloat16()
    outs = [torch.empty_like(x) for x in xs]
    it = [0]
    def nxt():
        it[0] = (it[0] + 1) % nb
        return it[0]
    report("rmsnorm", T, 2 * T * hid * 2, timeit(lambda: ops.rmsnorm(xs[nxt()], w, 1e-6, out=outs[it[0]])))
    report("fused_add_rmsnorm", T, 4 * T * hid * 2, timeit(lambda: ops.fused_add_rmsnorm(xs[nxt()], rs[it[0]], w, 1e-6)))
    gs = [torch.randn(T, 2 * inter, device=dev).bfloat16() for _ in range(nb)]
    go = [torch.empty(T, inter, device=dev, dtype=torch.bfloat16) for _ in range(nb)]
    report("silu_and_mul", T, 3 * T * inter * 2, timeit(lambda: ops.silu_and_mul(gs[nxt()], out=go[it[0]])))
    qkvs = [torch.randn(T, (Hq + 2 * Hkv) * D, device=dev).bfloat16() for _ in range(nb)]
    pos = torch.randint(0, 40960, (T,), device=dev, dtype=torch.int32)
    def rope():
        t = qkvs[nxt()]
        ops.rope_neox_(pos, t[:, : Hq * D], t[:, Hq * D: (Hq + Hkv) * D], cache, D)
    report("rope_neox", T, 2 * T * (Hq + Hkv) * D * 2 + T * D * 4, timeit(rope))
    slots = T + 64
    kv = torch.zeros(slots, 2, Hkv, D, device=dev, dtype=torch.bfloat16)
    loc = (torch.randperm(slots - 16, device=dev)[:T] + 16).to(torch.int32)
    qw = torch.ones(D, device=dev).bfloat16()
    def fused():
        ops.qknorm_rope_store_(qkvs[nxt()], qw, qw, 1e-6, pos, cache, Hq, Hkv, D, loc=loc, k_cache=kv[:, 0], v_cache=kv[:, 1])
    # qkv read once, q+k written back, k+v written to the pool, cos/sin row
    report("qknorm_rope_store", T, T * ((Hq + 2 * Hkv) * D * 2 + (Hq + Hkv) * D * 2 + 2 * Hkv * D * 2 + D * 4), timeit(fused))
    def qk_rms():
        t = qkvs[nxt()]
        ops.rmsnorm(t[:, : Hq * D].reshape(-1, D) if T <= 256 else t.view(-1, D)[: T * Hq], qw, 1e-6)
    logits = torch.randn(min(T, 256), 151936, device=dev).bfloat16()
    report("argmax", min(T, 256), logits.numel() * 2, timeit(lambda: ops.argmax(logits)))
