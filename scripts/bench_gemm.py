"""Decode-step GEMM shapes (M = batch) through the library paths torch offers: weight stored [N, K] (F.linear) or
[K, N] (matmul), bf16.  GB/s of weight bytes."""
import json, sys, os, itertools
import torch
import torch.nn.functional as F
dev = "cuda"
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 24576), "down": (12288, 4096), "lm_head": (4096, 151936)}
for M in (16, 64, 256):
    for name, (K, N) in shapes.items():
        x = torch.randn(M, K, device=dev).bfloat16()
        # several weight copies so that successive calls do not hit the 256 MB cache
        reps = max(1, int(600e6 // (K * N * 2)))
        Ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(reps)]
        Wts = [w.t().contiguous() for w in Ws]
        i = [0]
        def lin():
            i[0] = (i[0] + 1) % reps; return F.linear(x, Ws[i[0]])
        def mm():
            i[0] = (i[0] + 1) % reps; return x @ Wts[i[0]]
        a, b = t(lin), t(mm)
        gb = K * N * 2 / 1e3
        print(json.dumps(dict(M=M, op=name, K=K, N=N, linear_us=round(a, 1), linear_GBps=round(gb / a, 0), mm_us=round(b, 1), mm_GBps=round(gb / b, 0))), flush=True)
        del Ws, Wts
