"""Fused lm_head product + greedy reduction against F.linear + argmax: time per call and the rate at which the head's
weights stream (bytes of the weight matrix / time), Qwen3-8B head (151 936 x 4096) and its TP-8 shard."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_sglang_amd import ops  # noqa: E402


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = "cuda"
    torch.manual_seed(0)
    quick = os.environ.get("LH_QUICK") == "1"   # profiler passes: the full head at two batch sizes only
    for vocab, hidden in ([(151936, 4096)] if quick else [(151936, 4096), (18992, 4096), (151936, 1024)]):
        w = (torch.randn(vocab, hidden, device=dev) * 0.02).to(torch.bfloat16)
        for bs in ((32, 64) if quick else (1, 16, 32, 64, 128, 256)):
            x = torch.randn(bs, hidden, device=dev).to(torch.bfloat16)
            ws = ops.lm_head_argmax_workspace(bs, vocab, dev)
            fused = timed(lambda: ops.lm_head_argmax(x, w, workspace=ws))
            plain = timed(lambda: ops.argmax(torch.nn.functional.linear(x, w), return_values=True))
            wb = vocab * hidden * 2
            print(json.dumps({"vocab": vocab, "hidden": hidden, "bs": bs, "fused_us": round(fused * 1e3, 1), "linear_argmax_us": round(plain * 1e3, 1),
                              "fused_weight_GBps": round(wb / fused / 1e6, 1), "linear_argmax_weight_GBps": round(wb / plain / 1e6, 1),
                              "fused_TFLOPs": round(2 * bs * wb / 2 / fused / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
