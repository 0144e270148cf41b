"""Cost of always capturing the shared-prefix kernel: whole-model decode of a batch in which nothing is shared,
captured with and without it (ms per step)."""
import gc, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_sglang_amd.runtime import decode_loop
from mini_sglang_amd.runtime.qwen3 import Qwen3Config

for bs, ctx in ((256, 4096), (16, 4096)):
    for share in (False, True, False, True):
        mr, batch = decode_loop.build_decode_state(Qwen3Config.qwen3_8b(), bs, ctx, 16, 64, use_graph=True, share_prefix_in_graphs=share)
        dt = decode_loop.time_decode(mr, batch, 16, 4)
        print(json.dumps(dict(bs=bs, ctx=ctx, share_prefix_in_graphs=share, ms_per_step=round(dt / 16 * 1e3, 3))), flush=True)
        del mr, batch
        gc.collect(); torch.cuda.empty_cache()
