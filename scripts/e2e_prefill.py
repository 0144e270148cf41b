"""Whole-model prefill (extend) throughput: Qwen3-shaped stack, random weights, prompts of --len tokens."""
import sys, os, time, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mini_sglang_amd.runtime.qwen3 import Qwen3Config
from mini_sglang_amd.runtime.model_runner import ModelConfig, ModelRunner, ServerArgs
from mini_sglang_amd.runtime.batch_info import BatchInfo, Req, SamplingParams
from mini_sglang_amd.runtime import synth

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="8b"); ap.add_argument("--bs", type=int, default=1); ap.add_argument("--len", type=int, default=16384)
ap.add_argument("--ps", type=int, default=16); ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
torch.cuda.set_device(0)
cfg = {"8b": Qwen3Config.qwen3_8b, "0.6b": Qwen3Config.qwen3_0_6b, "32b": Qwen3Config.qwen3_32b}[a.model]()
cfg.max_position_embeddings = max(a.len + 64, 2048)
tokens = a.bs * a.len
args = ServerArgs(page_size=a.ps, device="cuda", disable_cuda_graph=True, max_num_reqs=a.bs, num_kv_tokens=(a.steps + 2) * (tokens + a.bs * a.ps),
                  weight_std=0.01, embed_std=0.3)
mr = ModelRunner(ModelConfig(cfg), args)
times = []
for it in range(a.steps + 1):
    reqs = [Req(i, synth.uniform_int(a.len, 100 * it + i, 0, cfg.vocab_size).tolist(), SamplingParams(0.0)) for i in range(a.bs)]
    batch = BatchInfo.init_new(reqs, mr.req_to_token_pool, mr.page_allocator, mr.kv_cache_pool)
    batch.prepare_for_extend()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    logits, _ = mr.forward_generate(batch)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
    for r in reqs:
        mr.req_to_token_pool.free(r.req_pool_idx)
dt = min(times[1:])
print(json.dumps(dict(model=a.model, bs=a.bs, prompt=a.len, ps=a.ps, ms=round(dt * 1e3, 2), tok_per_s=round(tokens / dt, 1))))
