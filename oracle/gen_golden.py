#!/usr/bin/env python3
"""Generate tests/golden/*.pt by running the REFERENCE's own torch-native path on seeded inputs.

Run in the build container only (the reference tree is read-only and does not travel):
    python oracle/gen_golden.py [--out tests/golden] [--ref /root/reference]

The reference package is imported in place, without executing its ``__init__`` (which pulls the serving
shell: zmq, setproctitle, uvloop): a namespace stub points ``mini_sglang.__path__`` at the tree and an empty
``zmq`` module satisfies utils/utils.py.  ``sgl_kernel`` and ``flash_attn`` are absent, so RMSNorm and RoPE
take their torch fallbacks — the "torch-native attention/norm path" the parity target names.  Inputs come
from this repo's counter-based generator (mini_sglang_amd.runtime.synth), so a fixture stores only seeds,
shapes and the reference's OUTPUTS.

Fixtures:
  ops.pt        RMSNorm (plain / fused-add), SiluAndMul, RotaryEmbedding outputs
  attention.pt  TorchNativeAttnBackend.forward_decode / forward_extend on hand-built pools and page tables
  e2e_*.pt      ModelRunner + BatchInfo greedy generation (the test_one_batch sequence) on a Qwen3-shaped
                model with this repo's deterministic weights: token streams, top-2 margins, logits slices,
                page tables and out_cache_loc sequences
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mini_sglang_amd.runtime import synth, workload  # noqa: E402
from mini_sglang_amd.runtime.qwen3 import Qwen3Config  # noqa: E402


def import_reference(ref_root: str):
    pkg = types.ModuleType("mini_sglang")
    pkg.__path__ = [os.path.join(ref_root, "mini_sglang")]
    sys.modules["mini_sglang"] = pkg
    sys.modules.setdefault("zmq", types.ModuleType("zmq"))
    import torch.distributed as dist

    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="gloo", rank=0, world_size=1)


# --------------------------------------------------------------------------------------------- per-op
OPS_CASES = {
    "rmsnorm": [(5, 128, 1), (3, 1024, 2), (2, 4096, 3), (2, 5120, 4)],             # rows, cols, seed
    "fused_add_rmsnorm": [(3, 1024, 5), (2, 4096, 6), (2, 5120, 7)],
    "silu_and_mul": [(3, 3072, 8), (2, 12288, 9)],                                   # rows, n, seed
    "rope": [(12, 4, 2, 128, 10), (7, 8, 1, 128, 11)],                               # T, Hq, Hkv, D, seed
}


def rope_positions(T: int, seed: int, max_pos: int = 40960) -> torch.Tensor:
    return torch.cat([torch.tensor([0, 1, max_pos - 1], dtype=torch.int32),
                      synth.uniform_int(T - 3, seed, 0, max_pos).to(torch.int32)])


def gen_ops(out_dir: str):
    from mini_sglang.layers.activation import SiluAndMul
    from mini_sglang.layers.layernorm import RMSNorm
    from mini_sglang.layers.rotary_embedding import RotaryEmbedding

    fx = {}
    for rows, cols, seed in OPS_CASES["rmsnorm"]:
        mod = RMSNorm(cols, eps=1e-6).to(torch.bfloat16)
        mod.weight.data.copy_(synth.normal((cols,), seed + 100, std=0.5) + 1)
        fx[f"rmsnorm/{rows}x{cols}/{seed}"] = mod(synth.normal((rows, cols), seed, std=2.0)).detach()
    for rows, cols, seed in OPS_CASES["fused_add_rmsnorm"]:
        mod = RMSNorm(cols, eps=1e-6).to(torch.bfloat16)
        mod.weight.data.copy_(synth.normal((cols,), seed + 100, std=0.5) + 1)
        y, r = mod(synth.normal((rows, cols), seed), synth.normal((rows, cols), seed + 200, std=3.0))
        fx[f"fused_add_rmsnorm/{rows}x{cols}/{seed}/y"] = y.detach()
        fx[f"fused_add_rmsnorm/{rows}x{cols}/{seed}/residual"] = r.detach()
    for rows, n, seed in OPS_CASES["silu_and_mul"]:
        fx[f"silu_and_mul/{rows}x{n}/{seed}"] = SiluAndMul()(synth.normal((rows, 2 * n), seed, std=2.0))
    for T, Hq, Hkv, D, seed in OPS_CASES["rope"]:
        rope = RotaryEmbedding(D, D, 40960, 1e6)
        q, k = synth.normal((T, Hq * D), seed), synth.normal((T, Hkv * D), seed + 300)
        qo, ko = rope(rope_positions(T, seed + 400), q, k)
        fx[f"rope/{T}x{Hq}x{Hkv}x{D}/{seed}/q"] = qo
        fx[f"rope/{T}x{Hq}x{Hkv}x{D}/{seed}/k"] = ko
        fx[f"rope/{T}x{Hq}x{Hkv}x{D}/{seed}/cache_rows"] = rope.cos_sin_cache[[0, 1, 40959]].clone()
    torch.save(fx, os.path.join(out_dir, "ops.pt"))
    print("ops.pt:", len(fx), "tensors")


# --------------------------------------------------------------------------------------------- attention
ATTN_DECODE_CASES = [  # Hq, Hkv, page_size, seq_lens, groups, prefix, seed
    (16, 8, 1, [1, 2, 17, 100, 1000], None, None, 20),
    (32, 8, 16, [1, 15, 16, 17, 300, 1029], None, None, 21),
    (32, 8, 16, [300, 256, 1000, 64, 65, 40], [0, 0, 0, 1, 1, 2], [256, 64, 0], 22),
    (8, 1, 64, [63, 64, 65, 2000], None, None, 23),
    (64, 8, 16, [5, 130], None, None, 24),
]
ATTN_EXTEND_CASES = [  # Hq, Hkv, page_size, prefix_lens, input_lens, seed
    (4, 2, 1, [0, 0, 7], [1, 33, 70], 30),
    (16, 8, 16, [0, 32, 256], [130, 31, 40], 31),
    (8, 1, 64, [0, 128], [64, 70], 32),
]
EXTEND_COL_STRIDE = 4  # extend fixtures keep every 4th output column (size)


def decode_case_tensors(Hq, Hkv, ps, seq, groups, prefix, seed):
    pool_pages = sum(-(-s // ps) for s in seq) + 8
    pt = workload.make_page_table(seq, ps, pool_pages, seed, groups=groups, group_prefix_lens=prefix)
    slots = (pool_pages + 1) * ps
    kc, vc = synth.normal((slots, Hkv, 128), seed + 1), synth.normal((slots, Hkv, 128), seed + 2)
    q = synth.normal((len(seq), Hq * 128), seed + 3)
    k_new, v_new = synth.normal((len(seq), Hkv, 128), seed + 4), synth.normal((len(seq), Hkv, 128), seed + 5)
    return pt, kc, vc, q, k_new, v_new


def extend_case_tensors(Hq, Hkv, ps, prefix, inp, seed):
    seq = [p + n for p, n in zip(prefix, inp)]
    pool_pages = sum(-(-s // ps) for s in seq) + 8
    pt = workload.make_page_table(seq, ps, pool_pages, seed)
    slots = (pool_pages + 1) * ps
    kc, vc = synth.normal((slots, Hkv, 128), seed + 1), synth.normal((slots, Hkv, 128), seed + 2)
    T = sum(inp)
    q = synth.normal((T, Hq * 128), seed + 3)
    k_new, v_new = synth.normal((T, Hkv, 128), seed + 4), synth.normal((T, Hkv, 128), seed + 5)
    return pt, kc, vc, q, k_new, v_new


def new_token_slots(pt, prefix, inp):
    """Slots of the tokens an extend writes: positions [prefix, prefix + n) of every request."""
    return torch.cat([pt.req_to_token[b, p: p + n] for b, (p, n) in enumerate(zip(prefix, inp))]).to(torch.int32)


def gen_attention(out_dir: str):
    from mini_sglang.layers.attention import Attention
    from mini_sglang.layers.attn.torch_attn_backend import TorchNativeAttnBackend
    from mini_sglang.managers.batch_info import BatchInfo, ForwardMode

    class Pool:  # the two methods the backend calls (token2kv.py:133-151)
        def __init__(self, kc, vc):
            self.k, self.v = kc, vc

        def set_kv_buffer(self, layer, loc, ck, cv):
            self.k[loc] = ck.view(-1, self.k.shape[1], self.k.shape[2])
            self.v[loc] = cv.view(-1, self.v.shape[1], self.v.shape[2])

        def get_key_buffer(self, _):
            return self.k

        def get_value_buffer(self, _):
            return self.v

    fx = {}
    for ci, (Hq, Hkv, ps, seq, groups, prefix, seed) in enumerate(ATTN_DECODE_CASES):
        pt, kc, vc, q, k_new, v_new = decode_case_tensors(Hq, Hkv, ps, seq, groups, prefix, seed)
        runner = types.SimpleNamespace(device="cpu", req_to_token_pool=types.SimpleNamespace(req_to_token=pt.req_to_token),
                                       kv_cache_pool=Pool(kc, vc))
        be = TorchNativeAttnBackend(runner)
        layer = Attention(Hq, 128, 128 ** -0.5, Hkv, layer_id=0)
        # decode semantics: the new token's K/V is stored at the request's LAST position first, then attended
        loc = torch.tensor([int(pt.req_to_token[b, s - 1]) for b, s in enumerate(seq)], dtype=torch.int32)
        batch = BatchInfo(forward_mode=ForwardMode.DECODE, seq_lens=pt.seq_lens, out_cache_loc=loc.long(),
                          req_pool_indices=torch.arange(len(seq), dtype=torch.int32), device="cpu")
        fx[f"decode/{ci}"] = be.forward(q, k_new, v_new, layer, batch)
    for ci, (Hq, Hkv, ps, prefix, inp, seed) in enumerate(ATTN_EXTEND_CASES):
        pt, kc, vc, q, k_new, v_new = extend_case_tensors(Hq, Hkv, ps, prefix, inp, seed)
        runner = types.SimpleNamespace(device="cpu", req_to_token_pool=types.SimpleNamespace(req_to_token=pt.req_to_token),
                                       kv_cache_pool=Pool(kc, vc))
        be = TorchNativeAttnBackend(runner)
        layer = Attention(Hq, 128, 128 ** -0.5, Hkv, layer_id=0)
        batch = BatchInfo(forward_mode=ForwardMode.EXTEND, seq_lens=pt.seq_lens, out_cache_loc=new_token_slots(pt, prefix, inp).long(),
                          req_pool_indices=torch.arange(len(inp), dtype=torch.int32),
                          prefix_seq_lens=torch.tensor(prefix, dtype=torch.int32),
                          input_seq_lens=torch.tensor(inp, dtype=torch.int32), device="cpu")
        fx[f"extend/{ci}"] = be.forward(q, k_new, v_new, layer, batch)[:, ::EXTEND_COL_STRIDE].clone()
    torch.save(fx, os.path.join(out_dir, "attention.pt"))
    print("attention.pt:", len(fx), "tensors")


# --------------------------------------------------------------------------------------------- end to end
E2E_CASES = {
    # name: (config kwargs, page_size, prompts (seed, lengths), new tokens, weight seed, weight std, embed std, successor head)
    "tiny_ps1": (dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512,
                      vocab_size=2048, max_position_embeddings=1024, tie_word_embeddings=True), 1, (41, [5, 19]), 64, 7, 0.05, 0.05, 0.0),
    "tiny_ps16": (dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512,
                       vocab_size=2048, max_position_embeddings=1024, tie_word_embeddings=False), 16, (42, [15, 16, 33]), 64, 8, 0.02, 0.3, 0.04),
    # first configuration of the benchmark family: Qwen3-0.6B shape, page_size 1, 256 greedy tokens, batch 2 (about 6 minutes of host time)
    "qwen3_0.6b_ps1": (dict(max_position_embeddings=2048, tie_word_embeddings=False), 1, (43, [16, 29]), 256, 11, 0.01, 0.3, 0.02),
    # the same shape with a RANDOM head (no successor structure): top-2 margins of a bf16 ulp occur, so two correct
    # implementations need not agree on the stream; compared teacher-forced (the fixture's tokens are fed back) on the
    # logits of all 256 steps
    "qwen3_0.6b_ps1_random_head": (dict(max_position_embeddings=2048, tie_word_embeddings=False, lm_head_std=0.005), 1, (47, [23, 9]), 256, 13,
                                   0.01, 0.3, 0.0),
    # Qwen3-8B layer shapes (hidden 4096, 32 q / 8 kv heads, intermediate 12288, full vocabulary) on four layers, page_size
    # 16, batch 3 with a prompt that ends on a page boundary: the widths the benchmark runs, at a depth the host can compute
    "qwen3_8b_l4_ps16": (dict(hidden_size=4096, num_hidden_layers=4, num_attention_heads=32, num_key_value_heads=8, intermediate_size=12288,
                              max_position_embeddings=2048, tie_word_embeddings=False), 16, (51, [40, 17, 64]), 40, 17, 0.006, 0.3, 0.02),
}
WIDE_COLS = 2048


def wide_slice(logits: torch.Tensor) -> torch.Tensor:
    """WIDE_COLS columns spread evenly over the vocabulary (every column when the vocabulary is that small)."""
    stride = max(1, logits.shape[-1] // WIDE_COLS)
    return logits[:, ::stride][:, :WIDE_COLS]


def e2e_prompts(seed: int, lengths, vocab: int):
    return [synth.uniform_int(n, seed + i, 0, vocab).tolist() for i, n in enumerate(lengths)]


def copy_weights_into_reference(ref_model, mine):
    """TP=1: the packed shapes are identical; copy parameter by parameter."""
    sd = dict(ref_model.named_parameters())
    with torch.no_grad():
        sd["model.embed_tokens.weight"].copy_(mine.embed_tokens)
        if not mine.config.tie_word_embeddings:
            sd["lm_head.weight"].copy_(mine.lm_head)
        sd["model.norm.weight"].copy_(mine.norm.weight)
        for i, layer in enumerate(mine.layers):
            p = f"model.layers.{i}."
            sd[p + "self_attn.qkv_proj.weight"].copy_(layer.self_attn.qkv_proj)
            sd[p + "self_attn.o_proj.weight"].copy_(layer.self_attn.o_proj)
            sd[p + "self_attn.q_norm.weight"].copy_(layer.self_attn.q_norm.weight)
            sd[p + "self_attn.k_norm.weight"].copy_(layer.self_attn.k_norm.weight)
            sd[p + "mlp.gate_up_proj.weight"].copy_(layer.mlp.gate_up_proj)
            sd[p + "mlp.down_proj.weight"].copy_(layer.mlp.down_proj)
            sd[p + "input_layernorm.weight"].copy_(layer.input_layernorm.weight)
            sd[p + "post_attention_layernorm.weight"].copy_(layer.post_attention_layernorm.weight)


def gen_e2e(out_dir: str, name: str, spec):
    import mini_sglang.managers.model_runner as ref_mr
    from mini_sglang.managers.batch_info import BatchInfo
    from mini_sglang.managers.req_info import Req
    from mini_sglang.managers.sampling_params import SamplingParams
    from mini_sglang.managers.server_args import ServerArgs
    from mini_sglang.mem_cache.req2token import ReqToTokenPool
    from mini_sglang.mem_cache.token2kv import MHAKVPool, PageAllocator
    from mini_sglang.utils.model_config import ModelConfig

    from oracle import modules_oracle as mo
    from mini_sglang_amd.runtime.qwen3 import Qwen3ForCausalLM, init_weights

    cfg_kw, ps, (pseed, plens), new_tokens, wseed, wstd, estd, succ = spec
    cfg = Qwen3Config(**cfg_kw)
    ref_mr.get_available_gpu_memory = lambda *a, **k: 0.0
    with tempfile.TemporaryDirectory() as d:
        hf = dict(architectures=["Qwen3ForCausalLM"], model_type="qwen3", hidden_size=cfg.hidden_size,
                  num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                  num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size,
                  vocab_size=cfg.vocab_size, max_position_embeddings=cfg.max_position_embeddings, rope_theta=cfg.rope_theta,
                  rms_norm_eps=cfg.rms_norm_eps, tie_word_embeddings=cfg.tie_word_embeddings, hidden_act="silu",
                  torch_dtype="bfloat16", attention_bias=False, eos_token_id=-1)
        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump(hf, f)
        args = ServerArgs(model=d, attention_backend="torch", page_size=ps, device="cpu", disable_cuda_graph=True,
                          dummy_load=True, random_seed=0, max_num_reqs=8)
        mc = ModelConfig(d)
        max_ctx = cfg.max_position_embeddings
        pages = 1024 // ps
        r2t = ReqToTokenPool(8, max_ctx, ps, "cpu")
        pa = PageAllocator(pages, ps, "cpu")
        kv = MHAKVPool(pages * ps, ps, torch.bfloat16, cfg.num_key_value_heads, cfg.head_dim, cfg.num_hidden_layers, "cpu")
        runner = ref_mr.ModelRunner(mc, args, gpu_id=0, tp_rank=0, tp_size=1, req_to_token_pool=r2t, page_allocator=pa, kv_cache_pool=kv)
    torch.set_default_dtype(torch.bfloat16)
    mine = Qwen3ForCausalLM(cfg, layers=mo)
    torch.set_default_dtype(torch.float32)
    init_weights(mine, wseed, std=wstd, embed_std=estd, successor_head=succ)
    copy_weights_into_reference(runner.model, mine)

    prompts = e2e_prompts(pseed, plens, cfg.vocab_size)
    sp = SamplingParams(temperature=0.0, max_new_tokens=new_tokens + 8, ignore_eos=True)
    reqs = [Req(str(i), p, sp) for i, p in enumerate(prompts)]
    batch = BatchInfo.init_new(reqs=reqs, req_to_token_pool=r2t, page_allocator=pa, kv_cache_pool=kv)
    margins, tops, slices, locs, wide = [], [], [], [], []

    def record(logits):
        wide.append(wide_slice(logits.next_token_logits).clone())
        lf = logits.next_token_logits.float()
        top2 = lf.topk(2, dim=-1).values
        margins.append((top2[:, 0] - top2[:, 1]).clone())
        tops.append(top2[:, 0].clone())
        slices.append(logits.next_token_logits[:, :64].clone())
        locs.append(batch.out_cache_loc.clone().to(torch.int32))

    with torch.inference_mode():
        batch.prepare_for_extend()
        logits = runner.forward_extend(batch)
        record(logits)
        runner.process_extend_result(batch, logits)
        for _ in range(new_tokens - 1):
            batch.prepare_for_decode()
            logits = runner.forward_decode(batch)
            record(logits)
            runner.process_decode_result(batch, logits)
    fx = {
        "tokens": torch.tensor([r.token_ids[len(p):] for r, p in zip(reqs, prompts)], dtype=torch.int32),
        "margins": torch.stack(margins),                      # [steps, bs] fp32 top-1 minus top-2 logit
        "top1_logits": torch.stack(tops),                     # [steps, bs] fp32 value of the winning logit
        "logits_slices": torch.stack(slices),                 # [steps, bs, 64] bf16
        "logits_wide": torch.stack(wide),                     # [steps, bs, 2048] bf16: every (vocab // 2048)-th column
        "out_cache_loc_first": locs[0], "out_cache_loc_decode": torch.stack(locs[1:]),
        "req_to_token_rows": torch.cat([r2t.req_to_token[r.req_pool_idx, : len(r.token_ids) - 1] for r in reqs]),
    }
    torch.save(fx, os.path.join(out_dir, f"e2e_{name}.pt"))
    m = fx["margins"]
    print(f"e2e_{name}.pt: tokens {tuple(fx['tokens'].shape)} min margin {m.min().item():.4f} median {m.median().item():.4f}",
          "distinct tokens per request", [len(set(t.tolist())) for t in fx["tokens"]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    import_reference(a.ref)
    torch.manual_seed(0)
    if a.only in ("", "ops"):
        gen_ops(a.out)
    if a.only in ("", "attention"):
        gen_attention(a.out)
    for name, spec in E2E_CASES.items():
        if a.only in ("", "e2e", name):
            gen_e2e(a.out, name, spec)


if __name__ == "__main__":
    main()
