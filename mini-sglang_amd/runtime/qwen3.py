"""A Qwen3-shaped module stack around the hot path — the call-site order of the reference model
(mini_sglang/models/qwen3.py:80-97, 122-126, 162-178, 196-207, 225-232) with the dense GEMMs left to
``F.linear`` (hipBLASLt) and everything else on this package's kernels.

It exists to drive the path end to end (greedy streams, decode tok/s, tensor-parallel runs); it is not a
weight-loading model zoo.  Tensor parallelism follows the reference: heads and MLP columns sharded over the
default process group, ``all_reduce`` after o_proj / down_proj (mini_sglang/layers/linear.py:244-248),
vocab-parallel embedding (embed_head.py:45-53) and an ``all_gather`` of logits
(logits_processor.py:61-64).

``layers`` is the namespace the layer classes come from (default: this package's HIP layers); the tests
pass a namespace of oracle-backed modules to run the identical stack through the CPU restatement.
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from ..comm import tp_all_gather, tp_all_reduce


@dataclasses.dataclass
class Qwen3Config:
    hidden_size: int = 1024
    num_hidden_layers: int = 28
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 3072
    vocab_size: int = 151936
    max_position_embeddings: int = 40960
    rope_theta: float = 1000000.0
    rms_norm_eps: float = 1e-6
    tie_word_embeddings: bool = True
    # synthetic-weight knob, not part of the architecture: std of a random (untied, non-successor) lm_head; None = the
    # std of the other matrices
    lm_head_std: Optional[float] = None

    @staticmethod
    def qwen3_0_6b(**kw):
        return Qwen3Config(**kw)

    @staticmethod
    def qwen3_8b(**kw):
        return Qwen3Config(hidden_size=4096, num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8,
                           intermediate_size=12288, tie_word_embeddings=False, **kw)

    @staticmethod
    def qwen3_32b(**kw):
        return Qwen3Config(hidden_size=5120, num_hidden_layers=64, num_attention_heads=64, num_key_value_heads=8,
                           intermediate_size=25600, tie_word_embeddings=False, **kw)


def _tp() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _default_layers():
    from .. import layers
    return layers


class Qwen3Attention(nn.Module):
    def __init__(self, cfg: Qwen3Config, layer_id: int, layers):
        super().__init__()
        _, tp = _tp()
        assert cfg.num_attention_heads % tp == 0 and cfg.num_key_value_heads % tp == 0
        self.num_heads = cfg.num_attention_heads // tp
        self.num_kv_heads = cfg.num_key_value_heads // tp
        self.head_dim = cfg.head_dim
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.eps = cfg.rms_norm_eps
        self.qkv_proj = nn.Parameter(torch.empty(self.q_size + 2 * self.kv_size, cfg.hidden_size))
        self.o_proj = nn.Parameter(torch.empty(cfg.hidden_size, self.q_size))
        self.rotary_emb = layers.get_rope(self.head_dim, rotary_dim=self.head_dim, max_position=cfg.max_position_embeddings,
                                          base=cfg.rope_theta)
        self.attn = layers.Attention(self.num_heads, self.head_dim, self.head_dim ** -0.5, self.num_kv_heads, layer_id=layer_id)
        self.q_norm = layers.RMSNorm(self.head_dim, eps=cfg.rms_norm_eps)
        self.k_norm = layers.RMSNorm(self.head_dim, eps=cfg.rms_norm_eps)
        # one kernel for q/k-norm + RoPE + KV store when the layer namespace offers it
        self.fused_qk = getattr(layers, "qknorm_rope_store", None)

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor, batch) -> torch.Tensor:
        qkv = F.linear(hidden_states, self.qkv_proj)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        if self.fused_qk is not None:
            k_cache, v_cache = batch.kv_cache_pool.get_kv_buffer(self.attn.layer_id)
            self.fused_qk(qkv, self.q_norm.weight.data, self.k_norm.weight.data, self.eps, positions,
                          self.rotary_emb.cos_sin_cache, self.num_heads, self.num_kv_heads, self.head_dim,
                          batch.out_cache_loc, k_cache, v_cache, self.rotary_emb.reference_rounding)
            o = self.attn(q, k, v, batch, save_kv_cache=False)
        else:  # the reference's op order, one module call per op (qwen3.py:88-95)
            q = self.q_norm(q.reshape(-1, self.head_dim)).view(q.shape)
            k = self.k_norm(k.reshape(-1, self.head_dim)).view(k.shape)
            q, k = self.rotary_emb(positions, q, k)
            o = self.attn(q, k, v, batch)
        out = F.linear(o, self.o_proj)
        if _tp()[1] > 1:
            tp_all_reduce(out)
        return out


class Qwen3MLP(nn.Module):
    def __init__(self, cfg: Qwen3Config, layers):
        super().__init__()
        _, tp = _tp()
        assert cfg.intermediate_size % tp == 0
        inter = cfg.intermediate_size // tp
        self.gate_up_proj = nn.Parameter(torch.empty(2 * inter, cfg.hidden_size))
        self.down_proj = nn.Parameter(torch.empty(cfg.hidden_size, inter))
        self.act_fn = layers.SiluAndMul()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.act_fn(F.linear(x, self.gate_up_proj))
        x = F.linear(x, self.down_proj)
        if _tp()[1] > 1:
            tp_all_reduce(x)
        return x


class Qwen3DecoderLayer(nn.Module):
    def __init__(self, cfg: Qwen3Config, layer_id: int, layers):
        super().__init__()
        self.self_attn = Qwen3Attention(cfg, layer_id, layers)
        self.mlp = Qwen3MLP(cfg, layers)
        self.input_layernorm = layers.RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)
        self.post_attention_layernorm = layers.RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)

    def forward(self, positions, hidden_states, batch, residual: Optional[torch.Tensor]):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, batch)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class Qwen3ForCausalLM(nn.Module):
    def __init__(self, cfg: Qwen3Config, layers=None):
        super().__init__()
        layers = layers or _default_layers()
        self.config = cfg
        rank, tp = _tp()
        assert cfg.vocab_size % tp == 0
        self.vocab_per_rank = cfg.vocab_size // tp
        self.vocab_start = rank * self.vocab_per_rank
        self.embed_tokens = nn.Parameter(torch.empty(self.vocab_per_rank, cfg.hidden_size))
        self.layers = nn.ModuleList([Qwen3DecoderLayer(cfg, i, layers) for i in range(cfg.num_hidden_layers)])
        self.norm = layers.RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)
        self.lm_head = self.embed_tokens if cfg.tie_word_embeddings else nn.Parameter(torch.empty(self.vocab_per_rank, cfg.hidden_size))

    def embed(self, input_ids: torch.Tensor) -> torch.Tensor:
        _, tp = _tp()
        if tp == 1:
            return F.embedding(input_ids, self.embed_tokens)
        mask = (input_ids >= self.vocab_start) & (input_ids < self.vocab_start + self.vocab_per_rank)
        y = F.embedding(mask * (input_ids - self.vocab_start), self.embed_tokens)
        y = mask.unsqueeze(1) * y
        tp_all_reduce(y)
        return y

    def forward_hidden(self, input_ids: torch.Tensor, positions: torch.Tensor, batch) -> torch.Tensor:
        """Final-norm hidden state of the token each request samples from: [bs, hidden]."""
        hidden_states = self.embed(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, batch, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        if batch.forward_mode.is_extend():
            last = torch.cumsum(batch.input_seq_lens, dim=0, dtype=torch.int32) - 1
            hidden_states = hidden_states[last.long()].contiguous()
        return hidden_states

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, batch) -> torch.Tensor:
        """Returns next-token logits [bs, vocab] in the model dtype (logits_processor.py:47-67)."""
        logits = F.linear(self.forward_hidden(input_ids, positions, batch), self.lm_head)
        _, tp = _tp()
        if tp > 1:
            logits = torch.cat(tp_all_gather(logits), dim=-1)
        return logits

    def forward_greedy(self, input_ids: torch.Tensor, positions: torch.Tensor, batch, argmax_fn, out: Optional[torch.Tensor] = None,
                       head_argmax_fn=None):
        """Greedy next tokens, int32 [bs], equal to argmax(forward(...).float(), -1) (sampler.py:14-15) without ever
        materialising the full-vocabulary logits under tensor parallelism: every rank reduces its own vocabulary
        shard with ``argmax_fn(local_logits) -> (index int32 [bs], value fp32 [bs])`` and the ranks exchange bs
        (value, index) pairs — 2 KB at batch 256 instead of the 78 MB all-gather of logits_processor.py:56-64.
        Ties between shards go to the lower rank, i.e. the lower token id, as a single argmax would decide.
        ``head_argmax_fn(hidden, lm_head_rows) -> (index, value)``, when given, replaces the head GEMM and ``argmax_fn``
        (logits_processor.py:56-59 + sampler.py:14-15 fused)."""
        hidden = self.forward_hidden(input_ids, positions, batch)
        if head_argmax_fn is not None:
            # head product and reduction in one pass: not even the rank's own [bs, vocab / tp] logits are written
            idx, val = head_argmax_fn(hidden, self.lm_head)
        else:
            idx, val = argmax_fn(F.linear(hidden, self.lm_head))
        idx = pick_sharded_argmax(idx, val, self.vocab_per_rank)
        if out is not None:
            out.copy_(idx)
            return out
        return idx


def pick_sharded_argmax(idx: torch.Tensor, val: torch.Tensor, vocab_per_rank: int) -> torch.Tensor:
    """Global greedy token from per-rank winners: ``idx`` int32 [bs] (index inside this rank's vocabulary shard) and
    ``val`` fp32 [bs] (its logit).  One all-gather of bs (value, index) pairs; the first maximum over ranks wins, so
    ties go to the lower token id and a NaN counts as the maximum, as torch.argmax over the full row decides."""
    _, tp = _tp()
    if tp == 1:
        return idx
    pair = torch.stack([val, idx.view(torch.float32)], dim=-1)   # index bits travel unchanged in an fp32 slot
    allp = torch.stack(tp_all_gather(pair))                      # [tp, bs, 2]
    win = torch.argmax(allp[..., 0], dim=0, keepdim=True)        # [1, bs]
    local = allp[..., 1].contiguous().view(torch.int32).gather(0, win)[0]
    return local + (win[0] * vocab_per_rank).to(torch.int32)


def init_weights(model: Qwen3ForCausalLM, seed: int, std: float = 0.02, embed_std: Optional[float] = None,
                 norm_jitter: float = 0.1, successor_head: float = 0.0) -> None:
    """Deterministic random weights, identical on every machine and for every TP degree: each FULL
    (unsharded) matrix is generated from its own seed and the local shard is cut out of it, following the
    reference's sharding (q/k/v and gate/up shards stacked per rank, linear.py:126-138, 182-205; row
    shards for o_proj / down_proj, linear.py:250-258; vocab rows, embed_head.py:37-43).

    successor_head = a > 0 (untied models only) sets lm_head row j to a * embedding row (j - 1): the residual
    stream carries the current token's embedding, so the greedy continuation of token t is t + 1 with a
    top-2 margin of many bf16 ulps at every step.  A purely random head over a 151 936-entry vocabulary
    has a top-2 gap of about one bf16 ulp at one step in ten, which makes 256-token greedy streams a coin
    toss between any two correct implementations; the successor head keeps the stream a meaningful check
    of positions, page tables and KV writes while every layer still shapes the logits that are compared."""
    from . import synth

    cfg = model.config
    rank, tp = _tp()
    dev = model.embed_tokens.device
    dt = model.embed_tokens.dtype
    D, H, Hkv = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
    hid, inter = cfg.hidden_size, cfg.intermediate_size

    def full(shape, s, sd=std):
        return synth.normal(shape, seed * 1000003 + s, std=sd, dtype=torch.float32, device=dev)

    def rows(t, n_total):  # this rank's contiguous row block
        per = n_total // tp
        return t[rank * per: (rank + 1) * per]

    with torch.no_grad():
        emb = full((cfg.vocab_size, hid), 1, embed_std if embed_std is not None else std)
        model.embed_tokens.copy_(rows(emb, cfg.vocab_size).to(dt))
        if not cfg.tie_word_embeddings:
            head = torch.roll(emb, shifts=1, dims=0) * successor_head if successor_head > 0 else full(
                (cfg.vocab_size, hid), 2, cfg.lm_head_std if cfg.lm_head_std is not None else std)
            model.lm_head.copy_(rows(head, cfg.vocab_size).to(dt))
        elif successor_head > 0:
            raise ValueError("successor_head needs an untied lm_head (tie_word_embeddings=False)")
        model.norm.weight.copy_((1.0 + full((hid,), 3, norm_jitter)).to(dt))
        for i, layer in enumerate(model.layers):
            b = 100 + i * 20
            at = layer.self_attn
            wq, wk, wv = full((H * D, hid), b + 1), full((Hkv * D, hid), b + 2), full((Hkv * D, hid), b + 3)
            at.qkv_proj.copy_(torch.cat([rows(wq, H * D), rows(wk, Hkv * D), rows(wv, Hkv * D)]).to(dt))
            wo = full((hid, H * D), b + 4)
            per = H * D // tp
            at.o_proj.copy_(wo[:, rank * per: (rank + 1) * per].to(dt))
            at.q_norm.weight.copy_((1.0 + full((D,), b + 5, norm_jitter)).to(dt))
            at.k_norm.weight.copy_((1.0 + full((D,), b + 6, norm_jitter)).to(dt))
            wg, wu = full((inter, hid), b + 7), full((inter, hid), b + 8)
            layer.mlp.gate_up_proj.copy_(torch.cat([rows(wg, inter), rows(wu, inter)]).to(dt))
            wd = full((hid, inter), b + 9)
            per_i = inter // tp
            layer.mlp.down_proj.copy_(wd[:, rank * per_i: (rank + 1) * per_i].to(dt))
            layer.input_layernorm.weight.copy_((1.0 + full((hid,), b + 10, norm_jitter)).to(dt))
            layer.post_attention_layernorm.weight.copy_((1.0 + full((hid,), b + 11, norm_jitter)).to(dt))
