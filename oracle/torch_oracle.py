"""CPU oracle: a restatement, in plain torch ops, of the reference's torch-native hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this module; it is used by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker / the CPU
baseline, never as the thing shipped or measured as the product.

Every function follows the reference file:line cited in its docstring (paths relative to the
reference root) and keeps the reference's rounding points: tensors are bf16 between ops, each
torch op computes in fp32 internally and rounds once.  Parity pin: tests/golden/*.pt hold
outputs of the reference's own modules (imported from the read-only reference tree by
oracle/gen_golden.py in the build container) on seeded inputs; tests/test_oracle_golden.py
checks this restatement against them bit for bit.

All functions are device agnostic (they run wherever their inputs live); the oracle proper is
their CPU execution.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- layers
def rmsnorm_ref(x: torch.Tensor, weight: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None):
    """mini_sglang/layers/layernorm.py:39-50 (the torch fallback taken when sgl_kernel is absent).

    Returns ``y`` or ``(y, residual')``.  The norm is taken over the UNROUNDED fp32 sum while the
    returned residual is that sum rounded to the input dtype (layernorm.py:40-46).
    """
    orig_dtype = x.dtype
    xf = x.to(torch.float32)
    if residual is not None:
        xf = xf + residual.to(torch.float32)
        residual = xf.to(orig_dtype)
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    y = (xf * weight).to(orig_dtype)
    if residual is not None:
        return y, residual
    return y


def silu_and_mul_ref(x: torch.Tensor) -> torch.Tensor:
    """mini_sglang/layers/activation.py:12-14."""
    a, b = x.chunk(2, dim=-1)
    return F.silu(a) * b


def build_cos_sin_cache(head_dim: int, max_position: int, base: float, device="cpu") -> torch.Tensor:
    """mini_sglang/layers/rotary_embedding.py:59-67: fp32 [max_position, head_dim] = cat(cos, sin)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(device)


def _apply_rotary(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """rotary_embedding.py:19-33: cos/sin are cast to x's dtype and the rotate-half runs in that dtype."""
    cos = cos.unsqueeze(-2).to(x.dtype)
    sin = sin.unsqueeze(-2).to(x.dtype)
    x1, x2 = torch.chunk(x, 2, dim=-1)
    y1 = x1 * cos - x2 * sin
    y2 = x1 * sin + x2 * cos
    return torch.cat((y1, y2), dim=-1)


def rope_ref(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, cos_sin_cache: torch.Tensor,
             head_dim: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """rotary_embedding.py:86-95 (torch fallback): q [T, Hq*D], k [T, Hkv*D] -> rotated copies."""
    T = positions.size(0)
    cos_sin = cos_sin_cache[positions.long()]
    cos, sin = cos_sin.chunk(2, dim=-1)
    qo = _apply_rotary(q.reshape(T, -1, head_dim), cos, sin).reshape(q.shape)
    ko = _apply_rotary(k.reshape(T, -1, head_dim), cos, sin).reshape(k.shape)
    return qo, ko


def rope_fp32_ref(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, cos_sin_cache: torch.Tensor,
                  head_dim: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Neox rotation in fp32 with one final rounding: the numerics of the fused kernel the reference calls
    when sgl_kernel is present (rotary_embedding.py:77-84).  No reference output pins this variant here
    (sgl_kernel is absent from this environment): parity unpinned, checked only against this formula."""
    T = positions.size(0)
    cos, sin = cos_sin_cache[positions.long()].chunk(2, dim=-1)
    cos, sin = cos.unsqueeze(-2), sin.unsqueeze(-2)

    def rot(x):
        xf = x.reshape(T, -1, head_dim).float()
        x1, x2 = xf.chunk(2, dim=-1)
        return torch.cat((x1 * cos - x2 * sin, x1 * sin + x2 * cos), dim=-1).to(x.dtype).reshape(x.shape)

    return rot(q), rot(k)


def store_kv_ref(k_cache: torch.Tensor, v_cache: torch.Tensor, loc: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> None:
    """mini_sglang/mem_cache/token2kv.py:141-142 (index_put into the per-layer slot pool)."""
    Hkv, D = k_cache.shape[1], k_cache.shape[2]
    k_cache[loc.long()] = k.reshape(-1, Hkv, D)
    v_cache[loc.long()] = v.reshape(-1, Hkv, D)


# --------------------------------------------------------------------------- attention
def paged_decode_ref(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, req_to_token: torch.Tensor,
                     req_pool_indices: torch.Tensor, seq_lens: torch.Tensor, scale: float) -> torch.Tensor:
    """mini_sglang/layers/attn/torch_attn_backend.py:114-182.

    q: [bs, Hq, D]; caches [slots, Hkv, D]; one SDPA call per request over the gathered K/V rows,
    non-causal, GQA via enable_gqa (torch_attn_backend.py:168-175, 234-250).
    """
    bs, Hq, D = q.shape
    Hkv = k_cache.shape[1]
    out = torch.empty_like(q)
    qh = q.movedim(0, 1)  # [Hq, bs, D]
    seq_list = seq_lens.tolist()
    idx_list = req_pool_indices.tolist()
    for i in range(bs):
        tok = req_to_token[idx_list[i], : seq_list[i]].long()
        kk = k_cache[tok].movedim(0, 1)  # [Hkv, S, D]
        vv = v_cache[tok].movedim(0, 1)
        o = F.scaled_dot_product_attention(qh[:, i : i + 1, :].unsqueeze(0), kk.unsqueeze(0), vv.unsqueeze(0),
                                           enable_gqa=(Hq != Hkv), scale=scale, is_causal=False)
        out[i : i + 1] = o.squeeze(0).movedim(1, 0)
    return out


def paged_extend_ref(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, req_to_token: torch.Tensor,
                     req_pool_indices: torch.Tensor, seq_lens: torch.Tensor, prefix_lens: torch.Tensor,
                     input_lens: torch.Tensor, scale: float) -> torch.Tensor:
    """mini_sglang/layers/attn/torch_attn_backend.py:28-111.

    q: [sum Sq, Hq, D] packed.  Per request the query run is padded in front to the full key length
    (the reference leaves those prefix rows uninitialised, :82-88; zeros here — their outputs are
    discarded, :109), one causal S x S SDPA runs, and rows [prefix:] are kept.
    """
    T, Hq, D = q.shape
    Hkv = k_cache.shape[1]
    out = torch.empty_like(q)
    qh = q.movedim(0, 1)  # [Hq, T, D]
    start = 0
    seq_list, pre_list, in_list = seq_lens.tolist(), prefix_lens.tolist(), input_lens.tolist()
    idx_list = req_pool_indices.tolist()
    for i in range(len(seq_list)):
        S, P, n = seq_list[i], pre_list[i], in_list[i]
        q_full = torch.zeros((Hq, S, D), dtype=q.dtype, device=q.device)
        q_full[:, P:, :] = qh[:, start : start + n, :]
        tok = req_to_token[idx_list[i], :S].long()
        kk = k_cache[tok].movedim(0, 1)
        vv = v_cache[tok].movedim(0, 1)
        o = F.scaled_dot_product_attention(q_full.unsqueeze(0), kk.unsqueeze(0), vv.unsqueeze(0),
                                           enable_gqa=(Hq != Hkv), scale=scale, is_causal=True)
        out[start : start + n] = o.squeeze(0).movedim(1, 0)[P:]
        start += n
    return out


def argmax_ref(logits: torch.Tensor) -> torch.Tensor:
    """mini_sglang/layers/sampler.py:14-15: greedy token = argmax over fp32 logits."""
    return logits.to(torch.float).argmax(dim=-1).to(torch.int32)


# --------------------------------------------------------------------------- fp64 ground truth (tests only)
def prepare_decode_ref(req_pool_indices, seq_lens, last_slots, free_pages, free_head: int, req_to_token: torch.Tensor,
                       page_size: int, req_to_page=None):
    """One decode step's batch preparation, request by request, as the reference does it
    (mini_sglang/managers/batch_info.py:247-302; page order: PageAllocator.alloc takes from the front of the
    free list, mini_sglang/mem_cache/token2kv.py:186-196).  Plain Python lists / ints in, written in place
    (``seq_lens``, ``last_slots``, ``req_to_token``); returns (positions, out_cache_loc, new free_head), or
    raises RuntimeError like the allocator when the free list runs out (nothing is modified then)."""
    ps = page_size
    need = sum(1 for old in seq_lens if ps == 1 or old % ps == 0)
    if need > len(free_pages) - free_head:
        raise RuntimeError(f"Not enough free pages available. need {need}, free {len(free_pages) - free_head}")
    positions, out_loc = [], []
    for i, old in enumerate(seq_lens):
        if ps == 1 or old % ps == 0:          # batch_info.py:277-283: a fresh page, first slot
            slot = int(free_pages[free_head]) * ps
            if req_to_page is not None and ps > 1:    # the page half of the table (req2token.py:43-51): page id of the new page
                req_to_page[int(req_pool_indices[i]), int(old) // ps] = int(free_pages[free_head])
            free_head += 1
        else:                                 # batch_info.py:286-291: the slot after the previous token's
            slot = int(last_slots[i]) + 1
        positions.append(int(old))
        out_loc.append(slot)
        req_to_token[int(req_pool_indices[i]), int(old)] = slot   # batch_info.py:298-300
        seq_lens[i] = int(old) + 1            # batch_info.py:262-263
        last_slots[i] = slot
    return positions, out_loc, free_head


def extend_plan_ref(cu_seqlens_q, prefix_lens, max_seqlen_q: int, q_tile: int = 128, k_tile: int = 64):
    """Work items of an extend launch with their cost, for checking the device-built launch order.  No counterpart
    in the reference (its backends leave block scheduling to their attention library); the cost of item
    b * tiles + qt is the number of `k_tile`-key tiles visible to query rows [q_tile * qt, q_tile * (qt + 1)) of
    request b under the reference's mask (torch_attn_backend.py:74-101: row i sees positions [0, prefix + i]).
    Returns (items, key_tiles) for the items that hold at least one query row."""
    tiles = -(-int(max_seqlen_q) // q_tile)
    items, cost = [], []
    for b in range(len(prefix_lens)):
        n, pre = int(cu_seqlens_q[b + 1]) - int(cu_seqlens_q[b]), int(prefix_lens[b])
        for qt in range(tiles):
            if qt * q_tile >= n:
                break
            visible = min(pre + n, pre + (qt + 1) * q_tile)
            items.append(b * tiles + qt)
            cost.append(-(-visible // k_tile))
    return items, cost


def decode_plan_ref(seq_lens, max_splits: int, chunk: int, tile: int = 16):
    """Work items of a split decode launch with their cost, for checking the device-built launch order (no
    counterpart in the reference, whose decode loops over requests on the host, torch_attn_backend.py:146-180).
    Item split * bs + b covers positions [split * chunk, min(S_b, (split + 1) * chunk)) of request b -- the last
    of `max_splits` splits takes everything that is left -- and costs its number of `tile`-token tiles.
    Returns (items, tiles) for the items that cover at least one position."""
    bs = len(seq_lens)
    chunk = max(tile, -(-chunk // tile) * tile)
    items, cost = [], []
    for split in range(max_splits):
        for b in range(bs):
            S = int(seq_lens[b])
            if split >= min(-(-S // chunk), max_splits):
                continue
            t0 = split * chunk
            t1 = S if split == max_splits - 1 else min(S, t0 + chunk)
            items.append(split * bs + b)
            cost.append(-(-(t1 - t0) // tile))
    return items, cost


def paged_attention_fp64(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, tok: torch.Tensor, scale: float,
                         q_positions: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Exact softmax(q K^T) V in float64 for ONE request: q [Sq, Hq, D], tok = its slot list [S].
    q_positions[i] = last key position row i may see (None = all keys: decode)."""
    Sq, Hq, D = q.shape
    Hkv = k_cache.shape[1]
    g = Hq // Hkv
    kk = k_cache[tok.long()].double().repeat_interleave(g, dim=1)  # [S, Hq, D]
    vv = v_cache[tok.long()].double().repeat_interleave(g, dim=1)
    s = torch.einsum("qhd,khd->hqk", q.double(), kk) * scale
    if q_positions is not None:
        kpos = torch.arange(kk.shape[0], device=q.device)
        s = s.masked_fill(kpos[None, None, :] > q_positions[None, :, None], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("hqk,khd->qhd", p, vv)
