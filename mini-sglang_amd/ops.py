"""Tensor-level wrappers over the C ABI: shape / dtype / device checks on the host, raw device
pointers and element strides into the library, work enqueued on torch's current stream.

Every function here runs a HIP kernel; none has a torch or CPU fallback.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _capi

_lib = _capi.lib
_check = _capi.check

DECODE_TILE = 16  # tokens per wave iteration of the decode kernel; split chunks are multiples of it

# msgl_set_variant's choices (include/msgl_hip.h): tests and measurements pin a kernel form with them; the library
# itself never reads its environment and a serving process leaves them all on "auto" (-1)
VARIANTS = {"decode_mfma": 0, "decode_mfma_waves": 1, "extend_heads": 2, "extend_wide": 3, "extend_ring": 4, "extend_key_split": 5,
            "decode_mfma_depth": 6}


def set_variant(name: str, value: Optional[int]) -> int:
    """Pin (or, with None / -1, release) one kernel-form choice process-wide; returns the previous value."""
    _need(name in VARIANTS, f"unknown variant {name!r} (one of {sorted(VARIANTS)})")
    return int(_lib.msgl_set_variant(VARIANTS[name], -1 if value is None else int(value)))


def reset_variants() -> None:
    for which in VARIANTS.values():
        _lib.msgl_set_variant(which, -1)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need(cond: bool, msg: str) -> None:
    if not cond:
        raise ValueError(msg)


def _dev_bf16(t: torch.Tensor, name: str) -> None:
    _need(t.is_cuda, f"{name} must live on the GPU (got {t.device})")
    _need(t.dtype == torch.bfloat16, f"{name} must be bfloat16 (got {t.dtype})")


def _dev_i32(t: torch.Tensor, name: str) -> None:
    _need(t.is_cuda, f"{name} must live on the GPU (got {t.device})")
    _need(t.dtype == torch.int32, f"{name} must be int32 (got {t.dtype})")
    _need(t.dim() == 1 and (t.numel() <= 1 or t.stride(0) == 1), f"{name} must be a contiguous 1-D tensor")


def _rows_view(t: torch.Tensor, name: str) -> Tuple[int, int, int]:
    """(rows, cols, row_stride) of a 2-D tensor whose last dim is dense."""
    _need(t.dim() == 2, f"{name} must be 2-D (got shape {tuple(t.shape)})")
    _need(t.shape[1] == 1 or t.stride(1) == 1, f"{name}: last dim must be dense")
    return t.shape[0], t.shape[1], (t.stride(0) if t.shape[0] > 1 else t.shape[1])


def _cache_stride(k_cache: torch.Tensor, v_cache: torch.Tensor, Hkv: int, D: int) -> int:
    """Slot stride (elements) of a [slots, Hkv, D] K/V buffer pair; heads must be dense per slot.
    Dense buffers and the [:, 0] / [:, 1] views of an interleaved [slots, 2, Hkv, D] pool both qualify."""
    _dev_bf16(k_cache, "k_cache"), _dev_bf16(v_cache, "v_cache")
    _need(k_cache.dim() == 3 and tuple(k_cache.shape[1:]) == (Hkv, D) and v_cache.shape == k_cache.shape,
          "k_cache / v_cache must be [slots, Hkv, D]")
    for t, n in ((k_cache, "k_cache"), (v_cache, "v_cache")):
        _need(t.stride(2) == 1 and t.stride(1) == D, f"{n}: [Hkv, D] must be dense per slot")
    _need(k_cache.stride(0) == v_cache.stride(0) and k_cache.stride(0) >= Hkv * D, "k_cache / v_cache slot strides differ")
    return k_cache.stride(0)


# ----------------------------------------------------------------------------- KV store
def store_kv(k: torch.Tensor, v: torch.Tensor, loc: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor) -> None:
    """k_cache[loc] = k; v_cache[loc] = v.  k, v: [T, Hkv, D] (row-strided views allowed)."""
    _dev_bf16(k, "k"), _dev_bf16(v, "v")
    _dev_i32(loc, "loc")
    T, Hkv, D = k.shape
    _need(v.shape == k.shape, "k and v shapes differ")
    cs = _cache_stride(k_cache, v_cache, Hkv, D)
    _need(loc.numel() == T, "loc length must equal the number of tokens")
    for t, n in ((k, "k"), (v, "v")):
        _need(t.stride(2) == 1 and t.stride(1) == D, f"{n}: heads must be dense [Hkv, D] per token")
    ks = k.stride(0) if T > 1 else Hkv * D
    vs = v.stride(0) if T > 1 else Hkv * D
    _check(_lib.msgl_store_kv(k.data_ptr(), ks, v.data_ptr(), vs, loc.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                              cs, T, Hkv, D, _stream()))


# ----------------------------------------------------------------------------- decode
def decode_workspace_bytes(bs: int, Hq: int, D: int, max_splits: int) -> int:
    return int(_lib.msgl_paged_decode_workspace_bytes(bs, Hq, D, max_splits))


def paged_decode(
    q: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    req_to_token: torch.Tensor,
    req_pool_indices: torch.Tensor,
    seq_lens: torch.Tensor,
    scale: float,
    page_size: int,
    *,
    out: Optional[torch.Tensor] = None,
    workspace: Optional[torch.Tensor] = None,
    max_splits: int = 1,
    chunk_tokens: int = 0,
    chunk_dev: Optional[torch.Tensor] = None,
    kv_stream: bool = True,
    reuse_lens: Optional[torch.Tensor] = None,
    plan: Optional[torch.Tensor] = None,
    start_lens: Optional[torch.Tensor] = None,
    extra_splits: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """Decode attention for q [bs, Hq, D] over the paged pool; returns [bs, Hq, D] bf16.
    plan: launch order from decode_plan for the same seq_lens / max_splits / chunk (ragged batches).
    start_lens / extra_splits: positions below start_lens[b] were handled by shared_prefix_decode, which left
    extra_splits[b] partial results in the same workspace; they are merged in.

    kv_stream: read K/V with the non-temporal hint, except positions below reuse_lens[b] (prefix slots other
    requests share), which stay cacheable.  A pure performance hint."""
    _dev_bf16(q, "q")
    _dev_i32(req_pool_indices, "req_pool_indices"), _dev_i32(seq_lens, "seq_lens")
    _need(q.dim() == 3 and q.stride(2) == 1 and q.stride(1) == q.shape[2], "q must be [bs, Hq, D] with dense heads")
    bs, Hq, D = q.shape
    _need(k_cache.dim() == 3, "k_cache must be [slots, Hkv, D]")
    Hkv = k_cache.shape[1]
    cs = _cache_stride(k_cache, v_cache, Hkv, D)
    _need(req_to_token.is_cuda and req_to_token.dtype == torch.int32 and req_to_token.dim() == 2
          and req_to_token.stride(1) == 1, "req_to_token must be an int32 [reqs, ctx] GPU tensor with dense rows")
    _need(req_pool_indices.numel() == bs and seq_lens.numel() == bs, "req_pool_indices / seq_lens must have bs entries")
    if out is None:
        out = torch.empty((bs, Hq, D), dtype=torch.bfloat16, device=q.device)
    else:
        _dev_bf16(out, "out")
        _need(out.shape == q.shape and out.stride(2) == 1 and out.stride(1) == D, "out must match q's layout")
    if max_splits > 1:
        need = decode_workspace_bytes(bs, Hq, D, max_splits)
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.uint8, device=q.device)
        _need(workspace.is_cuda and workspace.is_contiguous() and workspace.numel() * workspace.element_size() >= need,
              f"workspace must hold {need} bytes")
    if reuse_lens is not None:
        _dev_i32(reuse_lens, "reuse_lens")
        _need(reuse_lens.numel() == bs, "reuse_lens must have bs entries")
    if chunk_dev is not None:
        _dev_i32(chunk_dev, "chunk_dev")
    elif chunk_tokens <= 0:
        _need(max_splits == 1, "chunk_tokens is required when max_splits > 1")
        chunk_tokens = 1 << 30  # one split covers any sequence
    if plan is not None:
        _dev_i32(plan, "plan")
        _need(plan.numel() == bs * max_splits, "plan must have bs * max_splits entries")
    if extra_splits is not None:
        _need(start_lens is not None and workspace is not None, "extra_splits needs start_lens and the workspace the partials are in")
        _dev_i32(extra_splits, "extra_splits")
        _need(extra_splits.numel() == bs, "extra_splits must have bs entries")
    if start_lens is not None:
        _dev_i32(start_lens, "start_lens")
        _need(start_lens.numel() == bs, "start_lens must have bs entries")
    qs = q.stride(0) if bs > 1 else Hq * D
    os_ = out.stride(0) if bs > 1 else Hq * D
    _check(_lib.msgl_paged_decode(
        q.data_ptr(), qs, k_cache.data_ptr(), v_cache.data_ptr(), cs,
        req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(), seq_lens.data_ptr(),
        out.data_ptr(), os_,
        workspace.data_ptr() if workspace is not None else None,
        workspace.numel() * workspace.element_size() if workspace is not None else 0,
        bs, Hq, Hkv, D, page_size, float(scale), max_splits, int(chunk_tokens),
        chunk_dev.data_ptr() if chunk_dev is not None else None, 1 if kv_stream else 0,
        reuse_lens.data_ptr() if reuse_lens is not None else None,
        plan.data_ptr() if plan is not None else None,
        start_lens.data_ptr() if start_lens is not None else None,
        extra_splits.data_ptr() if extra_splits is not None else None, _stream()))
    return out


STATIC_WORKSPACE_BYTES = 32 << 20  # MSGL_PAGED_DECODE_STATIC_WORKSPACE_BYTES


def paged_decode_static(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, req_to_token: torch.Tensor,
                        req_pool_indices: torch.Tensor, seq_lens: torch.Tensor, scale: float, page_size: int, out: torch.Tensor, *,
                        max_splits: int = 1, chunk_tokens: int = 0, chunk_dev: Optional[torch.Tensor] = None, kv_stream: bool = True,
                        reuse_lens: Optional[torch.Tensor] = None, plan: Optional[torch.Tensor] = None) -> torch.Tensor:
    """paged_decode through the entry point that allocates nothing: the caller brings `out`, the split workspace is the
    library's built-in one (STATIC_WORKSPACE_BYTES per GPU; larger needs are refused).  Calls that overlap on one
    GPU share that workspace and race."""
    _dev_bf16(q, "q"), _dev_bf16(out, "out")
    _dev_i32(req_pool_indices, "req_pool_indices"), _dev_i32(seq_lens, "seq_lens")
    _need(q.dim() == 3 and q.stride(2) == 1 and q.stride(1) == q.shape[2], "q must be [bs, Hq, D] with dense heads")
    bs, Hq, D = q.shape
    _need(k_cache.dim() == 3, "k_cache must be [slots, Hkv, D]")
    Hkv = k_cache.shape[1]
    cs = _cache_stride(k_cache, v_cache, Hkv, D)
    _need(req_to_token.is_cuda and req_to_token.dtype == torch.int32 and req_to_token.dim() == 2
          and req_to_token.stride(1) == 1, "req_to_token must be an int32 [reqs, ctx] GPU tensor with dense rows")
    _need(req_pool_indices.numel() == bs and seq_lens.numel() == bs, "req_pool_indices / seq_lens must have bs entries")
    _need(out.shape == q.shape and out.stride(2) == 1 and out.stride(1) == D, "out must match q's layout")
    for t, name in ((reuse_lens, "reuse_lens"), (chunk_dev, "chunk_dev"), (plan, "plan")):
        if t is not None:
            _dev_i32(t, name)
    if chunk_dev is None and chunk_tokens <= 0:
        _need(max_splits == 1, "chunk_tokens is required when max_splits > 1")
        chunk_tokens = 1 << 30
    _need(plan is None or plan.numel() == bs * max_splits, "plan must have bs * max_splits entries")
    _check(_lib.msgl_paged_decode_static(
        q.data_ptr(), q.stride(0) if bs > 1 else Hq * D, k_cache.data_ptr(), v_cache.data_ptr(), cs,
        req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(), seq_lens.data_ptr(),
        out.data_ptr(), out.stride(0) if bs > 1 else Hq * D, bs, Hq, Hkv, D, page_size, float(scale), max_splits, int(chunk_tokens),
        chunk_dev.data_ptr() if chunk_dev is not None else None, 1 if kv_stream else 0,
        reuse_lens.data_ptr() if reuse_lens is not None else None, plan.data_ptr() if plan is not None else None, _stream()))
    return out


def shared_prefix_decode(
    q: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    req_to_token: torch.Tensor,
    req_pool_indices: torch.Tensor,
    seq_lens: torch.Tensor,
    start_lens: torch.Tensor,
    group_first: torch.Tensor,
    group_count: torch.Tensor,
    group_prefix: torch.Tensor,
    max_members: int,
    max_prefix: int,
    key_chunk: int,
    scale: float,
    workspace: torch.Tensor,
    max_splits: int,
    chunk_tokens: int = 0,
    chunk_dev: Optional[torch.Tensor] = None,
    items: Optional[torch.Tensor] = None,
    key_chunk_dev: Optional[torch.Tensor] = None,
    group_members: Optional[torch.Tensor] = None,
) -> None:
    """Attention of every group's decode queries over the group's shared prefix, once per group, left as split
    partials in `workspace` for the paged_decode call that follows with the same seq_lens / start_lens /
    max_splits / chunk and extra_splits[b] = ceil(prefix / key_chunk).  items / key_chunk_dev: the work list and
    slice length in device memory, for captured launches (shared_prefix_items builds the list).  group_members:
    batch indices of the members, group after group (group_first then indexes this list); None = the members of
    group g are the consecutive batch entries group_first[g] .. + group_count[g] - 1."""
    _dev_bf16(q, "q")
    for t, name in ((req_pool_indices, "req_pool_indices"), (seq_lens, "seq_lens"), (start_lens, "start_lens"),
                    (group_first, "group_first"), (group_count, "group_count"), (group_prefix, "group_prefix")):
        _dev_i32(t, name)
    _need(q.dim() == 3 and q.stride(2) == 1 and q.stride(1) == q.shape[2], "q must be [bs, Hq, D] with dense heads")
    bs, Hq, D = q.shape
    Hkv = k_cache.shape[1]
    cs = _cache_stride(k_cache, v_cache, Hkv, D)
    ng = group_first.numel()
    _need(group_count.numel() == ng and group_prefix.numel() == ng, "group arrays must have one entry per group")
    _need(seq_lens.numel() == bs and start_lens.numel() == bs and req_pool_indices.numel() == bs, "per-request arrays must have bs entries")
    _need(workspace.is_cuda and workspace.is_contiguous(), "workspace must be a contiguous GPU tensor")
    if chunk_dev is not None:
        _dev_i32(chunk_dev, "chunk_dev")
    if items is not None:
        _dev_i32(items, "items")
    if key_chunk_dev is not None:
        _dev_i32(key_chunk_dev, "key_chunk_dev")
    if group_members is not None:
        _dev_i32(group_members, "group_members")
    _check(_lib.msgl_shared_prefix_decode(
        q.data_ptr(), q.stride(0) if bs > 1 else Hq * D, k_cache.data_ptr(), v_cache.data_ptr(), cs,
        req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(), seq_lens.data_ptr(), start_lens.data_ptr(),
        group_first.data_ptr(), group_count.data_ptr(), group_prefix.data_ptr(),
        group_members.data_ptr() if group_members is not None else None, ng, int(max_members), int(max_prefix), int(key_chunk),
        workspace.data_ptr(), workspace.numel() * workspace.element_size(), bs, Hq, Hkv, D, float(scale), int(max_splits),
        int(chunk_tokens), chunk_dev.data_ptr() if chunk_dev is not None else None,
        items.data_ptr() if items is not None else None, items.numel() if items is not None else 0,
        key_chunk_dev.data_ptr() if key_chunk_dev is not None else None, _stream()))


def shared_prefix_items(groups, heads_per_kv_head: int, key_chunk: int, capacity: int) -> Optional[List[int]]:
    """Work list of shared_prefix_decode for `groups` = [(members, prefix length)]: one entry per (group, 128-row
    tile of members x heads, key slice), padded with -1 to `capacity`; None when the list does not fit."""
    out = []
    groups = [(0, len(members), plen) for members, plen in groups]
    for gi, (_, count, plen) in enumerate(groups):
        for rt in range(-(-count * heads_per_kv_head // 128)):
            for ks in range(-(-plen // key_chunk)):
                out.append(gi << 20 | rt << 14 | ks)
    if len(out) > capacity or len(groups) >= 2048 or any(-(-g[2] // key_chunk) > 0x3fff or -(-g[1] * heads_per_kv_head // 128) > 63 for g in groups):
        return None
    return out + [-1] * (capacity - len(out))


def decode_plan(seq_lens: torch.Tensor, max_splits: int, chunk_tokens: int = 0, chunk_dev: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, sorted_dev: Optional[torch.Tensor] = None,
                start_lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Launch order of paged_decode's (split, request) items for a ragged batch: full splits first, partial tails
    by falling length.  Built on the device from seq_lens; one call per step serves every layer.  sorted_dev: device
    word, 0 = write the identity order instead (decided at run time, for captured graphs)."""
    _dev_i32(seq_lens, "seq_lens")
    bs = seq_lens.numel()
    if chunk_dev is not None:
        _dev_i32(chunk_dev, "chunk_dev")
    if sorted_dev is not None:
        _dev_i32(sorted_dev, "sorted_dev")
    if out is None:
        out = torch.empty(bs * max_splits, dtype=torch.int32, device=seq_lens.device)
    else:
        _dev_i32(out, "out")
        _need(out.numel() == bs * max_splits, "plan must have bs * max_splits entries")
    _check(_lib.msgl_decode_plan(seq_lens.data_ptr(), bs, int(max_splits), int(chunk_tokens),
                                 chunk_dev.data_ptr() if chunk_dev is not None else None,
                                 sorted_dev.data_ptr() if sorted_dev is not None else None,
                                 start_lens.data_ptr() if start_lens is not None else None, out.data_ptr(), _stream()))
    return out


def pick_split(seq_lens_cpu_max: int, total_tokens: int, bs: int, max_splits: int, local_kv_heads: int = 8) -> int:
    """Tokens per split for the decode kernel: a multiple of DECODE_TILE such that the longest sequence fits
    max_splits splits.  Measured on MI355X (256 CUs): small batches run best with about one work group per CU,
    large uniform ones with two (long splits amortise a group's prologue and merge), ragged ones with about
    eight, so that short requests and the tails of long ones even out.  A split is never shorter than four
    16-token tiles per wave; with fewer than 8 local kv heads several waves share a head and a split."""
    waves_per_head = max(1, 8 // max(1, min(8, local_kv_heads)))
    min_chunk = max(128, 64 * waves_per_head)
    ragged = seq_lens_cpu_max * bs > 1.25 * total_tokens
    if ragged:
        target_groups = 2048
    else:
        # with at most four local kv heads (TP 2 / 4 / 8 shards) a group's 8 waves already split a chunk between them:
        # one group per CU and, when it fits, no second split to merge (measured on the headline shape per rank:
        # +1 % with four kv heads, +5-9 % with one)
        target_groups = 256 if (total_tokens <= 256 * 1024 or waves_per_head >= 2) else 512
    per_group = max(1, -(-total_tokens // target_groups))
    floor_ = -(-max(seq_lens_cpu_max, 1) // max_splits)
    chunk = max(per_group, floor_, min_chunk)
    return -(-chunk // DECODE_TILE) * DECODE_TILE


def planned_makespan(seq_lens, chunk: int, slots: int, group_overhead: int) -> int:
    """Tokens the busiest of `slots` concurrently running work groups reads when the (request, split) items of
    `seq_lens` cut at `chunk` are started longest first (what the hardware's in-order dispatch does with a launch
    ordered by decode_plan): the full splits are dealt round robin, then every partial tail, by falling length,
    goes to the group slot that is free first.  `group_overhead`: start-up and merge cost of one item, in tokens."""
    import heapq

    full = sum(s // chunk for s in seq_lens)
    tails = sorted((s % chunk for s in seq_lens if s % chunk), reverse=True)
    unit = chunk + group_overhead
    base, extra = divmod(full, slots)
    if not tails:
        return (base + (1 if extra else 0)) * unit
    if len(tails) <= slots - extra:  # every tail finds a slot that holds only `base` full splits
        return max(base * unit + tails[0] + group_overhead, (base + 1) * unit if extra else 0)
    loads = [(base + 1) * unit] * extra + [base * unit] * (slots - extra)
    heapq.heapify(loads)
    for t in tails:
        heapq.heapreplace(loads, loads[0] + t + group_overhead)
    return max(loads)


def pick_split_planned(seq_lens, max_splits: int, local_kv_heads: int = 8, slots: int = 256) -> int:
    """Tokens per split for a ragged batch launched in decode_plan order: the candidate (multiples of 256 tokens)
    whose longest-first schedule on `slots` compute units finishes first.  `seq_lens`: host-side lengths."""
    lens = [int(s) for s in seq_lens]
    waves_per_head = max(1, 8 // max(1, min(8, local_kv_heads)))
    lo = max(256, 64 * waves_per_head, -(-max(lens) // max_splits))
    lo = -(-lo // 256) * 256
    hi = max(lo, min(-(-max(lens) // 256) * 256, 8192))
    overhead = 128 * waves_per_head  # a group's prologue + its share of the merge, in tokens of this shard's rows
    step = 256 * max(1, (hi - lo) // (256 * 16))  # at most ~16 candidates
    best, best_cost = lo, None
    for chunk in range(lo, hi + 1, step):
        cost = planned_makespan(lens, chunk, slots, overhead)
        if sum(-(-s // chunk) for s in lens) > slots:
            cost *= 1.1  # measured: launches that fit the chip in one round beat what the list schedule predicts for the rest
        if best_cost is None or cost < best_cost:
            best, best_cost = chunk, cost
    return best


# ----------------------------------------------------------------------------- extend
def paged_extend(
    q: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    req_to_token: torch.Tensor,
    req_pool_indices: torch.Tensor,
    cu_seqlens_q: torch.Tensor,
    prefix_lens: torch.Tensor,
    max_seqlen_q: int,
    scale: float,
    page_size: int,
    *,
    out: Optional[torch.Tensor] = None,
    max_seqlen_k: int = 0,
    plan: Optional[torch.Tensor] = None,
    workspace: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """Extend attention for packed q [sum Sq, Hq, D]; K/V of the new tokens must already be stored.
    `max_seqlen_k` (longest prefix + new tokens, 0 = unknown) and `plan` (from extend_plan, same batch) only
    steer scheduling; the output does not depend on them.  `workspace` (extend_workspace_bytes(T, Hq, D) bytes,
    optional) lets a launch of few work items over long key ranges split the keys between work groups."""
    _dev_bf16(q, "q")
    _dev_i32(req_pool_indices, "req_pool_indices"), _dev_i32(cu_seqlens_q, "cu_seqlens_q"), _dev_i32(prefix_lens, "prefix_lens")
    _need(q.dim() == 3 and q.stride(2) == 1 and q.stride(1) == q.shape[2], "q must be [T, Hq, D] with dense heads")
    T, Hq, D = q.shape
    bs = req_pool_indices.numel()
    _need(cu_seqlens_q.numel() == bs + 1 and prefix_lens.numel() == bs, "cu_seqlens_q needs bs+1 and prefix_lens bs entries")
    _need(k_cache.dim() == 3, "k_cache must be [slots, Hkv, D]")
    Hkv = k_cache.shape[1]
    cs = _cache_stride(k_cache, v_cache, Hkv, D)
    _need(req_to_token.is_cuda and req_to_token.dtype == torch.int32 and req_to_token.dim() == 2
          and req_to_token.stride(1) == 1, "req_to_token must be an int32 [reqs, ctx] GPU tensor with dense rows")
    if out is None:
        out = torch.empty((T, Hq, D), dtype=torch.bfloat16, device=q.device)
    else:
        _dev_bf16(out, "out")
        _need(out.shape == q.shape and out.stride(2) == 1 and out.stride(1) == D, "out must match q's layout")
    qs = q.stride(0) if T > 1 else Hq * D
    os_ = out.stride(0) if T > 1 else Hq * D
    if plan is not None:
        _dev_i32(plan, "plan")
        _need(plan.numel() == extend_plan_items(bs, int(max_seqlen_q)), "plan was made for another bs / max_seqlen_q")
    if workspace is not None:
        _need(workspace.is_cuda and workspace.is_contiguous() and workspace.data_ptr() % 16 == 0, "workspace must be a contiguous GPU tensor")
    _check(_lib.msgl_paged_extend(
        q.data_ptr(), qs, k_cache.data_ptr(), v_cache.data_ptr(), cs,
        req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(), cu_seqlens_q.data_ptr(),
        prefix_lens.data_ptr(), out.data_ptr(), os_, bs, int(max_seqlen_q), int(max_seqlen_k),
        plan.data_ptr() if plan is not None else None, Hq, Hkv, D, page_size, float(scale),
        workspace.data_ptr() if workspace is not None else None,
        workspace.numel() * workspace.element_size() if workspace is not None else 0, T, _stream()))
    return out


def extend_workspace_bytes(total_tokens: int, Hq: int, D: int) -> int:
    return int(_lib.msgl_paged_extend_workspace_bytes(int(total_tokens), Hq, D))


def extend_key_splits(bs: int, max_seqlen_q: int, max_seqlen_k: int, Hq: int, Hkv: int) -> int:
    """Key runs per work item paged_extend takes for this shape when its workspace allows (1 = none)."""
    return int(_lib.msgl_paged_extend_key_splits(int(bs), int(max_seqlen_q), int(max_seqlen_k), int(Hq), int(Hkv)))


def extend_workspace_bytes_for(total_tokens: int, Hq: int, D: int, key_splits: int) -> int:
    """Workspace bytes of paged_extend for `key_splits` key runs (0 for one run)."""
    return int(_lib.msgl_paged_extend_workspace_bytes_for(int(total_tokens), int(Hq), int(D), int(key_splits)))


def extend_plan_items(bs: int, max_seqlen_q: int) -> int:
    return int(_lib.msgl_extend_plan_items(int(bs), int(max_seqlen_q)))


def extend_plan(cu_seqlens_q: torch.Tensor, prefix_lens: torch.Tensor, max_seqlen_q: int,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Launch order of paged_extend's work items for this batch, longest (most key tiles) first; computed on
    the device, so the lengths are not read back.  One call per forward pass serves every layer."""
    _dev_i32(cu_seqlens_q, "cu_seqlens_q"), _dev_i32(prefix_lens, "prefix_lens")
    bs = prefix_lens.numel()
    _need(cu_seqlens_q.numel() == bs + 1, "cu_seqlens_q needs bs+1 entries")
    n = extend_plan_items(bs, max_seqlen_q)
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=prefix_lens.device)
    else:
        _dev_i32(out, "out")
        _need(out.numel() == n, f"plan needs {n} entries")
    _check(_lib.msgl_extend_plan(cu_seqlens_q.data_ptr(), prefix_lens.data_ptr(), bs, int(max_seqlen_q), out.data_ptr(), _stream()))
    return out


# ----------------------------------------------------------------------------- norms / activation / rope
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev_bf16(x, "x"), _dev_bf16(weight, "weight")
    rows, cols, xs = _rows_view(x, "x")
    _need(weight.numel() == cols and weight.is_contiguous(), "weight must be a dense [cols] vector")
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    _, ocols, os_ = _rows_view(out, "out")
    _need(out.shape[0] == rows and ocols == cols, "out shape mismatch")
    _check(_lib.msgl_rmsnorm(x.data_ptr(), xs, weight.data_ptr(), out.data_ptr(), os_, rows, cols, float(eps), _stream()))
    return out


def fused_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> None:
    """In place: residual += x (rounded to bf16), x = rmsnorm(unrounded sum) * weight."""
    _dev_bf16(x, "x"), _dev_bf16(residual, "residual"), _dev_bf16(weight, "weight")
    _need(x.dim() == 2 and x.is_contiguous() and residual.shape == x.shape and residual.is_contiguous(),
          "x and residual must be contiguous 2-D tensors of equal shape")
    rows, cols = x.shape
    _need(weight.numel() == cols and weight.is_contiguous(), "weight must be a dense [cols] vector")
    _check(_lib.msgl_fused_add_rmsnorm(x.data_ptr(), residual.data_ptr(), weight.data_ptr(), rows, cols, float(eps), _stream()))


def silu_and_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev_bf16(x, "x")
    rows, cols2, xs = _rows_view(x, "x")
    _need(cols2 % 2 == 0, "last dim must be even")
    n = cols2 // 2
    if out is None:
        out = torch.empty((rows, n), dtype=torch.bfloat16, device=x.device)
    _need(out.shape == (rows, n) and out.is_contiguous(), "out must be contiguous [rows, n]")
    _check(_lib.msgl_silu_and_mul(x.data_ptr(), xs, out.data_ptr(), rows, n, _stream()))
    return out


def rope_neox_(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, cos_sin_cache: torch.Tensor, head_dim: int,
               reference_rounding: bool = True) -> None:
    """In place on q [T, Hq*D] and k [T, Hkv*D] (row-strided views allowed)."""
    _dev_bf16(q, "q"), _dev_bf16(k, "k"), _dev_i32(positions, "positions")
    _need(cos_sin_cache.is_cuda and cos_sin_cache.dtype == torch.float32 and cos_sin_cache.is_contiguous()
          and cos_sin_cache.shape[1] == head_dim, "cos_sin_cache must be a contiguous fp32 [max_pos, D] GPU tensor")
    T, qc, qs = _rows_view(q, "q")
    Tk, kc, ks = _rows_view(k, "k")
    _need(T == Tk == positions.numel(), "q, k and positions disagree on the token count")
    _need(qc % head_dim == 0 and kc % head_dim == 0, "row widths must be multiples of head_dim")
    _check(_lib.msgl_rope_neox(positions.data_ptr(), q.data_ptr(), qs, k.data_ptr(), ks, cos_sin_cache.data_ptr(), T,
                               qc // head_dim, kc // head_dim, head_dim, 0 if reference_rounding else 1, _stream()))


def qknorm_rope_store_(qkv: torch.Tensor, q_weight: torch.Tensor, k_weight: torch.Tensor, eps: float,
                       positions: torch.Tensor, cos_sin_cache: torch.Tensor, Hq: int, Hkv: int, head_dim: int,
                       loc: Optional[torch.Tensor] = None, k_cache: Optional[torch.Tensor] = None,
                       v_cache: Optional[torch.Tensor] = None, reference_rounding: bool = True) -> None:
    """Per-head q/k RMSNorm + RoPE in place on qkv [T, (Hq+2Hkv)*D]; optionally scatters k, v to the pool."""
    _dev_bf16(qkv, "qkv"), _dev_bf16(q_weight, "q_weight"), _dev_bf16(k_weight, "k_weight"), _dev_i32(positions, "positions")
    T, cols, qs = _rows_view(qkv, "qkv")
    _need(cols == (Hq + 2 * Hkv) * head_dim, "qkv width must be (Hq + 2*Hkv) * head_dim")
    _need(positions.numel() == T, "positions length must equal the number of tokens")
    _need(cos_sin_cache.is_cuda and cos_sin_cache.dtype == torch.float32 and cos_sin_cache.is_contiguous()
          and cos_sin_cache.shape[1] == head_dim, "cos_sin_cache must be a contiguous fp32 [max_pos, D] GPU tensor")
    _need(q_weight.numel() == head_dim and k_weight.numel() == head_dim, "norm weights must have head_dim entries")
    if loc is not None:
        _dev_i32(loc, "loc")
        _need(loc.numel() == T and k_cache is not None and v_cache is not None, "loc needs T entries and both caches")
        cs = _cache_stride(k_cache, v_cache, Hkv, head_dim)
    else:
        cs = Hkv * head_dim
    _check(_lib.msgl_qknorm_rope_store(
        qkv.data_ptr(), qs, q_weight.data_ptr(), k_weight.data_ptr(), float(eps), positions.data_ptr(),
        cos_sin_cache.data_ptr(), loc.data_ptr() if loc is not None else None,
        k_cache.data_ptr() if loc is not None else None, v_cache.data_ptr() if loc is not None else None,
        cs, T, Hq, Hkv, head_dim, 0 if reference_rounding else 1, _stream()))


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None, return_values: bool = False):
    """Greedy token per row of logits [bs, vocab] (bf16 or fp32); int32 [bs].  With return_values also the
    winning logit as fp32 [bs] (what tensor-parallel ranks exchange instead of their logits shards)."""
    _need(logits.is_cuda and logits.dtype in (torch.bfloat16, torch.float32), "logits must be bf16 or fp32 on the GPU")
    bs, vocab, rs = _rows_view(logits, "logits")
    if out is None:
        out = torch.empty(bs, dtype=torch.int32, device=logits.device)
    _dev_i32(out, "out")
    vals = torch.empty(bs, dtype=torch.float32, device=logits.device) if return_values else None
    _check(_lib.msgl_argmax_val(logits.data_ptr(), rs, logits.element_size(), out.data_ptr(),
                                vals.data_ptr() if vals is not None else None, bs, vocab, _stream()))
    return (out, vals) if return_values else out


def lm_head_argmax_workspace(bs: int, vocab: int, device) -> torch.Tensor:
    """Scratch tensor for ``lm_head_argmax`` at this batch size and (local) vocabulary; contents never matter."""
    return torch.empty(int(_lib.msgl_lm_head_argmax_workspace_bytes(int(bs), int(vocab))), dtype=torch.uint8, device=device)


def lm_head_argmax(hidden: torch.Tensor, weight: torch.Tensor, workspace: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None):
    """Greedy token and its logit straight from the final hidden state [bs, hidden] and the head's rows
    [vocab, hidden] (both bf16): ``(argmax, max)`` of ``F.linear(hidden, weight)`` per row, int32 [bs] and fp32 [bs], without
    the logits being written anywhere (see msgl_lm_head_argmax).  ``workspace``: from ``lm_head_argmax_workspace``
    (allocated here when omitted; a captured step passes its own)."""
    _need(hidden.is_cuda and weight.is_cuda and hidden.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16,
          "hidden and weight must be bf16 on the GPU")
    _need(hidden.dim() == 2 and weight.dim() == 2 and hidden.shape[1] == weight.shape[1] and hidden.stride(1) == 1 and weight.stride(1) == 1,
          "hidden [bs, k] and weight [vocab, k] with unit inner stride")
    bs, k = hidden.shape
    vocab = weight.shape[0]
    if workspace is None:
        workspace = lm_head_argmax_workspace(bs, vocab, hidden.device)
    if out is None:
        out = torch.empty(bs, dtype=torch.int32, device=hidden.device)
    _dev_i32(out, "out")
    vals = torch.empty(bs, dtype=torch.float32, device=hidden.device)
    _check(_lib.msgl_lm_head_argmax(hidden.data_ptr(), hidden.stride(0), weight.data_ptr(), weight.stride(0), out.data_ptr(), vals.data_ptr(),
                                    bs, vocab, k, workspace.data_ptr(), workspace.numel(), _stream()))
    return out, vals


def prepare_decode(req_pool_indices: torch.Tensor, seq_lens: torch.Tensor, last_slots: torch.Tensor, free_pages: torch.Tensor,
                   free_range: torch.Tensor, req_to_token: torch.Tensor, positions: torch.Tensor, out_cache_loc: torch.Tensor,
                   status: torch.Tensor, page_size: int, next_tokens: Optional[torch.Tensor] = None,
                   input_ids: Optional[torch.Tensor] = None, req_to_page: Optional[torch.Tensor] = None) -> None:
    """One decode step's batch preparation on the device (see msgl_prepare_decode): advances ``seq_lens``,
    ``last_slots``, ``free_range[0]`` and writes ``positions``, ``out_cache_loc``, the page table and (optionally)
    ``input_ids`` in place.  ``status`` (int32[1]) becomes non-zero if the step was refused."""
    bs = req_pool_indices.numel()
    for name, t, n in (("req_pool_indices", req_pool_indices, bs), ("seq_lens", seq_lens, bs), ("last_slots", last_slots, bs),
                       ("positions", positions, bs), ("out_cache_loc", out_cache_loc, bs), ("free_range", free_range, 2),
                       ("status", status, 1), ("free_pages", free_pages, free_pages.numel())):
        _dev_i32(t, name)
        _need(t.numel() == n, f"{name}: expected {n} elements, got {t.numel()}")
    _need(req_to_token.is_cuda and req_to_token.dtype == torch.int32 and req_to_token.dim() == 2 and req_to_token.stride(1) == 1,
          "req_to_token must be an int32 [rows, max_ctx] GPU tensor with contiguous rows")
    _need((next_tokens is None) == (input_ids is None), "next_tokens and input_ids go together")
    if next_tokens is not None:
        _dev_i32(next_tokens, "next_tokens")
        _dev_i32(input_ids, "input_ids")
        _need(next_tokens.numel() == bs and input_ids.numel() == bs, "next_tokens / input_ids: one entry per request")
    if req_to_page is not None:
        _need(req_to_page.is_cuda and req_to_page.dtype == torch.int32 and req_to_page.dim() == 2 and req_to_page.stride(1) == 1
              and req_to_page.shape[0] == req_to_token.shape[0], "req_to_page: int32 [rows, pages] on the GPU")
    _check(_lib.msgl_prepare_decode_paged(
        req_pool_indices.data_ptr(), seq_lens.data_ptr(), last_slots.data_ptr(), free_pages.data_ptr(), free_pages.numel(),
        free_range.data_ptr(), req_to_token.data_ptr(), req_to_token.stride(0),
        req_to_page.data_ptr() if req_to_page is not None else None, req_to_page.stride(0) if req_to_page is not None else 0,
        req_to_token.shape[1], positions.data_ptr(),
        out_cache_loc.data_ptr(), next_tokens.data_ptr() if next_tokens is not None else None,
        input_ids.data_ptr() if input_ids is not None else None, status.data_ptr(), bs, page_size, _stream()))


def stream_read(buf: torch.Tensor, sink: torch.Tensor, unroll: int = 8, groups_per_cu: int = 16, span_kib: int = 0) -> None:
    """Read every byte of `buf` once (16 B per lane, non-temporal, `unroll` loads in flight per lane, at most
    `groups_per_cu` work groups per compute unit) and store nothing: the benchmark's yardstick.  span_kib > 0: by
    contiguous spans of that many KiB, one work group per span (msgl_stream_read_span)."""
    _need(buf.is_cuda and buf.is_contiguous() and sink.is_cuda and sink.numel() * sink.element_size() >= 256, "GPU buffers expected")
    nbytes = buf.numel() * buf.element_size() // 16 * 16
    if span_kib > 0:
        _check(_lib.msgl_stream_read_span(buf.data_ptr(), nbytes, sink.data_ptr(), int(unroll), int(groups_per_cu), int(span_kib), _stream()))
    else:
        _check(_lib.msgl_stream_read_cfg(buf.data_ptr(), nbytes, sink.data_ptr(), int(unroll), int(groups_per_cu), _stream()))
