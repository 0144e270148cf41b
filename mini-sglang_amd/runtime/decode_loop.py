"""Steady-state decode loop on synthetic contexts: the measurement harness for "decode tok/s".

Builds a ModelRunner for a Qwen3-shaped model, gives every request a context of ``ctx`` tokens by writing a
page table (pages drawn in random order, optionally with shared page-aligned prefixes) and filling the pool
with random K/V — a real prefill of 256 x 8k tokens would only produce the same tensors slowly — and then
runs ordinary decode steps: ``prepare_for_decode`` -> ``forward_generate`` (hipGraph replay when captured) ->
greedy sampling -> append.  One step produces one token per request.
"""
from __future__ import annotations

import time
from typing import Optional

import torch

from .batch_info import BatchInfo, ForwardMode, Req, SamplingParams
from .model_runner import ModelConfig, ModelRunner, ServerArgs
from .qwen3 import Qwen3Config, _tp
from . import synth, workload


def build_decode_state(cfg: Qwen3Config, bs: int, ctx: int, page_size: int, steps_budget: int, *, device="cuda",
                       use_graph: bool = True, seed: int = 0, groups=None, group_prefix_lens=None, seq_lens=None,
                       kv_layout: str = "interleaved", max_ctx: Optional[int] = None, share_prefix_in_graphs: bool = True):
    rank, tp = _tp()
    ps = page_size
    seq_lens = list(seq_lens) if seq_lens is not None else [ctx] * bs
    if groups is None:
        ctx_pages = sum(-(-s // ps) for s in seq_lens)
    else:
        ctx_pages = sum(-(-s // ps) for s in seq_lens) - sum(
            (sum(1 for g in groups if g == gi) - 1) * (plen // ps) for gi, plen in enumerate(group_prefix_lens))
    grow_pages = bs * (-(-steps_budget // ps) + 1)       # pages the decode steps may take
    pool_pages = ctx_pages + grow_pages
    max_ctx = max_ctx or -(-(max(seq_lens) + steps_budget + ps) // ps) * ps
    cfg.max_position_embeddings = max(max_ctx, 2048)
    args = ServerArgs(page_size=ps, device=device, disable_cuda_graph=not use_graph, max_num_reqs=bs, max_capture_bs=bs,
                      random_seed=seed, num_kv_tokens=pool_pages * ps, kv_layout=kv_layout, weight_std=0.01, embed_std=0.3,
                      successor_head=0.02 if not cfg.tie_word_embeddings else 0.0,
                      share_prefix_in_graphs=share_prefix_in_graphs)
    mr = ModelRunner(ModelConfig(cfg), args, gpu_id=torch.cuda.current_device() if device != "cpu" else 0, tp_rank=rank, tp_size=tp)
    # contexts live in pages 1..ctx_pages (drawn in a seeded random order, the same table on every rank); the
    # allocator keeps pages ctx_pages+1.. for the tokens the decode steps append
    pt = workload.make_page_table(seq_lens, ps, ctx_pages, seed + 1, groups=groups, group_prefix_lens=group_prefix_lens,
                                  max_ctx=cfg.max_position_embeddings)
    mr.page_allocator.free_page = torch.arange(ctx_pages + 1, pool_pages + 1, dtype=torch.int32)
    rows = mr.req_to_token_pool.alloc(bs)
    mr.req_to_token_pool.load_rows(torch.tensor(rows, device=device), pt.req_to_token.to(device))
    for layer in range(cfg.num_hidden_layers):  # random K/V: a prefill would produce the same kind of tensor, slowly
        k, v = mr.kv_cache_pool.get_kv_buffer(layer)
        if kv_layout == "interleaved":
            k._base.normal_()
        else:
            k.normal_()
            v.normal_()
    reqs = []
    for i, S in enumerate(seq_lens):
        r = Req(i, synth.uniform_int(1, seed + 7 + i, 0, cfg.vocab_size).tolist() * 1, SamplingParams(0.0))
        r.token_ids = [0] * (S - 1) + r.token_ids  # only the length and the last token matter for decode
        r.req_pool_idx = rows[i]
        r.last_slot = int(pt.req_to_token[i, S - 1])
        if groups is not None and group_prefix_lens[groups[i]] > 0:  # as prepare_for_extend records a radix-cache hit
            P = group_prefix_lens[groups[i]]
            r.shared_key = (P, int(pt.req_to_token[i, 0]), int(pt.req_to_token[i, P - 1]))
        reqs.append(r)
    batch = BatchInfo.init_new(reqs, mr.req_to_token_pool, mr.page_allocator, mr.kv_cache_pool)
    batch.forward_mode = ForwardMode.DECODE
    batch.seq_lens_cpu = torch.tensor(seq_lens, dtype=torch.int32)
    batch.seq_lens = batch.seq_lens_cpu.to(device)
    batch.req_pool_indices = torch.tensor(rows, dtype=torch.int32, device=device)
    return mr, batch


def decode_steps(mr: ModelRunner, batch: BatchInfo, n: int) -> None:
    for _ in range(n):
        batch.prepare_for_decode()
        logits, _ = mr.forward_generate(batch)
        mr.process_result(batch, logits)


def time_decode(mr: ModelRunner, batch: BatchInfo, steps: int, warmup: int, barrier=None) -> float:
    """Seconds for `steps` decode steps (host preparation, graph replay, greedy sampling and the token read-back
    included), after `warmup` untimed steps."""
    decode_steps(mr, batch, warmup)
    if barrier is not None:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    decode_steps(mr, batch, steps)
    if barrier is not None:
        barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def time_decode_chained(mr: ModelRunner, batch: BatchInfo, steps: int, warmup: int, chain: int = 8, barrier=None) -> float:
    """Seconds for `steps` decode steps run `chain` at a time on the device (DeviceDecodeLoop: preparation,
    forward and greedy sampling captured together, one host synchronisation and token read-back per chain)."""
    from .device_decode import DeviceDecodeLoop

    loop = DeviceDecodeLoop(mr, batch, max_steps=max(chain, warmup))
    if warmup:
        loop.run(warmup)
        loop.sync_host()
    if barrier is not None:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    while done < steps:
        k = min(chain, steps - done)
        loop.run(k)
        loop.sync_host()
        done += k
    if barrier is not None:
        barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def time_decode_samples(mr: ModelRunner, batch: BatchInfo, samples: int, steps_each: int, warmup: int, barrier=None):
    """`samples` consecutive timings of `steps_each` decode steps (as time_decode), after `warmup` untimed steps:
    seconds per sample, in order (the contexts grow by a token per step, which the spread then includes)."""
    out = [time_decode(mr, batch, steps_each, warmup, barrier=barrier)]
    for _ in range(samples - 1):
        out.append(time_decode(mr, batch, steps_each, 0, barrier=barrier))
    return out


def time_decode_chained_samples(mr: ModelRunner, batch: BatchInfo, samples: int, chain: int, warmup: int, barrier=None):
    """`samples` timings of one device-side chain of `chain` decode steps each (see time_decode_chained)."""
    from .device_decode import DeviceDecodeLoop

    loop = DeviceDecodeLoop(mr, batch, max_steps=max(chain, warmup))
    if warmup:
        loop.run(warmup)
        loop.sync_host()
    out = []
    for _ in range(samples):
        if barrier is not None:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.run(chain)
        loop.sync_host()
        if barrier is not None:
            barrier()
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    return out
