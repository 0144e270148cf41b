"""Tensor-parallel path on the CPU: two ranks over gloo run the Qwen3-shaped stack with KV heads, MLP columns
and vocabulary sharded exactly as on GPUs (attention needs no collective; o_proj / down_proj all-reduce,
embedding all-reduce, logits all-gather) and must reproduce the single-rank greedy stream of the reference
fixture.  Oracle layers stand in for the HIP kernels, which need a GPU."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, steps, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import golden_cases as gc
        from oracle import modules_oracle as mo
        from mini_sglang_amd.mem_cache import MHAKVPool, PageAllocator, ReqToTokenPool
        from mini_sglang_amd.runtime.batch_info import BatchInfo, Req, SamplingParams
        from mini_sglang_amd.runtime.model_runner import ModelConfig, ModelRunner, ServerArgs
        from mini_sglang_amd.runtime.qwen3 import Qwen3Config
        from oracle import gen_golden as gg

        cfg_kw, ps, (pseed, plens), nt, wseed, wstd, estd, succ = gg.E2E_CASES[name]
        cfg = Qwen3Config(**cfg_kw)
        args = ServerArgs(page_size=ps, device="cpu", disable_cuda_graph=True, max_num_reqs=8, random_seed=wseed, weight_std=wstd,
                          embed_std=estd, successor_head=succ, kv_layout="split")
        pages = 1024 // ps
        r2t = ReqToTokenPool(8, cfg.max_position_embeddings, ps, "cpu")
        pa = PageAllocator(pages, ps)
        kv = MHAKVPool(pages * ps, ps, torch.bfloat16, cfg.num_key_value_heads // world, cfg.head_dim, cfg.num_hidden_layers, "cpu", layout="split")
        mr = ModelRunner(ModelConfig(cfg), args, tp_rank=rank, tp_size=world, req_to_token_pool=r2t, page_allocator=pa, kv_cache_pool=kv,
                         layers=mo, attn_backend_factory=mo.OracleAttnBackend)
        assert mr.model.layers[0].self_attn.num_kv_heads == cfg.num_key_value_heads // world
        prompts = gg.e2e_prompts(pseed, plens, cfg.vocab_size)
        reqs = [Req(i, p, SamplingParams(0.0)) for i, p in enumerate(prompts)]
        batch = BatchInfo.init_new(reqs, r2t, pa, kv)
        tops = []
        batch.prepare_for_extend()
        lg, _ = mr.forward_generate(batch)
        tops.append(lg.float().max(dim=-1).values)
        mr.process_result(batch, lg)
        def cpu_argmax(local):  # stands in for the HIP argmax: (index, value) of this rank's vocabulary shard
            i = local.float().argmax(dim=-1)
            return i.to(torch.int32), local.float().gather(1, i[:, None])[:, 0]

        for _ in range(steps - 1):
            batch.prepare_for_decode()
            lg, _ = mr.forward_generate(batch)
            tops.append(lg.float().max(dim=-1).values)
            # sampling on the shards (pairs exchanged instead of logits) picks the same tokens as the gathered logits
            sharded = mr.model.forward_greedy(batch.input_ids, batch.positions, batch, cpu_argmax)
            assert torch.equal(sharded.long(), lg.float().argmax(dim=-1))
            mr.process_result(batch, lg)
        # ties between shards go to the lower token id, NaN counts as the maximum
        from mini_sglang_amd.runtime.qwen3 import pick_sharded_argmax
        val = torch.tensor([1.0, 2.0 + rank, 5.0 - rank, float("nan") if rank == 1 else 9.0, 3.0])
        idx = torch.tensor([7, 3, 4, 2, 0], dtype=torch.int32)
        assert pick_sharded_argmax(idx, val, 100).tolist() == [7, 103, 4, 102, 0]
        toks = torch.tensor([r.token_ids[len(p):] for r, p in zip(reqs, prompts)], dtype=torch.int32)
        # every rank must hold the same logits after the all-gather, hence the same tokens
        gathered = [torch.empty_like(toks) for _ in range(world)]
        dist.all_gather(gathered, toks)
        assert all(torch.equal(g, toks) for g in gathered)
        if rank == 0:
            torch.save({"tokens": toks, "top1": torch.stack(tops)}, out)
    finally:
        dist.destroy_process_group()


def _spawn_with_deadline(fn, args, nprocs, seconds):
    """mp.spawn with a time limit of its own (no plugin needed): ranks still alive at the deadline are ended and
    the test fails."""
    import time

    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.monotonic() + seconds
    while not ctx.join(timeout=5):
        if time.monotonic() > deadline:
            for p in ctx.processes:
                if p.is_alive():
                    p.terminate()
            pytest.fail(f"ranks still running after {seconds} s")


def test_tp2_gloo_matches_reference_stream(tmp_path):
    import golden_cases as gc

    name, steps = "tiny_ps16", 24
    out = str(tmp_path / "tp2.pt")
    port = 29650 + os.getpid() % 200
    _spawn_with_deadline(_worker, (2, port, name, steps, out), 2, 600)
    got = torch.load(out, weights_only=True)
    fx = gc.load(f"e2e_{name}.pt")
    assert torch.equal(got["tokens"], fx["tokens"][:, :steps])
    # partial sums are reduced in a different order than one rank's GEMM: logits agree to bf16 rounding
    assert (got["top1"] - fx["top1_logits"][:steps]).abs().max().item() <= 5e-2
