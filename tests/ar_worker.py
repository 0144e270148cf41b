"""One rank of the direct-read all-reduce test (started by test_xgmi_allreduce_gpu.py, one process per rank).
Ranks share the visible GPUs round robin: on a one-GPU box they all sit on cuda:0 and reach each other's memory
through the same IPC mapping they would use across xGMI."""
import functools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def rank_input(numel, seed, r):
    g = torch.Generator().manual_seed(seed * 131 + r)
    return (torch.randn(numel, generator=g) * (1 + r)).bfloat16()


@functools.lru_cache(maxsize=None)
def expected_sum(numel, seed, world):
    acc = torch.zeros(numel, dtype=torch.float32)
    for r in range(world):  # rank order, fp32, one rounding: what the kernel is specified to do
        acc += rank_input(numel, seed, r).float()
    return acc.bfloat16()


def tp_model(rank, world, dev):
    """The Qwen3-shaped stack on HIP kernels, tensor parallel over `world` ranks, with every o_proj / down_proj /
    embedding all-reduce on the direct-read kernels: prints the greedy tokens of the small end-to-end fixture case."""
    import json

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import gen_golden as gg
    from mini_sglang_amd import comm
    from mini_sglang_amd.mem_cache import MHAKVPool, PageAllocator, ReqToTokenPool
    from mini_sglang_amd.runtime.batch_info import BatchInfo, Req, SamplingParams
    from mini_sglang_amd.runtime.model_runner import ModelConfig, ModelRunner, ServerArgs
    from mini_sglang_amd.runtime.qwen3 import Qwen3Config

    name, steps = "tiny_ps16", 24
    cfg_kw, ps, (pseed, plens), nt, wseed, wstd, estd, succ = gg.E2E_CASES[name]
    cfg = Qwen3Config(**cfg_kw)
    args = ServerArgs(page_size=ps, device=str(dev), disable_cuda_graph=True, max_num_reqs=8, random_seed=wseed, weight_std=wstd,
                      embed_std=estd, successor_head=succ)
    pages = 1024 // ps
    r2t = ReqToTokenPool(8, cfg.max_position_embeddings, ps, str(dev))
    pa = PageAllocator(pages, ps)
    kv = MHAKVPool(pages * ps, ps, torch.bfloat16, cfg.num_key_value_heads // world, cfg.head_dim, cfg.num_hidden_layers, str(dev))
    mr = ModelRunner(ModelConfig(cfg), args, gpu_id=dev.index, tp_rank=rank, tp_size=world, req_to_token_pool=r2t, page_allocator=pa,
                     kv_cache_pool=kv)
    ar = comm.enable_xgmi_allreduce(max_bytes=1 << 20)
    prompts = gg.e2e_prompts(pseed, plens, cfg.vocab_size)
    reqs = [Req(i, p, SamplingParams(0.0)) for i, p in enumerate(prompts)]
    batch = BatchInfo.init_new(reqs, r2t, pa, kv)
    batch.prepare_for_extend()
    lg, _ = mr.forward_generate(batch)
    mr.process_result(batch, lg)
    for _ in range(steps - 1):
        batch.prepare_for_decode()
        lg, _ = mr.forward_generate(batch)
        mr.process_result(batch, lg)
    ar.check()
    toks = [r.token_ids[len(p):] for r, p in zip(reqs, prompts)]
    print("TOKENS " + json.dumps(toks), flush=True)
    comm.disable_xgmi_allreduce()
    dist.destroy_process_group()
    sys.exit(0)


def main():
    import faulthandler

    faulthandler.dump_traceback_later(60, exit=True)  # a stuck rank says where, then leaves
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    from mini_sglang_amd.comm import XgmiAllReduce

    if len(sys.argv) > 1 and sys.argv[1] == "tp-model":
        return tp_model(rank, world, dev)
    comm = XgmiAllReduce(max_bytes=4 << 20)
    sizes = [8, 512, 3 * 4096 + 8, 131072 + 8, 1 << 20]
    bad = []
    # every size in both forms
    for form, one_shot_bytes in (("one", 1 << 40), ("two", 1), ("auto", 0)):
        comm.one_shot_bytes = one_shot_bytes
        for n in sizes:
            x = rank_input(n, n, rank).to(dev)
            if rank == 1:
                time.sleep(0.002)  # skew the ranks
            comm.all_reduce(x)
            if not torch.equal(x.cpu(), expected_sum(n, n, world)):
                bad.append((form, n))
    # back-to-back calls of changing size without synchronising: buffer reuse
    comm.one_shot_bytes = 0
    seq = [sizes[(3 * i) % len(sizes)] for i in range(24)]
    ins = [rank_input(n, 1000 + i, rank).to(dev) for i, n in enumerate(seq)]
    outs = [torch.empty_like(t) for t in ins]
    torch.cuda.synchronize()
    for t, o in zip(ins, outs):
        comm.all_reduce(t, out=o)
    torch.cuda.synchronize()
    for i, (n, o) in enumerate(zip(seq, outs)):
        if not torch.equal(o.cpu(), expected_sum(n, 1000 + i, world)):
            bad.append(("chain", i, n))
    # captured and replayed
    a, b = torch.zeros(4096 * 16, dtype=torch.bfloat16, device=dev), torch.zeros(1 << 20, dtype=torch.bfloat16, device=dev)
    for _ in range(2):
        comm.all_reduce(a), comm.all_reduce(b)
    torch.cuda.synchronize()
    dist.barrier()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        comm.all_reduce(a)
        comm.all_reduce(b)
        comm.all_reduce(a)  # the sum of the sums: world * sum
    for it in range(4):
        a.copy_(rank_input(a.numel(), 5000 + it, rank)), b.copy_(rank_input(b.numel(), 6000 + it, rank))
        g.replay()
        torch.cuda.synchronize()
        ea = expected_sum(a.numel(), 5000 + it, world)
        acc = torch.zeros(a.numel(), dtype=torch.float32)
        for _ in range(world):
            acc += ea.float()
        if not torch.equal(a.cpu(), acc.bfloat16()) or not torch.equal(b.cpu(), expected_sum(b.numel(), 6000 + it, world)):
            bad.append(("graph", it))
    comm.check()
    # timing, for the log only
    x = torch.zeros(256 * 4096, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        comm.all_reduce(x)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(50):
        comm.all_reduce(x)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    comm.close()
    print(f"rank {rank}/{world} on {dev}: {'OK' if not bad else 'MISMATCH ' + str(bad[:8])}; 2 MiB all-reduce {us:.1f} us", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if not bad else 1)


if __name__ == "__main__":
    main()
