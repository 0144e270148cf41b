"""One rank of the direct-read all-reduce test (started by test_xgmi_allreduce_gpu.py, one process per rank).
Ranks share the visible GPUs round robin: on a one-GPU box they all sit on cuda:0 and reach each other's memory
through the same IPC mapping they would use across xGMI.  Ranks that share a device issue every call step by step
with host barriers in between (the communicator picks that by itself): whether kernels of two processes run at
the same time on one GPU is up to the driver, and the one-launch form, in which a kernel waits for its peer's
kernel, is exercised only after a probe has shown that they do ("fused" mode below)."""
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
    print(f"rank {rank}: all-reduce {'step by step (ranks share a device)' if ar.stepped else 'in one launch'}", flush=True)
    prompts = gg.e2e_prompts(pseed, plens, cfg.vocab_size)
    reqs = [Req(i, p, SamplingParams(0.0)) for i, p in enumerate(prompts)]
    batch = BatchInfo.init_new(reqs, r2t, pa, kv)
    for step in range(steps):
        batch.prepare_for_extend() if step == 0 else batch.prepare_for_decode()
        lg, _ = mr.forward_generate(batch)
        try:
            ar.check()  # a wait that ran out is reported at the step it happened in
        except RuntimeError as e:
            print(f"rank {rank}: forward {step} ({'extend' if step == 0 else 'decode'}): {e}", flush=True)
            raise
        mr.process_result(batch, lg)
    toks = [r.token_ids[len(p):] for r, p in zip(reqs, prompts)]
    print("TOKENS " + json.dumps(toks), flush=True)
    comm.disable_xgmi_allreduce()
    dist.destroy_process_group()
    sys.exit(0)


SIZES = [8, 512, 3 * 4096 + 8, 131072 + 8, 1 << 20]


def every_size_in_both_forms(comm, rank, world, dev, bad):
    for form, one_shot_bytes in (("one", 1 << 40), ("two", 1), ("auto", 0)):
        comm.one_shot_bytes = one_shot_bytes
        for n in SIZES:
            x = rank_input(n, n, rank).to(dev)
            if rank == 1:
                time.sleep(0.002)  # skew the ranks
            comm.all_reduce(x)
            if not torch.equal(x.cpu(), expected_sum(n, n, world)):
                bad.append((form, n))
    comm.one_shot_bytes = 0


def chain_of_changing_sizes(comm, rank, world, dev, bad):
    """Back-to-back calls without synchronising in which the grid size and the form change from call to call
    (1 work group, 9-group two-step, 32-group one-step, ...) and one rank lags: what the staging buffers'
    reuse has to survive."""
    one_vec = 8
    seq = []
    for i in range(36):
        # (elements, one_shot_bytes): tiny one-step, two-step of 9 groups per slice, wide one-step, long two-step
        seq.append([(one_vec * 64, 1 << 40), (9 * 512 * 8 * world, 1), (32 * 512 * 8, 1 << 40), (1 << 20, 1)][(i * 7 + i // 5) % 4])
    ins = [rank_input(n, 1000 + i, rank).to(dev) for i, (n, _) in enumerate(seq)]
    outs = [torch.empty_like(t) for t in ins]
    torch.cuda.synchronize()
    dist.barrier()
    for i, (t, o, (_, osb)) in enumerate(zip(ins, outs, seq)):
        comm.one_shot_bytes = osb
        if rank == (i % world) and i % 3 == 0:
            time.sleep(0.001)  # this rank's launch comes late
        comm.all_reduce(t, out=o)
    comm.one_shot_bytes = 0
    torch.cuda.synchronize()
    for i, ((n, _), o) in enumerate(zip(seq, outs)):
        if not torch.equal(o.cpu(), expected_sum(n, 1000 + i, world)):
            bad.append(("chain", i, n))


def captured_and_replayed(comm, rank, world, dev, bad):
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


def everyone_agrees(ok: bool) -> bool:
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t[0]))


def main():
    import faulthandler

    faulthandler.dump_traceback_later(240, exit=True)  # a stuck rank says where, then leaves
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    from mini_sglang_amd.comm import XgmiAllReduce

    mode = sys.argv[1] if len(sys.argv) > 1 else "stepped"
    if mode == "tp-model":
        return tp_model(rank, world, dev)
    bad = []
    if mode == "stepped":
        comm = XgmiAllReduce(max_bytes=4 << 20)  # ranks on one device: step by step, chosen by the communicator
        if world > torch.cuda.device_count():
            assert comm.shared_device and comm.stepped, "ranks on one device must be recognised as such"
        comm.stepped = True
        every_size_in_both_forms(comm, rank, world, dev, bad)
        chain_of_changing_sizes(comm, rank, world, dev, bad)
        comm.check()
        us = 0.0
    else:  # "fused": the one-launch form, which needs the ranks' kernels to run side by side
        comm = XgmiAllReduce(max_bytes=4 << 20, stepped=False, timeout_ms=300)
        x = rank_input(512, 512, rank).to(dev)
        dist.barrier()
        comm.all_reduce(x)
        try:
            comm.check()
            ok = torch.equal(x.cpu(), expected_sum(512, 512, world))
        except RuntimeError:
            ok = False
        if not everyone_agrees(ok):
            print("NOT-CONCURRENT: kernels of the ranks' processes do not run side by side on this device", flush=True)
            comm.close()
            dist.destroy_process_group()
            sys.exit(0)
        comm.set_timeout_ms(500)
        every_size_in_both_forms(comm, rank, world, dev, bad)
        chain_of_changing_sizes(comm, rank, world, dev, bad)
        captured_and_replayed(comm, rank, world, dev, bad)
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
        # A wait that ran out after the probe had passed means that the processes' kernels stopped overlapping at some
        # point (the GPU's scheduler decides that, not this code): reported like a failed probe.  Wrong sums without
        # any wait having run out are a failure of the protocol.
        try:
            comm.check()
            waited_out = False
        except RuntimeError:
            waited_out = True
        if not everyone_agrees(not waited_out):
            print("NOT-CONCURRENT: a wait for the peer's kernel ran out after the probe had passed", flush=True)
            comm.close()
            dist.destroy_process_group()
            sys.exit(0)
    comm.close()
    print(f"rank {rank}/{world} on {dev} ({mode}): {'OK' if not bad else 'MISMATCH ' + str(bad[:8])}; 2 MiB all-reduce {us:.1f} us", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if not bad else 1)


if __name__ == "__main__":
    main()
