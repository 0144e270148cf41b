"""The attention backend driven through its hooks alone, in the order the reference's graph runner calls them
(mini_sglang/managers/cuda_graph_runner.py:84-171): static input buffers, ``init_cuda_graph_state(max_bs)`` once,
then per batch size, LARGEST FIRST, ``init_forward_metadata_capture_cuda_graph`` -> two eager warm-up forwards
-> capture; a replay pads the batch to the next captured size (lengths 1, out_cache_loc 0, the row indices of the
padded entries left stale), copies the inputs, calls ``init_forward_metadata_replay_cuda_graph`` and replays.
Nothing else of this repository's runner takes part, and nothing is poked into the backend from outside: whatever
the captured kernels need (the split workspace above all: bs * splits is not monotone in bs, 128 x 32 < 112 x 37)
has to come from the constructor and ``init_cuda_graph_state``.  Replayed outputs are compared with the oracle."""
import bisect
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

from oracle import torch_oracle as orc  # noqa: E402

DEV = "cuda"
D = 128


class HooksOnlyGraphRunner:
    """What a CudaGraphRunner does with a backend, for a 'model' that is one attention layer."""

    def __init__(self, backend, layer, forward, max_bs):
        from mini_sglang_amd.runtime.batch_info import BatchInfo, ForwardMode

        self.backend, self.layer, self.forward = backend, layer, forward
        self.BatchInfo, self.mode = BatchInfo, ForwardMode.DECODE
        i32 = dict(dtype=torch.int32, device=DEV)
        self.input_ids = torch.zeros(max_bs, **i32)
        self.positions = torch.zeros(max_bs, **i32)
        self.out_cache_loc = torch.zeros(max_bs, **i32)
        self.seq_lens = torch.ones(max_bs, **i32)
        self.req_pool_indices = torch.zeros(max_bs, **i32)
        backend.init_cuda_graph_state(max_bs)
        self.graph_bs = [1, 2, 4, 8] + list(range(16, max_bs + 1, 16))
        self.graphs, self.outputs, pool = {}, {}, None
        with torch.inference_mode():
            for bs in reversed(self.graph_bs):
                batch = self.BatchInfo(forward_mode=self.mode, input_ids=self.input_ids[:bs], positions=self.positions[:bs],
                                       seq_lens=self.seq_lens[:bs], out_cache_loc=self.out_cache_loc[:bs],
                                       req_pool_indices=self.req_pool_indices[:bs], attn_backend=backend)
                backend.init_forward_metadata_capture_cuda_graph(bs, batch.forward_mode)
                for _ in range(2):
                    torch.cuda.synchronize()
                    forward(batch)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    out = forward(batch)
                pool = g.pool()
                self.graphs[bs], self.outputs[bs] = g, out
                torch.cuda.synchronize()

    def replay(self, seq_lens_cpu, req_pool_indices, out_cache_loc):
        raw = seq_lens_cpu.numel()
        bs = self.graph_bs[bisect.bisect_left(self.graph_bs, raw)]
        if bs != raw:
            self.seq_lens.fill_(1)
            self.out_cache_loc.zero_()
        self.seq_lens[:raw].copy_(seq_lens_cpu)
        self.out_cache_loc[:raw].copy_(out_cache_loc)
        self.req_pool_indices[:raw].copy_(req_pool_indices)
        self.backend.init_forward_metadata_replay_cuda_graph(bs, req_pool_indices=self.req_pool_indices, seq_lens=self.seq_lens,
                                                             seq_lens_cpu=seq_lens_cpu, forward_mode=self.mode)
        self.graphs[bs].replay()
        return self.outputs[bs][:raw]


def test_backend_under_a_hooks_only_graph_runner_matches_oracle():
    from mini_sglang_amd.attn.hip_attn_backend import HipAttnBackend, splits_for_batch
    from mini_sglang_amd.layers.attn_layer import Attention
    from mini_sglang_amd.mem_cache import MHAKVPool, ReqToTokenPool
    from mini_sglang_amd.runtime import synth, workload

    Hq, Hkv, ps, max_bs, max_ctx = 16, 8, 16, 128, 4096
    assert 112 * splits_for_batch(112) > 128 * splits_for_batch(128)  # the case a lazily grown workspace gets wrong
    lens_all = [int(x) for x in synth.uniform_int(max_bs, 21, 40, 3500).tolist()]
    lens_all[0], lens_all[1], lens_all[5] = 3500, 1, 16
    pages = sum(-(-(n + 1) // ps) for n in lens_all) + 8
    pt = workload.make_page_table([n + 1 for n in lens_all], ps, pages, seed=4)  # room for the token each step stores
    r2t_pool = ReqToTokenPool(max_bs + 8, max_ctx, ps, DEV)
    r2t_pool.req_to_token[:max_bs, : pt.req_to_token.shape[1]] = pt.req_to_token.to(DEV)
    kv_pool = MHAKVPool(pages * ps, ps, torch.bfloat16, Hkv, D, 1, DEV)
    kv_host = synth.normal(((pages + 1) * ps, 2, Hkv, D), 9)
    kv_pool.k_buffer[0].copy_(kv_host[:, 0].to(DEV)), kv_pool.v_buffer[0].copy_(kv_host[:, 1].to(DEV))
    # what a reference ModelRunner exposes to a backend's constructor (model_runner.py:35-63), and nothing more
    runner = types.SimpleNamespace(device=DEV, req_to_token_pool=r2t_pool, kv_cache_pool=kv_pool, page_size=ps, tp_size=1,
                                   model_config=types.SimpleNamespace(max_context_len=max_ctx, num_heads=Hq, num_kv_heads=Hkv, head_dim=D))
    backend = HipAttnBackend(runner)
    layer = Attention(Hq, D, D ** -0.5, Hkv, 0)
    q_static = torch.zeros((max_bs, Hq * D), dtype=torch.bfloat16, device=DEV)
    k_static = torch.zeros((max_bs, Hkv * D), dtype=torch.bfloat16, device=DEV)
    v_static = torch.zeros((max_bs, Hkv * D), dtype=torch.bfloat16, device=DEV)

    def forward(batch):
        bs = batch.seq_lens.numel()
        return layer(q_static[:bs], k_static[:bs], v_static[:bs], batch)

    gr = HooksOnlyGraphRunner(backend, layer, forward, max_bs)
    ws_ptr, ws_bytes = backend._graph_workspace.data_ptr(), backend._graph_workspace.numel()
    from mini_sglang_amd import ops
    assert ws_bytes >= max(ops.decode_workspace_bytes(b, Hq, D, splits_for_batch(b)) for b in range(1, max_bs + 1))

    step = 0
    # the largest graph first, then one that pads 100 -> 112 (more work items than 128), then small ones, then 128 again
    for raw in (128, 100, 112, 33, 16, 3, 1, 128):
        step += 1
        lens = [n + 1 for n in lens_all[:raw]]                   # the step appends one token to every request
        sl = torch.tensor(lens, dtype=torch.int32)
        rows = torch.arange(raw, dtype=torch.int32)
        loc = pt.req_to_token[rows.long(), (sl - 1).long()]      # slot of the new token
        q = synth.normal((raw, Hq, D), 100 + step)
        k_new, v_new = synth.normal((raw, Hkv, D), 200 + step), synth.normal((raw, Hkv, D), 300 + step)
        q_static[:raw].copy_(q.view(raw, -1).to(DEV)), k_static[:raw].copy_(k_new.view(raw, -1).to(DEV))
        v_static[:raw].copy_(v_new.view(raw, -1).to(DEV))
        # an eager decode in between whose split count exceeds what any captured size uses must not disturb the graphs
        if raw == 112:
            from mini_sglang_amd.runtime.batch_info import BatchInfo, ForwardMode
            eb = BatchInfo(forward_mode=ForwardMode.DECODE, seq_lens=sl[:3].to(DEV), seq_lens_cpu=sl[:3],
                           out_cache_loc=loc[:3].to(DEV), req_pool_indices=rows[:3].to(DEV), attn_backend=backend)
            backend.init_forward_metadata(eb)
            backend.forward_metadata.max_splits, backend.forward_metadata.chunk_tokens = 3000, 16
            backend.forward_metadata.decode_plan = None
            with torch.inference_mode():
                backend.forward_decode(q_static[:3], k_static[:3].view(3, Hkv, D), v_static[:3].view(3, Hkv, D), layer, eb, save_kv_cache=False)
        out = gr.replay(sl, rows.to(DEV), loc.to(DEV)).view(raw, Hq, D).cpu()
        torch.cuda.synchronize()
        kv_host[loc.long(), 0], kv_host[loc.long(), 1] = k_new, v_new   # what the step stored
        ref = orc.paged_decode_ref(q, kv_host[:, 0], kv_host[:, 1], pt.req_to_token, rows, sl, D ** -0.5)
        err = (out.float() - ref.float()).abs().max().item()
        assert err <= 2e-2, f"replay of batch {raw}: max abs err {err} vs oracle"
        assert torch.equal(kv_pool.k_buffer[0][loc.long().to(DEV)].cpu(), k_new), "the step's K rows are in the pool"
        assert (backend._graph_workspace.data_ptr(), backend._graph_workspace.numel()) == (ws_ptr, ws_bytes), \
            "the workspace captured graphs point at must never be replaced"


def test_decode_tails_in_the_last_page_of_the_pool():
    """Every request's last page is among the pool's last pages and the K/V allocation ends with the pool: with 16-token
    tiles read from one page-table entry, a tile past a split's end must fall back on the last valid TILE, not on the
    last valid token (16 rows from there would run past the page and the allocation)."""
    from mini_sglang_amd import ops
    from mini_sglang_amd.runtime import synth

    ps, Hq, Hkv = 16, 32, 8
    lens = [1, 15, 17, 33, 250, 1000, 1009, 2047]
    pages_of = [-(-n // ps) for n in lens]
    total = sum(pages_of)
    slots = (total + 1) * ps                     # page 0 is the dummy page; nothing behind the last page
    r2t = torch.zeros((len(lens), 2048), dtype=torch.int32)
    # pages dealt from the END of the pool, each request's LAST page first
    nxt = total
    order = [[] for _ in lens]
    for b, n in enumerate(pages_of):
        order[b].append(nxt)
        nxt -= 1
    for b, n in enumerate(pages_of):
        for _ in range(n - 1):
            order[b].insert(0, nxt)
            nxt -= 1
    for b, pg in enumerate(order):
        for i, p_ in enumerate(pg):
            r2t[b, i * ps:(i + 1) * ps] = torch.arange(p_ * ps, (p_ + 1) * ps, dtype=torch.int32)
    kv = synth.normal((slots, 2, Hkv, D), 5)
    q = synth.normal((len(lens), Hq, D), 6)
    idx = torch.arange(len(lens), dtype=torch.int32)
    sl = torch.tensor(lens, dtype=torch.int32)
    kvd = kv.to(DEV)
    ref = orc.paged_decode_ref(q, kv[:, 0], kv[:, 1], r2t, idx, sl, D ** -0.5)
    for max_splits, chunk in ((1, 0), (8, 256), (64, 32), (128, 16)):
        out = ops.paged_decode(q.to(DEV), kvd[:, 0], kvd[:, 1], r2t.to(DEV), idx.to(DEV), sl.to(DEV), D ** -0.5, ps,
                               max_splits=max_splits, chunk_tokens=chunk).cpu()
        err = (out.float() - ref.float()).abs().max().item()
        assert err <= 2e-2, f"max_splits {max_splits}: max abs err {err}"
