"""Model runner: owns the model, the pools and the attention backend, and runs extend / decode forwards.

Mirrors the parts of the reference's ModelRunner that sit around the hot path
(mini_sglang/managers/model_runner.py:22-273): backend selection by name (107-128, with ``hip`` added),
pool construction (161-191), ``forward_extend`` / ``forward_decode`` / ``forward_generate`` (193-229) and
greedy / temperature sampling of the next token (231-273).
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import torch

from ..mem_cache import MHAKVPool, PageAllocator, ReqToTokenPool
from .batch_info import BatchInfo
from .qwen3 import Qwen3Config, Qwen3ForCausalLM, _tp, init_weights


@dataclasses.dataclass
class ServerArgs:
    """The subset of the reference's flags the path depends on (mini_sglang/managers/server_args.py:16-49)."""
    page_size: int = 1
    attention_backend: str = "hip"
    max_num_reqs: int = 1024
    max_capture_bs: int = 128
    device: str = "cuda"
    disable_cuda_graph: bool = False
    random_seed: int = 0
    num_kv_tokens: int = 0          # pool capacity in tokens (0: size from free HBM like the reference)
    gpu_memory_utilization: float = 0.9
    kv_layout: str = "interleaved"
    weight_std: float = 0.02
    embed_std: Optional[float] = None
    successor_head: float = 0.0     # see runtime.qwen3.init_weights
    share_prefix_in_graphs: bool = True   # captured decode steps attend shared radix prefixes once per group of requests


@dataclasses.dataclass
class ModelConfig:
    """mini_sglang/utils/model_config.py:7-22, from a Qwen3Config instead of a HF directory."""
    hf_config: Qwen3Config
    dtype: torch.dtype = torch.bfloat16

    def __post_init__(self):
        c = self.hf_config
        self.hidden_size, self.num_hidden_layers, self.head_dim = c.hidden_size, c.num_hidden_layers, c.head_dim
        self.num_heads, self.num_kv_heads, self.vocab_size = c.num_attention_heads, c.num_key_value_heads, c.vocab_size
        self.max_context_len = c.max_position_embeddings
        self.kv_cache_dtype = self.dtype


ATTENTION_BACKENDS = ("hip",)


class ModelRunner:
    def __init__(self, model_config: ModelConfig, server_args: ServerArgs, gpu_id: int = 0, tp_rank: int = 0, tp_size: int = 1,
                 req_to_token_pool: Optional[ReqToTokenPool] = None, page_allocator: Optional[PageAllocator] = None,
                 kv_cache_pool: Optional[MHAKVPool] = None, layers=None, attn_backend_factory=None):
        self.device = server_args.device
        self.gpu_id, self.tp_rank, self.tp_size = gpu_id, tp_rank, tp_size
        self.page_size = server_args.page_size
        self.model_config = model_config
        self.server_args = server_args
        assert _tp()[1] == tp_size, "tp_size must equal the default process group's world size"

        prev = torch.get_default_dtype()
        torch.set_default_dtype(model_config.dtype)
        try:
            self.model = Qwen3ForCausalLM(model_config.hf_config, layers=layers).to(self.device)
        finally:
            torch.set_default_dtype(prev)
        init_weights(self.model, server_args.random_seed, std=server_args.weight_std, embed_std=server_args.embed_std,
                     successor_head=server_args.successor_head)

        if req_to_token_pool is None:
            self.init_memory_pool()
        else:
            assert page_allocator is not None and kv_cache_pool is not None
            self.req_to_token_pool, self.page_allocator, self.kv_cache_pool = req_to_token_pool, page_allocator, kv_cache_pool

        self.attn_backend = None
        self.init_attn_backend(attn_backend_factory)
        self.graph_runner = None
        if not server_args.disable_cuda_graph:
            from .graph_runner import HipGraphRunner
            self.graph_runner = HipGraphRunner(self)

    # ------------------------------------------------------------------ set-up
    def init_attn_backend(self, factory=None):
        """String switch of the reference (model_runner.py:107-128) with the ``hip`` backend; ``factory`` lets a test
        install an oracle-backed backend without adding a product-side name for it."""
        if factory is not None:
            self.attn_backend = factory(self)
        elif self.server_args.attention_backend == "hip":
            from ..attn.hip_attn_backend import HipAttnBackend
            self.attn_backend = HipAttnBackend(self)
            self.attn_backend.share_prefix_in_graphs = bool(self.server_args.share_prefix_in_graphs)
        else:
            raise ValueError(f"Unsupported attention backend: {self.server_args.attention_backend}")

    def init_memory_pool(self):
        cfg, args = self.model_config, self.server_args
        ps = self.page_size
        num_tokens = args.num_kv_tokens
        if num_tokens <= 0:  # model_runner.py:130-159: what is left of HBM after the weights
            free, total = torch.cuda.mem_get_info(self.gpu_id)
            cell = (cfg.num_kv_heads // self.tp_size) * cfg.head_dim * cfg.num_hidden_layers * 2 * 2
            num_tokens = int(total * args.gpu_memory_utilization - (total - free)) // (cell * ps) * ps
        num_pages = num_tokens // ps
        self.page_allocator = PageAllocator(num_pages, ps)
        self.req_to_token_pool = ReqToTokenPool(args.max_num_reqs, cfg.max_context_len, ps, self.device)
        self.kv_cache_pool = MHAKVPool(num_pages * ps, ps, cfg.kv_cache_dtype, cfg.num_kv_heads // self.tp_size, cfg.head_dim,
                                       cfg.num_hidden_layers, self.device, layout=args.kv_layout)

    # ------------------------------------------------------------------ forwards
    @torch.inference_mode()
    def forward_extend(self, batch: BatchInfo) -> torch.Tensor:
        assert batch.forward_mode.is_extend()
        batch.attn_backend = self.attn_backend
        batch.attn_backend.init_forward_metadata(batch)
        return self.model(batch.input_ids, batch.positions, batch)

    @torch.inference_mode()
    def forward_decode(self, batch: BatchInfo) -> torch.Tensor:
        assert batch.forward_mode.is_decode()
        batch.attn_backend = self.attn_backend
        batch.attn_backend.init_forward_metadata(batch)
        return self.model(batch.input_ids, batch.positions, batch)

    @torch.inference_mode()
    def forward_generate(self, batch: BatchInfo):
        """(logits, ran_in_graph) — model_runner.py:211-229."""
        if self.graph_runner is not None and self.graph_runner.can_run(batch):
            batch.attn_backend = self.attn_backend
            return self.graph_runner.replay(batch), True
        if batch.forward_mode.is_extend():
            return self.forward_extend(batch), False
        return self.forward_decode(batch), False

    # ------------------------------------------------------------------ sampling
    def sample(self, logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
        """mini_sglang/layers/sampler.py:13-23: greedy where temperature == 0, else Gumbel-max sampling."""
        if bool((temperatures == 0).all()):
            if logits.is_cuda:
                from .. import ops
                return ops.argmax(logits).long()
            return logits.float().argmax(dim=-1)
        lf = logits.to(torch.float)
        greedy = lf.argmax(dim=-1)
        probs = torch.softmax(lf / temperatures.unsqueeze(1), dim=-1, dtype=torch.float)
        sampled = probs.div_(torch.empty_like(probs).exponential_(1) + 1e-10).argmax(dim=-1)
        return torch.where(temperatures == 0, greedy, sampled)

    def process_result(self, batch: BatchInfo, logits: torch.Tensor) -> List[int]:
        """Append one sampled token per request (model_runner.py:231-273)."""
        temps = torch.tensor([r.sampling_params.temperature for r in batch.reqs], dtype=torch.float32, device=logits.device)
        ids = self.sample(logits, temps).tolist()
        for r, t in zip(batch.reqs, ids):
            r.token_ids.append(int(t))
        return ids
