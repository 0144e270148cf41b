"""Layer modules with the reference's signatures, backed by HIP kernels only."""
from .activation import SiluAndMul
from .attn_layer import Attention
from .layernorm import RMSNorm
from .rotary_embedding import RotaryEmbedding, get_rope
from ..ops import qknorm_rope_store_ as _qknorm_rope_store_


def qknorm_rope_store(qkv, q_weight, k_weight, eps, positions, cos_sin_cache, num_heads, num_kv_heads, head_dim,
                      out_cache_loc, k_cache, v_cache, reference_rounding=True):
    """Fused replacement for the reference's per-head q_norm / k_norm -> rotary_emb -> set_kv_buffer sequence
    (mini_sglang/models/qwen3.py:88-94, mini_sglang/mem_cache/token2kv.py:141-142): one kernel over the qkv
    GEMM output, in place, that also scatters the rotated k and v rows into the layer's pool."""
    _qknorm_rope_store_(qkv, q_weight, k_weight, eps, positions, cos_sin_cache, num_heads, num_kv_heads, head_dim,
                        loc=out_cache_loc, k_cache=k_cache, v_cache=v_cache, reference_rounding=reference_rounding)

__all__ = ["RMSNorm", "SiluAndMul", "RotaryEmbedding", "get_rope", "Attention", "qknorm_rope_store"]
