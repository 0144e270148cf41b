"""Layer modules with the reference's signatures, backed by HIP kernels only."""
from .activation import SiluAndMul
from .attention import Attention
from .layernorm import RMSNorm
from .rotary_embedding import RotaryEmbedding, get_rope

__all__ = ["RMSNorm", "SiluAndMul", "RotaryEmbedding", "get_rope", "Attention"]
