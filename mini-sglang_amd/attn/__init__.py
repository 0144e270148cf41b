from .attn_backend import AttentionBackend

__all__ = ["AttentionBackend"]
