from .backend_api import AttentionBackend

__all__ = ["AttentionBackend"]
