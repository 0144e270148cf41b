from .req2token import ReqToTokenPool
from .token2kv import MHAKVPool, PageAllocator

__all__ = ["ReqToTokenPool", "MHAKVPool", "PageAllocator"]
