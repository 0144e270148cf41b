from .req2token import ReqToTokenPool
from .token2kv import MHAKVPool, PageAllocator

from .radix_cache import MatchResult, RadixCache

__all__ = ["ReqToTokenPool", "MHAKVPool", "PageAllocator", "RadixCache", "MatchResult"]
