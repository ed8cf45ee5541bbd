"""`from krasis.kv_cache import PagedKVCache, SequenceKVState` (python/krasis/kv_cache.py) -> the MI355X-native package's classes."""
from krasis_amd.kv_cache import PAGE_SIZE, TRTLLM_BLOCK_CONSTRAINT, PagedKVCache, SequenceKVState, unpage_into_store  # noqa: F401
