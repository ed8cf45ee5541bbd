"""Paged KV cache bookkeeping with the reference's class surface (python/krasis/kv_cache.py:38-272: `PagedKVCache`, `SequenceKVState`) and the hand-off of
a paged cache to the decode store (`CpuDecoder._copy_kv_cache`, decode_setup.py:653-711).

On MI355X a request's cache is ONE contiguous `[max_seq, kv_heads * head_dim]` allocation per layer inside the decode store (288 GB of HBM: nothing has to
be paged to make room for the experts), and the prompt pass of this library writes that flat cache directly.  What is here is the part of the reference a
CALLER sees: the page pool, the per-sequence page list with its index arrays, the store method of the combined MLA layout -- same constructor arguments,
same tensor shapes / dtypes, same error texts -- and `unpage_into_store`, which gathers a sequence's pages into the flat caches the decode store takes
(FP8-E4M3 pages are widened to FP16 exactly, or handed over as E4M3 bytes when the store runs `kr_decode_set_kv_dtype(FP8)`).  The attention kernels do
NOT read through a page table: a caller that keeps its own paged cache unpages once, at the prompt -> decode hand-off, as the reference does."""
import math
from typing import List, Optional, Sequence

import torch

PAGE_SIZE = 16                   # tokens per page (kv_cache.py:25)
TRTLLM_BLOCK_CONSTRAINT = 128    # block_num % (128 / page_size) == 0 (kv_cache.py:28)


class PagedKVCache:
    """A fixed pool of pages for a set of layers on one device (kv_cache.py:31-181).  `cfg` needs: is_mla / is_gqa, attention_type, and
    kv_lora_rank + qk_rope_head_dim (MLA) or num_key_value_heads + gqa_head_dim (GQA)."""

    def __init__(self, cfg, num_layers: int, device, max_pages: Optional[int] = None, kv_dtype: torch.dtype = torch.float8_e4m3fn,
                 page_size: int = PAGE_SIZE, combined: bool = False, max_mb: Optional[int] = None):
        self.cfg, self.num_layers, self.device, self.page_size, self.kv_dtype, self.combined = cfg, num_layers, device, page_size, kv_dtype, combined
        self.attention_type = cfg.attention_type
        if cfg.is_mla:
            self.ckv_dim, self.kpe_dim = cfg.kv_lora_rank, cfg.qk_rope_head_dim
            self.kv_cache_dim = self.ckv_dim + self.kpe_dim
            self.num_kv_heads = self.gqa_head_dim = None
        else:
            self.ckv_dim = self.kpe_dim = None
            self.num_kv_heads, self.gqa_head_dim = cfg.num_key_value_heads, cfg.gqa_head_dim
            self.kv_cache_dim = cfg.num_key_value_heads * cfg.gqa_head_dim * 2
        if max_pages is None:
            budget = (2000 if max_mb is None else max_mb) * 1024 * 1024
            max_pages = max(64, budget // self._bytes_per_page())
        self.max_pages = max_pages
        self.k_cache = self.v_cache = self.ckv_cache = self.kpe_cache = self.kv_cache = None
        z = lambda *shape: torch.zeros(*shape, dtype=kv_dtype, device=device)
        if cfg.is_gqa:
            self.k_cache = z(num_layers, max_pages, page_size, self.num_kv_heads, self.gqa_head_dim)
            self.v_cache = z(num_layers, max_pages, page_size, self.num_kv_heads, self.gqa_head_dim)
        elif combined:
            self.kv_cache = z(num_layers, max_pages, page_size, self.kv_cache_dim)
        else:
            self.ckv_cache = z(num_layers, max_pages, page_size, self.ckv_dim)
            self.kpe_cache = z(num_layers, max_pages, page_size, self.kpe_dim)
        self._free_pages: List[int] = list(range(max_pages))
        self._free_pages.reverse()          # pop from the end: pages are handed out in ascending order

    def _bytes_per_page(self) -> int:
        return self.page_size * self.kv_cache_dim * (1 if self.kv_dtype == torch.float8_e4m3fn else 2) * self.num_layers

    @property
    def max_context_tokens(self) -> int:
        return self.max_pages * self.page_size

    @property
    def free_page_count(self) -> int:
        return len(self._free_pages)

    def alloc_pages(self, n: int) -> List[int]:
        if n > len(self._free_pages):
            raise RuntimeError(f"KV cache exhausted: need {n} pages, have {len(self._free_pages)}")
        return [self._free_pages.pop() for _ in range(n)]

    def free_pages(self, pages: Sequence[int]) -> None:
        self._free_pages.extend(pages)

    def get_layer_caches(self, layer_offset: int):
        assert self.attention_type == "mla" and not self.combined
        return self.ckv_cache[layer_offset], self.kpe_cache[layer_offset]

    def get_combined_layer_cache(self, layer_offset: int) -> torch.Tensor:
        assert self.attention_type == "mla" and self.combined
        return self.kv_cache[layer_offset].unsqueeze(0)

    def get_gqa_layer_caches(self, layer_offset: int):
        assert self.attention_type == "gqa"
        return self.k_cache[layer_offset], self.v_cache[layer_offset]


class SequenceKVState:
    """The pages of one request and its length (kv_cache.py:184-272)."""

    def __init__(self, cache: PagedKVCache, seq_id: int = 0):
        self.cache, self.seq_id = cache, seq_id
        self.pages: List[int] = []
        self.seq_len = 0

    def ensure_capacity(self, new_tokens: int) -> None:
        need = (self.seq_len + new_tokens + self.cache.page_size - 1) // self.cache.page_size
        if need > len(self.pages):
            self.pages.extend(self.cache.alloc_pages(need - len(self.pages)))

    def advance(self, num_tokens: int) -> None:
        self.seq_len += num_tokens

    def free(self) -> None:
        if self.pages:
            self.cache.free_pages(self.pages)
            self.pages = []
            self.seq_len = 0

    def kv_indices(self, device) -> torch.Tensor:
        return torch.tensor(self.pages, dtype=torch.int32, device=device) if self.pages else torch.zeros(0, dtype=torch.int32, device=device)

    def kv_indptr(self, device) -> torch.Tensor:
        return torch.tensor([0, len(self.pages)], dtype=torch.int32, device=device)

    def kv_len_arr(self, device) -> torch.Tensor:
        return torch.tensor([self.seq_len], dtype=torch.int32, device=device)

    def last_page_len(self) -> int:
        if self.seq_len == 0:
            return 0
        rem = self.seq_len % self.cache.page_size
        return rem if rem > 0 else self.cache.page_size

    def last_page_len_tensor(self, device) -> torch.Tensor:
        return torch.tensor([self.last_page_len()], dtype=torch.int32, device=device)

    def block_tables(self, device, pad_to_multiple: int = 8) -> torch.Tensor:
        n = len(self.pages)
        c = TRTLLM_BLOCK_CONSTRAINT // self.cache.page_size
        padded = math.ceil(n / c) * c if n > 0 else c
        table = torch.full((1, padded), -1, dtype=torch.int32, device=device)
        if n > 0:
            table[0, :n] = torch.tensor(self.pages, dtype=torch.int32, device=device)
        return table

    def store_kv_combined(self, layer_offset: int, kv_combined: torch.Tensor, positions: torch.Tensor) -> None:
        assert self.cache.combined, "store_kv_combined requires combined cache"
        ps = self.cache.page_size
        pages = torch.tensor(self.pages, dtype=torch.long, device=kv_combined.device)
        self.cache.kv_cache[layer_offset, pages[positions.long() // ps], positions.long() % ps] = kv_combined.to(self.cache.kv_dtype)

    # ---- not in the reference class: the gather its CpuDecoder._copy_kv_cache does page by page (decode_setup.py:653-711)
    def unpage(self, paged_layer: torch.Tensor, out_dtype: Optional[torch.dtype] = torch.float16) -> torch.Tensor:
        """rows 0 .. seq_len-1 of one layer's paged tensor `[pages, page_size, ...]` as a flat `[seq_len, prod(...)]` tensor; out_dtype None keeps the
        page dtype (E4M3 bytes for a store that runs an FP8 cache), else the exact widening the reference applies (`.to(torch.float16)`)."""
        if self.seq_len == 0:
            return paged_layer.new_zeros((0, int(paged_layer[0, 0].numel())), dtype=out_dtype or paged_layer.dtype)
        n_pages = (self.seq_len + self.cache.page_size - 1) // self.cache.page_size
        idx = torch.tensor(self.pages[:n_pages], dtype=torch.long, device=paged_layer.device)
        flat = paged_layer[idx].reshape(n_pages * self.cache.page_size, -1)[: self.seq_len]
        return flat if out_dtype is None else flat.to(out_dtype)


def unpage_into_store(store, seq_state: SequenceKVState, layer_map, kv_max_seq: int, conv_states=None, recur_states=None, keep_alive: Optional[list] = None):
    """Hand a paged prompt-pass cache to the decode store (the GPU-paged -> flat copy of decode_setup.py:653-711 followed by set_decode_state).
    layer_map: one entry per decode layer -- None (linear-attention layer: no KV), or the KV layer offset inside `seq_state.cache`.  The flat caches are
    `[kv_max_seq, dim]` in the store's KV element type (FP16, or E4M3 bytes when the store was switched with set_kv_dtype(True) and the pages are FP8).
    Returns the list of (k, v) host arrays (keep them alive as long as the store uses them -- `keep_alive` gets them appended)."""
    import numpy as np
    cache = seq_state.cache
    fp8_store = bool(getattr(store, "_kv_fp8", False))
    ks, vs, held = [], [], []
    for off in layer_map:
        if off is None:
            ks.append(0); vs.append(0); continue
        if cache.attention_type == "gqa":
            a, b = cache.get_gqa_layer_caches(off)
        elif cache.combined:
            both = cache.kv_cache[off]
            a, b = both[..., : cache.ckv_dim], both[..., cache.ckv_dim:]
        else:
            a, b = cache.get_layer_caches(off)
        pair = []
        for t in (a, b):
            keep_bytes = fp8_store and t.dtype == torch.float8_e4m3fn
            flat = seq_state.unpage(t, None if keep_bytes else torch.float16).cpu().contiguous()
            host = flat.view(torch.uint8).numpy() if keep_bytes else flat.view(torch.int16).numpy().view(np.uint16)
            full = np.zeros((kv_max_seq, host.shape[1]), host.dtype)
            full[: host.shape[0]] = host
            pair.append(full)
        held.append(tuple(pair)); ks.append(pair[0].ctypes.data); vs.append(pair[1].ctypes.data)
    n = len(layer_map)
    conv = [(c.ctypes.data if c is not None else 0) for c in (conv_states or [None] * n)]
    rec = [(r.ctypes.data if r is not None else 0) for r in (recur_states or [None] * n)]
    store.set_decode_state(seq_state.seq_len, kv_max_seq, ks, vs, conv, rec)
    if keep_alive is not None:
        keep_alive.extend(held)
    return held
