"""krasis_amd.kv_cache: the reference's paged-KV bookkeeping classes (python/krasis/kv_cache.py:38-272) on CPU tensors -- pool sizing, page allocation
order, exhaustion message, FlashInfer / TRTLLM index arrays, the combined-MLA store -- and the unpaging gather of decode_setup.py:653-711.  When the
reference package is importable (this container: /root/reference/python) the two implementations are driven side by side and must agree call by call."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from krasis_amd.kv_cache import PagedKVCache, SequenceKVState, unpage_into_store


class Cfg:
    def __init__(self, mla):
        self.attention_type = "mla" if mla else "gqa"
        self.is_mla, self.is_gqa = mla, not mla
        self.kv_lora_rank, self.qk_rope_head_dim = 512, 64
        self.num_key_value_heads, self.gqa_head_dim = 2, 256


def test_pool_sizing_and_allocation():
    c = PagedKVCache(Cfg(False), num_layers=12, device="cpu", max_mb=8)
    per_page = 16 * (2 * 256 * 2) * 1 * 12                      # page_size x K+V dims x 1 byte (E4M3) x layers
    assert c.max_pages == max(64, 8 * 1024 * 1024 // per_page) and c.max_context_tokens == c.max_pages * 16
    assert c.k_cache.shape == (12, c.max_pages, 16, 2, 256) and c.k_cache.dtype == torch.float8_e4m3fn
    s = SequenceKVState(c)
    s.ensure_capacity(1); assert s.pages == [0]
    s.advance(16); s.ensure_capacity(1); assert s.pages == [0, 1]
    s.advance(5)
    assert s.last_page_len() == 5 and s.kv_indptr("cpu").tolist() == [0, 2] and s.kv_len_arr("cpu").tolist() == [21]
    bt = s.block_tables("cpu")
    assert bt.shape == (1, 8) and bt[0, :2].tolist() == [0, 1] and (bt[0, 2:] == -1).all()
    with pytest.raises(RuntimeError, match=r"KV cache exhausted: need \d+ pages, have \d+"):
        s.ensure_capacity(c.max_context_tokens)
    free_before = c.free_page_count
    s.free()
    assert c.free_page_count == free_before + 2 and s.seq_len == 0 and s.pages == []


def test_mla_layouts_and_combined_store():
    split = PagedKVCache(Cfg(True), 3, "cpu", max_pages=64, kv_dtype=torch.float16)
    ckv, kpe = split.get_layer_caches(1)
    assert ckv.shape == (64, 16, 512) and kpe.shape == (64, 16, 64)
    comb = PagedKVCache(Cfg(True), 3, "cpu", max_pages=64, kv_dtype=torch.float16, combined=True)
    assert comb.get_combined_layer_cache(2).shape == (1, 64, 16, 576)
    s = SequenceKVState(comb); s.ensure_capacity(40); s.advance(40)
    rows = torch.randn(40, 576).to(torch.float16)
    s.store_kv_combined(2, rows, torch.arange(40))
    assert torch.equal(s.unpage(comb.kv_cache[2]), rows)


def test_unpage_matches_the_reference_copy_loop_and_feeds_a_store():
    c = PagedKVCache(Cfg(False), 2, "cpu", max_pages=64)
    other = SequenceKVState(c, 1); other.ensure_capacity(20); other.advance(20)      # somebody else's pages come first: ours are not contiguous from 0
    s = SequenceKVState(c); s.ensure_capacity(37); s.advance(37)
    g = torch.Generator().manual_seed(3)
    c.k_cache.copy_((torch.randn(c.k_cache.shape, generator=g) * 0.5).to(torch.float8_e4m3fn)); c.v_cache.copy_((torch.randn(c.v_cache.shape, generator=g) * 0.5).to(torch.float8_e4m3fn))
    # the reference's loop (decode_setup.py:692-708), restated
    ref = torch.zeros(37, 512, dtype=torch.float16); t = 0
    for p in s.pages:
        n = min(16, 37 - t)
        if n <= 0:
            break
        ref[t:t + n] = c.k_cache[1][p, :n].to(torch.float16).reshape(n, -1); t += n
    assert torch.equal(s.unpage(c.k_cache[1]), ref)
    assert s.unpage(c.k_cache[1], None).dtype == torch.float8_e4m3fn

    class Store:                       # records what set_decode_state is given
        _kv_fp8 = False
        def set_decode_state(self, seq_len, kv_max, ks, vs, conv, rec): self.got = (seq_len, kv_max, ks, vs, conv, rec)
    st = Store(); keep = []
    held = unpage_into_store(st, s, [None, 0, 1], 64, keep_alive=keep)
    seq_len, kv_max, ks, vs, conv, rec = st.got
    assert (seq_len, kv_max) == (37, 64) and ks[0] == 0 and vs[0] == 0 and len(held) == 2 and keep == held
    k1 = held[1][0]
    assert k1.shape == (64, 512) and k1.dtype == np.uint16 and np.array_equal(k1[:37], ref.view(torch.int16).numpy().view(np.uint16)) and not k1[37:].any()
    st._kv_fp8 = True                  # a store with E4M3 caches takes the page bytes as they are
    held8 = unpage_into_store(st, s, [0], 64)
    assert held8[0][1].dtype == np.uint8 and np.array_equal(held8[0][1][:37], s.unpage(c.v_cache[0], None).view(torch.uint8).numpy())


@pytest.mark.skipif(not os.path.isdir("/root/reference/python/krasis"), reason="reference checkout not present")
def test_side_by_side_with_the_reference_classes():
    sys.modules.setdefault("krasis", types.ModuleType("krasis")).__path__ = ["/root/reference/python/krasis"]
    cfgmod = types.ModuleType("krasis.config"); cfgmod.ModelConfig = object; sys.modules["krasis.config"] = cfgmod
    import importlib.util
    spec = importlib.util.spec_from_file_location("krasis.kv_cache", "/root/reference/python/krasis/kv_cache.py")
    ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
    for mla, combined in ((False, False), (True, False), (True, True)):
        a = ref.PagedKVCache(Cfg(mla), 2, torch.device("cpu"), max_pages=48, kv_dtype=torch.float16, combined=combined)
        b = PagedKVCache(Cfg(mla), 2, "cpu", max_pages=48, kv_dtype=torch.float16, combined=combined)
        sa, sb = ref.SequenceKVState(a), SequenceKVState(b)
        for step in (1, 15, 1, 40, 100):
            sa.ensure_capacity(step); sb.ensure_capacity(step); sa.advance(step); sb.advance(step)
            assert sa.pages == sb.pages and sa.seq_len == sb.seq_len and sa.last_page_len() == sb.last_page_len()
            assert torch.equal(sa.kv_indices("cpu"), sb.kv_indices("cpu")) and torch.equal(sa.block_tables("cpu"), sb.block_tables("cpu"))
            assert a.free_page_count == b.free_page_count and a.max_context_tokens == b.max_context_tokens
        if combined:
            rows = torch.randn(sa.seq_len, 576).to(torch.float16); pos = torch.arange(sa.seq_len)
            sa.store_kv_combined(1, rows, pos); sb.store_kv_combined(1, rows, pos)
            assert torch.equal(a.kv_cache, b.kv_cache)
        with pytest.raises(RuntimeError) as ea:
            sa.ensure_capacity(10_000)
        with pytest.raises(RuntimeError) as eb:
            sb.ensure_capacity(10_000)
        assert str(ea.value) == str(eb.value)
        sa.free(); sb.free()
        assert a.free_page_count == b.free_page_count == 48
