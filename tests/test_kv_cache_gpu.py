"""A paged prompt-pass cache handed to the decode store (krasis_amd.kv_cache.unpage_into_store, the reference's CpuDecoder._copy_kv_cache,
decode_setup.py:653-711): decoding on the unpaged cache equals decoding on the same rows given flat, bit for bit -- FP16 pages, and E4M3 pages both
widened to FP16 (the reference's hand-off) and kept as E4M3 bytes for a store that runs an FP8 cache."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.test_decode_gpu import build

pytestmark = pytest.mark.gpu
F = np.float32


class Cfg:
    attention_type = "gqa"; is_mla = False; is_gqa = True
    def __init__(self, nkv, hd): self.num_key_value_heads, self.gqa_head_dim = nkv, hd


@pytest.mark.parametrize("pages_fp8,store_fp8", [(False, False), (True, False), (True, True)])
def test_decode_on_an_unpaged_cache_equals_decode_on_the_flat_rows(pages_fp8, store_fp8):
    from krasis_amd.kv_cache import PagedKVCache, SequenceKVState, unpage_into_store
    seq = 21
    def fresh():
        st, eng, orc, keep, d = build(seed=8, kinds=["gqa", "la", "gqa"], kv_max=64)
        if store_fp8:
            st.set_kv_dtype(True)
        return st, d, keep
    st_a, d, keep_a = fresh(); st_b, _, keep_b = fresh()
    gqa_layers = [i for i, k in enumerate(d["kinds"]) if k == "gqa"]
    dt = torch.float8_e4m3fn if pages_fp8 else torch.float16
    cache = PagedKVCache(Cfg(d["nkv"], d["hd"]), len(gqa_layers), "cuda", max_pages=32, kv_dtype=dt)
    other = SequenceKVState(cache, 1); other.ensure_capacity(30); other.advance(30)         # our pages start in the middle of the pool
    ss = SequenceKVState(cache); ss.ensure_capacity(seq); ss.advance(seq)
    g = torch.Generator(device="cuda").manual_seed(5)
    rows = {}
    for off, li in enumerate(gqa_layers):
        for which, paged in (("k", cache.k_cache), ("v", cache.v_cache)):
            r = (torch.randn((seq, d["nkv"], d["hd"]), device="cuda", generator=g) * 0.5).to(dt)
            for t in range(seq):
                paged[off, ss.pages[t // 16], t % 16] = r[t]
            rows[(li, which)] = r.reshape(seq, -1)
    conv = d["state"]["conv"]; recur = d["state"]["recur"]
    # A: through the paged hand-off
    unpage_into_store(st_a, ss, [gqa_layers.index(i) if i in gqa_layers else None for i in range(len(d["kinds"]))], d["kv_max"], conv, recur, keep_alive=keep_a)
    # B: the same rows written flat by hand in the store's element type
    ks, vs = [], []
    for li in range(len(d["kinds"])):
        if li not in gqa_layers:
            ks.append(0); vs.append(0); continue
        pair = []
        for which in ("k", "v"):
            r = rows[(li, which)]
            if store_fp8:
                host = r.view(torch.uint8).cpu().numpy()
            else:
                host = r.to(torch.float16).view(torch.int16).cpu().numpy().view(np.uint16)
            full = np.zeros((d["kv_max"], host.shape[1]), host.dtype); full[:seq] = host; pair.append(full)
        keep_b.append(pair); ks.append(pair[0].ctypes.data); vs.append(pair[1].ctypes.data)
    st_b.set_decode_state(seq, d["kv_max"], ks, vs, [(c.ctypes.data if c is not None else 0) for c in conv], [(r.ctypes.data if r is not None else 0) for r in recur])
    la = np.empty(d["V"], F); lb = np.empty(d["V"], F)
    tok = 3
    for pos in range(seq, seq + 4):
        st_a.decode_step(tok, pos, la.ctypes.data); st_b.decode_step(tok, pos, lb.ctypes.data)
        assert np.isfinite(la).all() and np.array_equal(la.view(np.uint32), lb.view(np.uint32)), pos
        tok = int(np.argmax(la))
