"""GPU parity of the full decode step (hybrid linear-attention + gated GQA + MoE with shared expert) against the oracle
driver: logits, sampled token and every piece of recurrent / KV state compared BIT FOR BIT."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.oracle_decode import OracleDecode
from tests.util import make_experts, upload

pytestmark = pytest.mark.gpu
F = np.float32


def _ptr(a):
    return a.ctypes.data


def build(seed=0, norm_bias_one=True, scoring=1, rsf=1.0, with_dense=False, wbits=4, kv_max=32, hd=64, nh=4, dims=None, kinds=None, la_heads=(2, 4), peaked=False, gguf=False,
          la_dkdv=(128, 128)):
    """peaked=True: a model whose next-token distribution is PEAKED without training -- large embeddings, lm_head tied to them (x 0.05): the logit of the
    current token stands ~8 above the rest, the layers (attention, router, experts) perturb it by an amount comparable to the noise floor, so on a token
    stream with repeats the perplexity is O(10) and a routing flip or a tolerance-mode rounding difference shows up in it (tests/test_tolerance_peaked_gpu.py)"""
    from krasis_amd import CpuDecodeStore, KrasisEngine, ModelConfig
    rng = np.random.default_rng(seed)
    H, V, E, k, I, SI = dims or (256, 512, 16, 4, 128, 128)
    (nk, nv), (dk, dv) = la_heads, la_dkdv
    nkv, d2 = 2, 8
    kinds = kinds or (["la", "gqa", "la"] + (["gqa"] if with_dense else []))
    nL = len(kinds)
    emb = ((rng.random((V, H)) - 0.5) * (3.0 if peaked else 0.2)).astype(F)
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, nL, 0, rsf))
    eng.set_routing_config("softmax" if scoring == 1 else "sigmoid", True, k, E, H)
    st = CpuDecodeStore(128, True, norm_bias_one); st.set_moe_store(eng)
    orc = OracleDecode(H, 1e-6, norm_bias_one, k, scoring, True, rsf, emb, V)
    keep = [emb]

    def W(rows, cols, scale=0.05):
        w = (rng.standard_normal((rows, cols)) * scale).astype(F); keep.append(w)
        return st.store_weight_f32(_ptr(w), rows, cols, wbits), orc.store_weight_f32(w, wbits)

    def N(n):
        w = (rng.random(n) * 0.2 + (0.0 if norm_bias_one else 0.9)).astype(F); keep.append(w)
        return st.store_norm_weight(_ptr(w), n), orc.store_norm(w)

    fin = N(H)
    if peaked:
        lmw = (emb * F(0.05)).astype(F); keep.append(lmw)
        lm = (st.store_weight_f32(_ptr(lmw), V, H, wbits), orc.store_weight_f32(lmw, wbits))
    else:
        lm = W(V, H)
    st.configure_decode(H, nL, 1e-6, fin[0], lm[0], V, k, scoring, True, rsf, _ptr(emb))
    orc.final_norm, orc.lm_head = fin[1], lm[1]
    cos = np.cos(np.arange(kv_max)[:, None] * (1.0 / 10000.0 ** (2 * np.arange(d2) / (2 * d2)))[None, :]).astype(F)
    sin = np.sin(np.arange(kv_max)[:, None] * (1.0 / 10000.0 ** (2 * np.arange(d2) / (2 * d2)))[None, :]).astype(F)
    keep += [cos, sin]; orc.rope = (cos, sin)
    state = dict(kv_k=[None] * nL, kv_v=[None] * nL, conv=[None] * nL, recur=[None] * nL)
    for li, kind in enumerate(kinds):
        n_in, n_post = N(H), N(H)
        L = dict(in_norm=n_in[1], post_norm=n_post[1], attn=kind)
        if kind == "la":
            hr = nv // nk; group_dim = 2 * dk + 2 * dv * hr; conv_dim = 2 * nk * dk + nv * dv
            qkvz = W(nk * group_dim, H, 0.08); ba = W(nk * 2 * hr, H, 0.3); out = W(H, nv * dv)
            conv_w = (rng.standard_normal(conv_dim * 4) * 0.4).astype(F); a_log = (rng.random(nv) * 1.5 - 1.0).astype(F)
            dt_bias = (rng.random(nv) - 0.5).astype(F); norm_w = (rng.random(nv * dv) + 0.5).astype(F)
            keep += [conv_w, a_log, dt_bias, norm_w]
            scale = F(1.0 / np.sqrt(dk))
            st.add_decode_la_layer(n_in[0], n_post[0], qkvz[0], ba[0], out[0], _ptr(conv_w), _ptr(a_log), _ptr(dt_bias), _ptr(norm_w),
                                   nk, nv, dk, dv, nv // nk, 4, float(scale))
            cs = ((rng.random(conv_dim * 4) - 0.5) * 0.2).astype(F); rs = ((rng.random(nv * dk * dv) - 0.5) * 0.02).astype(F)
            state["conv"][li] = cs; state["recur"][li] = rs
            L.update(qkvz=qkvz[1], ba=ba[1], out=out[1], conv_w=conv_w, a_log=a_log, dt_bias=dt_bias, norm_w=norm_w, nk=nk, nv=nv, dk=dk, dv=dv,
                     scale=scale, conv_state=cs.copy(), recur_state=rs.copy())
        else:
            q = W(nh * hd * 2, H, 0.08); kk = W(nkv * hd, H, 0.08); v = W(nkv * hd, H, 0.08); o = W(H, nh * hd)
            qn = (rng.random(hd) + 0.5).astype(F); kn = (rng.random(nkv * hd) + 0.5).astype(F); keep += [qn, kn]   # shared vs per-head norm weights
            sm = F(1.0 / np.sqrt(hd))
            st.add_decode_gqa_layer(n_in[0], n_post[0], q[0], kk[0], v[0], o[0], _ptr(qn), qn.size, _ptr(kn), kn.size, True, nh, nkv, hd, float(sm))
            kc = O.f32_to_f16_bits((rng.standard_normal((kv_max, nkv * hd)) * 0.5).astype(F)); vc = O.f32_to_f16_bits((rng.standard_normal((kv_max, nkv * hd)) * 0.5).astype(F))
            state["kv_k"][li] = kc; state["kv_v"][li] = vc
            L.update(q=q[1], k=kk[1], v=v[1], o=o[1], q_norm=qn, k_norm=kn, gated=True, nh=nh, nkv=nkv, hd=hd, sm_scale=sm, kv_k=kc.copy(), kv_v=vc.copy())
        if with_dense and li == nL - 1:
            DI = 384                                    # dense MLP with a padded intermediate (cols of down = 384)
            gw = W(DI - 40, H); uw = W(DI - 40, H)
            wd = (rng.standard_normal((H, DI)) * 0.05).astype(F); wd[:, DI - 40:] = 0; keep.append(wd)
            dw = (st.store_weight_f32(_ptr(wd), H, DI, wbits), orc.store_weight_f32(wd, wbits))
            st.set_decode_layer_dense(li, gw[0], uw[0], dw[0]); L.update(mlp="dense", gate_w=gw[1], up_w=uw[1], down_w=dw[1])
        else:
            if gguf:      # routed experts as native GGUF blocks (Q4_K gate / up; down Q4_K when I % 256 == 0, else Q8_0 -- the V2-Lite situation): prompt pass only
                from tests.test_gguf_gpu import make as make_gguf
                gu_t, dn_t = gguf if isinstance(gguf, tuple) else (O.Q4_K, O.Q4_K if I % 256 == 0 else O.Q8_0)
                experts = [make_gguf(rng, H, I, gu_t, dn_t) for _ in range(E)]
                for ei, ex in enumerate(experts):
                    eng.load_gguf_expert(li, ei, ex.gate, ex.up, ex.down, gu_t, dn_t, I)
            else:
                experts = make_experts(rng, E, H, I, wbits); upload(eng, li, experts)
            gate = ((rng.random((E, H)) - 0.5) * 0.1).astype(F); keep.append(gate)
            esc = ((rng.random(E) - 0.5) * 0.01).astype(F) if scoring == 0 else None
            eng.set_route_weight_f32(li, gate, None, esc)
            sgu = W(2 * SI, H); sd = W(H, SI); sg = W(1, H, 0.3)
            st.set_decode_layer_moe(li, li, li, sgu[0], sd[0], sg[0])
            L.update(mlp="moe", gate=gate, esc=esc, experts=experts, sgu=sgu[1], sd=sd[1], sg=sg[1])
        orc.layers.append(L)
    st.set_decode_rope(_ptr(cos), _ptr(sin), d2, kv_max)
    st.finalize_decode()
    z = lambda xs: [(_ptr(x) if x is not None else 0) for x in xs]
    reset = lambda: st.set_decode_state(5, kv_max, z(state["kv_k"]), z(state["kv_v"]), z(state["conv"]), z(state["recur"]))
    reset()
    return st, eng, orc, keep, dict(H=H, V=V, kinds=kinds, kv_max=kv_max, nkv=nkv, hd=hd, conv_dim=2 * nk * dk + nv * dv, nv=nv, dk=dk, dv=dv, reset=reset,
                                    state=state)


@pytest.mark.parametrize("cfg", [dict(), dict(norm_bias_one=False, scoring=0, rsf=2.5), dict(with_dense=True), dict(wbits=8, with_dense=True),
                                 dict(dims=(256, 512, 16, 4, 128, 384), seed=3),       # shared expert three times as wide as the routed ones (V2-Lite has 2 x)
                                 dict(la_dkdv=(64, 64), seed=4), dict(la_dkdv=(128, 64), la_heads=(2, 2), seed=5),    # the 64-wide key / value head forms of the in-projection epilogue and the per-value-head recurrence
                                 dict(dims=(256, 512, 512, 10, 128, 128), seed=6), dict(dims=(256, 512, 256, 8, 128, 128), seed=7, kinds=["la", "gqa"])])   # E = 512 / 256: the fused router's selection with the exponentials on four waves (8 / 4 values per lane)
@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("round5_forms", [1, 0])
def test_decode_step_bit_exact(cfg, graph, round5_forms):
    """round5_forms = 1 (default): linear-attention layers run the conv as the in-projection's epilogue and the recurrence one workgroup per VALUE head; the experts'
    down projection and the routing-order combine are one launch.  0: in-projection, then conv + recurrence in one launch per KEY head; down projection per slot, the
    combine inside the next norm launch.  Both forms bit-equal to the oracle (logits, conv state, recurrent state)."""
    st, eng, orc, keep, d = build(**cfg)
    st.set_use_graph(graph)
    st.set_option("la_heads", round5_forms)
    st.set_option("w2_combine", round5_forms)
    tok = 7
    for step, pos in enumerate([5, 6, 7]):
        logits = np.empty(d["V"], F)
        st.decode_step(tok, pos, logits.ctypes.data)
        ref = orc.step(tok, pos)
        assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (step, float(np.max(np.abs(logits - ref))))
        nxt = st.last_token()
        assert nxt == O.sample_greedy(ref)
        tok = nxt
    for li, kind in enumerate(d["kinds"]):
        L = orc.layers[li]
        if kind == "la":
            cs = np.empty(d["conv_dim"] * 4, F); rs = np.empty(d["nv"] * d["dk"] * d["dv"], F)
            st.get_decode_state(li, None, None, cs, rs)
            assert np.array_equal(cs.view(np.uint32), L["conv_state"].view(np.uint32))
            assert np.array_equal(rs.view(np.uint32), L["recur_state"].view(np.uint32))
        else:
            kc = np.empty((d["kv_max"], d["nkv"] * d["hd"]), np.uint16); vc = np.empty_like(kc)
            st.get_decode_state(li, kc, vc, None, None)
            assert np.array_equal(kc, L["kv_k"]) and np.array_equal(vc, L["kv_v"])


@pytest.mark.parametrize("dims,types", [((256, 512, 16, 4, 128, 128), True), ((512, 512, 8, 2, 256, 128), True), ((256, 512, 16, 4, 128, 128), (O.Q8_0, O.Q4_0)),
                                        ((512, 512, 8, 2, 256, 128), (O.Q4_0, O.Q8_0))])
@pytest.mark.parametrize("graph,fast", [(True, False), (False, False), (True, True)])
def test_decode_step_on_native_gguf_experts_bit_exact(dims, types, graph, fast):
    """kr_decode_step on a model whose ROUTED experts are native GGUF blocks (Q4_K gate / up with Q8_0 down at I = 128 or Q4_K down at I = 256 --
    the V2-Lite and QCN situations -- and Q8_0 / Q4_0 mixes, so every integer block kernel runs in both modes).  The reference drives such layers per layer from Python through moe_forward_gguf (moe.rs:990-1110,
    tests/test_gguf_native.py:47-57); the oracle driver is that control flow.  Inside the captured step: router, k routed experts through the block
    kernels on bf16(hidden) (decode.rs:3307), the decode store's shared expert on the f32 hidden, combine in routing order in the next norm launch.
    Logits, greedy token and every state tensor BIT FOR BIT, graph and eager; with KR_DECODE_FAST set the routed slots of the mode's gate|up and down launches
    walk the GGUF blocks (the block kernels' products, a row's blocks split over two waves, select and combine folded in): logits within the mode's 2e-3,
    same greedy token."""
    st, eng, orc, keep, d = build(dims=dims, gguf=types, seed=21)
    st.set_use_graph(graph)
    if fast:
        st.set_attention_mode(False, decode_fast=True)
    tok = 7
    for step, pos in enumerate([5, 6, 7]):
        logits = np.empty(d["V"], F)
        st.decode_step(tok, pos, logits.ctypes.data)
        ref = orc.step(tok, pos)
        if fast:
            assert float(np.abs(logits - ref).max() / np.abs(ref).max()) <= 2e-3
            assert int(np.argmax(logits)) == O.sample_greedy(ref)
        else:
            assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (step, float(np.max(np.abs(logits - ref))))
            assert st.last_token() == O.sample_greedy(ref)
        tok = O.sample_greedy(ref)
    if not fast:
        for li, kind in enumerate(d["kinds"]):
            L = orc.layers[li]
            if kind == "la":
                cs = np.empty(d["conv_dim"] * 4, F); rs = np.empty(d["nv"] * d["dk"] * d["dv"], F)
                st.get_decode_state(li, None, None, cs, rs)
                assert np.array_equal(cs.view(np.uint32), L["conv_state"].view(np.uint32))
                assert np.array_equal(rs.view(np.uint32), L["recur_state"].view(np.uint32))


@pytest.mark.parametrize("graph,fast", [(True, False), (True, True)])
def test_decode_step_native_gguf_at_qcn_widths(graph, fast):
    """VERDICT r4 next #3b: bench.py's `qcn-q4k-gguf` shape as a parity test -- hidden 2048, expert intermediate 512, Q4_K gate / up / down (8 super-blocks per
    gate row, 2 per down row), top-10 with renormalisation, shared expert 512 with its sigmoid gate; linear-attention + gated-GQA layers; expert count reduced
    to 24.  Exact: bit for bit against the oracle's moe_forward_gguf-driven decode; KR_DECODE_FAST: within the mode's 2e-3, same greedy token."""
    st, eng, orc, keep, d = build(dims=(2048, 512, 24, 10, 512, 512), gguf=True, seed=33, kinds=["la", "gqa"], hd=64)
    st.set_use_graph(graph)
    if fast:
        st.set_attention_mode(False, decode_fast=True)
    tok = 7
    for step, pos in enumerate([5, 6, 7]):
        logits = np.empty(d["V"], F)
        st.decode_step(tok, pos, logits.ctypes.data)
        ref = orc.step(tok, pos)
        if fast:
            err = float(np.abs(logits - ref).max() / np.abs(ref).max())
            assert 0.0 < err <= 2e-3, (step, err)
            assert int(np.argmax(logits)) == O.sample_greedy(ref)
        else:
            assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (step, float(np.max(np.abs(logits - ref))))
            assert st.last_token() == O.sample_greedy(ref)
        tok = O.sample_greedy(ref)


@pytest.mark.parametrize("hd", [64, 128, 256])
def test_decode_step_long_positions_bit_exact(hd):
    """positions past one / two 128-row stages of the attention kernel's K / V staging (decode.rs:4194 order kept across stages)"""
    st, eng, orc, keep, d = build(seed=5, kv_max=300, hd=hd)
    tok = 11
    for pos in [126, 127, 128, 129, 255, 256, 257, 299]:
        logits = np.empty(d["V"], F)
        st.decode_step(tok, pos, logits.ctypes.data)
        ref = orc.step(tok, pos)
        assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (pos, float(np.max(np.abs(logits - ref))))
        tok = O.sample_greedy(ref)


@pytest.mark.parametrize("hd,graph", [(64, True), (256, False), (128, True)])
def test_decode_step_split_attention_bit_exact(hd, graph):
    """kv_max_seq > 1024: decode attention runs as a scores launch over (heads x 256-position blocks) + a softmax / p.v launch; same bits"""
    st, eng, orc, keep, d = build(seed=8, kv_max=1300, hd=hd)
    st.set_use_graph(graph)
    tok = 5
    for pos in [3, 255, 256, 257, 700, 1023, 1024, 1299]:
        logits = np.empty(d["V"], F)
        st.decode_step(tok, pos, logits.ctypes.data)
        ref = orc.step(tok, pos)
        assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (pos, float(np.max(np.abs(logits - ref))))
        tok = O.sample_greedy(ref)


@pytest.mark.parametrize("hd,fp8,stream", [(64, False, True), (256, False, True), (128, True, True), (128, False, True), (256, True, True),
                                           (256, True, False), (128, True, False), (64, True, False), (256, "codes", False)])
def test_decode_step_streamed_attention_bit_exact(hd, fp8, stream, monkeypatch):
    """caches too long for an LDS-resident score row (> ~23 k positions) stream it from HBM in tiles; set_option("gqa_stream") forces that form
    on a cache the oracle can follow, positions across the 128-row stage and 4096-value tile boundaries.  stream=False: the same long
    random E4M3 cache through the LDS-resident form (hardware FP8 widening in the p.v chain at head_dim 128 / 256)"""
    st, eng, orc, keep, d = build(seed=12, kv_max=4400, hd=hd)
    if stream:
        st.set_option("gqa_stream", 1)
    if fp8:
        st.set_kv_dtype(True); O.set_kv_fp8(True)
    try:
        if fp8:
            rng = np.random.default_rng(5)
            def cache():
                if fp8 == "codes":      # V: every finite E4M3 code (subnormals, +-448, both zeros) through the hardware widening; K stays small so the softmax spreads
                    b = rng.integers(0, 256, (d["kv_max"], d["nkv"] * d["hd"])).astype(np.uint8)
                    b[(b & 0x7F) == 0x7F] = 0x3C
                    return b
                return O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["nkv"] * d["hd"])) * 0.5).astype(F))
            kv = {li: (O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["nkv"] * d["hd"])) * 0.5).astype(F)), cache())
                  for li, kind in enumerate(d["kinds"]) if kind == "gqa"}
            for li in kv:
                orc.layers[li]["kv_k"] = kv[li][0].astype(np.uint16); orc.layers[li]["kv_v"] = kv[li][1].astype(np.uint16)
            n = len(d["kinds"]); ptr = lambda a: a.ctypes.data
            st.set_decode_state(5, d["kv_max"], [ptr(kv[i][0]) if i in kv else 0 for i in range(n)], [ptr(kv[i][1]) if i in kv else 0 for i in range(n)],
                                [ptr(x) if x is not None else 0 for x in d["state"]["conv"]], [ptr(x) if x is not None else 0 for x in d["state"]["recur"]])
        tok = 5
        for pos in [3, 127, 128, 1500, 4095, 4096, 4399]:
            logits = np.empty(d["V"], F)
            st.decode_step(tok, pos, logits.ctypes.data)
            ref = orc.step(tok, pos)
            assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (pos, float(np.max(np.abs(logits - ref))))
            tok = O.sample_greedy(ref)
    finally:
        O.set_kv_fp8(False)


@pytest.mark.parametrize("dims,hd,nh", [((2048, 512, 72, 10, 512, 512), 256, 16),      # Qwen3-Coder-Next widths (hidden 2048, I 512, top-10, head_dim 256, 8 q / kv head)
                                         ((4096, 384, 72, 8, 1536, 1536), 128, 32)])     # Qwen3-235B widths (hidden 4096, I 1536, top-8, head_dim 128)
def test_decode_step_production_widths_bit_exact(dims, hd, nh):
    """one linear-attention + one GQA layer at the real models' widths (expert count reduced): the K = 2048 / 4096 paths of the
    cooperative matvec, the 32-chunk gate rows and the NV > 1 selection of the fused router, the 256-wide heads"""
    st, eng, orc, keep, d = build(seed=21, dims=dims, hd=hd, nh=nh, kv_max=48, kinds=["la", "gqa"])
    tok = 3
    for pos in [5, 6, 40]:
        logits = np.empty(d["V"], F)
        st.decode_step(tok, pos, logits.ctypes.data)
        ref = orc.step(tok, pos)
        assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (pos, float(np.max(np.abs(logits - ref))))
        tok = O.sample_greedy(ref)


def test_generate_batch_greedy_matches_oracle():
    st, eng, orc, keep, d = build(seed=3)
    toks = st.generate_batch(11, 5, 4)
    ref, tok = [], 11
    for i in range(4):
        tok = O.sample_greedy(orc.step(tok, 5 + i)); ref.append(tok)
    assert toks == ref
