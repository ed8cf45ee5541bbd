"""GPU parity of the MLA decode arm (DeepSeek-V2-Lite shape family: dense layer 0, MoE layers with an un-gated shared expert, MLA with
the direct query projection; plus the V3-style LoRA query path) against the oracle driver -- logits and both FP16 caches BIT FOR BIT."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.oracle_decode import OracleDecode
from tests.util import make_experts, upload

pytestmark = pytest.mark.gpu
F = np.float32


def _ptr(a):
    return a.ctypes.data


def build(seed=0, lora=False, klr=512, nh=4, kv_max=24, dims=None, gguf=False):
    from krasis_amd import CpuDecodeStore, KrasisEngine, ModelConfig
    rng = np.random.default_rng(seed)
    H, V, E, k, I, SI = dims or (256, 384, 8, 3, 128, 256)   # shared expert = 2 x I like V2-Lite (n_shared_experts = 2)
    nd, rd, vhd, qlr = 128, 64, 128, 384
    nL = 2
    emb = ((rng.random((V, H)) - 0.5) * 0.2).astype(F)
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, nL, 2, 1.0))
    eng.set_routing_config("softmax", False, k, E, H)       # V2-Lite: softmax, no top-k renormalisation
    st = CpuDecodeStore(128, True, False); st.set_moe_store(eng)
    orc = OracleDecode(H, 1e-6, False, k, 1, False, 1.0, emb, V)
    keep = [emb]

    def W(rows, cols, scale=0.05):
        w = (rng.standard_normal((rows, cols)) * scale).astype(F); keep.append(w)
        return st.store_weight_f32(_ptr(w), rows, cols, 4), orc.store_weight_f32(w, 4)

    def N(n):
        w = (rng.random(n) * 0.2 + 0.9).astype(F); keep.append(w)
        return st.store_norm_weight(_ptr(w), n), orc.store_norm(w)

    fin = N(H); lm = W(V, H)
    st.configure_decode(H, nL, 1e-6, fin[0], lm[0], V, k, 1, False, 1.0, _ptr(emb))
    orc.final_norm, orc.lm_head = fin[1], lm[1]
    half = rd // 2
    ang = np.arange(kv_max)[:, None] * (1.0 / 10000.0 ** (2 * np.arange(half) / rd))[None, :]
    cos, sin = np.cos(ang).astype(F), np.sin(ang).astype(F); keep += [cos, sin]
    state_k, state_v = [None] * nL, [None] * nL
    for li in range(nL):
        n_in, n_post = N(H), N(H)
        kv_a = W(klr + rd, H, 0.08); o = W(H, nh * vhd)
        if lora:
            q_a = W(qlr, H, 0.08); q_b = W(nh * (nd + rd), qlr, 0.08); q = (None, None)
            qan = (rng.random(qlr) + 0.5).astype(F); keep.append(qan)
        else:
            q = W(nh * (nd + rd), H, 0.08); q_a = q_b = (None, None); qan = None
        w_kc = O.f32_to_bf16((rng.standard_normal((nh, nd, klr)) * 0.06).astype(F)); w_vc = O.f32_to_bf16((rng.standard_normal((nh, vhd, klr)) * 0.06).astype(F))
        kvn = (rng.random(klr) + 0.5).astype(F); keep += [w_kc, w_vc, kvn]
        sm = F(1.0 / np.sqrt(nd + rd))
        st.add_decode_mla_layer(n_in[0], n_post[0], kv_a[0], o[0], q[0], q_a[0], q_b[0], _ptr(w_kc), w_kc.size, _ptr(w_vc), w_vc.size, _ptr(kvn), klr,
                                _ptr(qan) if qan is not None else 0, qlr if qan is not None else 0, _ptr(cos), _ptr(sin), half, kv_max, nh, klr, nd, rd, vhd,
                                float(sm))
        ck = O.f32_to_f16_bits((rng.standard_normal((kv_max, klr)) * 0.5).astype(F)); kp = O.f32_to_f16_bits((rng.standard_normal((kv_max, rd)) * 0.5).astype(F))
        state_k[li], state_v[li] = ck, kp
        L = dict(in_norm=n_in[1], post_norm=n_post[1], attn="mla", kv_a=kv_a[1], o=o[1], q=q[1], q_a=q_a[1], q_b=q_b[1], q_a_norm=qan, kv_a_norm=kvn,
                 w_kc=O.bf16_to_f32(w_kc), w_vc=O.bf16_to_f32(w_vc), cos=cos, sin=sin, nh=nh, klr=klr, nd=nd, rd=rd, vhd=vhd, sm_scale=sm,
                 ckv=ck.copy(), kpe=kp.copy())
        if li == 0:                                          # first_k_dense_replace = 1
            DI = 384
            gw = W(DI, H); uw = W(DI, H); dw = W(H, DI)
            st.set_decode_layer_dense(li, gw[0], uw[0], dw[0]); L.update(mlp="dense", gate_w=gw[1], up_w=uw[1], down_w=dw[1])
        else:
            if gguf:      # routed experts as native GGUF blocks: Q4_K gate / up, down Q4_K when I % 256 == 0 else Q8_0 -- the int4cpu build of V2-Lite (I = 1408)
                from tests.test_gguf_gpu import make as make_gguf
                dn_t = O.Q4_K if I % 256 == 0 else O.Q8_0
                experts = [make_gguf(rng, H, I, O.Q4_K, dn_t) for _ in range(E)]
                for ei, ex in enumerate(experts):
                    eng.load_gguf_expert(li, ei, ex.gate, ex.up, ex.down, O.Q4_K, dn_t, I)
            else:
                experts = make_experts(rng, E, H, I); upload(eng, li, experts)
            gate = ((rng.random((E, H)) - 0.5) * 0.1).astype(F); keep.append(gate)
            eng.set_route_weight_f32(li, gate, None, None)
            sgu = W(2 * SI, H); sd = W(H, SI)
            st.set_decode_layer_moe(li, li, li, sgu[0], sd[0], None)
            L.update(mlp="moe", gate=gate, experts=experts, sgu=sgu[1], sd=sd[1])
        orc.layers.append(L)
    st.finalize_decode()
    z = lambda xs: [(_ptr(x) if x is not None else 0) for x in xs]
    st.set_decode_state(5, kv_max, [0] * nL, [0] * nL, [0] * nL, [0] * nL, z(state_k), z(state_v))
    return st, eng, orc, keep, dict(V=V, nL=nL, kv_max=kv_max, klr=klr, rd=rd)


@pytest.mark.parametrize("cfg", [dict(), dict(lora=True, seed=2), dict(klr=256, nh=3, seed=4)])
@pytest.mark.parametrize("graph", [True, False])
def test_mla_decode_step_bit_exact(cfg, graph):
    st, eng, orc, keep, d = build(**cfg)
    st.set_use_graph(graph)
    tok = 9
    for step, pos in enumerate([5, 6, 7]):
        logits = np.empty(d["V"], F)
        st.decode_step(tok, pos, logits.ctypes.data)
        ref = orc.step(tok, pos)
        assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (step, float(np.max(np.abs(logits - ref))))
        tok = st.last_token()
        assert tok == O.sample_greedy(ref)
    for li in range(d["nL"]):
        ck = np.empty((d["kv_max"], d["klr"]), np.uint16); kp = np.empty((d["kv_max"], d["rd"]), np.uint16)
        st.get_decode_state(li, ck, kp, None, None)
        assert np.array_equal(ck, orc.layers[li]["ckv"]) and np.array_equal(kp, orc.layers[li]["kpe"])


@pytest.mark.parametrize("cfg,chunk,depth", [(dict(), 4, 2), (dict(lora=True, seed=2), 3, 1), (dict(klr=256, nh=3, seed=4), 16, 2)])
def test_mla_prompt_pass_equals_sequential_decode(cfg, chunk, depth):
    """kr_decode_prefill over MLA layers (batched projections + the decode launches with a token dimension) leaves logits and both
    FP16 caches bit-identical to feeding the same tokens through decode_step one by one."""
    toks = [3, 17, 99, 250, 7, 7, 41, 300, 12, 5, 64]
    st, eng, orc, keep, d = build(**cfg)
    seq = np.empty(d["V"], F)
    for i, t in enumerate(toks):
        st.decode_step(t, 5 + i, seq.ctypes.data)
    caches = []
    for li in range(d["nL"]):
        ck = np.empty((d["kv_max"], d["klr"]), np.uint16); kp = np.empty((d["kv_max"], d["rd"]), np.uint16)
        st.get_decode_state(li, ck, kp, None, None); caches.append((ck, kp))
    st2, eng2, orc2, keep2, d2 = build(**cfg)                  # same seed -> same weights and initial caches
    st2.set_prefill_chunk(chunk); st2.set_prefill_depth(depth)
    pf = np.empty(d["V"], F); st2.prefill(toks, 5, pf.ctypes.data)
    assert np.array_equal(pf.view(np.uint32), seq.view(np.uint32)), float(np.max(np.abs(pf - seq)))
    assert st2.last_token() == st.last_token()
    for li in range(d["nL"]):
        ck = np.empty((d["kv_max"], d["klr"]), np.uint16); kp = np.empty((d["kv_max"], d["rd"]), np.uint16)
        st2.get_decode_state(li, ck, kp, None, None)
        assert np.array_equal(ck, caches[li][0]) and np.array_equal(kp, caches[li][1])


@pytest.mark.parametrize("cfg", [dict(), dict(lora=True, seed=2)])
def test_mla_fp8_latent_cache_bit_exact(cfg):
    """FP8-E4M3 compressed-KV / rope caches (set_kv_dtype): decode steps, both caches and the prompt pass against the oracle's E4M3 twin"""
    st, eng, orc, keep, d = build(**cfg)
    st.set_kv_dtype(True); O.set_kv_fp8(True)
    try:
        rng = np.random.default_rng(9)
        nL = d["nL"]
        ck = [O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["klr"])) * 0.5).astype(F)) for _ in range(nL)]
        kp = [O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["rd"])) * 0.5).astype(F)) for _ in range(nL)]
        for li in range(nL):
            orc.layers[li]["ckv"] = ck[li].astype(np.uint16); orc.layers[li]["kpe"] = kp[li].astype(np.uint16)   # oracle keeps one byte per u16 slot
        reset = lambda: st.set_decode_state(5, d["kv_max"], [0] * nL, [0] * nL, [0] * nL, [0] * nL, [_ptr(x) for x in ck], [_ptr(x) for x in kp])
        reset()
        tok = 9
        for step, pos in enumerate([5, 6, 7]):
            logits = np.empty(d["V"], F)
            st.decode_step(tok, pos, logits.ctypes.data)
            ref = orc.step(tok, pos)
            assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (step, float(np.max(np.abs(logits - ref))))
            tok = st.last_token()
        for li in range(nL):
            a = np.empty((d["kv_max"], d["klr"]), np.uint8); b = np.empty((d["kv_max"], d["rd"]), np.uint8)
            st.get_decode_state(li, a, b, None, None)
            assert np.array_equal(a, orc.layers[li]["ckv"].astype(np.uint8)) and np.array_equal(b, orc.layers[li]["kpe"].astype(np.uint8))
        # prompt pass with FP8 latent caches == decoding the prompt token by token
        toks = [3, 9, 27, 81, 5, 15, 200]
        reset()
        seq = np.empty(d["V"], F)
        for i, t in enumerate(toks):
            st.decode_step(t, 8 + i, seq.ctypes.data)
        reset()
        st.set_prefill_chunk(3)
        pf = np.empty(d["V"], F); st.prefill(toks, 8, pf.ctypes.data)
        assert np.array_equal(pf.view(np.uint32), seq.view(np.uint32))
    finally:
        O.set_kv_fp8(False)


@pytest.mark.parametrize("cfg,fp8", [(dict(kv_max=600, nh=11), False), (dict(kv_max=600, lora=True, seed=2, klr=256), False), (dict(kv_max=560, nh=9, seed=6), True)])
def test_mla_long_cache_split_attention_bit_exact(cfg, fp8):
    """kv_max_seq > 512: decode scores run in their own launch that shares every staged latent row between 8 heads (position blocks x head
    groups, ragged last head group), softmax + weighted sum per head; positions around the 64-row stage boundaries"""
    st, eng, orc, keep, d = build(**cfg)
    if fp8:
        st.set_kv_dtype(True); O.set_kv_fp8(True)
    try:
        nL = d["nL"]
        if fp8:
            rng = np.random.default_rng(4)
            ck = [O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["klr"])) * 0.5).astype(F)) for _ in range(nL)]
            kp = [O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["rd"])) * 0.5).astype(F)) for _ in range(nL)]
            for li in range(nL):
                orc.layers[li]["ckv"] = ck[li].astype(np.uint16); orc.layers[li]["kpe"] = kp[li].astype(np.uint16)
            st.set_decode_state(5, d["kv_max"], [0] * nL, [0] * nL, [0] * nL, [0] * nL, [_ptr(x) for x in ck], [_ptr(x) for x in kp])
        tok = 9
        for pos in [5, 62, 63, 64, 65, 127, 128, 191, d["kv_max"] - 1]:
            logits = np.empty(d["V"], F)
            st.decode_step(tok, pos, logits.ctypes.data)
            ref = orc.step(tok, pos)
            assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (pos, float(np.max(np.abs(logits - ref))))
            tok = O.sample_greedy(ref)
    finally:
        O.set_kv_fp8(False)


@pytest.mark.parametrize("cfg,fp8", [(dict(kv_max=9000, nh=5, seed=8), False), (dict(kv_max=8300, nh=3, seed=9, klr=256), True),
                                     (dict(kv_max=24000, nh=2, seed=10), False)])
def test_mla_streamed_score_row_bit_exact(cfg, fp8):
    """long decode caches: the score row stays in HBM (written by the head-shared scores launch); the softmax streams it -- max, exp in place,
    position-ordered sum over 1024-value tiles -- and the weighted sum runs on producer / consumer waves over column-major latent stages
    (FP16 and E4M3).  A randomly filled cache, positions around the 64-row stages and the tile edges, up to 24 000 positions"""
    st, eng, orc, keep, d = build(**cfg)
    if fp8:
        st.set_kv_dtype(True); O.set_kv_fp8(True)
    try:
        nL = d["nL"]
        rng = np.random.default_rng(14)
        conv = O.f32_to_e4m3 if fp8 else O.f32_to_bf16
        ck = [conv((rng.standard_normal((d["kv_max"], d["klr"])) * 0.5).astype(F)) for _ in range(nL)]
        kp = [conv((rng.standard_normal((d["kv_max"], d["rd"])) * 0.5).astype(F)) for _ in range(nL)]
        for li in range(nL):
            orc.layers[li]["ckv"] = ck[li].astype(np.uint16); orc.layers[li]["kpe"] = kp[li].astype(np.uint16)
        st.set_decode_state(5, d["kv_max"], [0] * nL, [0] * nL, [0] * nL, [0] * nL, [_ptr(x) for x in ck], [_ptr(x) for x in kp])
        tok = 9
        for pos in [5, 63, 64, 4094, 4095, 4096, 4131, 8191, 8192, d["kv_max"] - 1]:   # 24000 positions: streamed without the hook
            logits = np.empty(d["V"], F)
            st.decode_step(tok, pos, logits.ctypes.data)
            ref = orc.step(tok, pos)
            assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (pos, float(np.max(np.abs(logits - ref))))
            tok = O.sample_greedy(ref)
    finally:
        O.set_kv_fp8(False)


def test_mla_production_widths_bit_exact():
    """DeepSeek-V2-Lite widths: hidden 2048, 16 heads, kv_lora 512, expert intermediate 1408 (11 quantization groups: the odd-group
    padding of the lane-tiled layout), shared expert 2816, top-6 (expert count reduced to 12); decode steps + prompt pass"""
    st, eng, orc, keep, d = build(seed=31, nh=16, kv_max=40, dims=(2048, 384, 12, 6, 1408, 2816))
    tok = 4
    for pos in [5, 6, 30]:
        logits = np.empty(d["V"], F)
        st.decode_step(tok, pos, logits.ctypes.data)
        ref = orc.step(tok, pos)
        assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (pos, float(np.max(np.abs(logits - ref))))
        tok = O.sample_greedy(ref)
    st2, eng2, orc2, keep2, d2 = build(seed=31, nh=16, kv_max=40, dims=(2048, 384, 12, 6, 1408, 2816))
    toks = [3, 17, 99, 250, 7, 7, 41, 300, 12]
    seq = np.empty(d["V"], F)
    st2.set_prefill_chunk(4)
    pf = np.empty(d["V"], F); st2.prefill(toks, 5, pf.ctypes.data)
    st3, eng3, orc3, keep3, d3 = build(seed=31, nh=16, kv_max=40, dims=(2048, 384, 12, 6, 1408, 2816))
    for i, t in enumerate(toks):
        st3.decode_step(t, 5 + i, seq.ctypes.data)
    assert np.array_equal(pf.view(np.uint32), seq.view(np.uint32)), float(np.max(np.abs(pf - seq)))


def test_mla_geometry_errors():
    from krasis_amd import CpuDecodeStore, KrasisEngine, ModelConfig
    eng = KrasisEngine(); eng.configure(ModelConfig(256, 128, 8, 2, 1, 0, 1.0))
    st = CpuDecodeStore(128, True, False); st.set_moe_store(eng)
    with pytest.raises(RuntimeError):                        # configure_decode first (decode.rs:2153-2154)
        st.add_decode_mla_layer(0, 0, 0, 0, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 512, 128, 64, 128, 0.1)


@pytest.mark.parametrize("cfg", [dict(), dict(lora=True, seed=2), dict(klr=256, nh=3, seed=4), dict(dims=(2048, 512, 16, 6, 256, 512), nh=16, seed=9),
                                 dict(dims=(512, 384, 8, 6, 1408, 2816), nh=4, seed=13)])
def test_mla_decode_step_tolerance_mode(cfg):
    """KR_DECODE_FAST on MLA layers (round 4): the input add + RMSNorm folded into the kv_a | q (LoRA: kv_a | q_a) projection launch, the o projection on the
    K-split tree-sum matvec fed by the f32 w_vc output, the MoE block on the mode's three launches; absorb / scores / weighted sum / w_vc keep the exact
    kernels.  STATED TOLERANCE (the mode's): logits within 2e-3 of the oracle driver (max |diff| / max |ref|), same greedy token, latent caches within
    3e-3; the fourth case has V2-Lite's widths (H 2048, 16 heads), the last its expert widths (I = 1408: an odd group count, two waves per routed slot and
    four for the twice-as-wide shared expert in the down + combine launch; the absorption / latent norm / w_vc launches are the mode's tree-sum forms)."""
    st, eng, orc, keep, d = build(**cfg)
    st.set_attention_mode(False, decode_fast=True)
    tok = 9
    for step, pos in enumerate([5, 6, 7, 8]):
        logits = np.empty(d["V"], F)
        st.decode_step(tok, pos, logits.ctypes.data)
        ref = orc.step(tok, pos)
        err = float(np.abs(logits - ref).max() / np.abs(ref).max())
        assert np.isfinite(logits).all() and err <= 2e-3, (step, err)
        assert err > 0.0 or step > 0                      # the tolerance kernels ran (another summation order)
        assert int(np.argmax(logits)) == O.sample_greedy(ref)
        tok = O.sample_greedy(ref)
    for li in range(d["nL"]):
        ck = np.empty((d["kv_max"], d["klr"]), np.uint16); kp = np.empty((d["kv_max"], d["rd"]), np.uint16)
        st.get_decode_state(li, ck, kp, None, None)
        a = ck.view(np.float16).astype(F); b = orc.layers[li]["ckv"].view(np.float16).astype(F)
        assert float(np.abs(a - b).max() / np.abs(b).max()) <= 3e-3


@pytest.mark.parametrize("graph,fast", [(True, False), (False, False), (True, True)])
def test_mla_native_gguf_decode_at_v2lite_widths(graph, fast):
    """VERDICT r4 next #3a: the shape bench.py's `v2lite-q4k-gguf` configuration runs, as a parity test -- MLA layers at V2-Lite's widths (hidden 2048, 16 heads,
    kv_lora 512), routed experts as NATIVE GGUF blocks with I = 1408 (Q4_K gate / up; Q8_0 down because 1408 is not a multiple of 256: 44 blocks per row), the
    shared expert twice as wide (2816, transposed INT4, un-gated), top-6 without renormalisation; expert count reduced to 10.  Exact mode: logits and greedy
    token BIT FOR BIT against the oracle's moe_forward_gguf-driven decode (moe.rs:990-1110, gguf_kernels.rs:690-756), graph and eager.  KR_DECODE_FAST: the
    routed slots of the mode's two expert launches walk the GGUF blocks beside a shared expert on two waves (`gguf && ps > 1` in kr_launch_fw2): logits within
    the mode's 2e-3, same greedy token."""
    st, eng, orc, keep, d = build(seed=41, nh=16, kv_max=40, dims=(2048, 384, 10, 6, 1408, 2816), gguf=True)
    st.set_use_graph(graph)
    if fast:
        st.set_attention_mode(False, decode_fast=True)
    tok = 4
    for step, pos in enumerate([5, 6, 30]):
        logits = np.empty(d["V"], F)
        st.decode_step(tok, pos, logits.ctypes.data)
        ref = orc.step(tok, pos)
        if fast:
            err = float(np.abs(logits - ref).max() / np.abs(ref).max())
            assert np.isfinite(logits).all() and 0.0 < err <= 2e-3, (step, err)
            assert int(np.argmax(logits)) == O.sample_greedy(ref)
        else:
            assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (step, float(np.max(np.abs(logits - ref))))
            assert st.last_token() == O.sample_greedy(ref)
        tok = O.sample_greedy(ref)


@pytest.mark.parametrize("klr,fp8", [(512, False), (256, False), (512, True), (256, True)])
def test_mla_tolerance_attention_short_cache(klr, fp8):
    """KR_DECODE_FAST, MLA over a short cache (kv_max_seq <= 1024): scores + softmax + weighted sum of the latent rows as ONE tree-sum launch per head
    (kr_fmla_kernel: 16 lanes per position, one wave per cache row, request batches of 32 positions) instead of the exact-order staged launch.  Positions on both
    sides of the batch boundaries up to the end of a 200-position cache, FP16 and E4M3 latent caches, kv_lora 512 / 256: logits within the mode's bound of the
    oracle driver (2e-3 with FP16 caches, 5e-3 with E4M3: docs/design/11), same greedy token, and within the same bound of the mode's exact-order attention launch
    (kr_decode_set_option "gqa_fused" 0) on the same steps."""
    outs = {}
    try:
        for fused in (1, 0):
            st, eng, orc, keep, d = build(seed=7, klr=klr, nh=4, kv_max=200)
            nL = d["nL"]
            if fp8:
                st.set_kv_dtype(True); O.set_kv_fp8(True)
                rng = np.random.default_rng(9)
                ck = [O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["klr"])) * 0.5).astype(F)) for _ in range(nL)]
                kp = [O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["rd"])) * 0.5).astype(F)) for _ in range(nL)]
                for li in range(nL):
                    orc.layers[li]["ckv"] = ck[li].astype(np.uint16); orc.layers[li]["kpe"] = kp[li].astype(np.uint16)
                st.set_decode_state(5, d["kv_max"], [0] * nL, [0] * nL, [0] * nL, [0] * nL, [_ptr(x) for x in ck], [_ptr(x) for x in kp])
                keep += ck + kp
            st.set_attention_mode(False, decode_fast=True)
            st.set_option("gqa_fused", fused)
            bound = 5e-3 if fp8 else 2e-3
            tok = 9; lg = []
            for pos in [0, 1, 15, 16, 31, 32, 33, 63, 64, 65, 127, 128, 199]:
                logits = np.empty(d["V"], F)
                st.decode_step(tok, pos, logits.ctypes.data)
                ref = orc.step(tok, pos)
                err = float(np.abs(logits - ref).max() / np.abs(ref).max())
                assert np.isfinite(logits).all() and err <= bound, (fused, pos, err)
                assert int(np.argmax(logits)) == O.sample_greedy(ref), (fused, pos)
                tok = O.sample_greedy(ref); lg.append(logits)
            outs[fused] = lg
        for a, b in zip(outs[1], outs[0]):
            assert float(np.abs(a - b).max() / np.abs(b).max()) <= (5e-3 if fp8 else 2e-3)
    finally:
        O.set_kv_fp8(False)
