"""GPU parity: router logits / scores / top-k ids are BIT-EXACT against the oracle (both reference tie rules)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
RULE_ENGINE, RULE_DECODE = 0, 1


def _engine(E, H, k, scoring="softmax", norm=True):
    from krasis_amd import KrasisEngine, ModelConfig
    eng = KrasisEngine()
    eng.configure(ModelConfig(H, 128, E, k, 1))
    eng.set_routing_config(scoring, norm, k, E, H)
    return eng


@pytest.mark.parametrize("E,H,k,scoring,bf16_gate", [
    (512, 2048, 10, "softmax", False),   # Qwen3-Coder-Next router, f32 gate
    (512, 2048, 10, "softmax", True),    # bf16-exact gate -> bf16 storage in HBM
    (64, 2048, 6, "softmax", True),      # DeepSeek-V2-Lite
    (256, 1024, 8, "sigmoid", False),
])
def test_decode_rule_bit_exact(E, H, k, scoring, bf16_gate):
    rng = np.random.default_rng(E + H)
    gate = ((rng.random((E, H), dtype=np.float32) - 0.5) * 0.04).astype(np.float32)   # +-0.02 (decode.rs:5181)
    if bf16_gate:
        gate = O.bf16_to_f32(O.f32_to_bf16(gate)).reshape(E, H)
    esc = ((rng.random(E, dtype=np.float32) - 0.5) * 0.01).astype(np.float32) if scoring == "sigmoid" else None
    eng = _engine(E, H, k, scoring)
    eng.set_route_weight_f32(0, gate, None, esc)
    m = 7
    x = (rng.random((m, H), dtype=np.float32) - 0.5).astype(np.float32)
    ids, w, lg = eng.route(0, x, m, RULE_DECODE, want_logits=True)
    for t in range(m):
        r_ids, r_w, r_lg = O.route_decode(gate, x[t], k, 0 if scoring == "sigmoid" else 1, True, None, esc)
        assert np.array_equal(lg[t].view(np.uint32), r_lg.view(np.uint32)), "logits"
        assert np.array_equal(ids[t], r_ids), "top-k ids"
        assert np.array_equal(w[t].view(np.uint32), r_w.view(np.uint32)), "weights"


@pytest.mark.parametrize("E,H,m,bf16_gate,use_bias", [(512, 2048, 100, True, False), (72, 256, 70, False, True), (64, 2048, 33, True, True), (200, 512, 64, False, False)])
def test_decode_rule_batch_logits_on_mfma_bit_exact(E, H, m, bf16_gate, use_bias):
    """batches of >= 32 tokens take the f32-MFMA logits kernel (kr_route_mfma.hip): 16 chains per (token, expert) as 16 accumulators, the
    reference's fold tree afterwards -- the same bits as the GEMV, hence as moe_route_matmul_avx2 (partial token / expert tiles included)"""
    rng = np.random.default_rng(E + m)
    gate = ((rng.random((E, H), dtype=np.float32) - 0.5) * 0.04).astype(np.float32)
    if bf16_gate:
        gate = O.bf16_to_f32(O.f32_to_bf16(gate)).reshape(E, H)
    bias = ((rng.random(E, dtype=np.float32) - 0.5) * 0.1).astype(np.float32) if use_bias else None
    k = 6
    eng = _engine(E, H, k, "softmax")
    eng.set_route_weight_f32(0, gate, bias, None)
    x = ((rng.random((m, H), dtype=np.float32) - 0.5) * 2).astype(np.float32)
    ids, w, lg = eng.route(0, x, m, RULE_DECODE, want_logits=True)
    for t in range(m):
        r_ids, r_w, r_lg = O.route_decode(gate, x[t], k, 1, True, bias, None)
        assert np.array_equal(lg[t].view(np.uint32), r_lg.view(np.uint32)), ("logits", t)
        assert np.array_equal(ids[t], r_ids) and np.array_equal(w[t].view(np.uint32), r_w.view(np.uint32)), t


def test_decode_rule_ties_follow_heap_order():
    E, H, k = 64, 128, 4
    gate = np.zeros((E, H), np.float32)
    vals = np.zeros(E, np.float32); vals[[3, 9, 20, 33, 47, 60]] = [2, 2, 2, 1, 2, 1]   # injected ties
    gate[:, 0] = vals
    x = np.zeros((1, H), np.float32); x[0, 0] = 1.0
    eng = _engine(E, H, k, "sigmoid", False)
    eng.set_route_weight_f32(0, gate)
    ids, w = eng.route(0, x, 1, RULE_DECODE)
    r_ids, r_w, _ = O.route_decode(gate, x[0], k, 0, False)
    assert np.array_equal(ids[0], r_ids) and np.array_equal(w[0].view(np.uint32), r_w.view(np.uint32))
    # all-equal scores (e.g. a zero / padded token): pure heap-order result
    x0 = np.zeros((1, H), np.float32)
    ids0, _ = eng.route(0, x0, 1, RULE_DECODE)
    assert np.array_equal(ids0[0], O.route_decode(gate, x0[0], k, 0, False)[0])


@pytest.mark.parametrize("sigmoid", [False, True])
def test_engine_rule_bit_exact(sigmoid):
    E, H, k = 128, 1024, 8
    rng = np.random.default_rng(17)
    gate = O.f32_to_bf16(((rng.random((E, H), dtype=np.float32) - 0.5) * 0.04).astype(np.float32)).reshape(E, H)
    bias = ((rng.random(E, dtype=np.float32) - 0.5) * 0.01).astype(np.float32)
    eng = _engine(E, H, k, "sigmoid" if sigmoid else "softmax")
    eng.set_routing_weights(0, gate.tobytes(), bias.tobytes())
    act = O.f32_to_bf16((rng.random((3, H), dtype=np.float32) - 0.5).astype(np.float32)).reshape(3, H)
    ids, w = eng.route(0, act, 3, RULE_ENGINE)
    for t in range(3):
        r_ids, r_w = O.route_engine(gate, act[t], k, sigmoid, True, bias)
        assert np.array_equal(ids[t], r_ids)
        assert np.array_equal(w[t].view(np.uint32), r_w.view(np.uint32))


@pytest.mark.parametrize("E,H,m,use_bias", [(512, 2048, 300, False), (72, 256, 70, True), (130, 512, 45, True),
                                                (200, 256, 300, True), (130, 256, 700, False), (512, 1024, 7100, True)])      # the workgroup-tiled kernel: 64-token tiles (H = 256: no k-split form) and 128-token tiles (7100 tokens x 512 experts), partial tiles
def test_decode_rule_batch_logits_tolerance_form(E, H, m, use_bias):
    """kr_moe_set_gemm_mode(e, 1) / KR_GEMM_FAST: the batch logits on the bf16 MFMA with x split into hi + lo bf16 (kr_route_logits_fast_kernel) instead of
    the 16 f32 chains.  STATED TOLERANCE: |fast - exact| <= 2e-5 * sum_k |x_k g_k| per logit (x carried to 2^-17, products summed in another order);
    the top-k ids equal the exact kernel's wherever the k-th and (k + 1)-th logits are further apart than twice that bound (partial token / expert tiles
    included).  An f32 gate (not bf16-exact) keeps the exact kernel: bit-identical."""
    from krasis_amd._lib import check
    rng = np.random.default_rng(E + m)
    gate = O.bf16_to_f32(O.f32_to_bf16(((rng.random((E, H), dtype=np.float32) - 0.5) * 0.04).astype(np.float32))).reshape(E, H)
    bias = ((rng.random(E, dtype=np.float32) - 0.5) * 0.1).astype(np.float32) if use_bias else None
    k = 6
    eng = _engine(E, H, k, "softmax")
    eng.set_route_weight_f32(0, gate, bias, None)
    x = ((rng.random((m, H), dtype=np.float32) - 0.5) * 2).astype(np.float32)
    ids0, w0, lg0 = eng.route(0, x, m, RULE_DECODE, want_logits=True)
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, 1))
    ids1, w1, lg1 = eng.route(0, x, m, RULE_DECODE, want_logits=True)
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, 0))
    bound = 2e-5 * (np.abs(x).astype(np.float64) @ np.abs(gate).astype(np.float64).T)
    err = np.abs(lg1.astype(np.float64) - lg0.astype(np.float64))
    assert (err <= bound).all(), float((err / bound).max())
    assert err.max() > 0.0                                   # the tolerance kernel ran (another summation order), not the exact one
    srt = np.sort(lg0, axis=1)[:, ::-1]
    clear = (srt[:, k - 1] - srt[:, k]) > 2 * bound.max(axis=1)
    assert clear.sum() >= m // 2
    assert np.array_equal(ids0[clear], ids1[clear])
    # an f32 gate that is not bf16-exact: the launcher refuses, the exact kernel runs
    gate32 = ((rng.random((E, H), dtype=np.float32) - 0.5) * 0.04).astype(np.float32)
    eng.set_route_weight_f32(0, gate32, bias, None)
    a = eng.route(0, x, m, RULE_DECODE, want_logits=True)[2]
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, 1))
    b = eng.route(0, x, m, RULE_DECODE, want_logits=True)[2]
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, 0))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
