"""GPU parity: HIP decode-MoE path vs the oracle, through the C ABI.  Bit-exact (f32 outputs compared as bits)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import make_experts, rand_bf16, upload

pytestmark = pytest.mark.gpu


def _engine(H, I, E, k, n_shared=0, rsf=1.0, layers=1, swiglu=0.0, alpha=0.0):
    from krasis_amd import KrasisEngine, ModelConfig
    eng = KrasisEngine()
    eng.configure(ModelConfig(H, I, E, k, layers, n_shared, rsf, swiglu, alpha))
    return eng


def _oracle_moe(experts, ids, w, act, shared=None, rsf=1.0, swiglu=0.0, alpha=0.0):
    sel = [(experts[i], wi) for i, wi in zip(ids, w) if i >= 0]
    return O.moe_forward_unified([s[0] for s in sel], [s[1] for s in sel], act, shared, rsf, swiglu, alpha, O.SIG_POLY5_DIV)


@pytest.mark.parametrize("H,I,E,k,bits", [
    (256, 128, 8, 2, 4),
    (256, 384, 8, 3, 4),      # odd number of groups in the down projection (pair padding)
    (2048, 512, 16, 10, 4),   # Qwen3-Coder-Next expert shape
    (2048, 1408 - 128, 8, 6, 4),  # 10 groups
    (512, 256, 8, 4, 8),      # INT8 path
    (2048, 512, 12, 10, 8),
])
def test_moe_forward_bit_exact(H, I, E, k, bits):
    rng = np.random.default_rng(H + I + E + bits)
    experts = make_experts(rng, E, H, I, bits)
    eng = _engine(H, I, E, k)
    upload(eng, 0, experts)
    for trial in range(3):
        act = rand_bf16(rng, H, 1.0 if trial else 0.01)
        ids = rng.choice(E, k, replace=False).astype(np.int32)
        w = rng.random(k).astype(np.float32); w /= w.sum()
        got = np.frombuffer(eng.moe_forward(0, act.tobytes(), ids.tolist(), w.tolist()), np.float32)
        ref = _oracle_moe(experts, ids, w, act)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), np.max(np.abs(got - ref))


def test_shared_expert_rsf_and_skip_ids():
    H, I, E, k = 512, 256, 8, 4
    rng = np.random.default_rng(7)
    experts = make_experts(rng, E, H, I)
    shared = make_experts(rng, 1, H, 2 * I)[0]     # n_shared_experts = 2 -> shared intermediate = 2*I
    eng = _engine(H, I, E, k, n_shared=2, rsf=2.5)
    upload(eng, 0, experts, shared)
    B = 5
    act = rand_bf16(rng, (B, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(B)]).astype(np.int32)
    ids[1, 2] = -1; ids[3, :] = -1                    # -1 = skip (moe.rs:2904)
    w = rng.random((B, k)).astype(np.float32)
    out = np.empty((B, H), np.uint16)
    eng.forward_moe_direct(0, act.ctypes.data, ids.ctypes.data, w.ctypes.data, out.ctypes.data, B, k)
    for b in range(B):
        ref = _oracle_moe(experts, ids[b], w[b], act[b], shared, 2.5)
        assert np.array_equal(out[b], O.f32_to_bf16(ref)), b   # bf16 RNE of the f32 result (moe.rs:2947)


def test_gptoss_activation_variant():
    H, I, E, k = 256, 128, 4, 2
    rng = np.random.default_rng(11)
    experts = make_experts(rng, E, H, I, scale=0.3)
    eng = _engine(H, I, E, k, swiglu=7.0, alpha=1.702)
    upload(eng, 0, experts)
    act = rand_bf16(rng, H, 2.0); ids = np.array([1, 3], np.int32); w = np.array([0.6, 0.4], np.float32)
    got = np.frombuffer(eng.moe_forward(0, act.tobytes(), ids.tolist(), w.tolist()), np.float32)
    ref = _oracle_moe(experts, ids, w, act, swiglu=7.0, alpha=1.702)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_retile_roundtrip_and_errors():
    H, I, E, k = 256, 128, 4, 2
    rng = np.random.default_rng(3)
    experts = make_experts(rng, E, H, I)
    eng = _engine(H, I, E, k)
    with pytest.raises(RuntimeError):
        eng.moe_forward(0, b"\0" * (H * 2), [0], [1.0])          # nothing uploaded yet
    upload(eng, 0, experts)
    w13, w13s, w2, w2s = eng.download_expert(0, 2)
    assert np.array_equal(w13, experts[2].w13) and np.array_equal(w13s, experts[2].w13_scales)
    assert np.array_equal(w2, experts[2].w2) and np.array_equal(w2s, experts[2].w2_scales)
    with pytest.raises(ValueError):
        eng.moe_forward(0, b"\0" * 10, [0], [1.0])                 # wrong activation size (moe.rs:1790)
    with pytest.raises(ValueError):
        eng.moe_forward(0, b"\0" * (H * 2), [0, 1], [1.0])         # len mismatch (moe.rs:1798)
    with pytest.raises(ValueError):
        eng.moe_forward(5, b"\0" * (H * 2), [0], [1.0])            # bad layer


def test_synthetic_full_size_sampled_parity():
    """Full Qwen3-Coder-Next layer (512 experts, GPU-generated synthetic weights): the selected experts are read back,
    un-tiled and fed to the oracle -- the same bits on both sides."""
    H, I, E, k = 2048, 512, 512, 10
    eng = _engine(H, I, E, k, n_shared=1)
    eng.fill_synthetic(4, seed=1234)
    rng = np.random.default_rng(5)
    act = O.f32_to_bf16((rng.random(H, dtype=np.float32) - 0.5))
    ids = rng.choice(E, k, replace=False).astype(np.int32)
    w = rng.random(k).astype(np.float32); w /= w.sum()
    got = np.frombuffer(eng.moe_forward(0, act.tobytes(), ids.tolist(), w.tolist()), np.float32)
    sel = [O.UnifiedExpert(*eng.download_expert(0, int(i)), H, I) for i in ids]
    sh = O.UnifiedExpert(*eng.download_expert(0, -1), H, I)
    ref = O.moe_forward_unified(sel, w, act, sh, 1.0)
    assert np.isfinite(got).all() and np.abs(got).max() > 0
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_reduce_sum_bf16():
    import torch
    eng = _engine(256, 128, 4, 2)
    rng = np.random.default_rng(9)
    ins = [rand_bf16(rng, 1000) for _ in range(3)]
    dev = [torch.from_numpy(a.view(np.int16)).cuda() for a in ins]
    out = torch.empty(1000, dtype=torch.int16, device="cuda")
    eng.reduce_sum_bf16([d.data_ptr() for d in dev], out.data_ptr(), 1000)
    torch.cuda.synchronize(); eng.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint16), O.reduce_sum_bf16(ins))


def test_weight_export_api_of_the_reference_engine():
    """KrasisEngine.get_expert_w13_packed/_scales, get_expert_w2_packed/_scales, get_experts_batch, get_experts_all_batch, write_experts_{all,range}_into[_pinned],
    get_shared_expert_weights, write_shared_expert_into[_pinned] (moe.rs:1972-2709): byte layouts = the reference's Marlin GPU format -- the exported
    experts, uploaded into a second engine through kr_upload_expert_marlin, give a bit-identical moe_forward; range / batch / into / pinned variants agree
    byte for byte; the reference's error classes (ValueError on short buffers and bad ranges, RuntimeError without shared experts)."""
    from krasis_amd import KrasisEngine, ModelConfig
    from krasis_amd._lib import check
    from oracle import oracle as O  # noqa: F401  (test infrastructure only)
    H, I, E, k = 256, 128, 6, 2
    rng = np.random.default_rng(3)
    experts = make_experts(rng, E, H, I); sh = make_experts(rng, 1, H, I)[0]
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1, 1, 1.5)); upload(eng, 0, experts, sh)
    p13 = eng.get_expert_w13_packed(0); s13 = eng.get_expert_w13_scales(0); p2 = eng.get_expert_w2_packed(0); s2 = eng.get_expert_w2_scales(0)
    per = [len(x) // E for x in (p13, s13, p2, s2)]
    assert per[0] == 2 * I * H // 8 * 4 and per[1] == (H // 128) * 2 * I * 2 and per[2] == H * I // 8 * 4 and per[3] == (I // 128) * H * 2
    assert eng.get_expert_w13_packed(0, 2, 5) == p13[2 * per[0]:5 * per[0]]
    assert eng.get_experts_batch(0, [4, 1], "w2_scales") == s2[4 * per[3]:5 * per[3]] + s2[per[3]:2 * per[3]]
    a = eng.get_experts_all_batch(0, [3, 0])
    assert a[0] == p13[3 * per[0]:4 * per[0]] + p13[:per[0]] and a[3] == s2[3 * per[3]:4 * per[3]] + s2[:per[3]]
    bufs = [bytearray(E * n) for n in per]
    eng.write_experts_all_into(0, *bufs)
    assert [bytes(b) for b in bufs] == [p13, s13, p2, s2]
    bufs = [bytearray(2 * n) for n in per]
    eng.write_experts_range_into(0, 1, 3, *bufs)
    assert bytes(bufs[2]) == p2[per[2]:3 * per[2]]
    pinned = [np.zeros(2 * n, np.uint8) for n in per]
    eng.write_experts_range_into_pinned(0, 1, 3, *[v for b in pinned for v in (b.ctypes.data, b.size)])
    assert pinned[0].tobytes() == p13[per[0]:3 * per[0]] and pinned[3].tobytes() == s2[per[3]:3 * per[3]]
    with pytest.raises(ValueError, match="too small"):
        eng.write_experts_all_into(0, bytearray(10), bufs[1], bufs[2], bufs[3])
    with pytest.raises(ValueError, match="Invalid range"):
        eng.write_experts_range_into(0, 3, 3, *bufs)
    with pytest.raises(ValueError, match="Unknown weight_type"):
        eng.get_experts_batch(0, [0], "w3")
    shw = eng.get_shared_expert_weights(0)
    sb = [bytearray(len(x)) for x in shw]
    eng.write_shared_expert_into(0, *sb)
    assert [bytes(b) for b in sb] == list(shw)
    # round trip through the Marlin upload: same bits out of the MoE operator
    e2 = KrasisEngine(); e2.configure(ModelConfig(H, I, E, k, 1, 1, 1.5))
    for x in range(E):
        w = [np.frombuffer(t[x * n:(x + 1) * n], dt).copy() for t, n, dt in zip((p13, s13, p2, s2), per, (np.uint32, np.uint16, np.uint32, np.uint16))]
        check(e2._lib.kr_upload_expert_marlin(e2._h, 0, x, I, w[0].ctypes.data, w[1].ctypes.data, w[2].ctypes.data, w[3].ctypes.data, 4))
    w = [np.frombuffer(t, dt).copy() for t, dt in zip(shw, (np.uint32, np.uint16, np.uint32, np.uint16))]
    check(e2._lib.kr_upload_expert_marlin(e2._h, 0, -1, I, w[0].ctypes.data, w[1].ctypes.data, w[2].ctypes.data, w[3].ctypes.data, 4))
    e2._cpu_bits = e2._gpu_bits = 4
    act = rand_bf16(rng, H); ids = [5, 2]; wt = [0.7, 0.3]
    y1 = np.frombuffer(eng.moe_forward(0, act.tobytes(), ids, wt), np.float32); y2 = np.frombuffer(e2.moe_forward(0, act.tobytes(), ids, wt), np.float32)
    assert np.array_equal(y1.view(np.uint32), y2.view(np.uint32))
    e3 = KrasisEngine(); e3.configure(ModelConfig(H, I, E, k, 1, 0, 1.0)); upload(e3, 0, experts)
    with pytest.raises(RuntimeError, match="No shared experts"):
        e3.get_shared_expert_weights(0)
