"""Regression tests for the round-1 advisor findings: queued steps without a host sync, last_token ordering, stale
prefill nibble sums after a late upload, forward_moe_routed on the legacy stream, expert-id range handling."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.test_decode_gpu import build
from tests.util import make_experts, rand_bf16, upload

pytestmark = pytest.mark.gpu
F = np.float32


def test_queued_steps_without_host_sync_match_sequential():
    """kr_decode_step with logits_out = NULL / device pointer never syncs: N steps are queued back to back.  (token, position) are
    kernel arguments of a by-value launch, so step i must still see ITS token and position (the pinned-slot form raced)."""
    import torch
    st, eng, orc, keep, d = build(seed=4)
    toks = [7, 3, 11, 2, 9, 4, 1, 8]
    seq = []
    for i, t in enumerate(toks):
        lg = np.empty(d["V"], F); st.decode_step(t, 5 + i, lg.ctypes.data); seq.append(lg.copy())
    d["reset"]()
    dev = torch.empty((len(toks), d["V"]), dtype=torch.float32, device="cuda")
    for i, t in enumerate(toks):                      # no host synchronisation between the calls
        st.decode_step(t, 5 + i, dev[i].data_ptr())
    eng.synchronize()
    got = dev.cpu().numpy()
    for i in range(len(toks)):
        assert np.array_equal(got[i].view(np.uint32), seq[i].view(np.uint32)), i
    # and with no logits pointer at all: last_token() must wait for the step that writes it
    d["reset"]()
    for i, t in enumerate(toks):
        st.decode_step(t, 5 + i)
    assert st.last_token() == O.sample_greedy(seq[-1])


def test_last_token_after_prefill_without_output_pointer():
    st, eng, orc, keep, d = build(seed=6)
    toks = [3, 1, 4, 1, 5, 9, 2, 6, 5]
    lg = None
    for i, t in enumerate(toks):
        lg = np.empty(d["V"], F); st.decode_step(t, 5 + i, lg.ctypes.data)
    want = O.sample_greedy(lg)
    for _ in range(3):                                # repeated: a stale read would show up as a mismatch on some iteration
        d["reset"]()
        assert st.prefill(toks, 5) == want


def test_negative_position_rejected():
    st, eng, orc, keep, d = build(seed=2)
    with pytest.raises(ValueError):
        st.decode_step(1, -1)


def test_upload_after_first_prefill_refreshes_nibble_sums():
    """MatSet.wsum (per-group nibble sums of the MFMA path) is built on the first kr_moe_prefill for ALL experts; an expert uploaded or
    replaced afterwards must not keep stale sums (kr_upload_expert_unified / kr_fill_layer_synthetic now drop them)."""
    import torch
    from krasis_amd import GpuPrefillManager, KrasisEngine, ModelConfig
    H, I, E, k = 256, 128, 8, 2
    rng = np.random.default_rng(31)
    experts = make_experts(rng, E, H, I)
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1))
    upload(eng, 0, experts[: E - 1])                   # expert E-1 is not loaded at first use
    M = 96
    x = rand_bf16(rng, (M, H))
    ids = np.stack([rng.choice(E - 1, k, replace=False) for _ in range(M)]).astype(np.int32)
    w = rng.random((M, k)).astype(F)
    mgr = GpuPrefillManager(eng, k)
    xd = torch.from_numpy(x.view(np.int16)).cuda().view(torch.bfloat16); wd = torch.from_numpy(w).cuda()
    mgr.forward(0, xd, torch.from_numpy(ids).cuda(), wd, routed_only=True)     # builds the sums
    # now load the missing expert and REPLACE expert 0
    new0 = make_experts(rng, 1, H, I)[0]
    eng.load_unified_expert(0, E - 1, experts[E - 1].w13, experts[E - 1].w13_scales, experts[E - 1].w2, experts[E - 1].w2_scales, 4, 4)
    eng.load_unified_expert(0, 0, new0.w13, new0.w13_scales, new0.w2, new0.w2_scales, 4, 4)
    experts[0] = new0
    ids2 = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32); ids2[:, 0] = np.where(np.arange(M) % 2, 0, E - 1); ids2[:, 1] = 1 + (np.arange(M) % (E - 2))
    out = mgr.forward(0, xd, torch.from_numpy(ids2).cuda(), wd, routed_only=True)
    torch.cuda.synchronize(); eng.synchronize()
    got = out.view(torch.int16).cpu().numpy().view(np.uint16)
    for t in range(0, M, 7):
        ref = O.moe_forward_unified([experts[i] for i in ids2[t]], w[t], x[t])
        assert np.array_equal(got[t], O.f32_to_bf16(ref)), t


def test_forward_moe_routed_on_legacy_stream_and_host_id_checks():
    from krasis_amd import KrasisEngine, ModelConfig
    H, I, E, k = 256, 128, 8, 2
    rng = np.random.default_rng(17)
    experts = make_experts(rng, E, H, I)
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1))
    upload(eng, 0, experts)
    eng.set_routing_config("softmax", True, k, E, H)
    gate = rand_bf16(rng, (E, H), 0.1)
    eng.set_routing_weights(0, gate.tobytes())
    act = rand_bf16(rng, H)
    a = np.empty(H, np.uint16); b = np.empty(H, np.uint16)
    eng.forward_moe_routed(0, act.ctypes.data, a.ctypes.data)                 # engine stream
    eng.forward_moe_routed(0, act.ctypes.data, b.ctypes.data, stream=1)       # (void*)1 = legacy default stream
    assert np.array_equal(a, b)
    from krasis_amd import _lib
    ids, w = eng.route(0, np.ascontiguousarray(act[None, :]), 1, rule=_lib.KR_ROUTE_RULE_ENGINE)
    ref = O.moe_forward_unified([experts[i] for i in ids[0]], w[0], act)
    assert np.array_equal(a, O.f32_to_bf16(ref))
    # host-resident ids are range-checked (the reference would panic on the index)
    with pytest.raises(ValueError):
        eng.moe_forward(0, act.tobytes(), [0, E], [0.5, 0.5])


def test_device_ids_out_of_range_are_skipped():
    import torch
    from krasis_amd import KrasisEngine, ModelConfig
    H, I, E, k = 256, 128, 8, 3
    rng = np.random.default_rng(19)
    experts = make_experts(rng, E, H, I)
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1))
    upload(eng, 0, experts)
    act = rand_bf16(rng, (2, H)); w = rng.random((2, k)).astype(F)
    ids = np.array([[1, E + 5, 3], [2, 4, 1 << 20]], np.int32)
    out = torch.empty((2, H), dtype=torch.int16, device="cuda")
    ad, idd, wd = torch.from_numpy(act.view(np.int16)).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(w).cuda()
    eng.forward_moe_direct(0, ad.data_ptr(), idd.data_ptr(), wd.data_ptr(), out.data_ptr(), 2, k)
    eng.synchronize(); torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16)
    for b in range(2):
        sel = [(experts[i], wi) for i, wi in zip(ids[b], w[b]) if 0 <= i < E]
        ref = O.moe_forward_unified([s[0] for s in sel], [s[1] for s in sel], act[b])
        assert np.array_equal(got[b], O.f32_to_bf16(ref)), b
