"""GPU parity of the native GGUF block experts (Q4_K / Q8_0 / Q4_0 integer path, Q5_0 / Q6_K scalar path) against the oracle's
restatement of src/gguf_kernels.rs + moe_forward_gguf (src/moe.rs:990).  Bit-exact.  (The reference has no test for these kernels:
the oracle side is 'parity unpinned', see oracle/README.md.)"""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import rand_bf16

pytestmark = pytest.mark.gpu
BB = {O.Q4_K: (256, 144), O.Q8_0: (32, 34), O.Q4_0: (32, 18), O.Q5_0: (32, 22), O.Q6_K: (256, 210)}


def rand_blocks(rng, t, rows, K):
    be, bb = BB[t]; nb = K // be
    raw = rng.integers(0, 256, size=(rows, nb, bb), dtype=np.uint8)
    d = (rng.random((rows, nb)) * 4e-3 + 5e-4).astype(np.float16).view(np.uint16)
    if t == O.Q6_K:
        raw[:, :, 208] = d & 0xFF; raw[:, :, 209] = d >> 8
    else:
        raw[:, :, 0] = d & 0xFF; raw[:, :, 1] = d >> 8
        if t == O.Q4_K:
            dm = (rng.random((rows, nb)) * 2e-3 + 1e-4).astype(np.float16).view(np.uint16)
            raw[:, :, 2] = dm & 0xFF; raw[:, :, 3] = dm >> 8
    return np.ascontiguousarray(raw.reshape(rows, nb * bb))


def make(rng, H, I, gu_t, dn_t):
    return O.GgufExpert(rand_blocks(rng, gu_t, I, H), rand_blocks(rng, gu_t, I, H), rand_blocks(rng, dn_t, H, I), gu_t, dn_t, H, I)


@pytest.mark.parametrize("H,I,gu_t,dn_t", [
    (512, 256, O.Q4_K, O.Q4_K),
    (2048, 512, O.Q4_K, O.Q4_K),       # QCN expert shape, Q4_K
    (512, 96 * 2, O.Q4_K, O.Q8_0),     # intermediate not a multiple of 256 -> 32-block type for down (V2-Lite situation)
    (512, 224, O.Q4_0, O.Q4_0),
    (256, 160, O.Q8_0, O.Q5_0),        # scalar f32 fallback for the down projection
    (512, 256, O.Q4_K, O.Q6_K),
    (2048, 512, O.Q8_0, O.Q8_0),       # QCN Q8 config
])
def test_moe_forward_gguf_bit_exact(H, I, gu_t, dn_t):
    from krasis_amd import KrasisEngine, ModelConfig
    rng = np.random.default_rng(H + I + gu_t * 7 + dn_t)
    E, k = 6, 3
    experts = [make(rng, H, I, gu_t, dn_t) for _ in range(E)]
    shared = make(rng, H, I, gu_t, dn_t)
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1, 1, 1.5))
    for e, ex in enumerate(experts):
        eng.load_gguf_expert(0, e, ex.gate, ex.up, ex.down, gu_t, dn_t, I)
    eng.load_gguf_expert(0, -1, shared.gate, shared.up, shared.down, gu_t, dn_t, I)
    assert eng.has_gguf()
    B = 4
    act = rand_bf16(rng, (B, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(B)]).astype(np.int32)
    ids[2, 1] = -1
    w = rng.random((B, k)).astype(np.float32)
    got = np.frombuffer(eng.moe_forward(0, act[0].tobytes(), ids[0].tolist(), w[0].tolist()), np.float32)
    ref = O.moe_forward_gguf([experts[i] for i in ids[0]], w[0], act[0], shared, 1.5)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), float(np.max(np.abs(got - ref)))
    out = np.empty((B, H), np.uint16)
    eng.forward_moe_direct(0, act.ctypes.data, ids.ctypes.data, w.ctypes.data, out.ctypes.data, B, k)
    for b in range(B):
        sel = [(experts[i], wi) for i, wi in zip(ids[b], w[b]) if i >= 0]
        ref = O.moe_forward_gguf([s[0] for s in sel], [s[1] for s in sel], act[b], shared, 1.5)
        assert np.array_equal(out[b], O.f32_to_bf16(ref)), b


@pytest.mark.parametrize("H,I,gu_t,dn_t,shared", [
    (2048, 512, O.Q4_K, O.Q4_K, True),      # QCN expert shape, Q4_K super-blocks on the MFMA GEMM
    (512, 192, O.Q4_K, O.Q8_0, True),       # V2-Lite situation: intermediate not a multiple of 256 -> Q8_0 down, K % 256 != 0 tail stage
    (2048, 512, O.Q8_0, O.Q8_0, False),     # QCN Q8 config
    (512, 224, O.Q4_0, O.Q4_0, True),       # no MFMA form: the batch walks the bit-exact streaming kernels
])
def test_moe_prefill_gguf_mfma(H, I, gu_t, dn_t, shared):
    """kr_moe_prefill on a native-GGUF layer (M >= 64): Q4_K / Q8_0 blocks staged raw in LDS, int8 MFMA per 32-wide sub-block and activation
    digit, per-sub-block scale / min epilogue.  The integer sums are exact; the f32 chain runs once per output over the sub-blocks instead of
    the AVX2 kernel's 8 lane chains + hsum (kr_gguf_prefill.hip header), so the result is compared with the bit-exact kr_moe_forward on the
    same layer at a STATED TOLERANCE: |diff| <= 2e-5 * max|ref| (measured on MI355X: 4.6e-7 Q4_K/Q4_K at the QCN shape, 2.5e-6 Q4_K + Q8_0 down).
    Types without an MFMA form must stay bit-identical."""
    import os
    import torch
    from krasis_amd import GpuPrefillManager, KrasisEngine, ModelConfig, _lib
    from krasis_amd._lib import check
    rng = np.random.default_rng(H * 3 + I + gu_t + dn_t)
    E, k, M = 6, 3, 200
    experts = [make(rng, H, I, gu_t, dn_t) for _ in range(E)]
    sh = make(rng, H, I, gu_t, dn_t) if shared else None
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1, 1 if shared else 0, 1.5))
    for e, ex in enumerate(experts):
        eng.load_gguf_expert(0, e, ex.gate, ex.up, ex.down, gu_t, dn_t, I)
    if shared:
        eng.load_gguf_expert(0, -1, sh.gate, sh.up, sh.down, gu_t, dn_t, I)
    act = rand_bf16(rng, (M, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32)
    ids[5, 1] = -1; ids[77, :] = -1
    w = rng.random((M, k)).astype(np.float32)
    xd = torch.from_numpy(act.view(np.int16)).cuda(); idd = torch.from_numpy(ids).cuda(); wd = torch.from_numpy(w).cuda()
    ref = torch.empty((M, H), dtype=torch.float32, device="cuda"); got = torch.empty_like(ref)
    st = torch.cuda.current_stream().cuda_stream or 1
    check(eng._lib.kr_moe_forward(eng._h, 0, xd.data_ptr(), idd.data_ptr(), wd.data_ptr(), ref.data_ptr(), M, k, _lib.KR_OUT_F32, 0, st))
    check(eng._lib.kr_moe_prefill(eng._h, 0, xd.data_ptr(), idd.data_ptr(), wd.data_ptr(), got.data_ptr(), M, k, _lib.KR_OUT_F32, 0, st))
    torch.cuda.synchronize()
    r, g = ref.cpu().numpy(), got.cpu().numpy()
    # the streaming reference itself against the oracle on a few rows (bit for bit)
    for b in (0, 5, 77, M - 1):
        sel = [(experts[i], wi) for i, wi in zip(ids[b], w[b]) if i >= 0]
        if not sel and sh is None:
            assert not r[b].any() and not g[b].any()          # every slot skipped, no shared expert: zeros
            continue
        orc = O.moe_forward_gguf([s[0] for s in sel], [s[1] for s in sel], act[b], sh, 1.5)
        assert np.array_equal(r[b].view(np.uint32), orc.view(np.uint32)), b
    err = float(np.abs(g - r).max()); scale = float(np.abs(r).max())
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/r02_gguf_prefill_err.txt", "a") as f:
            f.write(f"H={H} I={I} types=({gu_t},{dn_t}) max|diff|={err:.3e} max|ref|={scale:.3e} rel={err / scale:.3e} rms_rel={float(np.sqrt(((g - r) ** 2).mean()) / np.sqrt((r ** 2).mean())):.3e}\n")
    if gu_t == O.Q4_0:
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
    else:
        assert np.isfinite(g).all() and err <= 2e-5 * scale, (err, scale)
    # bf16 output + routed_only through the operator class
    mgr = GpuPrefillManager(eng, k)
    ob = mgr.forward(0, xd.view(torch.bfloat16), idd, wd, routed_only=True)
    rb = torch.empty((M, H), dtype=torch.float32, device="cuda")
    check(eng._lib.kr_moe_forward(eng._h, 0, xd.data_ptr(), idd.data_ptr(), wd.data_ptr(), rb.data_ptr(), M, k, _lib.KR_OUT_F32, 1, st))
    torch.cuda.synchronize()
    d2 = (ob.float() - rb).abs().max().item()
    assert d2 <= 2 ** -7 * rb.abs().max().item() + 2e-5 * scale       # one bf16 rounding on top


def test_synthetic_gguf_fill_and_prefill_runs():
    """kr_fill_layer_synthetic_gguf (bench leg): finite outputs, decode and prompt-pass forms agree at the stated tolerance"""
    import torch
    from krasis_amd import KrasisEngine, ModelConfig, _lib
    from krasis_amd._lib import check
    H, I, E, k, M = 2048, 512, 16, 4, 96
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1))
    eng.fill_synthetic_gguf(O.Q4_K, O.Q4_K, seed=5)
    rng = np.random.default_rng(2)
    act = rand_bf16(rng, (M, H), 0.5); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32)
    w = rng.random((M, k)).astype(np.float32)
    xd = torch.from_numpy(act.view(np.int16)).cuda(); idd = torch.from_numpy(ids).cuda(); wd = torch.from_numpy(w).cuda()
    ref = torch.empty((M, H), dtype=torch.float32, device="cuda"); got = torch.empty_like(ref)
    st = torch.cuda.current_stream().cuda_stream or 1
    check(eng._lib.kr_moe_forward(eng._h, 0, xd.data_ptr(), idd.data_ptr(), wd.data_ptr(), ref.data_ptr(), M, k, _lib.KR_OUT_F32, 0, st))
    check(eng._lib.kr_moe_prefill(eng._h, 0, xd.data_ptr(), idd.data_ptr(), wd.data_ptr(), got.data_ptr(), M, k, _lib.KR_OUT_F32, 0, st))
    torch.cuda.synchronize()
    r, g = ref.cpu().numpy(), got.cpu().numpy()
    assert np.isfinite(r).all() and np.abs(r).max() > 0
    assert np.abs(g - r).max() <= 2e-5 * np.abs(r).max()


@pytest.mark.parametrize("H,I,E,k,M,shared,gu_t,dn_t", [
    (2048, 512, 6, 3, 200, True, O.Q4_K, O.Q4_K), (512, 256, 8, 4, 700, False, O.Q4_K, O.Q4_K), (1024, 512, 4, 2, 64, True, O.Q4_K, O.Q4_K),
    (512, 384, 8, 4, 300, True, O.Q4_K, O.Q8_0),      # the V2-Lite situation: Q8_0 down over an ODD number of 128-k groups (half-empty last stage)
    (2048, 512, 6, 3, 200, False, O.Q8_0, O.Q8_0),    # QCN Q8_0 everywhere
    (512, 1408, 4, 2, 150, False, O.Q4_K, O.Q8_0),    # V2-Lite's intermediate size itself (11 groups)
])
def test_moe_prefill_q4k_tolerance_form(H, I, E, k, M, shared, gu_t, dn_t):
    """kr_moe_set_gemm_mode(1) on a native Q4_K layer: the prompt-pass experts on the f16 matrix cores from the re-tiled copy of the super-blocks
    (kr_gq_repack_kernel -> kr_pfh_gemm_kernel<..., G = 1>): nibbles de-quantized in registers with the sub-block scale d * sc_j folded in (rounded once
    to f16), the offsets 8 d sc_j - dmin mn_j as K / 32 extra k-columns against the rows' per-32 sums, libm SiLU, f32 accumulation over the whole k
    range.  Yardstick: the exact path of the same layer (kr_moe_forward, bit-identical to the oracle's moe_forward_gguf -- checked on a few rows).
    STATED TOLERANCE: relative RMS error of the expert outputs <= 1.5e-3, max |diff| <= 1e-2 * max |ref| (f16 activations 2^-11, f16 sub-block scales
    2^-12; the INT4-g128 tolerance form sits at 2-4e-4 with its exact bf16 scales).  Q8_0 matrices (kr_gq8_repack_kernel -> BITS = 8, G = 1): int8 quants
    de-quantized exactly, one f16 scale 16 d per 32-wide block picked per k-step, no offset columns.  The error must also be ABOVE the exact block
    GEMM's (~1e-6): a silent fall-back to the exact form would pass every upper bound."""
    import os
    import torch
    from krasis_amd import KrasisEngine, ModelConfig, _lib
    from krasis_amd._lib import check
    rng = np.random.default_rng(H + I + M)
    experts = [make(rng, H, I, gu_t, dn_t) for _ in range(E)]
    sh = make(rng, H, I, gu_t, dn_t) if shared else None
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1, 1 if shared else 0, 1.5))
    for e, ex in enumerate(experts):
        eng.load_gguf_expert(0, e, ex.gate, ex.up, ex.down, gu_t, dn_t, I)
    if shared:
        eng.load_gguf_expert(0, -1, sh.gate, sh.up, sh.down, gu_t, dn_t, I)
    act = rand_bf16(rng, (M, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32)
    ids[5, 1] = -1; ids[min(77, M - 1), :] = -1
    w = rng.random((M, k)).astype(np.float32)
    xd = torch.from_numpy(act.view(np.int16)).cuda(); idd = torch.from_numpy(ids).cuda(); wd = torch.from_numpy(w).cuda()
    ref = torch.empty((M, H), dtype=torch.float32, device="cuda"); got = torch.empty_like(ref)
    st = torch.cuda.current_stream().cuda_stream or 1
    check(eng._lib.kr_moe_forward(eng._h, 0, xd.data_ptr(), idd.data_ptr(), wd.data_ptr(), ref.data_ptr(), M, k, _lib.KR_OUT_F32, 0, st))
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, 1))
    for rep in range(2):        # second call: the copy exists, nothing is built
        check(eng._lib.kr_moe_prefill(eng._h, 0, xd.data_ptr(), idd.data_ptr(), wd.data_ptr(), got.data_ptr(), M, k, _lib.KR_OUT_F32, 0, st))
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, 0))
    torch.cuda.synchronize()
    r, g = ref.cpu().numpy(), got.cpu().numpy()
    b = 0
    sel = [(experts[i], wi) for i, wi in zip(ids[b], w[b]) if i >= 0]
    orc = O.moe_forward_gguf([s[0] for s in sel], [s[1] for s in sel], act[b], sh, 1.5)
    assert np.array_equal(r[b].view(np.uint32), orc.view(np.uint32))
    rms = float(np.sqrt(((g - r) ** 2).mean()) / np.sqrt((r ** 2).mean())); mx = float(np.abs(g - r).max() / np.abs(r).max())
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/r03_q4k_fast_err.txt", "a") as f:
            f.write(f"GGUF tolerance form types {gu_t}/{dn_t} H={H} I={I} E={E} k={k} M={M} shared={shared}: rel RMS {rms:.3e}  max|diff|/max|ref| {mx:.3e}\n")
    assert np.isfinite(g).all() and rms <= 1.5e-3 and mx <= 1e-2, (rms, mx)
    assert rms > 2e-5, ("the tolerance form did not run", rms)
    if not shared:
        assert not g[min(77, M - 1)].any()            # every slot skipped: zeros


@pytest.mark.parametrize("which", ["gate", "up", "down"])
def test_tolerance_copy_follows_a_later_upload(which):
    """the KR_GEMM_FAST copy of a native-GGUF layer (gg_ensure_fast) is derived data.  An expert uploaded AFTER the first tolerance-mode
    prompt pass must be what the next tolerance pass computes with -- including `up`, whose columns live behind the gate set's copy."""
    import torch
    from krasis_amd import KrasisEngine, ModelConfig, _lib
    from krasis_amd._lib import check
    H, I, E, k, M = 512, 256, 4, 2, 96
    rng = np.random.default_rng(99)
    experts = [make(rng, H, I, O.Q4_K, O.Q4_K) for _ in range(E)]
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1, 0, 1.0))
    for e, ex in enumerate(experts):
        eng.load_gguf_expert(0, e, ex.gate, ex.up, ex.down, O.Q4_K, O.Q4_K, I)
    act = rand_bf16(rng, (M, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32)
    w = rng.random((M, k)).astype(np.float32)
    xd = torch.from_numpy(act.view(np.int16)).cuda(); idd = torch.from_numpy(ids).cuda(); wd = torch.from_numpy(w).cuda()
    st = torch.cuda.current_stream().cuda_stream or 1

    def run(fast):
        out = torch.empty((M, H), dtype=torch.float32, device="cuda")
        check(eng._lib.kr_moe_set_gemm_mode(eng._h, 1 if fast else 0))
        check(eng._lib.kr_moe_prefill(eng._h, 0, xd.data_ptr(), idd.data_ptr(), wd.data_ptr(), out.data_ptr(), M, k, _lib.KR_OUT_F32, 1, st))
        check(eng._lib.kr_moe_set_gemm_mode(eng._h, 0))
        torch.cuda.synchronize()
        return out.cpu().numpy()

    before = run(True)                                    # builds the tolerance copy
    bytes_with_copy = eng.device_bytes()
    new = make(rng, H, I, O.Q4_K, O.Q4_K)                 # replace ONE matrix of expert 1
    ex = experts[1]
    g, u, d = (new.gate if which == "gate" else ex.gate), (new.up if which == "up" else ex.up), (new.down if which == "down" else ex.down)
    eng.load_gguf_expert(0, 1, g, u, d, O.Q4_K, O.Q4_K, I)
    assert eng.device_bytes() < bytes_with_copy           # the stale copy's bytes are released and un-counted
    exact, fast = run(False), run(True)
    rms = float(np.sqrt(((fast - exact) ** 2).mean()) / np.sqrt((exact ** 2).mean()))
    changed = float(np.sqrt(((fast - before) ** 2).mean()) / np.sqrt((before ** 2).mean()))
    assert rms <= 1.5e-3, ("tolerance pass still reads the stale copy", rms)
    assert changed > 1e-2, changed                        # the upload did change the layer's output
