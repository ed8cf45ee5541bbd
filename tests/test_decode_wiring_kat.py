"""Known-answer tests of the MoE block's WIRING in decode_step (src/decode.rs:3286-3402) on inputs small enough to follow by hand.  The arithmetic
inside the operators is pinned elsewhere (tests/test_oracle_kat.py, tests/test_golden*.py); what a misreading of the Rust could still get wrong is
which tensor feeds what:
  (1) routed experts read bf16(hidden) (decode.rs:3307-3309), the shared expert and its gate row read the F32 hidden (:3356-3359, :3381-3384);
  (2) `moe_output *= rsf` scales the ROUTED sum only (:3338-3340), before `hidden = moe_output + shared_out` (:3396-3399);
  (3) the shared expert's output is multiplied by 1 / (1 + exp(-gate_row . hidden)) (:3379-3393).
Each is an exact identity here (powers of two, values that round the way the construction says), plus one output computed from first principles.
No GPU: the driver under test is tests/oracle_decode.py (the HIP decode step equals it bit for bit, tests/test_decode_gpu.py)."""
import numpy as np

from oracle import oracle as O
from tests.oracle_decode import OracleDecode

F = np.float32
H = I = 128


def _expert(gv, uv, dv):
    """gate / up: every intermediate unit = v * x[0]; down: every output = v * sum_i h[i]; v in {-8..7} / 8 is INT4-exact with bf16 scale 1/8"""
    g = np.zeros((I, H), F); g[:, 0] = gv
    u = np.zeros((I, H), F); u[:, 0] = uv
    d = np.full((H, I), dv, F)
    return O.unified_from_bf16(O.f32_to_bf16(g), O.f32_to_bf16(u), O.f32_to_bf16(d), 128, 4)


def _driver(rsf, with_shared=True, sg_val=0.0):
    emb = np.zeros((4, H), F)
    d = OracleDecode(H, 1e-6, False, 1, 1, True, rsf, emb, 4)
    L = dict(experts=[_expert(0.875, 0.875, 0.875), _expert(0.5, 0.75, 0.25)])
    if with_shared:
        gu = np.zeros((2 * I, H), F); gu[:I, 0] = 0.625; gu[I:, 0] = 0.75
        L["sgu"] = d.store_weight_f32(gu); L["sd"] = d.store_weight_f32(np.full((H, I), 0.375, F))
        sg = np.zeros((1, H), F); sg[0, 0] = sg_val
        L["sg"] = d.store_weight_f32(sg)
    return d, L


def _hid(x0):
    h = np.zeros(H, F); h[0] = x0
    return h


def test_routed_experts_see_bf16_hidden_shared_sees_f32():
    # 1 + 2^-8 is a tie between the bf16 neighbours 1.0 and 1 + 2^-7: RNE -> 1.0; 1 + 2^-8 + 2^-9 rounds up to 1 + 2^-7
    assert O.bf16_to_f32(O.f32_to_bf16(_hid(1.00390625)))[0] == F(1.0)
    assert O.bf16_to_f32(O.f32_to_bf16(_hid(1.005859375)))[0] == F(1.0078125)
    d, L = _driver(1.0, with_shared=False)
    a = d.moe_block(L, _hid(1.0), ([0], [1.0])); b = d.moe_block(L, _hid(1.00390625), ([0], [1.0])); c = d.moe_block(L, _hid(1.005859375), ([0], [1.0]))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))          # the routed expert cannot tell 1 + 2^-8 from 1.0 ...
    assert not np.array_equal(a, c)                                       # ... but sees the next bf16 value
    d, L = _driver(1.0, with_shared=True, sg_val=0.0)
    sa = d.moe_block(L, _hid(1.0), ([0], [0.0])); sb = d.moe_block(L, _hid(1.00390625), ([0], [0.0]))      # routed weight 0: only the shared expert is left
    assert not np.array_equal(sa, sb)                                     # the shared expert reads the f32 hidden: 2^-8 changes its INT16 scale


def test_rsf_scales_the_routed_sum_only():
    d1, L1 = _driver(1.0, sg_val=0.0); d2, L2 = _driver(2.0, sg_val=0.0)
    h = _hid(0.75)
    shared_only = d1.moe_block(L1, h, ([0], [0.0]))
    o1 = d1.moe_block(L1, h, ([0, 1], [0.75, 0.25])); o2 = d2.moe_block(L2, h, ([0, 1], [0.75, 0.25]))
    dn, Ln = _driver(1.0, with_shared=False)
    routed = dn.moe_block(Ln, h, ([0, 1], [0.75, 0.25]))
    assert np.array_equal(o1.view(np.uint32), (routed + shared_only).astype(F).view(np.uint32))                    # hidden = moe_output + shared_out
    assert np.array_equal(o2.view(np.uint32), ((routed * F(2.0)).astype(F) + shared_only).astype(F).view(np.uint32))   # rsf = 2: exact doubling of the routed part only


def test_shared_expert_is_scaled_by_the_libm_sigmoid_of_its_gate_row():
    h = _hid(0.5)
    d0, L0 = _driver(1.0, sg_val=0.0)
    half = d0.moe_block(L0, h, ([0], [0.0]))                 # gate row 0 -> sigmoid(0) = 0.5 exactly
    dn, Ln = _driver(1.0, sg_val=0.0); del Ln["sg"]
    full = dn.moe_block(Ln, h, ([0], [0.0]))                 # no gate row: unscaled
    assert np.array_equal(half.view(np.uint32), (full * F(0.5)).astype(F).view(np.uint32))
    dg, Lg = _driver(1.0, sg_val=0.875)                      # gate = 0.875 * 0.5 through the INT4 x INT16 matvec
    gv = dg.matvec(Lg["sg"], h)[0]
    want = (full * F(1.0 / (1.0 + np.exp(-np.float64(gv))))).astype(F)
    got = dg.moe_block(Lg, h, ([0], [0.0]))
    assert np.allclose(got, want, rtol=2e-7, atol=0)         # libm expf vs float64 exp: last bit


def test_one_routed_output_from_first_principles():
    """expert 0 on hidden = e0: x = bf16(1.0); INT16 digits: q0 = 32767, a_scale = f32(1 / 32767); gate = up = f32(f32(7 * 32767) * (0.125 * a_scale))
    (avx2.rs:1162-1176: one fma per group from zero); h_i = silu_poly5(gate) * up for every i, digits all 32767 with scale f32(h / 32767);
    y_j = f32(f32(7 * 32767 * 128) * (0.125 * h_scale)).  Every weight is 0.875 = q 7 x bf16 scale 0.125: quantize_int4 (marlin.rs:145) maps the
    row maximum to q = 7 with scale bf16(amax / 7), exact only when amax / 7 is a bf16 number."""
    d, L = _driver(1.0, with_shared=False)
    got = d.moe_block(L, _hid(1.0), ([0], [1.0]))
    a_scale = F(F(1.0) / F(32767.0))
    gate = F(F(7 * 32767) * F(F(0.125) * a_scale))
    hval = F(F(gate * F(O.sigmoid(float(gate), O.SIG_POLY5_DIV))) * gate)
    h_scale = F(hval / F(32767.0))
    y = F(F(7 * 32767 * 128) * F(F(0.125) * h_scale))
    assert np.all(got == got[0])
    assert got[0] == y, (float(got[0]), float(y))
