"""ctypes binding for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (krasis_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkrasis_oracle.so")

F32, F16, Q4_0, Q5_0, Q8_0, Q4_K, Q5_K, Q6_K, BF16 = 0, 1, 2, 6, 8, 12, 13, 14, 30
SIG_POLY5_DIV, SIG_POLY5_RCPNR, SIG_LIBM, SIG_POLY5_SCALAR = 0, 1, 2, 3


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "krasis_oracle.c")
    hdr = os.path.join(_HERE, "krasis_oracle.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkrasis_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.kro_bf16_to_f32.restype = C.c_float
        _lib.kro_f16_to_f32.restype = C.c_float
        _lib.kro_sigmoid.restype = C.c_float
        _lib.kro_f32_to_bf16.restype = C.c_uint16
        _lib.kro_f32_to_f16.restype = C.c_uint16
        _lib.kro_ggml_block_size.restype = C.c_size_t
        _lib.kro_ggml_block_bytes.restype = C.c_size_t
        _lib.kro_repack_tiled_u32.restype = C.c_size_t
        _lib.kro_repack_tiled_u16.restype = C.c_size_t
        _lib.kro_xs_next.restype = C.c_uint64
        _lib.kro_xs_next_u32.restype = C.c_uint32
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dt)


# ---------------------------------------------------------------- conversions
def f32_to_bf16(x) -> np.ndarray:
    """marlin.rs:25 RNE (vectorised; identical bit rule)."""
    b = _c(x, np.float32).view(np.uint32)
    r = b + (np.uint32(0x7FFF) + ((b >> np.uint32(16)) & np.uint32(1)))
    return (r >> np.uint32(16)).astype(np.uint16)


def bf16_to_f32(x) -> np.ndarray:
    return (_c(x, np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def f32_to_f16_bits(x) -> np.ndarray:
    return _c(x, np.float32).astype(np.float16).view(np.uint16)


# ---------------------------------------------------------------- A: codecs
def block_size(t: int) -> int:
    return int(lib().kro_ggml_block_size(C.c_int(t)))


def block_bytes(t: int) -> int:
    return int(lib().kro_ggml_block_bytes(C.c_int(t)))


def get_scale_min_k4(j: int, scales: np.ndarray):
    sc, mn = C.c_uint8(), C.c_uint8()
    s = _c(scales, np.uint8)
    lib().kro_get_scale_min_k4(C.c_int(j), _p(s), C.byref(sc), C.byref(mn))
    return sc.value, mn.value


def dequantize(t: int, data: np.ndarray, n: int) -> np.ndarray:
    d = _c(data, np.uint8)
    out = np.empty(n, np.float32)
    rc = lib().kro_dequantize(C.c_int(t), _p(d), C.c_size_t(n), _p(out))
    if rc != 0:
        raise ValueError(f"Dequantization not implemented for type {t}")
    return out


def quantize_int4(w_bf16: np.ndarray, gs: int = 128):
    w = _c(w_bf16, np.uint16)
    rows, cols = w.shape
    packed = np.empty((rows, cols // 8), np.uint32)
    scales = np.empty((rows, cols // gs), np.uint16)
    lib().kro_quantize_int4(_p(w), rows, cols, gs, _p(packed), _p(scales))
    return packed, scales


def quantize_int8(w_bf16: np.ndarray, gs: int = 128):
    w = _c(w_bf16, np.uint16)
    rows, cols = w.shape
    data = np.empty((rows, cols), np.int8)
    scales = np.empty((rows, cols // gs), np.uint16)
    lib().kro_quantize_int8(_p(w), rows, cols, gs, _p(data), _p(scales))
    return data, scales


def quantize_and_pack_legacy(w_bf16_kn: np.ndarray, gs: int = 128, bits: int = 4):
    """The reference's LEGACY GPU quantizer `_quantize_and_pack_gpu` (python/krasis/gpu_prefill.py:240-291), restated in numpy: w [K, N] bf16 ->
    (packed int32 [K / vals, N], scale bf16 [K / gs, N]).  Differs from the live rule (marlin.rs:145-207 = quantize_int4 above) in three places
    (trap 2): scale = max(|max| / qmax, |min| / (qmax + 1)) per group instead of amax / qmax; the DIVISION uses the unrounded f32 scale
    (the bf16 rounding happens only on the stored copy); round-half-EVEN (torch.round) instead of half-away.  Pinned by tests/golden/legacy_quant.npz
    (outputs of the imported reference function).  No product path uses this rule: kr_upload_expert_bf16 implements marlin.rs."""
    w = bf16_to_f32(_c(w_bf16_kn, np.uint16)).reshape(w_bf16_kn.shape).astype(np.float32)
    K, N = w.shape
    qmax = 7.0 if bits == 4 else 127.0
    g = w.reshape(K // gs, gs, N)
    mx = g.max(axis=1, keepdims=True); mn = g.min(axis=1, keepdims=True)
    scale = np.maximum(np.abs(mx) / np.float32(qmax), np.abs(mn) / np.float32(qmax + 1.0)).astype(np.float32)
    scale = np.maximum(scale, np.float32(1e-10))
    q = np.rint(g / scale)                                  # numpy rint == torch.round: half to even
    q = np.clip(q, -(qmax + 1.0), qmax).astype(np.int32) + int(qmax + 1)
    q = q.reshape(K, N)
    vals = 8 if bits == 4 else 4
    qv = q.reshape(K // vals, vals, N)
    packed = qv[:, 0, :].copy()
    for i in range(1, vals):
        packed |= qv[:, i, :] << (bits * i)
    return packed.astype(np.int32), f32_to_bf16(scale.reshape(K // gs, N)).reshape(K // gs, N)


def dequantize_int4(packed, scales, gs=128):
    packed = _c(packed, np.uint32); scales = _c(scales, np.uint16)
    rows, pc = packed.shape
    out = np.empty((rows, pc * 8), np.float32)
    lib().kro_dequantize_int4(_p(packed), _p(scales), rows, pc * 8, gs, _p(out))
    return out


def dequantize_int8(data, scales, gs=128):
    data = _c(data, np.int8); scales = _c(scales, np.uint16)
    rows, cols = data.shape
    out = np.empty((rows, cols), np.float32)
    lib().kro_dequantize_int8(_p(data), _p(scales), rows, cols, gs, _p(out))
    return out


def quantize_f32_to_transposed_int4(w: np.ndarray, gs: int = 128):
    w = _c(w, np.float32); rows, cols = w.shape
    pt = np.empty((cols // 8, rows), np.uint32); st = np.empty((cols // gs, rows), np.uint16)
    lib().kro_quantize_f32_to_transposed_int4(_p(w), rows, cols, gs, _p(pt), _p(st))
    return pt, st


def quantize_f32_to_transposed_int8(w: np.ndarray, gs: int = 128):
    w = _c(w, np.float32); rows, cols = w.shape
    dt = np.empty((cols, rows), np.int8); st = np.empty((cols // gs, rows), np.uint16)
    lib().kro_quantize_f32_to_transposed_int8(_p(w), rows, cols, gs, _p(dt), _p(st))
    return dt, st


def transpose_int4(packed, scales, gs=128):
    packed = _c(packed, np.uint32); scales = _c(scales, np.uint16)
    n, pk = packed.shape
    pt = np.empty((pk, n), np.uint32); st = np.empty((scales.shape[1], n), np.uint16)
    lib().kro_transpose_int4(_p(packed), _p(scales), n, pk * 8, gs, _p(pt), _p(st))
    return pt, st


def transpose_int8(data, scales, gs=128):
    data = _c(data, np.int8); scales = _c(scales, np.uint16)
    n, k = data.shape
    dt = np.empty((k, n), np.int8); st = np.empty((scales.shape[1], n), np.uint16)
    lib().kro_transpose_int8(_p(data), _p(scales), n, k, gs, _p(dt), _p(st))
    return dt, st


def repack_tiled_u32(src: np.ndarray):
    src = _c(src, np.uint32); kr, n = src.shape
    nt = (n + 255) // 256
    dst = np.empty((nt, kr, 256), np.uint32)
    lib().kro_repack_tiled_u32(_p(src), kr, n, _p(dst))
    return dst


def repack_tiled_u16(src: np.ndarray):
    src = _c(src, np.uint16); kr, n = src.shape
    nt = (n + 255) // 256
    dst = np.empty((nt, kr, 256), np.uint16)
    lib().kro_repack_tiled_u16(_p(src), kr, n, _p(dst))
    return dst


# ---------------------------------------------------------------- B: activation quantizers
def quant_act_int16_bf16(x_bf16, gs=128):
    x = _c(x_bf16, np.uint16); k = x.size
    q = np.empty(k, np.int16); s = np.empty(k // gs, np.float32)
    lib().kro_quant_act_int16_bf16(_p(x), k, gs, _p(q), _p(s))
    return q, s


def quant_act_int16_f32(x, gs=128):
    x = _c(x, np.float32); k = x.size
    q = np.empty(k, np.int16); s = np.empty(k // gs, np.float32)
    lib().kro_quant_act_int16_f32(_p(x), k, gs, _p(q), _p(s))
    return q, s


def gguf_quant_bf16(x_bf16):
    x = _c(x_bf16, np.uint16); k = x.size
    q = np.empty(k, np.int16); s = np.empty(k // 32, np.float32); sm = np.empty(k // 32, np.int32)
    lib().kro_gguf_quant_bf16(_p(x), k, _p(q), _p(s), _p(sm))
    return q, s, sm


def gguf_quant_f32(x):
    x = _c(x, np.float32); k = x.size
    q = np.empty(k, np.int16); s = np.empty(k // 32, np.float32); sm = np.empty(k // 32, np.int32)
    lib().kro_gguf_quant_f32(_p(x), k, _p(q), _p(s), _p(sm))
    return q, s, sm


def sigmoid(x: float, mode: int = SIG_POLY5_DIV) -> float:
    return float(lib().kro_sigmoid(C.c_float(x), C.c_int(mode)))


def silu_quant_int16(gate, up, gs=128, mode=SIG_POLY5_DIV):
    gate = _c(gate, np.float32); up = _c(up, np.float32); n = gate.size
    h = np.empty(n, np.float32); q = np.empty(n, np.int16); s = np.empty(n // gs, np.float32)
    lib().kro_silu_quant_int16(_p(gate), _p(up), n, gs, mode, _p(h), _p(q), _p(s))
    return h, q, s


def fast_silu_mul(gate, up, mode=SIG_POLY5_DIV):
    gate = _c(gate, np.float32); up = _c(up, np.float32); n = gate.size
    out = np.empty(n, np.float32)
    lib().kro_fast_silu_mul(_p(gate), _p(up), n, mode, _p(out))
    return out


# ---------------------------------------------------------------- C: matvecs
def matvec_int4_t(packed_t, scales_t, a, a_s, gs=128):
    packed_t = _c(packed_t, np.uint32); scales_t = _c(scales_t, np.uint16)
    a = _c(a, np.int16); a_s = _c(a_s, np.float32)
    kr, n = packed_t.shape
    out = np.empty(n, np.float32)
    lib().kro_matvec_int4_t(_p(packed_t), _p(scales_t), _p(a), _p(a_s), kr * 8, n, gs, _p(out))
    return out


def matvec_int8_t(data_t, scales_t, a, a_s, gs=128):
    data_t = _c(data_t, np.int8); scales_t = _c(scales_t, np.uint16)
    a = _c(a, np.int16); a_s = _c(a_s, np.float32)
    k, n = data_t.shape
    out = np.empty(n, np.float32)
    lib().kro_matvec_int8_t(_p(data_t), _p(scales_t), _p(a), _p(a_s), k, n, gs, _p(out))
    return out


def matvec_int4_rowmajor(packed, scales, a, a_s, gs=128):
    packed = _c(packed, np.uint32); scales = _c(scales, np.uint16)
    a = _c(a, np.int16); a_s = _c(a_s, np.float32)
    n, pk = packed.shape
    out = np.empty(n, np.float32)
    lib().kro_matvec_int4_rowmajor(_p(packed), _p(scales), _p(a), _p(a_s), pk * 8, n, gs, _p(out))
    return out


def gguf_matvec_int(t, w, a, a_s, a_sum, n, k):
    w = _c(w, np.uint8); a = _c(a, np.int16); a_s = _c(a_s, np.float32); a_sum = _c(a_sum, np.int32)
    out = np.empty(n, np.float32)
    lib().kro_gguf_matvec_int(t, _p(w), _p(a), _p(a_s), _p(a_sum), n, k, _p(out))
    return out


def gguf_set_avx2(on: bool):
    """Q4_K / Q8_0 integer rows through the AVX2 + OpenMP forms (same bits as the scalar lane form; tests/test_oracle_avx2.py)"""
    lib().kro_gguf_set_avx2(1 if on else 0)


def gguf_matvec_f32(t, w, x, n, k):
    w = _c(w, np.uint8); x = _c(x, np.float32)
    out = np.empty(n, np.float32)
    lib().kro_gguf_matvec_f32(t, _p(w), _p(x), n, k, _p(out))
    return out


def matvec_int4_tiled_avx2(packed_tiled, scales_tiled, a, a_s, k, n, gs=128, parallel=True):
    out = np.zeros(n, np.float32)
    lib().kro_matvec_int4_tiled_avx2(_p(_c(packed_tiled, np.uint32)), _p(_c(scales_tiled, np.uint16)),
                                     _p(_c(a, np.int16)), _p(_c(a_s, np.float32)), k, n, gs, _p(out), int(parallel))
    return out


def tile_expert(e: "UnifiedExpert") -> "UnifiedExpert":
    """repack_experts_to_tiled (moe.rs:1435): [K/8,N] -> [N/256,K/8,256] for an INT4 expert."""
    assert e.num_bits == 4 and e.w2_bits == 4
    return UnifiedExpert(repack_tiled_u32(e.w13), repack_tiled_u16(e.w13_scales), repack_tiled_u32(e.w2),
                         repack_tiled_u16(e.w2_scales), e.hidden, e.inter, e.gs, 4, 4)


def moe_forward_unified_tiled_avx2(experts_tiled, weights, act_bf16, mode=SIG_POLY5_DIV):
    n = len(experts_tiled)
    cs = [e.c() for e in experts_tiled]
    arr = (C.POINTER(_UnifiedExpertC) * n)(*[C.pointer(c) for c in cs])
    w = _c(weights, np.float32); act = _c(act_bf16, np.uint16)
    out = np.zeros(experts_tiled[0].hidden, np.float32)
    lib().kro_moe_forward_unified_tiled_avx2(arr, _p(w), n, _p(act), mode, _p(out))
    return out


def set_num_threads(n: int) -> None:
    lib().kro_set_num_threads(int(n))


def num_threads() -> int:
    return int(lib().kro_num_threads())


# ---------------------------------------------------------------- D/E: experts
class _UnifiedExpertC(C.Structure):
    _fields_ = [("w13", C.c_void_p), ("w13_scales", C.c_void_p), ("w2", C.c_void_p), ("w2_scales", C.c_void_p),
                ("hidden", C.c_int), ("inter", C.c_int), ("gs", C.c_int), ("num_bits", C.c_int), ("w2_bits", C.c_int),
                ("gate_bias", C.c_void_p), ("up_bias", C.c_void_p), ("down_bias", C.c_void_p)]


class _GgufExpertC(C.Structure):
    _fields_ = [("gate", C.c_void_p), ("up", C.c_void_p), ("down", C.c_void_p),
                ("gate_up_type", C.c_int), ("down_type", C.c_int), ("hidden", C.c_int), ("inter", C.c_int)]


@dataclass
class UnifiedExpert:
    """weights/mod.rs:287 -- CPU transposed layout: w13 [K/8,2N] u32 (int4) or [K,2N] i8; w2 [N/8,K] / [N,K]."""
    w13: np.ndarray
    w13_scales: np.ndarray
    w2: np.ndarray
    w2_scales: np.ndarray
    hidden: int
    inter: int
    gs: int = 128
    num_bits: int = 4
    w2_bits: int = 4
    gate_bias: Optional[np.ndarray] = None
    up_bias: Optional[np.ndarray] = None
    down_bias: Optional[np.ndarray] = None

    def c(self) -> _UnifiedExpertC:
        for name in ("w13", "w13_scales", "w2", "w2_scales"):
            assert getattr(self, name).flags["C_CONTIGUOUS"]
        s = _UnifiedExpertC()
        s.w13 = self.w13.ctypes.data; s.w13_scales = self.w13_scales.ctypes.data
        s.w2 = self.w2.ctypes.data; s.w2_scales = self.w2_scales.ctypes.data
        s.hidden, s.inter, s.gs, s.num_bits, s.w2_bits = self.hidden, self.inter, self.gs, self.num_bits, self.w2_bits
        s.gate_bias = None if self.gate_bias is None else self.gate_bias.ctypes.data
        s.up_bias = None if self.up_bias is None else self.up_bias.ctypes.data
        s.down_bias = None if self.down_bias is None else self.down_bias.ctypes.data
        return s


@dataclass
class GgufExpert:
    """weights/mod.rs:252 -- raw GGUF blocks, gate/up [inter, hidden], down [hidden, inter]."""
    gate: np.ndarray
    up: np.ndarray
    down: np.ndarray
    gate_up_type: int
    down_type: int
    hidden: int
    inter: int

    def c(self) -> _GgufExpertC:
        s = _GgufExpertC()
        s.gate = self.gate.ctypes.data; s.up = self.up.ctypes.data; s.down = self.down.ctypes.data
        s.gate_up_type, s.down_type, s.hidden, s.inter = self.gate_up_type, self.down_type, self.hidden, self.inter
        return s


def unified_from_bf16(gate_bf16, up_bf16, down_bf16, gs=128, num_bits=4, w2_bits=None) -> UnifiedExpert:
    """quantize_int4/int8 (marlin.rs) + UnifiedExpertWeights::from_expert_weights (weights/mod.rs:329,403)."""
    w2_bits = w2_bits or num_bits
    inter, hidden = gate_bf16.shape

    def q(w, bits):
        if bits == 4:
            p, s = quantize_int4(w, gs); return transpose_int4(p, s, gs)
        d, s = quantize_int8(w, gs); return transpose_int8(d, s, gs)
    gp, gsc = q(gate_bf16, num_bits); up_, usc = q(up_bf16, num_bits); dp, dsc = q(down_bf16, w2_bits)
    w13 = np.ascontiguousarray(np.concatenate([gp, up_], axis=1))
    w13s = np.ascontiguousarray(np.concatenate([gsc, usc], axis=1))
    return UnifiedExpert(w13, w13s, np.ascontiguousarray(dp), np.ascontiguousarray(dsc), hidden, inter, gs, num_bits, w2_bits)


def expert_forward_unified(e: UnifiedExpert, a, a_s, swiglu_limit=0.0, alpha=0.0, mode=SIG_POLY5_DIV):
    a = _c(a, np.int16); a_s = _c(a_s, np.float32)
    out = np.empty(e.hidden, np.float32)
    ec = e.c()
    lib().kro_expert_forward_unified(C.byref(ec), _p(a), _p(a_s), C.c_float(swiglu_limit), C.c_float(alpha), mode, _p(out))
    return out


def expert_forward_gguf(e: GgufExpert, act_bf16):
    act = _c(act_bf16, np.uint16)
    out = np.empty(e.hidden, np.float32)
    ec = e.c()
    lib().kro_expert_forward_gguf(C.byref(ec), _p(act), _p(out))
    return out


def moe_forward_unified(experts: Sequence[UnifiedExpert], weights, act_bf16, shared: Optional[UnifiedExpert] = None,
                        rsf: float = 1.0, swiglu_limit=0.0, alpha=0.0, mode=SIG_POLY5_DIV):
    n = len(experts)
    cs = [e.c() for e in experts]
    arr = (C.POINTER(_UnifiedExpertC) * max(n, 1))(*[C.pointer(c) for c in cs])
    w = _c(weights, np.float32); act = _c(act_bf16, np.uint16)
    hidden = (experts[0] if n else shared).hidden
    out = np.zeros(hidden, np.float32)
    sc = shared.c() if shared is not None else None
    lib().kro_moe_forward_unified(arr, _p(w), n, C.byref(sc) if sc is not None else None, C.c_float(rsf), _p(act),
                                  C.c_float(swiglu_limit), C.c_float(alpha), mode, _p(out))
    return out


def moe_forward_gguf(experts: Sequence[GgufExpert], weights, act_bf16, shared: Optional[GgufExpert] = None, rsf=1.0):
    n = len(experts)
    cs = [e.c() for e in experts]
    arr = (C.POINTER(_GgufExpertC) * max(n, 1))(*[C.pointer(c) for c in cs])
    w = _c(weights, np.float32); act = _c(act_bf16, np.uint16)
    hidden = (experts[0] if n else shared).hidden
    out = np.zeros(hidden, np.float32)
    sc = shared.c() if shared is not None else None
    lib().kro_moe_forward_gguf(arr, _p(w), n, C.byref(sc) if sc is not None else None, C.c_float(rsf), _p(act), _p(out))
    return out


def moe_prefill_bf16(experts: Sequence[UnifiedExpert], x_bf16, ids, w, rsf=1.0):
    x = _c(x_bf16, np.uint16); ids = _c(ids, np.int32); w = _c(w, np.float32)
    m, topk = ids.shape
    cs = [e.c() for e in experts]
    arr = (C.POINTER(_UnifiedExpertC) * len(cs))(*[C.pointer(c) for c in cs])
    out = np.zeros((m, experts[0].hidden), np.uint16)
    lib().kro_moe_prefill_bf16(arr, len(cs), _p(x), _p(ids), _p(w), m, topk, C.c_float(rsf), _p(out))
    return out


# ---------------------------------------------------------------- F: routers
def topk_indices(values, k):
    v = _c(values, np.float32); out = np.empty(k, np.int32)
    lib().kro_topk_indices(_p(v), v.size, k, _p(out))
    return out


def route_matmul(gate_f32, hidden_f32):
    g = _c(gate_f32, np.float32); h = _c(hidden_f32, np.float32)
    ne, hd = g.shape
    out = np.empty(ne, np.float32)
    lib().kro_route_matmul(_p(g), _p(h), ne, hd, _p(out))
    return out


def route_score_topk(logits, topk, scoring=1, norm_topk=True, e_score_corr=None):
    lg = _c(logits, np.float32).copy(); ne = lg.size
    esc = None if e_score_corr is None else _c(e_score_corr, np.float32)
    scores = np.zeros(ne, np.float32); ids = np.empty(topk, np.int32); w = np.empty(topk, np.float32)
    lib().kro_route_score_topk(_p(lg), ne, _p(esc), scoring, int(norm_topk), topk, _p(scores), _p(ids), _p(w))
    return ids, w, scores


def route_decode(gate_f32, hidden_f32, topk, scoring=1, norm_topk=True, bias=None, e_score_corr=None):
    """decode.rs:3291-3298: matmul, + bias, score_topk."""
    lg = route_matmul(gate_f32, hidden_f32)
    if bias is not None:
        lg = (lg + _c(bias, np.float32)).astype(np.float32)
    ids, w, _ = route_score_topk(lg, topk, scoring, norm_topk, e_score_corr)
    return ids, w, lg


def route_engine(gate_bf16, act_bf16, topk, sigmoid=False, norm_topk=True, corr_bias=None, swiglu_limit=0.0):
    g = _c(gate_bf16, np.uint16); a = _c(act_bf16, np.uint16)
    ne, hd = g.shape
    b = None if corr_bias is None else _c(corr_bias, np.float32)
    ids = np.empty(topk, np.int32); w = np.empty(topk, np.float32)
    lib().kro_route_engine(_p(g), _p(a), ne, hd, _p(b), int(sigmoid), int(norm_topk), topk, C.c_float(swiglu_limit), _p(ids), _p(w))
    return ids, w


# ---------------------------------------------------------------- G: decode ops
def fused_add_rmsnorm(hidden, residual, w, eps, first, bias_one=False):
    h = _c(hidden, np.float32).copy(); r = _c(residual, np.float32).copy(); w = _c(w, np.float32)
    lib().kro_fused_add_rmsnorm(_p(h), _p(r), _p(w), h.size, C.c_float(eps), int(first), int(bias_one))
    return h, r


def la_conv(qkvz, ba, conv_state, conv_w, a_log, dt_bias, scale, nk, nv, dk, dv, kd=4, mode=SIG_POLY5_DIV):
    qkvz = _c(qkvz, np.float32); ba = _c(ba, np.float32); cs = _c(conv_state, np.float32).copy()
    q = np.empty(nv * dk, np.float32); k = np.empty(nv * dk, np.float32); v = np.empty(nv * dv, np.float32)
    z = np.empty(nv * dv, np.float32); g = np.empty(nv, np.float32); beta = np.empty(nv, np.float32)
    lib().kro_la_conv(_p(qkvz), _p(ba), _p(cs), _p(_c(conv_w, np.float32)), _p(_c(a_log, np.float32)),
                      _p(_c(dt_bias, np.float32)), C.c_float(scale), _p(q), _p(k), _p(v), _p(z), _p(g), _p(beta),
                      nk, nv, dk, dv, kd, mode)
    return dict(q=q, k=k, v=v, z=z, g=g, beta=beta, conv_state=cs)


def la_recurrent(state, q, k, v, g, beta, nv, dk, dv):
    st = _c(state, np.float32).copy(); out = np.empty(nv * dv, np.float32)
    lib().kro_la_recurrent(_p(st), _p(_c(q, np.float32)), _p(_c(k, np.float32)), _p(_c(v, np.float32)),
                           _p(_c(g, np.float32)), _p(_c(beta, np.float32)), _p(out), nv, dk, dv)
    return out, st


def gated_rmsnorm_silu(recur, z, w, nv, dv, eps, mode=SIG_POLY5_DIV):
    out = np.empty(nv * dv, np.float32)
    lib().kro_gated_rmsnorm_silu(_p(_c(recur, np.float32)), _p(_c(z, np.float32)), _p(_c(w, np.float32)), _p(out),
                                 nv, dv, C.c_float(eps), mode)
    return out



# ---- stand-alone CpuDecodeStore operators (decode.rs:473-890): scalar loops + libm exp (kro_op_*) ----
def op_rmsnorm(x, w, eps, bias_one=False):
    x = _c(x, np.float32); out = np.empty_like(x)
    lib().kro_op_rmsnorm(_p(x), _p(_c(w, np.float32)), _p(out), x.size, C.c_float(eps), int(bias_one))
    return out


def op_silu_mul(gate, up):
    gate = _c(gate, np.float32); out = np.empty_like(gate)
    lib().kro_op_silu_mul(_p(gate), _p(_c(up, np.float32)), _p(out), gate.size)
    return out


def op_gated_rmsnorm_silu(x, z, w, eps, nv, dv):
    out = np.empty(nv * dv, np.float32)
    lib().kro_op_gated_rmsnorm_silu(_p(_c(x, np.float32)), _p(_c(z, np.float32)), _p(_c(w, np.float32)), _p(out), C.c_float(eps), nv, dv)
    return out


def op_la_conv(qkvz, ba, conv_state, conv_w, a_log, dt_bias, scale, nk, nv, dk, dv, hr, kd):
    """-> (q, k, v, z, g, beta, conv_state_new); conv_state [conv_dim, kd] is copied, the copy is updated"""
    cs = _c(conv_state, np.float32).copy()
    q = np.empty(nv * dk, np.float32); k = np.empty(nv * dk, np.float32); v = np.empty(nv * dv, np.float32); z = np.empty(nv * dv, np.float32)
    g = np.empty(nv, np.float32); beta = np.empty(nv, np.float32)
    lib().kro_op_la_conv(_p(_c(qkvz, np.float32)), _p(_c(ba, np.float32)), _p(cs), _p(_c(conv_w, np.float32)), _p(_c(a_log, np.float32)), _p(_c(dt_bias, np.float32)),
                         C.c_float(scale), _p(q), _p(k), _p(v), _p(z), _p(g), _p(beta), nk, nv, dk, dv, hr, kd)
    return q, k, v, z, g, beta, cs


def gqa_step(q_in, k, v, q_norm, k_norm, gated, nh, nkv, hd, eps, rope_cos, rope_sin, k_cache, v_cache, position, sm_scale):
    """Returns (attn_out, k_cache, v_cache) with caches updated at `position`."""
    q_in = _c(q_in, np.float32); k = _c(k, np.float32).copy(); v = _c(v, np.float32).copy()
    kc = _c(k_cache, np.uint16).copy(); vc = _c(v_cache, np.uint16).copy()
    rc = _c(rope_cos, np.float32); rs = _c(rope_sin, np.float32)
    qn = None if q_norm is None else _c(q_norm, np.float32)
    kn = None if k_norm is None else _c(k_norm, np.float32)
    out = np.empty(nh * hd, np.float32)
    lib().kro_gqa_step(_p(q_in), _p(k), _p(v), _p(qn), 0 if qn is None else qn.size, _p(kn), 0 if kn is None else kn.size,
                       int(gated), nh, nkv, hd, C.c_float(eps), _p(rc), _p(rs), rc.shape[1], _p(kc), _p(vc),
                       kc.shape[0], position, C.c_float(sm_scale), _p(out))
    return out, kc, vc


def set_kv_fp8(on: bool) -> None:
    lib().kro_set_kv_fp8(int(on))


def f32_to_e4m3(x) -> np.ndarray:
    x = _c(x, np.float32); L = lib(); L.kro_f32_to_e4m3.restype = C.c_uint8
    return np.array([L.kro_f32_to_e4m3(C.c_float(float(v))) for v in x.reshape(-1)], np.uint8).reshape(x.shape)


def e4m3_to_f32(b) -> np.ndarray:
    b = _c(b, np.uint8); L = lib(); L.kro_e4m3_to_f32.restype = C.c_float
    return np.array([L.kro_e4m3_to_f32(int(v)) for v in b.reshape(-1)], np.float32).reshape(b.shape)


def rmsnorm_seq(x, w, eps):
    x = _c(x, np.float32).copy()
    lib().kro_rmsnorm_seq(_p(x), _p(_c(w, np.float32)), x.size, C.c_float(eps))
    return x


def mla_step(kv_out, q_full, kv_a_norm, w_kc, w_vc, rope_cos, rope_sin, nh, klr, nd, rd, vhd, eps, sm_scale, ckv_cache, kpe_cache, position):
    """Returns (v_projected [nh*vhd], ckv_cache, kpe_cache) with the caches updated at `position` (FP16 bits as u16)."""
    kv = _c(kv_out, np.float32).copy(); q = _c(q_full, np.float32).copy()
    ck = _c(ckv_cache, np.uint16).copy(); kp = _c(kpe_cache, np.uint16).copy()
    out = np.empty(nh * vhd, np.float32)
    lib().kro_mla_step(_p(kv), _p(q), _p(_c(kv_a_norm, np.float32)), _p(_c(w_kc, np.float32)), _p(_c(w_vc, np.float32)),
                       _p(_c(rope_cos, np.float32)), _p(_c(rope_sin, np.float32)), nh, klr, nd, rd, vhd, C.c_float(eps), C.c_float(sm_scale),
                       _p(ck), _p(kp), position, _p(out))
    return out, ck, kp


def cross_entropy_nll(logits, label: int) -> float:
    """-log softmax(logits)[label]: torch.nn.functional.cross_entropy(reduction="none") of the perplexity harness
    (perplexity/measure_ppl.py:218-227), restated in float64 (numpy); the HIP kernel is held to 4e-6 absolute against it"""
    x = np.asarray(logits, np.float64)
    mx = float(x.max())
    return float(np.log(np.exp(x - mx).sum()) + mx - x[int(label)])


def sample_greedy(logits) -> int:
    lg = _c(logits, np.float32)
    return int(lib().kro_sample_greedy(_p(lg), lg.size))


def sample_from_logits(logits, temperature, top_k, top_p, rng_state: int):
    """Returns (token, new_rng_state); the logits copy is scaled in place like the reference."""
    lg = _c(logits, np.float32).copy()
    st = C.c_uint64(rng_state)
    lib().kro_sample_from_logits.restype = C.c_int
    tok = lib().kro_sample_from_logits(_p(lg), lg.size, C.c_float(temperature), int(top_k), C.c_float(top_p), C.byref(st))
    return int(tok), int(st.value)


def reduce_sum_bf16(inputs: Sequence[np.ndarray]) -> np.ndarray:
    ins = [_c(a, np.uint16) for a in inputs]
    n = ins[0].size
    arr = (C.c_void_p * len(ins))(*[a.ctypes.data for a in ins])
    out = np.empty(n, np.uint16)
    lib().kro_reduce_sum_bf16(arr, len(ins), C.c_size_t(n), _p(out))
    return out.reshape(ins[0].shape)


# ---------------------------------------------------------------- synthetic generator (decode.rs:4356)
class Xorshift64:
    SEED = 0x12345678ABCDEF01  # decode.rs:4898

    def __init__(self, seed: int = SEED):
        class _S(C.Structure):
            _fields_ = [("state", C.c_uint64)]
        self._s = _S(seed if seed != 0 else 0xDEADBEEF)

    def next_u64(self) -> int:
        return int(lib().kro_xs_next(C.byref(self._s)))

    def next_u32(self) -> int:
        return int(lib().kro_xs_next_u32(C.byref(self._s)))

    def fill_u32(self, n) -> np.ndarray:
        out = np.empty(n, np.uint32)
        lib().kro_xs_fill_u32(C.byref(self._s), _p(out), C.c_size_t(out.size))
        return out

    def fill_scales_bf16(self, n) -> np.ndarray:
        out = np.empty(n, np.uint16)
        lib().kro_xs_fill_bf16_scales(C.byref(self._s), _p(out), C.c_size_t(out.size))
        return out

    def fill_f32(self, n, amp: float) -> np.ndarray:
        out = np.empty(n, np.float32)
        lib().kro_xs_fill_f32_uniform(C.byref(self._s), _p(out), C.c_size_t(out.size), C.c_float(amp))
        return out
