"""Host-side de-quantizers for the re-quantizing GGUF load path (`gguf_native=False`, the reference's default and the only GGUF mode its
CLI reaches: weights/mod.rs:3375-3420 -> streaming_build_cpu_cache_from_gguf :3592 -> gguf_expert_from_f32 :4063).

Each function restates `dequant_*` of src/gguf.rs (:546-866, dispatcher dequantize_raw_data :872) in float32 numpy with the reference's
operation order (one rounding per `*` / `-`, no fused multiply-add), so the f32 values -- and therefore the bf16 values handed to
quantize_int4 / quantize_int8 -- carry the reference's bits.  Load-time code: it runs once per expert tensor on the host, the quantizer and
everything after it run on the GPU (kr_upload_expert_bf16)."""
import numpy as np

F32 = np.float32
# ggml type ids (gguf.rs:56-85)
GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q5_0, GGML_Q8_0, GGML_Q4_K, GGML_Q5_K, GGML_Q6_K, GGML_BF16 = 0, 1, 2, 6, 8, 12, 13, 14, 30


def cpu_bits(dtype: int):
    """gguf_type_to_cpu_bits (weights/mod.rs:26-42): the INT width a ggml type is re-quantized to."""
    if dtype in (GGML_Q4_0, 3, GGML_Q4_K):            # Q4_0 | Q4_1 | Q4_K
        return 4
    if dtype in (GGML_Q5_0, 7, GGML_Q5_K):            # Q5_0 | Q5_1 | Q5_K -> rounds down to 4
        return 4
    if dtype in (GGML_Q6_K, GGML_Q8_0, 9, 15, GGML_F16, GGML_BF16, GGML_F32):   # Q6_K | Q8_0 | Q8_1 | Q8_K | F16 | BF16 | F32
        return 8
    if dtype in (10, 11):                             # Q2_K | Q3_K
        return 4
    raise ValueError(f"ggml type {dtype} has no INT4 / INT8 mapping")


def _f16(b):
    return np.ascontiguousarray(b).view(np.float16).astype(F32)


def _blocks(data, n, qk, bb, name):
    nb = n // qk
    data = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
    if data.size < nb * bb:
        raise ValueError(f"{name} data too short: {data.size} < {nb * bb}")
    return data[: nb * bb].reshape(nb, bb), nb


def dequant_f32(data, n):
    return np.frombuffer(data, np.uint8)[: n * 4].view(F32).copy() if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)[: n * 4].view(F32).copy()


def dequant_f16(data, n):
    d = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
    return _f16(d[: n * 2])


def dequant_bf16(data, n):
    d = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
    return (np.ascontiguousarray(d[: n * 2]).view(np.uint16).astype(np.uint32) << 16).view(F32)


def dequant_q8_0(data, n):      # gguf.rs:573-593: out = d * q
    b, nb = _blocks(data, n, 32, 34, "Q8_0")
    d = _f16(b[:, 0:2])                                   # [nb, 1]
    q = b[:, 2:34].view(np.int8).astype(F32)
    return (d * q).reshape(-1)


def _nibbles32(qs):             # elements 0..15 = low nibbles, 16..31 = high nibbles (gguf.rs:620-624, 655-659)
    return np.concatenate([qs & 0x0F, qs >> 4], axis=1)


def dequant_q4_0(data, n):      # gguf.rs:639-663: out = d * (nibble - 8)
    b, nb = _blocks(data, n, 32, 18, "Q4_0")
    d = _f16(b[:, 0:2])
    q = _nibbles32(b[:, 2:18]).astype(np.int32) - 8
    return (d * q.astype(F32)).reshape(-1)


def dequant_q5_0(data, n):      # gguf.rs:599-633: out = d * ((nibble | bit << 4) - 16)
    b, nb = _blocks(data, n, 32, 22, "Q5_0")
    d = _f16(b[:, 0:2])
    qh = np.ascontiguousarray(b[:, 2:6]).view(np.uint32)  # [nb, 1]
    bit = ((qh >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.uint8)
    q = (_nibbles32(b[:, 6:22]) | (bit << 4)).astype(np.int32) - 16
    return (d * q.astype(F32)).reshape(-1)


def _scale_min_k4(scales):      # get_scale_min_k4 (gguf.rs:669-677) for j = 0..7 -> ([nb, 8] sc, [nb, 8] mn)
    s = scales.astype(np.uint8)
    sc = np.empty((s.shape[0], 8), np.uint8); mn = np.empty_like(sc)
    sc[:, 0:4] = s[:, 0:4] & 63; mn[:, 0:4] = s[:, 4:8] & 63
    sc[:, 4:8] = (s[:, 8:12] & 0x0F) | ((s[:, 0:4] >> 6) << 4)
    mn[:, 4:8] = (s[:, 8:12] >> 4) | ((s[:, 4:8] >> 6) << 4)
    return sc, mn


def dequant_q4_k(data, n):      # gguf.rs:684-741: out = (d * sc) * q - (dmin * mn)
    b, nb = _blocks(data, n, 256, 144, "Q4_K")
    d = _f16(b[:, 0:2]); dmin = _f16(b[:, 2:4])
    sc, mn = _scale_min_k4(b[:, 4:16])
    ds = d * sc.astype(F32); ms = dmin * mn.astype(F32)  # [nb, 8]
    qs = b[:, 16:144].reshape(nb, 4, 32)                  # 4 x 64 elements: 32 low nibbles then 32 high nibbles
    q = np.stack([qs & 0x0F, qs >> 4], axis=2).reshape(nb, 8, 32).astype(F32)
    return (ds[:, :, None] * q - ms[:, :, None]).reshape(-1)


def dequant_q5_k(data, n):      # gguf.rs:749-806
    b, nb = _blocks(data, n, 256, 176, "Q5_K")
    d = _f16(b[:, 0:2]); dmin = _f16(b[:, 2:4])
    sc, mn = _scale_min_k4(b[:, 4:16])
    ds = d * sc.astype(F32); ms = dmin * mn.astype(F32)
    qh = b[:, 16:48]                                      # [nb, 32]: bit 2g of byte l -> first half of group g, bit 2g + 1 -> second half
    qs = b[:, 48:176].reshape(nb, 4, 32)
    lo = qs & 0x0F; hi = qs >> 4
    g = np.arange(4, dtype=np.uint8)[None, :, None]
    b_lo = ((qh[:, None, :] >> (2 * g)) & 1).astype(np.uint8) * 16
    b_hi = ((qh[:, None, :] >> (2 * g + 1)) & 1).astype(np.uint8) * 16
    q = np.stack([(lo + b_lo), (hi + b_hi)], axis=2).reshape(nb, 8, 32).astype(F32)
    return (ds[:, :, None] * q - ms[:, :, None]).reshape(-1)


def dequant_q6_k(data, n):      # gguf.rs:814-866: out = (d * sc) * q with sc[0], sc[2], sc[4], sc[6] of each half (as the reference reads them)
    b, nb = _blocks(data, n, 256, 210, "Q6_K")
    ql = b[:, 0:128].reshape(nb, 2, 64); qh = b[:, 128:192].reshape(nb, 2, 32); sc = b[:, 192:208].view(np.int8).reshape(nb, 2, 8)
    d = _f16(b[:, 208:210])                               # [nb, 1]
    l0 = ql[:, :, 0:32]; l1 = ql[:, :, 32:64]
    q1 = ((l0 & 0x0F) | (((qh >> 0) & 3) << 4)).astype(np.int8).astype(np.int32) - 32
    q2 = ((l1 & 0x0F) | (((qh >> 2) & 3) << 4)).astype(np.int8).astype(np.int32) - 32
    q3 = ((l0 >> 4) | (((qh >> 4) & 3) << 4)).astype(np.int8).astype(np.int32) - 32
    q4 = ((l1 >> 4) | (((qh >> 6) & 3) << 4)).astype(np.int8).astype(np.int32) - 32
    dd = d[:, :, None]                                    # [nb, 1, 1]
    out = np.empty((nb, 2, 4, 32), F32)
    for k, q in enumerate((q1, q2, q3, q4)):
        out[:, :, k, :] = (dd * sc[:, :, 2 * k : 2 * k + 1].astype(F32)) * q.astype(F32)
    return out.reshape(-1)


_TABLE = {GGML_F32: dequant_f32, GGML_F16: dequant_f16, GGML_BF16: dequant_bf16, GGML_Q4_0: dequant_q4_0, GGML_Q5_0: dequant_q5_0,
          GGML_Q4_K: dequant_q4_k, GGML_Q5_K: dequant_q5_k, GGML_Q6_K: dequant_q6_k, GGML_Q8_0: dequant_q8_0}


def dequantize_raw_data(dtype: int, data, n_elements: int) -> np.ndarray:
    """gguf.rs:872 -- raw ggml blocks -> f32 [n_elements]"""
    fn = _TABLE.get(dtype)
    if fn is None:
        raise ValueError(f"Dequantization not implemented for ggml type {dtype}")
    return fn(data, n_elements)


def f32_to_bf16(x: np.ndarray) -> np.ndarray:
    """marlin.rs:25-30: round to nearest even on the raw bits (wrapping add, NaN payloads included)"""
    bits = np.ascontiguousarray(x, F32).view(np.uint32)
    return ((bits + (np.uint32(0x7FFF) + ((bits >> 16) & 1))) >> 16).astype(np.uint16)
