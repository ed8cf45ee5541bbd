"""KrasisEngine -- host mirror of the reference's PyO3 class (src/moe.rs:1377-3296) over the HIP C ABI.

Same method names, argument meaning and error behaviour as `krasis.KrasisEngine`; the arithmetic runs in
libkrasis_hip.so on an MI355X.  Raw-pointer methods take integer addresses exactly like the reference
(`forward_moe_direct`, moe.rs:2843) -- host or device addresses are both accepted.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import ModelConfigC, check, load_library

MAX_TOPK = 32  # moe.rs:2899


@dataclass
class ModelConfig:
    """Subset of config.json relevant to MoE (src/weights/mod.rs:51-70)."""
    hidden_size: int
    moe_intermediate_size: int
    n_routed_experts: int
    num_experts_per_tok: int
    num_moe_layers: int
    n_shared_experts: int = 0
    routed_scaling_factor: float = 1.0
    swiglu_limit: float = 0.0
    activation_alpha: float = 0.0
    group_size: int = 128

    def to_c(self) -> ModelConfigC:
        return ModelConfigC(self.hidden_size, self.moe_intermediate_size, self.n_routed_experts,
                            self.num_experts_per_tok, self.num_moe_layers, self.n_shared_experts, self.group_size,
                            self.routed_scaling_factor, self.swiglu_limit, self.activation_alpha)


def _addr(a) -> int:
    if a is None:
        return 0
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if hasattr(a, "data_ptr"):  # torch tensor (host or device)
        assert a.is_contiguous()
        return a.data_ptr()
    raise TypeError(f"cannot take the address of {type(a)}")


class KrasisEngine:
    def __init__(self, parallel: bool = True, num_threads: Optional[int] = None, skip_shared_experts: bool = False,
                 device: int = 0):
        # `parallel` / `num_threads` configure the reference's rayon pool (moe.rs:1497); the GPU path has no
        # host thread pool, the flags are kept for signature compatibility.
        self._lib = load_library()
        self._h = C.c_void_p()
        self._cfg: Optional[ModelConfig] = None
        self._parallel = parallel
        self._skip_shared = skip_shared_experts
        self._device = device
        self._cpu_bits = 4
        self._gpu_bits = 4
        self._pending = None
        self._routing_cfg = None
        self._has_gguf = False

    # ------------------------------------------------------------------ lifetime
    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._lib.kr_engine_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _need(self, msg="Model not loaded — call load() first"):
        if not self._h.value:
            raise RuntimeError(msg)

    def configure(self, cfg: ModelConfig, num_bits: int = 4) -> None:
        """Create the device engine for a model shape (what `load()` does after parsing config.json, moe.rs:1539)."""
        if self._h.value:
            self._lib.kr_engine_destroy(self._h)
            self._h = C.c_void_p()
        cc = cfg.to_c()
        check(self._lib.kr_engine_create(self._device, C.byref(cc), C.byref(self._h)))
        self._cfg = cfg
        self._cpu_bits = self._gpu_bits = num_bits

    def load(self, model_dir: str, group_size=None, max_layers=None, start_layer=None, num_bits=None,
             cpu_num_bits=None, gpu_num_bits=None, gguf_path=None, gguf_native: bool = False) -> None:
        """KrasisEngine.load (moe.rs:1538).  BF16 safetensors -> GPU-side quantize_int4/int8 -> HBM; with `gguf_path`: GGUF blocks
        de-quantized and re-quantized to INT4 / INT8-g128 (`gguf_native=False`, the reference's default) or kept native
        (`gguf_native=True`).  cpu_num_bits / gpu_num_bits collapse to ONE resident copy (there is no separate CPU store);
        when both are given the decode (cpu) precision wins, as it determines the numerics of `moe_forward`."""
        from .weight_store import load_from_gguf, load_from_hf
        if group_size not in (None, 128):
            raise ValueError(f"group_size {group_size} unsupported")
        bits = cpu_num_bits or num_bits or gpu_num_bits or 4
        if bits not in (4, 8):
            raise ValueError(f"cpu_num_bits must be 4 or 8, got {bits}")
        if gguf_path is not None:
            import os
            load_from_gguf(self, gguf_path, os.path.join(model_dir, "config.json"), gguf_native=gguf_native, max_layers=max_layers, start_layer=start_layer)
        else:
            load_from_hf(self, model_dir, bits, max_layers=max_layers, start_layer=start_layer)

    # ------------------------------------------------------------------ weights
    def load_unified_expert(self, layer: int, expert: int, w13, w13_scales, w2, w2_scales, num_bits: int = 4,
                            w2_bits: Optional[int] = None) -> None:
        """Upload one expert given in the reference CPU layout (UnifiedExpertWeights, weights/mod.rs:287);
        expert = -1 is the shared expert."""
        self._need()
        w2_bits = w2_bits or num_bits
        inter = w2.shape[0] * (8 if w2_bits == 4 else 1)
        for a in (w13, w13_scales, w2, w2_scales):
            assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"]
        check(self._lib.kr_upload_expert_unified(self._h, layer, expert, inter, _addr(w13), _addr(w13_scales), num_bits,
                                                 _addr(w2), _addr(w2_scales), w2_bits))

    def load_gguf_expert(self, layer: int, expert: int, gate: np.ndarray, up: np.ndarray, down: np.ndarray, gate_up_type: int,
                         down_type: int, inter: int) -> None:
        """Upload one expert as raw GGUF blocks (GgufExpertWeights, weights/mod.rs:252): gate/up [inter, hidden], down [hidden, inter],
        row-major blocks; expert = -1 is the shared expert.  This is the reference's `gguf_native=True` store."""
        self._need()
        for a in (gate, up, down):
            assert isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
        check(self._lib.kr_upload_expert_gguf(self._h, layer, expert, inter, _addr(gate), _addr(up), gate_up_type, _addr(down), down_type))
        self._has_gguf = True

    def fill_synthetic(self, bits: int = 4, seed: int = 0x12345678ABCDEF01, layers: Optional[Sequence[int]] = None) -> None:
        """Synthetic experts with bench_decode_synthetic's value distribution (decode.rs:4379-4392), generated on the GPU."""
        self._need()
        for layer in (layers if layers is not None else range(self._cfg.num_moe_layers)):
            check(self._lib.kr_fill_layer_synthetic(self._h, layer, bits, (seed + layer * 0x9E3779B97F4A7C15) & (2**64 - 1)))
        self._cpu_bits = self._gpu_bits = bits

    def fill_synthetic_gguf(self, gate_up_type: int = 12, down_type: int = 12, seed: int = 0x12345678ABCDEF01, layers: Optional[Sequence[int]] = None) -> None:
        """Synthetic native-GGUF experts (ggml type ids: Q4_K = 12, Q8_0 = 8) with the block distribution SURVEY 8d defines, generated on the GPU."""
        self._need()
        for layer in (layers if layers is not None else range(self._cfg.num_moe_layers)):
            check(self._lib.kr_fill_layer_synthetic_gguf(self._h, layer, gate_up_type, down_type, (seed + layer * 0x9E3779B97F4A7C15) & (2**64 - 1)))
        self._has_gguf = True

    def download_expert(self, layer: int, expert: int, bits: int = 4, w2_bits: Optional[int] = None):
        """Read an expert back in the reference layout (inverse re-tiling)."""
        self._need()
        w2_bits = w2_bits or bits
        c = self._cfg
        inter = c.moe_intermediate_size * (c.n_shared_experts if expert == -1 else 1)
        H = c.hidden_size
        w13 = np.empty((H // 8, 2 * inter), np.uint32) if bits == 4 else np.empty((H, 2 * inter), np.int8)
        w13s = np.empty((H // 128, 2 * inter), np.uint16)
        w2 = np.empty((inter // 8, H), np.uint32) if w2_bits == 4 else np.empty((inter, H), np.int8)
        w2s = np.empty((inter // 128, H), np.uint16)
        check(self._lib.kr_download_expert_unified(self._h, layer, expert, _addr(w13), _addr(w13s), _addr(w2), _addr(w2s)))
        return w13, w13s, w2, w2s

    # ------------------------------------------------------------------ forward (reference API)
    def moe_forward(self, moe_layer_idx: int, activation_bf16: bytes, expert_indices: List[int],
                    expert_weights: List[float]) -> bytes:
        """moe.rs:1775 -- single token; returns f32 bytes [hidden*4]."""
        self._need()
        H = self._cfg.hidden_size
        if len(activation_bf16) != H * 2:
            raise ValueError(f"Expected {H * 2} bytes (hidden_size={H} × 2), got {len(activation_bf16)}")
        if len(expert_indices) != len(expert_weights):
            raise ValueError(f"expert_indices len ({len(expert_indices)}) != expert_weights len ({len(expert_weights)})")
        act = np.frombuffer(activation_bf16, np.uint16).copy()
        ids = np.asarray(expert_indices, np.int32)
        w = np.asarray(expert_weights, np.float32)
        out = np.empty(H, np.float32)
        check(self._lib.kr_moe_forward(self._h, moe_layer_idx, _addr(act), _addr(ids), _addr(w), _addr(out), 1, len(ids),
                                       _lib.KR_OUT_F32, int(self._skip_shared), None))
        return out.tobytes()

    def forward_moe_direct(self, moe_layer_idx: int, activation_ptr: int, topk_ids_ptr: int, topk_weights_ptr: int,
                           output_ptr: int, batch_size: int, topk: int, stream: int = 0) -> None:
        """moe.rs:2843 -- raw pointers, bf16 in / bf16 out, ids -1 = skip."""
        self._need("Model not loaded")
        if topk > MAX_TOPK:
            raise ValueError(f"topk {topk} exceeds MAX_TOPK {MAX_TOPK}")
        check(self._lib.kr_moe_forward(self._h, moe_layer_idx, activation_ptr, topk_ids_ptr, topk_weights_ptr, output_ptr,
                                       batch_size, topk, _lib.KR_OUT_BF16, int(self._skip_shared), stream or None))

    def submit_forward(self, moe_layer_idx: int, activation_bf16: bytes, topk_ids_i32: bytes, topk_weights_f32: bytes,
                       batch_size: int) -> None:
        """moe.rs:2722 -- asynchronous in the reference (one in-flight job); here the launch is stream-async."""
        self._need("Model not loaded")
        H = self._cfg.hidden_size
        if len(activation_bf16) != batch_size * H * 2:
            raise ValueError(f"Expected {batch_size * H * 2} activation bytes, got {len(activation_bf16)}")
        topk = len(topk_ids_i32) // (4 * batch_size)
        if len(topk_weights_f32) != len(topk_ids_i32):
            raise ValueError("topk_ids / topk_weights size mismatch")
        act = np.frombuffer(activation_bf16, np.uint16).copy()
        ids = np.frombuffer(topk_ids_i32, np.int32).copy()
        w = np.frombuffer(topk_weights_f32, np.float32).copy()
        out = np.empty(batch_size * H, np.uint16)
        check(self._lib.kr_moe_forward(self._h, moe_layer_idx, _addr(act), _addr(ids), _addr(w), _addr(out), batch_size,
                                       topk, _lib.KR_OUT_BF16, int(self._skip_shared), None))
        self._pending = out

    def sync_forward(self) -> bytes:
        """moe.rs:2809 -- bf16 bytes [batch*hidden*2]."""
        if self._pending is None:
            raise RuntimeError("No forward submitted")
        out, self._pending = self._pending, None
        return out.tobytes()

    # ------------------------------------------------------------------ routing (moe.rs:2959-3246)
    def set_routing_config(self, scoring_func: str, norm_topk_prob: bool, topk: int, n_experts: int, hidden_size: int,
                           num_layers: int = 0) -> None:
        self._need("Model not loaded")
        code = {"sigmoid": _lib.KR_SCORE_SIGMOID, "softmax": _lib.KR_SCORE_SOFTMAX}.get(scoring_func)
        if code is None:
            raise ValueError(f"unknown scoring_func {scoring_func!r}")
        check(self._lib.kr_set_routing_config(self._h, code, int(norm_topk_prob), topk, n_experts, hidden_size))
        self._routing_cfg = (scoring_func, norm_topk_prob, topk, n_experts, hidden_size)

    def set_routing_weights(self, moe_layer_idx: int, gate_weight_bf16: bytes, correction_bias_f32: Optional[bytes] = None) -> None:
        self._need("Model not loaded")
        if self._routing_cfg is None:
            raise RuntimeError("Routing config not set")
        _, _, _, E, H = self._routing_cfg
        if len(gate_weight_bf16) != E * H * 2:
            raise ValueError(f"gate weight: expected {E * H * 2} bytes, got {len(gate_weight_bf16)}")
        g = np.frombuffer(gate_weight_bf16, np.uint16).copy()
        b = None if correction_bias_f32 is None else np.frombuffer(correction_bias_f32, np.float32).copy()
        check(self._lib.kr_set_routing_weights(self._h, moe_layer_idx, _addr(g), 0, None, _addr(b) or None))

    def forward_moe_routed(self, moe_layer_idx: int, activation_ptr: int, output_ptr: int, stream: int = 0) -> None:
        """moe.rs:3050 -- router + experts for one token; `stream` (extra): 0 = engine stream, 1 = legacy default stream, else a hipStream_t."""
        self._need("Model not loaded")
        check(self._lib.kr_forward_moe_routed(self._h, moe_layer_idx, activation_ptr, output_ptr, stream or None))

    # decode-graph router (CpuDecodeStore.store_route_weight + moe_route, decode.rs:895,1086)
    def set_route_weight_f32(self, moe_layer_idx: int, gate_f32: np.ndarray, bias: Optional[np.ndarray] = None,
                             e_score_corr: Optional[np.ndarray] = None) -> None:
        self._need("Model not loaded")
        g = np.ascontiguousarray(gate_f32, np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        c = None if e_score_corr is None else np.ascontiguousarray(e_score_corr, np.float32)
        check(self._lib.kr_set_routing_weights(self._h, moe_layer_idx, _addr(g), 1, _addr(b) or None, _addr(c) or None))

    def set_route_weight_synthetic(self, moe_layer_idx: int, seed: int = 0x12345678ABCDEF01, amp: float = 0.02, round_bf16: bool = True) -> None:
        """Router gate of bench_decode_synthetic (decode.rs:5181): xorshift64 uniform +-amp, optionally truncated to bf16 like a checkpoint."""
        self._need("Model not loaded")
        check(self._lib.kr_set_routing_weights_synthetic(self._h, moe_layer_idx, seed & (2**64 - 1), amp, int(round_bf16)))

    def route(self, moe_layer_idx: int, x, m: int, rule: int = _lib.KR_ROUTE_RULE_DECODE, want_logits: bool = False):
        """Returns (ids int32 [m,k], weights f32 [m,k][, logits f32 [m,E]]) as numpy arrays."""
        self._need("Model not loaded")
        _, _, k, E, _ = self._routing_cfg
        ids = np.empty((m, k), np.int32); w = np.empty((m, k), np.float32)
        lg = np.empty((m, E), np.float32) if want_logits else None
        check(self._lib.kr_route_topk(self._h, moe_layer_idx, _addr(x), m, rule, _addr(ids), _addr(w), _addr(lg) or None, None))
        return (ids, w, lg) if want_logits else (ids, w)

    # ------------------------------------------------------------------ collectives helper (moe.rs:2505)
    def reduce_sum_bf16(self, input_ptrs: Sequence[int], output_ptr: int, num_elements: int, stream: int = 0) -> None:
        if not input_ptrs:
            return
        self._need("Model not loaded")
        arr = (C.c_void_p * len(input_ptrs))(*input_ptrs)
        check(self._lib.kr_reduce_sum_bf16(self._h, arr, len(input_ptrs), output_ptr, num_elements, stream or None))

    def synchronize(self) -> None:
        self._need()
        check(self._lib.kr_synchronize(self._h))

    # ------------------------------------------------------------------ getters (moe.rs:1874-1965)
    def num_moe_layers(self) -> int:
        self._need("Model not loaded"); return self._cfg.num_moe_layers

    def hidden_size(self) -> int:
        self._need("Model not loaded"); return self._cfg.hidden_size

    def num_experts(self) -> int:
        self._need("Model not loaded"); return self._cfg.n_routed_experts

    def top_k(self) -> int:
        self._need("Model not loaded"); return self._cfg.num_experts_per_tok

    def group_size(self) -> int:
        self._need("Model not loaded"); return self._cfg.group_size

    def intermediate_size(self) -> int:
        self._need("Model not loaded"); return self._cfg.moe_intermediate_size

    def cpu_num_bits(self) -> int:
        self._need("Model not loaded"); return self._cpu_bits

    def gpu_num_bits(self) -> int:
        self._need("Model not loaded"); return self._gpu_bits

    def is_parallel(self) -> bool:
        return self._parallel

    def is_marlin_format(self) -> bool:
        self._need("Model not loaded"); return False  # no Marlin layout on CDNA

    def has_unified(self) -> bool:
        self._need("Model not loaded"); return True

    def has_gguf(self) -> bool:
        self._need("Model not loaded"); return self._has_gguf

    def marlin_w2_padded_n(self) -> int:
        """weights/mod.rs:942-949."""
        self._need("Model not loaded")
        h, i = self._cfg.hidden_size, self._cfg.moe_intermediate_size
        return h + 64 if (h == i and h % 256 != 0) else h

    def device_bytes(self) -> int:
        self._need(); return int(self._lib.kr_engine_device_bytes(self._h))

    # ------------------------------------------------------------------ weight export (moe.rs:1972-2709)
    # The reference hands its GPU-side (Marlin) expert tensors back to Python so that the prompt pass can DMA them to VRAM.  Here the experts are
    # already resident in HBM in the lane-tiled layout; these methods convert an expert back to the reference's Marlin GPU format on demand
    # (kr_download_expert_marlin: un-tile -> marlin_repack, word-for-word what the reference's cache holds) -- same names, argument meaning, byte
    # layouts, length checks and exception classes, so a caller of the PyO3 class finds them where it expects them.
    def _marlin_sizes(self, shared: bool = False):
        """bytes per expert of (w13_packed, w13_scales, w2_packed, w2_scales) in the reference's Marlin GPU format (weights/mod.rs:506-640)"""
        c = self._cfg
        H, gs = c.hidden_size, c.group_size if getattr(c, "group_size", 0) else 128
        inter = c.moe_intermediate_size * (c.n_shared_experts if shared else 1)
        bits = self._gpu_bits or 4
        n2 = H + 64 if (H == inter and H % 256 != 0) else H
        per_word = 8 if bits == 4 else 4                       # weights per u32
        return (2 * inter * H // per_word * 4, (H // gs) * 2 * inter * 2, n2 * inter // per_word * 4, (inter // gs) * n2 * 2)

    def _need_gpu_weights(self):
        self._need("Model not loaded")
        if self._has_gguf and not self._gpu_bits:
            raise RuntimeError("GPU weights not available")

    def _export_one(self, layer: int, expert: int, shared: bool = False):
        p13, s13, p2, s2 = self._marlin_sizes(shared)
        a, b, c, d = (np.empty(p13 // 4, np.uint32), np.empty(s13 // 2, np.uint16), np.empty(p2 // 4, np.uint32), np.empty(s2 // 2, np.uint16))
        check(self._lib.kr_download_expert_marlin(self._h, layer, -1 if shared else expert, _addr(a), _addr(b), _addr(c), _addr(d)))
        return a, b, c, d

    def _check_layer(self, moe_layer_idx: int):
        if not 0 <= moe_layer_idx < self._cfg.num_moe_layers:
            raise ValueError(f"moe_layer_idx {moe_layer_idx} out of range")

    def _get_range(self, moe_layer_idx: int, start, end, which: int) -> bytes:
        self._need_gpu_weights(); self._check_layer(moe_layer_idx)
        s = 0 if start is None else start
        e = self._cfg.n_routed_experts if end is None else end
        if s > e or e > self._cfg.n_routed_experts:
            raise ValueError(f"Invalid range [{s}, {e}), layer has {self._cfg.n_routed_experts} experts")
        return b"".join(self._export_one(moe_layer_idx, x)[which].tobytes() for x in range(s, e))

    def get_expert_w13_packed(self, moe_layer_idx: int, start=None, end=None) -> bytes:
        """moe.rs:1973 -- bytes of [end-start, K//16, 2N * bits/2] u32 (Marlin), K = hidden, N = 2 * intermediate"""
        return self._get_range(moe_layer_idx, start, end, 0)

    def get_expert_w13_scales(self, moe_layer_idx: int, start=None, end=None) -> bytes:
        return self._get_range(moe_layer_idx, start, end, 1)       # moe.rs:2005

    def get_expert_w2_packed(self, moe_layer_idx: int, start=None, end=None) -> bytes:
        return self._get_range(moe_layer_idx, start, end, 2)       # moe.rs:2037

    def get_expert_w2_scales(self, moe_layer_idx: int, start=None, end=None) -> bytes:
        return self._get_range(moe_layer_idx, start, end, 3)       # moe.rs:2069

    _WEIGHT_TYPES = {"w13_packed": 0, "w13_scales": 1, "w2_packed": 2, "w2_scales": 3}

    def get_experts_batch(self, moe_layer_idx: int, expert_ids: Sequence[int], weight_type: str) -> bytes:
        """moe.rs:2112 -- one weight type of the listed experts, contiguous"""
        self._need_gpu_weights(); self._check_layer(moe_layer_idx)
        if weight_type not in self._WEIGHT_TYPES:
            raise ValueError(f"Unknown weight_type: {weight_type}")
        for x in expert_ids:
            if not 0 <= x < self._cfg.n_routed_experts:
                raise ValueError(f"expert index {x} out of range")
        w = self._WEIGHT_TYPES[weight_type]
        return b"".join(self._export_one(moe_layer_idx, x)[w].tobytes() for x in expert_ids)

    def get_experts_all_batch(self, moe_layer_idx: int, expert_ids: Sequence[int]):
        """moe.rs:2202 -- (w13_packed, w13_scales, w2_packed, w2_scales) of the listed experts"""
        self._need_gpu_weights(); self._check_layer(moe_layer_idx)
        parts = [self._export_one(moe_layer_idx, x) for x in expert_ids]
        return tuple(b"".join(p[w].tobytes() for p in parts) for w in range(4))

    def _write_into(self, moe_layer_idx: int, ids: Sequence[int], bufs, names, shared: bool = False):
        per = self._marlin_sizes(shared)
        n = 1 if shared else len(ids)
        views = []
        for buf, nm, pe in zip(bufs, names, per):
            mv = memoryview(buf).cast("B")
            if len(mv) < n * pe:
                raise ValueError(f"{nm} too small: {len(mv)} < {n * pe}")            # moe.rs:2286-2300
            views.append(mv)
        for i, x in enumerate([0] if shared else ids):
            parts = self._export_one(moe_layer_idx, x, shared)
            for mv, part, pe in zip(views, parts, per):
                mv[i * pe:(i + 1) * pe] = part.tobytes()

    def write_experts_all_into(self, moe_layer_idx: int, w13p_buf, w13s_buf, w2p_buf, w2s_buf) -> None:
        """moe.rs:2259 -- every expert of the layer into caller-owned writable buffers (bytearray / numpy); ValueError when one is too small"""
        self._need_gpu_weights(); self._check_layer(moe_layer_idx)
        self._write_into(moe_layer_idx, range(self._cfg.n_routed_experts), (w13p_buf, w13s_buf, w2p_buf, w2s_buf), ("w13p_buf", "w13s_buf", "w2p_buf", "w2s_buf"))

    def write_experts_range_into(self, moe_layer_idx: int, start: int, end: int, w13p_buf, w13s_buf, w2p_buf, w2s_buf) -> None:
        """moe.rs:2346"""
        self._need_gpu_weights(); self._check_layer(moe_layer_idx)
        if start >= end or end > self._cfg.n_routed_experts:
            raise ValueError(f"Invalid range [{start}, {end}), layer has {self._cfg.n_routed_experts} experts")
        self._write_into(moe_layer_idx, range(start, end), (w13p_buf, w13s_buf, w2p_buf, w2s_buf), ("w13p_buf", "w13s_buf", "w2p_buf", "w2s_buf"))

    @staticmethod
    def _raw(ptr: int, n: int):
        import ctypes as C
        return (C.c_ubyte * n).from_address(ptr)

    def write_experts_range_into_pinned(self, moe_layer_idx: int, start: int, end: int, w13p_ptr: int, w13p_len: int, w13s_ptr: int, w13s_len: int,
                                        w2p_ptr: int, w2p_len: int, w2s_ptr: int, w2s_len: int) -> None:
        """moe.rs:2432 -- the same through raw (pinned) host addresses + lengths"""
        self._need_gpu_weights(); self._check_layer(moe_layer_idx)
        if start >= end or end > self._cfg.n_routed_experts:
            raise ValueError(f"Invalid range [{start}, {end}), layer has {self._cfg.n_routed_experts} experts")
        per = self._marlin_sizes(); n = end - start
        for nm, ln, pe in zip(("w13p", "w13s", "w2p", "w2s"), (w13p_len, w13s_len, w2p_len, w2s_len), per):
            if ln < n * pe:
                raise ValueError(f"{nm} buffer too small")                        # moe.rs:2455-2466
        bufs = (self._raw(w13p_ptr, n * per[0]), self._raw(w13s_ptr, n * per[1]), self._raw(w2p_ptr, n * per[2]), self._raw(w2s_ptr, n * per[3]))
        self._write_into(moe_layer_idx, range(start, end), bufs, ("w13p", "w13s", "w2p", "w2s"))

    def _need_shared(self):
        self._need("Model not loaded")
        if not self._cfg.n_shared_experts:
            raise RuntimeError("No shared experts")

    def get_shared_expert_weights(self, moe_layer_idx: int):
        """moe.rs:2675"""
        self._need_shared(); self._check_layer(moe_layer_idx)
        return tuple(p.tobytes() for p in self._export_one(moe_layer_idx, -1, shared=True))

    def write_shared_expert_into(self, moe_layer_idx: int, w13p_buf, w13s_buf, w2p_buf, w2s_buf) -> None:
        """moe.rs:2582"""
        self._need_shared(); self._check_layer(moe_layer_idx)
        self._write_into(moe_layer_idx, [0], (w13p_buf, w13s_buf, w2p_buf, w2s_buf), ("w13p_buf", "w13s_buf", "w2p_buf", "w2s_buf"), shared=True)

    def write_shared_expert_into_pinned(self, moe_layer_idx: int, w13p_ptr: int, w13p_len: int, w13s_ptr: int, w13s_len: int, w2p_ptr: int, w2p_len: int,
                                        w2s_ptr: int, w2s_len: int) -> None:
        """moe.rs:2631"""
        self._need_shared(); self._check_layer(moe_layer_idx)
        per = self._marlin_sizes(shared=True)
        for nm, ln, pe in zip(("w13p", "w13s", "w2p", "w2s"), (w13p_len, w13s_len, w2p_len, w2s_len), per):
            if ln < pe:
                raise ValueError(f"{nm} buffer too small")
        bufs = (self._raw(w13p_ptr, per[0]), self._raw(w13s_ptr, per[1]), self._raw(w2p_ptr, per[2]), self._raw(w2s_ptr, per[3]))
        self._write_into(moe_layer_idx, [0], bufs, ("w13p", "w13s", "w2p", "w2s"), shared=True)
