"""CpuDecodeStore -- host mirror of the reference's decode-graph class (src/decode.rs:193-3602) over the HIP C ABI.

The class keeps the reference's NAME and builder methods so `decode_setup.py`-style callers are source compatible; the
graph it builds runs on the MI355X (all weights / KV / recurrent state in HBM, one hipGraph replay per token).
Pointer arguments are integer HOST addresses exactly like the reference (decode.rs:280-286).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import check, load_library
from .engine import KrasisEngine, _addr


class CpuDecodeStore:
    def __init__(self, group_size: int = 128, parallel: bool = True, norm_bias_one: bool = False, device: Optional[int] = None):
        # Like the reference (decode.rs:229) the store exists BEFORE an engine is bound: weights, norms and router gates can be stored first and
        # set_moe_store(engine) may come last (decode_setup.py:1010) -- or first, as the synthetic benchmark does.  Until then the store runs on
        # the current HIP device with a bare engine inside the library.
        self._lib = load_library()
        self._group_size = group_size
        self._norm_bias_one = norm_bias_one
        self._h = C.c_void_p()
        if device is None:
            check(self._lib.kr_decode_create(None, group_size, int(norm_bias_one), C.byref(self._h)))
        else:     # a named device: the store must live where the engine it will be bound to lives
            check(self._lib.kr_decode_create_on(int(device), group_size, int(norm_bias_one), C.byref(self._h)))
        self._engine: Optional[KrasisEngine] = None
        self._vocab = 0
        self._n_layers = 0
        self._n_weights = 0
        self._routes: list = []          # route_id -> (gate f32 [E,H], bias | None, e_score_corr | None)   (store_route_weight)
        self._route_layer: dict = {}     # route_id -> engine MoE layer (set_decode_layer_moe)
        self._route_cfg = None           # (scoring code, norm_topk_prob, topk) from configure_decode
        self._routes_pushed: set = set()

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._lib.kr_decode_destroy(self._h); self._h = C.c_void_p()
        except Exception:
            pass

    # set_moe_store (decode.rs:2250): the engine owns the routed experts and the routers; bound at any point of the build
    def set_moe_store(self, engine: KrasisEngine) -> None:
        engine._need("Model not loaded")
        if self._engine is not None and self._engine is not engine:
            raise RuntimeError("MoE store already set")
        check(self._lib.kr_decode_set_moe_store(self._h, engine._h))
        self._engine = engine
        self._push_routes()

    def _push_routes(self) -> None:
        """router gates handed to store_route_weight live in the engine's router store (one per MoE layer): pushed once the engine, the routing
        configuration (configure_decode) and the route_id -> moe_layer_idx mapping (set_decode_layer_moe) are all known"""
        eng = self._engine
        if eng is None or self._route_cfg is None:
            return
        for rid, layer in self._route_layer.items():
            if rid in self._routes_pushed or rid >= len(self._routes):
                continue
            gate, bias, esc = self._routes[rid]
            E, H = gate.shape
            if eng._routing_cfg is None:
                sf, norm, topk = self._route_cfg
                check(self._lib.kr_set_routing_config(eng._h, sf, int(norm), topk, E, H))
                eng._routing_cfg = ({0: "sigmoid", 1: "softmax", 2: "swiglu"}[sf], norm, topk, E, H)
            eng.set_route_weight_f32(layer, gate, bias, esc)
            self._routes_pushed.add(rid)

    def _need(self):
        if not self._h.value:
            raise RuntimeError("decode store was destroyed")

    # ------------------------------------------------------------------ weights
    def store_weight_f32(self, data_ptr: int, rows: int, cols: int, num_bits: int = 4) -> int:
        self._need()
        if num_bits not in (4, 8):
            raise ValueError(f"num_bits must be 4 or 8, got {num_bits}")
        wid = C.c_int()
        check(self._lib.kr_decode_store_weight_f32(self._h, data_ptr, rows, cols, num_bits, C.byref(wid)))
        self._n_weights += 1
        return wid.value

    def store_weight_synthetic(self, rows: int, cols: int, num_bits: int = 4, seed: int = 1) -> int:
        self._need()
        wid = C.c_int()
        check(self._lib.kr_decode_store_weight_synthetic(self._h, rows, cols, num_bits, seed, C.byref(wid)))
        self._n_weights += 1
        return wid.value

    def download_weight(self, wid: int, rows: int, cols: int, num_bits: int = 4):
        self._need()
        packed = np.empty((cols // 8, rows), np.uint32) if num_bits == 4 else np.empty((cols, rows), np.int8)
        scales = np.empty((cols // 128, rows), np.uint16)
        check(self._lib.kr_decode_download_weight(self._h, wid, _addr(packed), _addr(scales)))
        return packed, scales

    def store_norm_weight(self, data_ptr: int, size: int) -> int:
        self._need()
        nid = C.c_int()
        check(self._lib.kr_decode_store_norm_weight(self._h, data_ptr, size, C.byref(nid)))
        return nid.value

    def store_route_weight(self, data_ptr: int, num_experts: int, hidden_dim: int, bias_ptr: Optional[int] = None, bias_len: int = 0,
                           e_score_corr_ptr: Optional[int] = None, e_score_corr_len: int = 0) -> int:
        """decode.rs:895 -- f32 gate [E, H] (+ optional bias / e_score_correction [E]) -> route_id.  The data is copied (the caller may free
        its tensors, decode_setup.py:563-567)."""
        self._need()
        if not data_ptr:
            raise ValueError("null gate pointer")
        rd = lambda p, n: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n,)).copy()
        gate = rd(data_ptr, num_experts * hidden_dim).reshape(num_experts, hidden_dim)
        bias = rd(bias_ptr, bias_len) if bias_ptr and bias_len else None
        esc = rd(e_score_corr_ptr, e_score_corr_len) if e_score_corr_ptr and e_score_corr_len else None
        self._routes.append((gate, bias, esc))
        # the device copy serves the stand-alone moe_route (decode.rs:955); the engine's routers get theirs through _push_routes
        rid = C.c_int()
        check(self._lib.kr_decode_store_route_weight(self._h, _addr(gate), num_experts, hidden_dim, _addr(bias) if bias is not None else None,
                                                     _addr(esc) if esc is not None else None, C.byref(rid)))
        assert rid.value == len(self._routes) - 1
        return rid.value

    def num_weights(self) -> int:
        return self._n_weights

    def num_route_weights(self) -> int:
        return len(self._routes)

    def total_bytes(self) -> int:
        return self.device_bytes()

    def weight_bytes(self, weight_id: int) -> int:
        """decode.rs:1107 -- packed words * 4 + bf16 scales * 2 of one stored weight"""
        return int(self._lib.kr_decode_weight_bytes(self._h, weight_id))

    def self_addr(self) -> int:
        """decode.rs:269 -- raw address of the store for GIL-free callers: here the kr_decode_store* handle of the C ABI"""
        return int(self._h.value or 0)

    def consolidate_weights_mmap(self) -> None:
        """decode.rs:2341 consolidates the CPU engine's weights into one NUMA-interleaved mmap; HBM allocations need no such step."""

    # ------------------------------------------------------------------ stand-alone operators (decode.rs:328-1086)
    # Pointers are plain addresses like the reference's (tensor.data_ptr()); host AND device addresses are accepted.  The arithmetic runs on
    # the GPU (csrc/kr_decode_standalone.cpp); results are bit-identical to the reference methods.
    def matmul(self, weight_id: int, input_ptr: int, output_ptr: int) -> None:
        check(self._lib.kr_decode_matmul(self._h, weight_id, input_ptr, output_ptr))

    def matmul_batch(self, weight_ids: Sequence[int], input_ptr: int, output_ptrs: Sequence[int]) -> None:
        if len(weight_ids) != len(output_ptrs):
            raise ValueError("weight_ids and output_ptrs must have same length")          # decode.rs:370
        n = len(weight_ids)
        ids = (C.c_int * max(n, 1))(*weight_ids); outs = (C.c_void_p * max(n, 1))(*output_ptrs)
        check(self._lib.kr_decode_matmul_batch(self._h, ids, n, input_ptr, outs))

    def fused_add_rmsnorm(self, hidden_ptr: int, residual_ptr: int, weight_ptr: int, eps: float, size: int, first_call: bool) -> None:
        if not weight_ptr:
            raise ValueError("null weight pointer")
        check(self._lib.kr_decode_fused_add_rmsnorm(self._h, hidden_ptr, residual_ptr, weight_ptr, -1, eps, size, int(first_call)))

    def fused_add_rmsnorm_id(self, hidden_ptr: int, residual_ptr: int, norm_id: int, eps: float, size: int, first_call: bool) -> None:
        check(self._lib.kr_decode_fused_add_rmsnorm(self._h, hidden_ptr, residual_ptr, None, norm_id, eps, size, int(first_call)))

    def rmsnorm(self, input_ptr: int, weight_ptr: int, eps: float, output_ptr: int, size: int) -> None:
        check(self._lib.kr_decode_rmsnorm(self._h, input_ptr, weight_ptr, eps, output_ptr, size))

    def silu_mul(self, gate_ptr: int, up_ptr: int, output_ptr: int, size: int) -> None:
        check(self._lib.kr_decode_silu_mul(self._h, gate_ptr, up_ptr, output_ptr, size))

    def fused_shared_expert(self, gate_up_wid: int, down_wid: int, input_ptr: int, output_ptr: int) -> None:
        check(self._lib.kr_decode_fused_shared_expert(self._h, gate_up_wid, down_wid, input_ptr, output_ptr))

    def linear_attention_recurrent(self, state_ptr: int, q_ptr: int, k_ptr: int, v_ptr: int, g_ptr: int, beta_ptr: int, output_ptr: int,
                                   nv: int, dk: int, dv: int) -> None:
        check(self._lib.kr_decode_linear_attention_recurrent(self._h, state_ptr, q_ptr, k_ptr, v_ptr, g_ptr, beta_ptr, output_ptr, nv, dk, dv))

    def gated_rmsnorm_silu(self, x_ptr: int, z_ptr: int, norm_weight_ptr: int, output_ptr: int, eps: float, nv: int, dv: int) -> None:
        check(self._lib.kr_decode_gated_rmsnorm_silu(self._h, x_ptr, z_ptr, norm_weight_ptr, output_ptr, eps, nv, dv))

    def linear_attention_conv(self, qkvz_ptr: int, ba_ptr: int, conv_state_ptr: int, conv_weight_ptr: int, a_log_ptr: int, dt_bias_ptr: int, scale: float,
                              q_out_ptr: int, k_out_ptr: int, v_out_ptr: int, z_out_ptr: int, g_out_ptr: int, beta_out_ptr: int,
                              nk: int, nv: int, dk: int, dv: int, hr: int, kernel_dim: int) -> None:
        check(self._lib.kr_decode_linear_attention_conv(self._h, qkvz_ptr, ba_ptr, conv_state_ptr, conv_weight_ptr, a_log_ptr, dt_bias_ptr, scale,
                                                        q_out_ptr, k_out_ptr, v_out_ptr, z_out_ptr, g_out_ptr, beta_out_ptr, nk, nv, dk, dv, hr, kernel_dim))

    def moe_route(self, route_id: int, hidden_ptr: int, topk_ids_out_ptr: int, topk_weights_out_ptr: int, topk: int, scoring_func: int,
                  norm_topk_prob: bool) -> None:
        check(self._lib.kr_decode_moe_route(self._h, route_id, hidden_ptr, topk_ids_out_ptr, topk_weights_out_ptr, topk, scoring_func, int(norm_topk_prob)))

    # ------------------------------------------------------------------ cancellation / timing (decode.rs:253-265)
    @property
    def last_decode_elapsed_s(self) -> float:
        return float(self._lib.kr_decode_last_elapsed_s(self._h))

    def cancel(self) -> None:
        check(self._lib.kr_decode_cancel(self._h))

    def reset_cancel(self) -> None:
        check(self._lib.kr_decode_reset_cancel(self._h))

    def repack_to_tiled(self) -> None:
        """decode.rs repack_to_tiled: the CPU engine re-tiles its weights for bandwidth; the HBM layout is already lane-tiled."""

    # ------------------------------------------------------------------ graph
    def configure_decode(self, hidden_size: int, num_layers: int, eps: float, final_norm_id: int, lm_head_wid: int, vocab_size: int,
                         topk: int, scoring_func: int, norm_topk_prob: bool, routed_scaling_factor: float, embedding_ptr: int,
                         synth_seed: int = 0) -> None:
        self._need()
        check(self._lib.kr_decode_configure(self._h, hidden_size, num_layers, eps, final_norm_id, lm_head_wid, vocab_size, topk,
                                            scoring_func, int(norm_topk_prob), routed_scaling_factor, embedding_ptr or None, synth_seed))
        self._vocab, self._n_layers = vocab_size, num_layers
        self._route_cfg = (scoring_func, bool(norm_topk_prob), topk)

    def add_decode_la_layer(self, input_norm_id, post_attn_norm_id, in_proj_qkvz_wid, in_proj_ba_wid, out_proj_wid, conv_weight_ptr,
                            a_log_ptr, dt_bias_ptr, norm_weight_ptr, nk, nv, dk, dv, hr, kernel_dim, scale) -> None:
        """decode.rs:2036 (same positional arguments; hr = value heads per key head)"""
        self._need()
        if hr * nk != nv:
            raise ValueError(f"head ratio {hr} does not match {nv} value / {nk} key heads")
        check(self._lib.kr_decode_add_la_layer(self._h, input_norm_id, post_attn_norm_id, in_proj_qkvz_wid, in_proj_ba_wid, out_proj_wid,
                                               conv_weight_ptr, a_log_ptr, dt_bias_ptr, norm_weight_ptr, nk, nv, dk, dv, kernel_dim, scale))

    def add_decode_gqa_layer(self, input_norm_id, post_attn_norm_id, q_proj_wid, k_proj_wid, v_proj_wid, o_proj_wid, q_norm_ptr, q_norm_len,
                             k_norm_ptr, k_norm_len, gated, num_heads, num_kv_heads, head_dim, sm_scale) -> None:
        self._need()
        check(self._lib.kr_decode_add_gqa_layer(self._h, input_norm_id, post_attn_norm_id, q_proj_wid, k_proj_wid, v_proj_wid, o_proj_wid,
                                                q_norm_ptr or None, q_norm_len, k_norm_ptr or None, k_norm_len, int(gated), num_heads,
                                                num_kv_heads, head_dim, sm_scale))

    def add_decode_mla_layer(self, input_norm_id, post_attn_norm_id, kv_a_proj_wid, o_proj_wid, q_proj_wid, q_a_proj_wid, q_b_proj_wid,
                             w_kc_ptr, w_kc_len, w_vc_ptr, w_vc_len, kv_a_norm_ptr, kv_a_norm_len, q_a_norm_ptr, q_a_norm_len, rope_cos_ptr,
                             rope_sin_ptr, rope_len, rope_max_seq, num_heads, kv_lora_rank, qk_nope_dim, qk_rope_dim, v_head_dim, sm_scale) -> None:
        """decode.rs:2131 (same positional arguments; None for an absent projection id)."""
        self._need()
        f = lambda v: -1 if v is None else v
        check(self._lib.kr_decode_add_mla_layer(self._h, input_norm_id, post_attn_norm_id, kv_a_proj_wid, o_proj_wid, f(q_proj_wid),
                                                f(q_a_proj_wid), f(q_b_proj_wid), w_kc_ptr, w_kc_len, w_vc_ptr, w_vc_len, kv_a_norm_ptr,
                                                kv_a_norm_len, q_a_norm_ptr or None, q_a_norm_len, rope_cos_ptr, rope_sin_ptr, rope_max_seq,
                                                num_heads, kv_lora_rank, qk_nope_dim, qk_rope_dim, v_head_dim, sm_scale))

    def set_decode_layer_moe(self, layer_idx: int, route_id: int, moe_layer_idx: int, shared_gate_up_wid: Optional[int] = None,
                             shared_down_wid: Optional[int] = None, shared_gate_wid: Optional[int] = None) -> None:
        self._need()
        f = lambda v: -1 if v is None else v
        check(self._lib.kr_decode_set_layer_moe(self._h, layer_idx, moe_layer_idx, f(shared_gate_up_wid), f(shared_down_wid), f(shared_gate_wid)))
        self._route_layer[route_id] = moe_layer_idx
        self._push_routes()

    def set_decode_layer_dense(self, layer_idx: int, gate_proj_wid: int, up_proj_wid: int, down_proj_wid: int) -> None:
        self._need()
        check(self._lib.kr_decode_set_layer_dense(self._h, layer_idx, gate_proj_wid, up_proj_wid, down_proj_wid))

    def set_decode_rope(self, cos_ptr: int, sin_ptr: int, half_dim: int, max_seq: int) -> None:
        self._need()
        check(self._lib.kr_decode_set_rope(self._h, cos_ptr, sin_ptr, half_dim, max_seq))

    def set_kv_dtype(self, fp8_e4m3: bool) -> None:
        """GQA KV cache element type: FP16 (reference CPU decode, default) or FP8-E4M3 (reference GPU cache, kv_cache.py:38)."""
        self._need(); check(self._lib.kr_decode_set_kv_dtype(self._h, 1 if fp8_e4m3 else 0))
        self._kv_fp8 = bool(fp8_e4m3)

    def set_attention_mode(self, fast: bool, gemm_fast: bool = False, decode_fast: bool = False) -> None:
        """fast False (default): the reference's sequential softmax / p.v order (bit-exact).  True: split-KV / flash attention and the chunked
        delta rule -- tolerance mode (logits within ~1e-4 relative).  gemm_fast True: the GEMMs of the prompt pass in the tolerance form as well
        (f16 activations, f32 accumulation over the whole k range: the dataflow of the reference's GPU prompt pass); decode steps are unaffected.
        decode_fast True (KR_DECODE_FAST): decode steps on the tolerance-mode kernels -- the reference's products, tree reductions instead of its
        sequential chains, norms / top-k / activation / combine folded into the matvec launches; the prompt pass is unaffected."""
        self._need(); check(self._lib.kr_decode_set_attention_mode(self._h, (1 if fast else 0) | (2 if gemm_fast else 0) | (4 if decode_fast else 0)))

    def set_option(self, name: str, value: int) -> None:
        """test / tuning hooks by name ("gqa_stream", "pfm_timing")"""
        self._need(); check(self._lib.kr_decode_set_option(self._h, name.encode(), int(value)))

    def finalize_decode(self) -> None:
        self._need()
        self._push_routes()
        check(self._lib.kr_decode_finalize(self._h))

    def set_decode_state(self, seq_len: int, kv_max_seq: int, kv_k_ptrs: Sequence[int], kv_v_ptrs: Sequence[int],
                         conv_state_ptrs: Sequence[int], recur_state_ptrs: Sequence[int], mla_ckv_ptrs=None, mla_kpe_ptrs=None) -> None:
        self._need()
        n = self._n_layers
        mk = lambda xs: (C.c_void_p * n)(*[(x or None) for x in xs])
        if mla_ckv_ptrs is not None:   # MLA layers keep their caches in the kv_k / kv_v slots of the C ABI
            kv_k_ptrs = [a or b for a, b in zip(list(kv_k_ptrs) + [0] * n, mla_ckv_ptrs)][:n]
            kv_v_ptrs = [a or b for a, b in zip(list(kv_v_ptrs) + [0] * n, mla_kpe_ptrs)][:n]
        check(self._lib.kr_decode_set_state(self._h, seq_len, kv_max_seq, mk(kv_k_ptrs), mk(kv_v_ptrs), mk(conv_state_ptrs), mk(recur_state_ptrs)))

    def fill_state_synthetic(self, kv_max_seq: int, seed: int = 7) -> None:
        self._need()
        check(self._lib.kr_decode_fill_state_synthetic(self._h, kv_max_seq, seed))

    def get_decode_state(self, layer: int, kv_k=None, kv_v=None, conv_state=None, recur_state=None) -> None:
        self._need()
        check(self._lib.kr_decode_get_state(self._h, layer, _addr(kv_k) or None, _addr(kv_v) or None, _addr(conv_state) or None,
                                            _addr(recur_state) or None))

    # ------------------------------------------------------------------ run
    def decode_step(self, token_id: int, position: int, output_ptr: int = 0, stream: int = 0) -> None:
        self._need()
        check(self._lib.kr_decode_step(self._h, token_id, position, output_ptr or None, stream or None))

    def prefill(self, tokens: Sequence[int], start_pos: int = 0, output_ptr: int = 0, stream: int = 0) -> int:
        """Whole-model prompt pass on the GPU (kr_decode_prefill): equivalent to calling decode_step for every prompt token, which is what
        replaces model.server_prefill + CpuDecoder.prepare (decode_setup.py:232-278).  Returns the greedy sample of the last position."""
        self._need()
        arr = (C.c_int32 * len(tokens))(*tokens)
        check(self._lib.kr_decode_prefill(self._h, arr, len(tokens), start_pos, output_ptr or None, stream or None))
        return self.last_token()

    def prefill_nll(self, tokens: Sequence[int], start_pos: int = 0, output_ptr: int = 0, stream: int = 0) -> np.ndarray:
        """Scoring prompt pass (kr_decode_prefill_nll): the prompt pass plus, per position i < n-1, the next-token negative log-likelihood
        -log softmax(logits_i)[tokens[i+1]] -- model.forward(return_all_logits=True) + cross_entropy(reduction="none") of the perplexity
        harness (perplexity/measure_ppl.py:212-227) without materialising [n, vocab].  Returns f32 [n-1]."""
        self._need()
        if len(tokens) < 2:
            raise ValueError(f"Need at least 2 tokens, got {len(tokens)}")
        arr = (C.c_int32 * len(tokens))(*tokens)
        nll = np.empty(len(tokens) - 1, np.float32)
        check(self._lib.kr_decode_prefill_nll(self._h, arr, len(tokens), start_pos, nll.ctypes.data, output_ptr or None, stream or None))
        return nll

    def reset_decode_state(self, kv_max_seq: int) -> None:
        """fresh request: zeroed KV / latent caches, conv and recurrent states (measure_ppl.py:199-206)"""
        self._need()
        check(self._lib.kr_decode_reset_state(self._h, kv_max_seq))

    def set_prefill_chunk(self, chunk: int) -> None:
        """Tokens per chunk of the prompt pass (0 = default); chunks alternate between two streams."""
        self._need(); check(self._lib.kr_decode_set_prefill_chunk(self._h, chunk))

    def set_prefill_depth(self, depth: int) -> None:
        """Chunks of the prompt pass in flight (streams / scratch arenas), 1..8; 0 = default."""
        self._need(); check(self._lib.kr_decode_set_prefill_depth(self._h, depth))

    def generate_batch(self, first_token: int, start_pos: int, max_tokens: int, temperature: float = 0.0, top_k: int = 0, top_p: float = 1.0,
                       stop_ids: Sequence[int] = (), presence_penalty: float = 0.0, rng_seed: int = 0) -> List[int]:
        """decode.rs:3525 -- decode loop + sampler on the GPU; `rng_seed` (extra, 0 = wall clock like the reference) makes draws reproducible."""
        self._need()
        out = (C.c_int * max(max_tokens, 1))(); n = C.c_int()
        stops = (C.c_int * max(len(stop_ids), 1))(*stop_ids)
        check(self._lib.kr_decode_generate(self._h, first_token, start_pos, max_tokens, temperature, top_k, top_p, stops, len(stop_ids),
                                           presence_penalty, rng_seed, out, C.byref(n), None))
        return list(out[: n.value])

    def generate_stream(self, first_token: int, start_position: int, max_tokens: int, temperature: float, top_k: int, top_p: float,
                        stop_ids: Sequence[int], tokenizer, presence_penalty: float, on_token, rng_seed: int = 0) -> int:
        """decode.rs:3611 -- the cancellable loop of the reference's Rust server.  on_token(token_id, text, finish_reason) -> bool (False cancels);
        finish_reason is None, "stop", "length" or "cancelled".  `tokenizer` is anything with decode(ids, skip_special_tokens=...) (tokenizers /
        transformers) or None (text = "").  Returns the number of generated tokens."""
        self._need()
        from ._lib import TOKEN_CB
        reasons = {0: None, 1: "stop", 2: "length", 3: "cancelled"}
        err: list = []

        def cb(token, reason, _user):
            try:
                text = ""
                if tokenizer is not None and reason != 3:
                    try:
                        text = tokenizer.decode([int(token)], skip_special_tokens=True)
                    except TypeError:
                        text = tokenizer.decode([int(token)])
                return 1 if on_token(int(token), text or "", reasons[reason]) else 0
            except BaseException as e:        # an exception must not unwind through the C frame
                err.append(e)
                return 0

        cfn = TOKEN_CB(cb)
        n = C.c_int()
        stops = (C.c_int * max(len(stop_ids), 1))(*stop_ids)
        check(self._lib.kr_decode_generate_stream(self._h, first_token, start_position, max_tokens, temperature, top_k, top_p, stops, len(stop_ids),
                                                  presence_penalty, rng_seed, cfn, None, C.byref(n), None))
        if err:
            raise err[0]
        return n.value

    def sample(self, temperature: float, top_k: int = 0, top_p: float = 1.0, presence_penalty: float = 0.0, rng_seed: int = 0, reset_seen: bool = False) -> int:
        """sample_from_logits (decode.rs:3718) on the logits of the last decode_step / prefill."""
        self._need()
        t = C.c_int()
        check(self._lib.kr_decode_sample(self._h, temperature, top_k, top_p, presence_penalty, rng_seed, int(reset_seen), C.byref(t), None))
        return t.value

    def last_token(self) -> int:
        t = C.c_int(); check(self._lib.kr_decode_last_token(self._h, C.byref(t))); return t.value

    def set_use_graph(self, enable: bool) -> None:
        self._need(); check(self._lib.kr_decode_set_use_graph(self._h, int(enable)))

    def read_hidden(self, n: int) -> np.ndarray:
        out = np.empty(n, np.float32); check(self._lib.kr_decode_read_buffer(self._h, 0, _addr(out), n)); return out

    def read_router(self, n_experts: int, topk: int):
        """(logits [E], ids [k], weights [k]) the router of the LAST MoE layer produced in the most recent decode step (test / debug aid)"""
        lg = np.empty(n_experts, np.float32); ids = np.empty(topk, np.int32); w = np.empty(topk, np.float32)
        check(self._lib.kr_decode_read_buffer(self._h, 4, _addr(lg), n_experts))
        check(self._lib.kr_decode_read_buffer(self._h, 2, _addr(ids), topk))
        check(self._lib.kr_decode_read_buffer(self._h, 3, _addr(w), topk))
        return lg, ids, w

    def device_bytes(self) -> int:
        self._need(); return int(self._lib.kr_decode_device_bytes(self._h))

    def profile_step(self, token_id: int, position: int):
        """one un-graphed decode step with HIP events around every launch -> [(ms, launches)] per kernel kind (include/krasis_hip.h lists the kinds);
        the stand-in for KRASIS_CPU_DECODE_TIMING's per-op buckets (decode.rs:3477-3517)"""
        self._need()
        ms = (C.c_double * 16)(); cnt = (C.c_long * 16)()
        check(self._lib.kr_decode_profile_step(self._h, token_id, position, ms, cnt, 16))
        return [(ms[i], cnt[i]) for i in range(16)]
