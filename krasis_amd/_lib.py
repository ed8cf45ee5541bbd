"""ctypes loader for libkrasis_hip.so (the C ABI declared in include/krasis_hip.h)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("KRASIS_HIP_LIB") or os.path.join(_HERE, "libkrasis_hip.so")   # KRASIS_HIP_LIB: probe builds of the same library (tools/probes), never a fallback

KR_OK, KR_ERR_STATE, KR_ERR_VALUE, KR_ERR_IO, KR_ERR_HIP = 0, 1, 2, 3, 4
KR_OUT_F32, KR_OUT_BF16 = 0, 1
KR_SCORE_SIGMOID, KR_SCORE_SOFTMAX, KR_SCORE_TOPK_SOFTMAX = 0, 1, 2
KR_ROUTE_RULE_ENGINE, KR_ROUTE_RULE_DECODE = 0, 1

# every symbol include/krasis_hip.h declares (checked by tests/test_abi.py without a GPU)
SYMBOLS = [
    "kr_last_error", "kr_version", "kr_alloc_count_total", "kr_engine_create", "kr_engine_destroy", "kr_engine_get_config",
    "kr_engine_device_bytes", "kr_upload_expert_unified", "kr_upload_expert_bf16", "kr_upload_expert_gguf", "kr_fill_layer_synthetic", "kr_fill_layer_synthetic_gguf",
    "kr_download_expert_unified", "kr_marlin_repack", "kr_marlin_unpack", "kr_upload_expert_marlin", "kr_download_expert_marlin", "kr_moe_forward", "kr_moe_prefill", "kr_set_routing_config", "kr_set_routing_weights", "kr_set_routing_weights_synthetic",
    "kr_route_topk", "kr_forward_moe_routed", "kr_reduce_sum_bf16", "kr_combine_rows", "kr_ep_unique_id", "kr_ep_init", "kr_ep_destroy", "kr_moe_prefill_ep", "kr_ep_comm_ranks", "kr_ep_max_int", "kr_ep_allreduce_f32", "kr_ep_loopback_create", "kr_ep_loopback_destroy", "kr_ep_init_loopback", "kr_moe_set_prefill_pairs", "kr_moe_set_gemm_mode", "kr_synchronize", "kr_set_profiling",
    "kr_get_profile", "kr_decode_create", "kr_decode_set_moe_store", "kr_decode_destroy", "kr_decode_store_weight_f32", "kr_decode_store_weight_synthetic",
    "kr_decode_download_weight", "kr_decode_store_norm_weight", "kr_decode_configure", "kr_decode_add_la_layer",
    "kr_decode_add_gqa_layer", "kr_decode_add_mla_layer", "kr_decode_prefill", "kr_decode_prefill_nll", "kr_decode_reset_state", "kr_decode_generate", "kr_decode_sample", "kr_sample_order", "kr_decode_set_prefill_chunk", "kr_decode_set_prefill_depth", "kr_decode_set_layer_moe", "kr_decode_set_layer_dense", "kr_decode_set_rope", "kr_decode_finalize", "kr_decode_set_kv_dtype", "kr_decode_set_attention_mode", "kr_decode_set_option", "kr_decode_create_on",
    "kr_decode_set_state", "kr_decode_fill_state_synthetic", "kr_decode_get_state", "kr_decode_step", "kr_decode_generate_greedy",
    "kr_decode_last_token", "kr_decode_set_use_graph", "kr_decode_read_buffer", "kr_decode_device_bytes", "kr_decode_profile_step",
    "kr_decode_generate_stream", "kr_decode_cancel", "kr_decode_reset_cancel", "kr_decode_last_elapsed_s", "kr_decode_matmul", "kr_decode_matmul_batch",
    "kr_decode_fused_add_rmsnorm", "kr_decode_rmsnorm", "kr_decode_silu_mul", "kr_decode_fused_shared_expert", "kr_decode_linear_attention_recurrent",
    "kr_decode_gated_rmsnorm_silu", "kr_decode_linear_attention_conv", "kr_decode_store_route_weight", "kr_decode_moe_route", "kr_decode_num_route_weights",
    "kr_decode_weight_bytes",
]
TOKEN_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_void_p)      # kr_token_cb(token, finish_reason, user) -> continue?


class KrasisHipError(RuntimeError):
    """HIP runtime failure or missing native library (no CPU fallback exists)."""


class ModelConfigC(C.Structure):
    _fields_ = [("hidden_size", C.c_int), ("moe_intermediate_size", C.c_int), ("n_routed_experts", C.c_int),
                ("num_experts_per_tok", C.c_int), ("num_moe_layers", C.c_int), ("n_shared_experts", C.c_int),
                ("group_size", C.c_int), ("routed_scaling_factor", C.c_float), ("swiglu_limit", C.c_float),
                ("activation_alpha", C.c_float)]


_lib = None


def lib_path() -> str:
    return _LIB


def load_library() -> C.CDLL:
    """Load libkrasis_hip.so; raises KrasisHipError if it was not built (build with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise KrasisHipError(f"{_LIB} is missing: build it with `make -C krasis_amd/csrc` (hipcc, gfx950). "
                             "krasis_amd has no CPU fallback.")
    # PyTorch-ROCm ships its own libamdhip64; import it first so this library binds to the SAME HIP runtime
    # (one context, torch device pointers valid here).  Loading the system runtime next to torch's breaks both.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - C/C++ consumers link the system runtime directly
        pass
    lib = C.CDLL(_LIB)
    lib.kr_last_error.restype = C.c_char_p
    lib.kr_alloc_count_total.restype = C.c_long; lib.kr_alloc_count_total.argtypes = []
    lib.kr_engine_device_bytes.restype = C.c_size_t
    lib.kr_engine_device_bytes.argtypes = [C.c_void_p]
    lib.kr_engine_create.argtypes = [C.c_int, C.POINTER(ModelConfigC), C.POINTER(C.c_void_p)]
    lib.kr_engine_destroy.argtypes = [C.c_void_p]
    lib.kr_engine_destroy.restype = None
    lib.kr_engine_get_config.argtypes = [C.c_void_p, C.POINTER(ModelConfigC)]
    lib.kr_upload_expert_unified.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_int]
    lib.kr_upload_expert_bf16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.kr_upload_expert_gguf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_int]
    lib.kr_fill_layer_synthetic.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64]
    lib.kr_fill_layer_synthetic_gguf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64]
    lib.kr_download_expert_unified.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.kr_marlin_repack.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.kr_marlin_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.kr_upload_expert_marlin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.kr_download_expert_marlin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.kr_moe_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.kr_moe_prefill.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.kr_set_routing_config.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.kr_set_routing_weights.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.kr_set_routing_weights_synthetic.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_float, C.c_int]
    lib.kr_route_topk.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
    lib.kr_forward_moe_routed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.kr_reduce_sum_bf16.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.kr_moe_set_prefill_pairs.argtypes = [C.c_void_p, C.c_int]
    lib.kr_moe_set_gemm_mode.argtypes = [C.c_void_p, C.c_int]
    lib.kr_combine_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.kr_ep_unique_id.argtypes = [C.c_void_p]
    lib.kr_ep_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.kr_ep_destroy.argtypes = [C.c_void_p]
    lib.kr_ep_comm_ranks.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.kr_ep_max_int.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]
    lib.kr_ep_allreduce_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.kr_ep_loopback_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.kr_ep_loopback_destroy.argtypes = [C.c_void_p]
    lib.kr_ep_init_loopback.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.kr_moe_prefill_ep.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.kr_synchronize.argtypes = [C.c_void_p]
    lib.kr_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.kr_get_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.kr_decode_create.argtypes = [vp, ci, ci, C.POINTER(vp)]
    lib.kr_decode_set_moe_store.argtypes = [vp, vp]
    lib.kr_decode_destroy.argtypes = [vp]; lib.kr_decode_destroy.restype = None
    lib.kr_decode_store_weight_f32.argtypes = [vp, vp, ci, ci, ci, C.POINTER(ci)]
    lib.kr_decode_store_weight_synthetic.argtypes = [vp, ci, ci, ci, C.c_uint64, C.POINTER(ci)]
    lib.kr_decode_download_weight.argtypes = [vp, ci, vp, vp]
    lib.kr_decode_store_norm_weight.argtypes = [vp, vp, ci, C.POINTER(ci)]
    lib.kr_decode_configure.argtypes = [vp, ci, ci, cf, ci, ci, ci, ci, ci, ci, cf, vp, C.c_uint64]
    lib.kr_decode_add_la_layer.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, cf]
    lib.kr_decode_add_gqa_layer.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, ci, vp, ci, ci, ci, ci, ci, cf]
    lib.kr_decode_prefill.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.kr_decode_prefill_nll.argtypes = [vp, vp, ci, ci, vp, vp, vp]
    lib.kr_decode_reset_state.argtypes = [vp, ci]
    lib.kr_decode_generate.argtypes = [vp, ci, ci, ci, cf, ci, cf, vp, ci, cf, C.c_uint64, vp, vp, vp]
    lib.kr_decode_sample.argtypes = [vp, cf, ci, cf, cf, C.c_uint64, ci, vp, vp]
    lib.kr_sample_order.argtypes = [vp, ci, ci, vp]
    lib.kr_decode_set_prefill_chunk.argtypes = [vp, ci]
    lib.kr_decode_set_prefill_depth.argtypes = [vp, ci]
    lib.kr_decode_add_mla_layer.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, vp, C.c_size_t, vp, C.c_size_t, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci,
                                            ci, cf]
    lib.kr_decode_set_layer_moe.argtypes = [vp, ci, ci, ci, ci, ci]
    lib.kr_decode_set_layer_dense.argtypes = [vp, ci, ci, ci, ci]
    lib.kr_decode_set_rope.argtypes = [vp, vp, vp, ci, ci]
    lib.kr_decode_finalize.argtypes = [vp]
    lib.kr_decode_set_kv_dtype.argtypes = [vp, ci]
    lib.kr_decode_set_attention_mode.argtypes = [vp, ci]
    lib.kr_decode_set_option.argtypes = [vp, C.c_char_p, ci]
    lib.kr_decode_set_state.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    lib.kr_decode_fill_state_synthetic.argtypes = [vp, ci, C.c_uint64]
    lib.kr_decode_get_state.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.kr_decode_step.argtypes = [vp, ci, ci, vp, vp]
    lib.kr_decode_generate_greedy.argtypes = [vp, ci, ci, ci, vp, ci, vp, C.POINTER(ci), vp]
    lib.kr_decode_last_token.argtypes = [vp, C.POINTER(ci)]
    lib.kr_decode_set_use_graph.argtypes = [vp, ci]
    lib.kr_decode_read_buffer.argtypes = [vp, ci, vp, ci]
    lib.kr_decode_profile_step.argtypes = [vp, ci, ci, C.POINTER(C.c_double), C.POINTER(C.c_long), ci]
    lib.kr_decode_device_bytes.argtypes = [vp]; lib.kr_decode_device_bytes.restype = C.c_size_t
    lib.kr_decode_generate_stream.argtypes = [vp, ci, ci, ci, cf, ci, cf, vp, ci, cf, C.c_uint64, TOKEN_CB, vp, vp, vp]
    lib.kr_decode_cancel.argtypes = [vp]; lib.kr_decode_reset_cancel.argtypes = [vp]
    lib.kr_decode_last_elapsed_s.argtypes = [vp]; lib.kr_decode_last_elapsed_s.restype = C.c_double
    lib.kr_decode_matmul.argtypes = [vp, ci, vp, vp]
    lib.kr_decode_matmul_batch.argtypes = [vp, vp, ci, vp, vp]
    lib.kr_decode_fused_add_rmsnorm.argtypes = [vp, vp, vp, vp, ci, cf, ci, ci]
    lib.kr_decode_rmsnorm.argtypes = [vp, vp, vp, cf, vp, ci]
    lib.kr_decode_silu_mul.argtypes = [vp, vp, vp, vp, ci]
    lib.kr_decode_fused_shared_expert.argtypes = [vp, ci, ci, vp, vp]
    lib.kr_decode_linear_attention_recurrent.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci]
    lib.kr_decode_gated_rmsnorm_silu.argtypes = [vp, vp, vp, vp, vp, cf, ci, ci]
    lib.kr_decode_linear_attention_conv.argtypes = [vp, vp, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci]
    lib.kr_decode_store_route_weight.argtypes = [vp, vp, ci, ci, vp, vp, vp]
    lib.kr_decode_moe_route.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci]
    lib.kr_decode_num_route_weights.argtypes = [vp]
    lib.kr_decode_weight_bytes.argtypes = [vp, ci]; lib.kr_decode_weight_bytes.restype = C.c_size_t
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Map C-ABI status codes onto the exception classes the reference raises."""
    if rc == KR_OK:
        return
    msg = load_library().kr_last_error().decode("utf-8", "replace")
    if rc == KR_ERR_STATE:
        raise RuntimeError(msg)          # PyRuntimeError in the reference
    if rc == KR_ERR_VALUE:
        raise ValueError(msg)            # PyValueError
    if rc == KR_ERR_IO:
        raise IOError(msg)               # PyIOError
    raise KrasisHipError(msg)
