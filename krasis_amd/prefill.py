"""GpuPrefillManager -- host mirror of the reference's prefill operator (python/krasis/gpu_prefill.py:4374-4484).

Only the operator contract is kept: `forward(moe_layer_idx, hidden[M,K] bf16, topk_ids[M,k] i32, topk_weights[M,k] f32,
routed_only)` returning bf16 [M,K] = rsf * routed + shared (or the bare routed sum when routed_only).  None of the reference's
VRAM-scarcity machinery (layer-grouped expert DMA, LRU / hot-cached-static experts, pinned staging) exists here: every expert
is resident in the MI355X's HBM and the arithmetic is a token sort + int8-MFMA grouped GEMM in libkrasis_hip.so.
Tensors are torch tensors on the engine's device; PyTorch is only the allocator.
"""
from __future__ import annotations

from . import _lib
from ._lib import check
from .engine import KrasisEngine

# batches below this go through the streaming (memory-bound) decode kernels, as in the reference's gpu_prefill_threshold split
PREFILL_MIN_TOKENS = 48


class GpuPrefillManager:
    def __init__(self, engine: KrasisEngine, num_experts_per_tok: int | None = None):
        engine._need("Model not loaded")
        self.engine = engine
        self.top_k = num_experts_per_tok or engine.top_k()

    def forward(self, moe_layer_idx: int, hidden, topk_ids, topk_weights, routed_only: bool = False):
        import torch
        if hidden.dtype != torch.bfloat16:
            raise ValueError(f"hidden must be bfloat16, got {hidden.dtype}")
        if not hidden.is_cuda:
            raise ValueError("hidden must live on the GPU")
        M, K = hidden.shape
        if K != self.engine.hidden_size():
            raise ValueError(f"hidden dim {K} != model hidden_size {self.engine.hidden_size()}")
        hidden = hidden.contiguous()
        ids = topk_ids.to(torch.int32).contiguous()
        w = topk_weights.to(torch.float32).contiguous()
        if ids.shape != w.shape or ids.shape[0] != M:
            raise ValueError("topk_ids / topk_weights shape mismatch")
        out = torch.empty((M, K), dtype=torch.bfloat16, device=hidden.device)
        st = torch.cuda.current_stream(hidden.device).cuda_stream or 1   # 1 = legacy default stream in the C ABI
        eng = self.engine
        fn = eng._lib.kr_moe_prefill if M >= PREFILL_MIN_TOKENS else eng._lib.kr_moe_forward
        check(fn(eng._h, moe_layer_idx, hidden.data_ptr(), ids.data_ptr(), w.data_ptr(), out.data_ptr(), M, ids.shape[1],
                 _lib.KR_OUT_BF16, int(routed_only), st))
        return out
