"""Sliding-window perplexity over the HIP prompt pass -- the host-side mirror of the reference's harness
(perplexity/measure_ppl.py:154-297, `evaluate_perplexity`): same windows, same scored region, same result dictionary.

The reference runs model.forward(window, return_all_logits=True) on its GPU prefill path and torch cross_entropy on the [W, V] logits;
here one call of CpuDecodeStore.prefill_nll scores a window on the device (logits bit-identical to the decode step's, log-sum-exp per
row fused behind the lm_head GEMM), so only W-1 floats per window come back to the host.
"""
from __future__ import annotations

import math
import time
from typing import Optional, Sequence


def window_plan(total_tokens: int, window_size: int, stride: int):
    """[(begin, end, score_start)] exactly as measure_ppl.py:191-247: windows start every `stride` tokens, hold up to `window_size`, the first
    one scores every shifted position, later ones only shifted positions >= stride - 1 (tokens not scored by an earlier window)"""
    if window_size < 2 or stride < 1:
        raise ValueError(f"window_size must be >= 2 and stride >= 1, got {window_size}, {stride}")
    plan = []
    for begin in range(0, total_tokens - 1, stride):
        end = min(begin + window_size, total_tokens)
        if end - begin < 2:
            break
        plan.append((begin, end, 0 if begin == 0 else stride - 1))
    return plan


def evaluate_perplexity(store, tokens: Sequence[int], window_size: int, stride: int, max_tokens: Optional[int] = None) -> dict:
    """store: a finalized CpuDecodeStore.  Every window starts from a fresh request state (reset_decode_state), positions 0..W-1."""
    if max_tokens is not None:
        tokens = tokens[:max_tokens]
    total_tokens = len(tokens)
    if total_tokens < 2:
        raise ValueError(f"Need at least 2 tokens, got {total_tokens}")
    total_nll, total_scored, num_windows = 0.0, 0, 0
    t_start = time.perf_counter()
    for begin, end, score_start in window_plan(total_tokens, window_size, stride):
        store.reset_decode_state(window_size)
        loss_per_pos = store.prefill_nll(list(tokens[begin:end]), 0)          # [W-1]: position i predicts token begin + i + 1
        scored = loss_per_pos[score_start:]
        if scored.shape[0] > 0:
            total_nll += float(scored.sum(dtype="float32"))                    # torch: scored_loss.sum().item() in f32
            total_scored += int(scored.shape[0])
        num_windows += 1
    elapsed_s = time.perf_counter() - t_start
    if total_scored == 0:
        raise ValueError("No tokens scored \u2014 check window/stride settings")
    mean_loss = total_nll / total_scored
    return {
        "perplexity": math.exp(mean_loss),
        "bits_per_char": mean_loss / math.log(2),
        "mean_loss": mean_loss,
        "total_nll": total_nll,
        "num_tokens_scored": total_scored,
        "num_tokens_total": total_tokens,
        "num_windows": num_windows,
        "window_size": window_size,
        "stride": stride,
        "elapsed_s": elapsed_s,
    }
