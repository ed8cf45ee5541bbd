"""bench_decode_synthetic (src/decode.rs:4618-5560): decode_step speed on synthetic weights of a model's exact shapes, driven by its config.json.

Same protocol as the reference: weights are pseudo-random words with bf16 scales, the router gate is the xorshift64 stream at +-0.02
(decode.rs:5181), kv_max_seq = 256 (decode.rs:5031), `warmup` untimed steps at positions 10, 11, ... and `num_steps` timed steps of token 0 at
positions (10 + warmup + i) % (kv_max_seq - 1) (decode.rs:5490-5513), the result printed to stderr as ms/token and tok/s.  The arguments that
steer the CPU engine (`num_threads`, `tiled`: rayon pool size, TILE_N re-tiling) are accepted and ignored: the HBM layout is fixed.  With
`timing` the per-kind launch times of one un-graphed step (HIP events) are printed in place of KRASIS_CPU_DECODE_TIMING's buckets.
"""
from __future__ import annotations

import json
import os
import sys
import time
from typing import Optional

import numpy as np

from .decode_setup import ModelArch
from .decode_store import CpuDecodeStore
from .engine import KrasisEngine, ModelConfig

KINDS = ["embed", "fused_add_rmsnorm", "proj_matvec", "la_conv", "la_recurrent", "gated_rmsnorm_silu", "attention", "route_logits",
         "route_select", "moe_w13", "moe_w2", "moe_combine", "lm_head", "argmax", "shared_gate"]
SEED = 0x12345678ABCDEF01          # decode.rs:4898


def build_synthetic(arch: ModelArch, num_bits: int = 4, max_experts: int = 0, device: int = 0, kv_max_seq: int = 256, kv_fp8: bool = False,
                    max_layers: Optional[int] = None):
    """-> (engine, store, keepalive).  Every layer of `arch` with the reference's synthetic fill; `max_experts` caps the experts allocated per
    layer (decode.rs:4737-4742: routing still covers the capped set only)."""
    H, V = arch.hidden_size, arch.vocab_size
    L = arch.num_hidden_layers if max_layers is None else min(max_layers, arch.num_hidden_layers)
    E = arch.n_routed_experts if not (0 < max_experts < arch.n_routed_experts) else max_experts
    k = min(arch.num_experts_per_tok, E) if E else 0
    moe_layers = [i for i in range(L) if arch.is_moe_layer(i)]
    n_sh = arch.n_shared_experts
    sh_inter = arch.effective_shared_expert_intermediate if n_sh else 0
    eng = KrasisEngine(device=device)
    scoring = {"sigmoid": 0, "softmax": 1}.get(arch.scoring_func, 1)
    if moe_layers:
        eng.configure(ModelConfig(H, arch.moe_intermediate_size, E, k, len(moe_layers), 0, arch.routed_scaling_factor))
        eng.fill_synthetic(num_bits, seed=SEED)
        eng.set_routing_config(arch.scoring_func if arch.scoring_func in ("sigmoid", "softmax") else "softmax", arch.norm_topk_prob, k, E, H)
    st = CpuDecodeStore(128, True, arch.norm_bias_one)
    if moe_layers:
        st.set_moe_store(eng)
    rng = np.random.default_rng(1234)
    keep, seed = [], [100]

    def W(rows, cols):
        seed[0] += 1
        return st.store_weight_synthetic(rows, (cols + 127) // 128 * 128, num_bits, seed[0])

    def N(n):
        w = ((rng.random(n, dtype=np.float32) - 0.5) * 0.04 + (0.0 if arch.norm_bias_one else 1.0)).astype(np.float32); keep.append(w)   # decode.rs:4936
        return st.store_norm_weight(w.ctypes.data, n)

    fin, lm = N(H), W(V, H)
    st.configure_decode(H, L, arch.rms_norm_eps, fin, lm, V, max(k, 1), scoring, arch.norm_topk_prob, arch.routed_scaling_factor, 0, synth_seed=777)
    nh, nkv, hd = arch.num_attention_heads, arch.num_key_value_heads, arch.head_dim
    gated = (not arch.is_mla) and arch.partial_rotary_factor < 1.0            # decode.rs:4699
    if arch.is_mla:
        klr, nd, rd, vhd = arch.kv_lora_rank, arch.qk_nope_head_dim, arch.qk_rope_head_dim, arch.v_head_dim
        half = rd // 2
        ang = np.arange(kv_max_seq)[:, None] * (1.0 / arch.rope_theta ** (2 * np.arange(half) / rd))[None, :]
        mcos, msin = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32); keep += [mcos, msin]
    moe_i = 0
    for l in range(L):
        n_in, n_post = N(H), N(H)
        if arch.layer_type(l) == "linear_attention":
            nk, nv, dk, dv, kd = arch.linear_num_key_heads, arch.linear_num_value_heads, arch.linear_key_head_dim, arch.linear_value_head_dim, arch.linear_conv_kernel_dim
            hr = nv // nk; group_dim = 2 * dk + 2 * dv * hr; conv_dim = 2 * nk * dk + nv * dv
            qkvz, ba, out = W(nk * group_dim, H), W(nk * 2 * hr, H), W(H, nv * dv)
            cw = ((rng.random(conv_dim * kd, dtype=np.float32) - 0.5) * 0.1).astype(np.float32)                                   # decode.rs:5142
            a_log = ((rng.random(nv, dtype=np.float32) - 0.5) * 2.0).astype(np.float32); dtb = (rng.random(nv, dtype=np.float32) - 0.5).astype(np.float32)
            nw = (rng.random(nv * dv, dtype=np.float32) + 0.5).astype(np.float32); keep += [cw, a_log, dtb, nw]
            st.add_decode_la_layer(n_in, n_post, qkvz, ba, out, cw.ctypes.data, a_log.ctypes.data, dtb.ctypes.data, nw.ctypes.data, nk, nv, dk, dv, hr, kd,
                                   1.0 / dk ** 0.5)
        elif arch.is_mla:
            kv_a, o = W(klr + rd, H), W(H, nh * vhd)
            if arch.has_q_lora:
                q, qa, qb = None, W(arch.q_lora_rank, H), W(nh * (nd + rd), arch.q_lora_rank)
                qan = (rng.random(arch.q_lora_rank) + 0.5).astype(np.float32); keep.append(qan)
            else:
                q, qa, qb, qan = W(nh * (nd + rd), H), None, None, None
            w_kc = ((rng.standard_normal((nh, nd, klr)) * 0.05).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)          # decode.rs:5057
            w_vc = ((rng.standard_normal((nh, vhd, klr)) * 0.05).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
            kvn = (rng.random(klr) + 0.5).astype(np.float32); keep += [w_kc, w_vc, kvn]
            st.add_decode_mla_layer(n_in, n_post, kv_a, o, q, qa, qb, w_kc.ctypes.data, w_kc.size, w_vc.ctypes.data, w_vc.size, kvn.ctypes.data, klr,
                                    qan.ctypes.data if qan is not None else 0, arch.q_lora_rank if qan is not None else 0,
                                    mcos.ctypes.data, msin.ctypes.data, half, kv_max_seq, nh, klr, nd, rd, vhd, float(1.0 / np.sqrt(nd + rd)))
        else:
            qw, kw, vw, ow = W(nh * hd * (2 if gated else 1), H), W(nkv * hd, H), W(nkv * hd, H), W(H, nh * hd)
            qn = (rng.random(hd, dtype=np.float32) + 0.5).astype(np.float32); kn = (rng.random(hd, dtype=np.float32) + 0.5).astype(np.float32); keep += [qn, kn]
            st.add_decode_gqa_layer(n_in, n_post, qw, kw, vw, ow, qn.ctypes.data, hd, kn.ctypes.data, hd, gated, nh, nkv, hd, 1.0 / hd ** 0.5)
        if arch.is_moe_layer(l):
            eng.set_route_weight_synthetic(moe_i, SEED, 0.02, True)                                                                 # decode.rs:5181
            if n_sh:
                sgu, sd = W(2 * sh_inter, H), W(H, sh_inter)
                sg = W(1, H) if arch.shared_expert_intermediate_size else None      # qwen3_next: sigmoid gate on the shared expert
                st.set_decode_layer_moe(l, moe_i, moe_i, sgu, sd, sg)
            else:
                st.set_decode_layer_moe(l, moe_i, moe_i, None, None, None)
            moe_i += 1
        else:
            DI = (arch.intermediate_size + 127) // 128 * 128
            st.set_decode_layer_dense(l, W(DI, H), W(DI, H), W(H, DI))
    if not arch.is_mla:
        half = hd // 2                                                                                                              # decode.rs:5381: full rotary
        pos = np.arange(kv_max_seq, dtype=np.float32)[:, None]
        freq = (1.0 / (arch.rope_theta ** (2.0 * np.arange(half, dtype=np.float32) / hd))).astype(np.float32)[None, :]
        cos, sin = np.cos(pos * freq).astype(np.float32), np.sin(pos * freq).astype(np.float32); keep += [cos, sin]
        st.set_decode_rope(cos.ctypes.data, sin.ctypes.data, half, kv_max_seq)
    st.finalize_decode()
    st.set_kv_dtype(kv_fp8)
    st.fill_state_synthetic(kv_max_seq, seed=4242)
    return eng, st, keep


def bench_decode_synthetic(config_path: str, num_steps: int = 100, warmup: int = 5, timing: bool = False, num_bits: int = 4, max_experts: int = 0,
                           num_threads: int = 40, tiled: bool = True, device: int = 0, max_layers: Optional[int] = None):
    """decode.rs:4618.  Prints the reference's report lines to stderr; additionally returns {"ms_per_token", "tok_per_s", "steps"} (the reference returns None)."""
    if num_bits not in (4, 8):
        raise ValueError("num_bits must be 4 or 8")
    try:
        with open(config_path) as f:
            json.load(f)
    except OSError as e:
        raise IOError(f"Failed to read {config_path}: {e}")                     # decode.rs:4652
    except json.JSONDecodeError as e:
        raise ValueError(f"Invalid JSON: {e}")                                  # decode.rs:4654
    model_dir = os.path.dirname(os.path.abspath(config_path))
    if os.path.basename(config_path) != "config.json":                          # ModelArch reads <dir>/config.json
        import shutil, tempfile
        model_dir = tempfile.mkdtemp(prefix="krasis_synth_")
        shutil.copy(config_path, os.path.join(model_dir, "config.json"))
    arch = ModelArch.from_model_path(model_dir)
    kv_max_seq = 256
    err = lambda *a: print(*a, file=sys.stderr, flush=True)
    err(f"=== Synthetic Decode Benchmark (MI355X, device {device}; num_threads={num_threads} / tiled={tiled} steer the CPU engine only) ===")
    t0 = time.perf_counter()
    eng, st, keep = build_synthetic(arch, num_bits, max_experts, device, kv_max_seq, max_layers=max_layers)
    err(f"Total weight memory: {(eng.device_bytes() + st.device_bytes()) / 1e9:.1f} GB")
    err(f"Data pre-generation: {time.perf_counter() - t0:.1f}s")
    err(f"\nRunning {warmup} warmup + {num_steps} timed steps...")
    for i in range(warmup):
        if 10 + i >= kv_max_seq:
            break
        st.decode_step(0, 10 + i)
    eng.synchronize()
    t_start = time.perf_counter()
    for i in range(num_steps):
        st.decode_step(0, (10 + warmup + i) % (kv_max_seq - 1))
    st.last_token()                                                             # drains the stream
    elapsed = time.perf_counter() - t_start
    ms, tps = elapsed / max(num_steps, 1) * 1e3, num_steps / elapsed if elapsed > 0 else 0.0
    err(f"\n=== RESULTS ({num_steps} steps) ===")
    err(f"Total: {elapsed:.2f}s")
    err(f"Per token: {ms:.1f} ms")
    err(f"Speed: {tps:.2f} tok/s")
    out = {"ms_per_token": ms, "tok_per_s": tps, "steps": num_steps}
    if timing:
        kinds = st.profile_step(0, 20)
        err("\nPer-kind launch time of one un-graphed step (ms):")
        for name, (t, n) in zip(KINDS, kinds):
            if n:
                err(f"  {name:20s} {t:8.3f}  ({n} launches)")
        out["per_kind_ms"] = {name: t for name, (t, n) in zip(KINDS, kinds) if n}
    return out
