"""Tiny random-weight HF checkpoints (config.json + model.safetensors, BF16) in the layouts of the two model families of BASELINE.json:
a Qwen3-Next-shaped hybrid (gated-delta-net linear attention + gated GQA + MoE with a sigmoid-gated shared expert, `(1 + w)` norms) and a
DeepSeek-V2-Lite-shaped MLA model (dense layer 0, un-gated shared experts, YaRN RoPE).  Test infrastructure: the tensors are what
`krasis_amd.decode_setup.CpuDecoder` loads, and the same dictionary (already rounded to bf16) lets the tests build the oracle side
independently of the loader."""
import json
import os

import numpy as np

F = np.float32


def _bf16(x):
    x = np.ascontiguousarray(x, F)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(F)


def _save(path, cfg, tensors):
    import torch
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    save_file({k: torch.from_numpy(v).to(torch.bfloat16).contiguous() for k, v in tensors.items()}, os.path.join(path, "model.safetensors"))


def make_qcn_tiny(path, seed=0, layers=4):
    rng = np.random.default_rng(seed)
    H, V, E, k, I, SI = 256, 320, 8, 2, 128, 128
    nk, nv, dk, dv, kd = 2, 4, 128, 128, 4
    nh, nkv, hd = 4, 2, 64
    cfg = dict(model_type="qwen3_next", hidden_size=H, num_hidden_layers=layers, num_attention_heads=nh, num_key_value_heads=nkv, head_dim=hd,
               vocab_size=V, rms_norm_eps=1e-6, rope_theta=10000.0, partial_rotary_factor=0.25, full_attention_interval=2, num_experts=E,
               num_experts_per_tok=k, moe_intermediate_size=I, shared_expert_intermediate_size=SI, norm_topk_prob=True, decoder_sparse_step=1,
               linear_conv_kernel_dim=kd, linear_key_head_dim=dk, linear_num_key_heads=nk, linear_value_head_dim=dv, linear_num_value_heads=nv,
               tie_word_embeddings=False, intermediate_size=512)
    t = {}
    r = lambda *s, a=0.05: _bf16(rng.standard_normal(s) * a)
    t["model.embed_tokens.weight"] = r(V, H, a=0.1); t["model.norm.weight"] = r(H, a=0.1); t["lm_head.weight"] = r(V, H)
    key_dim, val_dim = nk * dk, nv * dv
    for l in range(layers):
        p = f"model.layers.{l}"
        t[f"{p}.input_layernorm.weight"] = r(H, a=0.1); t[f"{p}.post_attention_layernorm.weight"] = r(H, a=0.1)
        if (l + 1) % 2 == 0:
            a = f"{p}.self_attn"
            t[f"{a}.q_proj.weight"] = r(nh * hd * 2, H, a=0.08); t[f"{a}.k_proj.weight"] = r(nkv * hd, H, a=0.08); t[f"{a}.v_proj.weight"] = r(nkv * hd, H, a=0.08)
            t[f"{a}.o_proj.weight"] = r(H, nh * hd); t[f"{a}.q_norm.weight"] = r(hd, a=0.2); t[f"{a}.k_norm.weight"] = r(hd, a=0.2)
        else:
            a = f"{p}.linear_attn"
            t[f"{a}.in_proj_qkvz.weight"] = r(2 * key_dim + 2 * val_dim, H, a=0.08); t[f"{a}.in_proj_ba.weight"] = r(2 * nv, H, a=0.3)
            t[f"{a}.conv1d.weight"] = r(2 * key_dim + val_dim, 1, kd, a=0.4); t[f"{a}.A_log"] = _bf16(rng.random(nv) * 1.5 - 1.0)
            t[f"{a}.dt_bias"] = _bf16(rng.random(nv) - 0.5); t[f"{a}.norm.weight"] = _bf16(rng.random(dv) + 0.5); t[f"{a}.out_proj.weight"] = r(H, val_dim)
        m = f"{p}.mlp"
        t[f"{m}.gate.weight"] = _bf16((rng.random((E, H)) - 0.5) * 0.1)
        for e in range(E):
            t[f"{m}.experts.{e}.gate_proj.weight"] = r(I, H); t[f"{m}.experts.{e}.up_proj.weight"] = r(I, H); t[f"{m}.experts.{e}.down_proj.weight"] = r(H, I)
        t[f"{m}.shared_expert.gate_proj.weight"] = r(SI, H); t[f"{m}.shared_expert.up_proj.weight"] = r(SI, H); t[f"{m}.shared_expert.down_proj.weight"] = r(H, SI)
        t[f"{m}.shared_expert_gate.weight"] = r(1, H, a=0.3)
    _save(path, cfg, t)
    return cfg, t


def make_v2lite_tiny(path, seed=1, layers=3):
    rng = np.random.default_rng(seed)
    H, V, E, k, I = 256, 320, 8, 3, 128
    nh, klr, nd, rd, vhd, DI = 4, 256, 128, 64, 128, 344          # dense intermediate 344: not a multiple of 128 -> the loader's column padding
    cfg = dict(model_type="deepseek_v2", hidden_size=H, num_hidden_layers=layers, num_attention_heads=nh, num_key_value_heads=nh, vocab_size=V,
               rms_norm_eps=1e-6, rope_theta=10000.0, kv_lora_rank=klr, q_lora_rank=None, qk_nope_head_dim=nd, qk_rope_head_dim=rd, v_head_dim=vhd,
               n_routed_experts=E, num_experts_per_tok=k, moe_intermediate_size=I, n_shared_experts=2, first_k_dense_replace=1, intermediate_size=DI,
               routed_scaling_factor=1.0, scoring_func="softmax", norm_topk_prob=False, tie_word_embeddings=False,
               rope_scaling=dict(type="yarn", factor=40.0, original_max_position_embeddings=4096, beta_fast=32.0, beta_slow=1.0, mscale=0.707,
                                 mscale_all_dim=0.707))
    t = {}
    r = lambda *s, a=0.05: _bf16(rng.standard_normal(s) * a)
    t["model.embed_tokens.weight"] = r(V, H, a=0.1); t["model.norm.weight"] = _bf16(rng.random(H) * 0.2 + 0.9); t["lm_head.weight"] = r(V, H)
    for l in range(layers):
        p = f"model.layers.{l}"
        t[f"{p}.input_layernorm.weight"] = _bf16(rng.random(H) * 0.2 + 0.9); t[f"{p}.post_attention_layernorm.weight"] = _bf16(rng.random(H) * 0.2 + 0.9)
        a = f"{p}.self_attn"
        t[f"{a}.q_proj.weight"] = r(nh * (nd + rd), H, a=0.08); t[f"{a}.kv_a_proj_with_mqa.weight"] = r(klr + rd, H, a=0.08)
        t[f"{a}.kv_a_layernorm.weight"] = _bf16(rng.random(klr) + 0.5); t[f"{a}.kv_b_proj.weight"] = r(nh * (nd + vhd), klr, a=0.06)
        t[f"{a}.o_proj.weight"] = r(H, nh * vhd)
        m = f"{p}.mlp"
        if l == 0:
            t[f"{m}.gate_proj.weight"] = r(DI, H); t[f"{m}.up_proj.weight"] = r(DI, H); t[f"{m}.down_proj.weight"] = r(H, DI)
        else:
            t[f"{m}.gate.weight"] = _bf16((rng.random((E, H)) - 0.5) * 0.1)
            for e in range(E):
                t[f"{m}.experts.{e}.gate_proj.weight"] = r(I, H); t[f"{m}.experts.{e}.up_proj.weight"] = r(I, H); t[f"{m}.experts.{e}.down_proj.weight"] = r(H, I)
            t[f"{m}.shared_experts.gate_proj.weight"] = r(2 * I, H); t[f"{m}.shared_experts.up_proj.weight"] = r(2 * I, H)
            t[f"{m}.shared_experts.down_proj.weight"] = r(H, 2 * I)
    _save(path, cfg, t)
    return cfg, t
