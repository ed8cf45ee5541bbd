"""Checkpoint -> decode graph -> perplexity, end to end: `krasis_amd.decode_setup.CpuDecoder` -- the counterpart of the
reference's decode_setup.py:120-230,824-1018 + weight_loader.py -- loads a tiny random-weight HF checkpoint of each BASELINE model family
(tests/tiny_checkpoints.py); the decode step, the prompt pass and `evaluate_perplexity` (perplexity/measure_ppl.py:154-297) are then checked
against the oracle driver built INDEPENDENTLY from the same tensors (the test restates the loader's transformations: BF16 -> f32,
`(1 + w)` norms, conv squeeze, norm tiling, kv_b_proj split, gate||up fusion, column padding, RoPE / YaRN tables).  Logits bit for bit.
A real checkpoint needs nothing but its path."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from tests.oracle_decode import OracleDecode
from tests.tiny_checkpoints import make_qcn_tiny, make_v2lite_tiny

F = np.float32


# ---------------------------------------------------------------------------------------------------------------- CPU: config + reader
def test_model_arch_from_tiny_checkpoints(tmp_path):
    from krasis_amd.decode_setup import CheckpointReader, ModelArch
    make_qcn_tiny(str(tmp_path / "qcn")); make_v2lite_tiny(str(tmp_path / "v2l"))
    r = CheckpointReader(str(tmp_path / "qcn"))
    a = ModelArch.from_model_path(str(tmp_path / "qcn"), list(r.weight_map))
    assert a.norm_bias_one and not a.is_mla and a.layer_types == ["linear_attention", "full_attention", "linear_attention", "full_attention"]
    assert (a.n_routed_experts, a.num_experts_per_tok, a.n_shared_experts, a.effective_shared_expert_intermediate) == (8, 2, 1, 128)
    assert a.rotary_dim == 16 and a.head_dim == 64 and a.first_k_dense_replace == 0 and a.layers_prefix == "model" and not a.tie_word_embeddings
    w = r.f32("model.layers.0.linear_attn.conv1d.weight")
    assert w.dtype == np.float32 and w.shape == (2 * 2 * 128 + 4 * 128, 1, 4)
    b = ModelArch.from_model_path(str(tmp_path / "v2l"), list(CheckpointReader(str(tmp_path / "v2l")).weight_map))
    assert b.is_mla and not b.has_q_lora and b.first_k_dense_replace == 1 and b.is_moe_layer(1) and not b.is_moe_layer(0)
    assert (b.kv_lora_rank, b.qk_nope_head_dim, b.qk_rope_head_dim, b.v_head_dim, b.rotary_dim) == (256, 128, 64, 128, 64)
    assert b.effective_shared_expert_intermediate == 256 and not b.norm_bias_one and b.layer_types is None


# ---------------------------------------------------------------------------------------------------------------- oracle side from raw tensors
def _pad(w, align=128):
    w = np.ascontiguousarray(w, F)
    if w.shape[1] % align == 0:
        return w
    out = np.zeros((w.shape[0], (w.shape[1] + align - 1) // align * align), F); out[:, : w.shape[1]] = w
    return out


def _experts(t, prefix, E):
    return [O.unified_from_bf16(O.f32_to_bf16(t[f"{prefix}.experts.{e}.gate_proj.weight"]), O.f32_to_bf16(t[f"{prefix}.experts.{e}.up_proj.weight"]),
                                O.f32_to_bf16(t[f"{prefix}.experts.{e}.down_proj.weight"]), 128, 4) for e in range(E)]


def oracle_from_qcn(cfg, t, max_seq):
    import torch
    H, V, L, E, k = cfg["hidden_size"], cfg["vocab_size"], cfg["num_hidden_layers"], cfg["num_experts"], cfg["num_experts_per_tok"]
    nk, nv, dk, dv = cfg["linear_num_key_heads"], cfg["linear_num_value_heads"], cfg["linear_key_head_dim"], cfg["linear_value_head_dim"]
    nh, nkv, hd = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    orc = OracleDecode(H, cfg["rms_norm_eps"], False, k, 1, True, 1.0, t["model.embed_tokens.weight"], V)     # (1 + w) folded into the stored norms
    one = F(1.0)
    orc.final_norm = orc.store_norm(t["model.norm.weight"] + one); orc.lm_head = orc.store_weight_f32(t["lm_head.weight"])
    dim = int(hd * cfg["partial_rotary_factor"])
    fr = torch.outer(torch.arange(max_seq, dtype=torch.float32), 1.0 / (cfg["rope_theta"] ** (torch.arange(0, dim, 2).float() / dim)))
    orc.rope = (np.ascontiguousarray(fr.cos().numpy()), np.ascontiguousarray(fr.sin().numpy()))
    for l in range(L):
        p = f"model.layers.{l}"
        Ld = dict(in_norm=orc.store_norm(t[f"{p}.input_layernorm.weight"] + one), post_norm=orc.store_norm(t[f"{p}.post_attention_layernorm.weight"] + one))
        if (l + 1) % cfg["full_attention_interval"] == 0:
            a = f"{p}.self_attn"
            Ld.update(attn="gqa", q=orc.store_weight_f32(t[f"{a}.q_proj.weight"]), k=orc.store_weight_f32(t[f"{a}.k_proj.weight"]),
                      v=orc.store_weight_f32(t[f"{a}.v_proj.weight"]), o=orc.store_weight_f32(t[f"{a}.o_proj.weight"]),
                      q_norm=t[f"{a}.q_norm.weight"] + one, k_norm=t[f"{a}.k_norm.weight"] + one, gated=True, nh=nh, nkv=nkv, hd=hd,
                      sm_scale=F(1.0 / math.sqrt(hd)), kv_k=np.zeros((max_seq, nkv * hd), np.uint16), kv_v=np.zeros((max_seq, nkv * hd), np.uint16))
        else:
            a = f"{p}.linear_attn"
            conv_dim = 2 * nk * dk + nv * dv
            Ld.update(attn="la", qkvz=orc.store_weight_f32(t[f"{a}.in_proj_qkvz.weight"]), ba=orc.store_weight_f32(t[f"{a}.in_proj_ba.weight"]),
                      out=orc.store_weight_f32(t[f"{a}.out_proj.weight"]), conv_w=np.ascontiguousarray(t[f"{a}.conv1d.weight"][:, 0, :]).reshape(-1),
                      a_log=t[f"{a}.A_log"], dt_bias=t[f"{a}.dt_bias"], norm_w=np.tile(t[f"{a}.norm.weight"], nv), nk=nk, nv=nv, dk=dk, dv=dv,
                      scale=F(1.0 / math.sqrt(dk)), conv_state=np.zeros(conv_dim * 4, F), recur_state=np.zeros(nv * dk * dv, F))
        m = f"{p}.mlp"
        gu = np.concatenate([t[f"{m}.shared_expert.gate_proj.weight"], t[f"{m}.shared_expert.up_proj.weight"]], 0)
        Ld.update(mlp="moe", gate=t[f"{m}.gate.weight"], experts=_experts(t, m, E), sgu=orc.store_weight_f32(gu),
                  sd=orc.store_weight_f32(t[f"{m}.shared_expert.down_proj.weight"]), sg=orc.store_weight_f32(t[f"{m}.shared_expert_gate.weight"]))
        orc.layers.append(Ld)
    return orc


def oracle_from_v2lite(cfg, t, max_seq):
    import torch
    H, V, L, E, k = cfg["hidden_size"], cfg["vocab_size"], cfg["num_hidden_layers"], cfg["n_routed_experts"], cfg["num_experts_per_tok"]
    nh, klr, nd, rd, vhd = cfg["num_attention_heads"], cfg["kv_lora_rank"], cfg["qk_nope_head_dim"], cfg["qk_rope_head_dim"], cfg["v_head_dim"]
    orc = OracleDecode(H, cfg["rms_norm_eps"], False, k, 1, False, 1.0, t["model.embed_tokens.weight"], V)
    orc.final_norm = orc.store_norm(t["model.norm.weight"]); orc.lm_head = orc.store_weight_f32(t["lm_head.weight"])
    rc = cfg["rope_scaling"]; theta = cfg["rope_theta"]
    freqs = 1.0 / (theta ** (torch.arange(0, rd, 2).float() / rd))                                   # decode_setup.py:722-747 (YaRN)
    low = max(0, math.floor(rd * math.log(rc["original_max_position_embeddings"] / (rc["beta_fast"] * 2 * math.pi)) / (2 * math.log(theta))))
    high = min(rd // 2 - 1, math.ceil(rd * math.log(rc["original_max_position_embeddings"] / (rc["beta_slow"] * 2 * math.pi)) / (2 * math.log(theta))))
    ramp = torch.clamp((torch.arange(rd // 2).float() - low) / max(high - low, 0.001), 0, 1)
    mask = 1.0 - ramp
    freqs = (freqs / rc["factor"]) * (1 - mask) + freqs * mask
    fr = torch.outer(torch.arange(max_seq, dtype=torch.float32), freqs)
    cos, sin = np.ascontiguousarray(fr.cos().numpy()), np.ascontiguousarray(fr.sin().numpy())
    mscale = 0.1 * rc["mscale_all_dim"] * math.log(rc["factor"]) + 1.0
    sm = F((1.0 / math.sqrt(nd + rd)) * mscale * mscale)
    for l in range(L):
        p = f"model.layers.{l}"; a = f"{p}.self_attn"
        kv_b = t[f"{a}.kv_b_proj.weight"].reshape(nh, nd + vhd, klr)
        Ld = dict(in_norm=orc.store_norm(t[f"{p}.input_layernorm.weight"]), post_norm=orc.store_norm(t[f"{p}.post_attention_layernorm.weight"]), attn="mla",
                  kv_a=orc.store_weight_f32(t[f"{a}.kv_a_proj_with_mqa.weight"]), o=orc.store_weight_f32(t[f"{a}.o_proj.weight"]),
                  q=orc.store_weight_f32(t[f"{a}.q_proj.weight"]), q_a=None, q_b=None, q_a_norm=None, kv_a_norm=t[f"{a}.kv_a_layernorm.weight"],
                  w_kc=np.ascontiguousarray(kv_b[:, :nd, :]), w_vc=np.ascontiguousarray(kv_b[:, nd:, :]), cos=cos, sin=sin, nh=nh, klr=klr, nd=nd, rd=rd,
                  vhd=vhd, sm_scale=sm, ckv=np.zeros((max_seq, klr), np.uint16), kpe=np.zeros((max_seq, rd), np.uint16))
        m = f"{p}.mlp"
        if l < cfg["first_k_dense_replace"]:
            Ld.update(mlp="dense", gate_w=orc.store_weight_f32(t[f"{m}.gate_proj.weight"]), up_w=orc.store_weight_f32(t[f"{m}.up_proj.weight"]),
                      down_w=orc.store_weight_f32(_pad(t[f"{m}.down_proj.weight"])))
        else:
            gu = np.concatenate([t[f"{m}.shared_experts.gate_proj.weight"], t[f"{m}.shared_experts.up_proj.weight"]], 0)
            Ld.update(mlp="moe", gate=t[f"{m}.gate.weight"], experts=_experts(t, m, E), sgu=orc.store_weight_f32(gu),
                      sd=orc.store_weight_f32(t[f"{m}.shared_experts.down_proj.weight"]), sg=None)
        orc.layers.append(Ld)
    return orc


def _float64_nll(logits, label):
    x = logits.astype(np.float64)
    m = x.max()
    return float(m + np.log(np.exp(x - m).sum()) - x[label])


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["qcn", "v2lite"])
def test_checkpoint_to_decode_to_perplexity(tmp_path, family):
    from krasis_amd import evaluate_perplexity
    from krasis_amd.decode_setup import CpuDecoder
    path = str(tmp_path / family)
    cfg, t = (make_qcn_tiny if family == "qcn" else make_v2lite_tiny)(path)
    max_seq = 32
    dec = CpuDecoder(path)
    dec.init_weights(max_rope_seq=max_seq)
    assert dec._store.num_weights() > 0 and dec.engine.num_moe_layers() == cfg["num_hidden_layers"] - cfg.get("first_k_dense_replace", 0)
    orc = (oracle_from_qcn if family == "qcn" else oracle_from_v2lite)(cfg, t, max_seq)
    V = cfg["vocab_size"]
    toks = [5, 17, 200, 3, 3, 99, 250, 41, 7, 300, 12, 64]
    # ---- token-by-token decode from a fresh state: logits bit for bit
    dec.prepare(max_seq)
    ref_logits = []
    for i, tk in enumerate(toks):
        got = np.empty(V, F); dec._store.decode_step(tk, i, got.ctypes.data)
        ref = orc.step(tk, i); ref_logits.append(ref)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (family, i, float(np.abs(got - ref).max()))
    # ---- prompt pass + generate: same first token as the oracle's greedy sample
    out = dec.generate(toks, 3)
    assert out[0] == O.sample_greedy(ref_logits[-1])
    nxt = O.sample_greedy(orc.step(out[0], len(toks)))
    assert out[1] == nxt
    # ---- perplexity harness over the loaded checkpoint vs a float64 cross-entropy of the oracle's logits (one window = the whole text)
    r = evaluate_perplexity(dec._store, toks, len(toks), len(toks) // 2)
    want = [_float64_nll(ref_logits[i], toks[i + 1]) for i in range(len(toks) - 1)]
    assert r["num_windows"] >= 1 and r["num_tokens_total"] == len(toks)
    dec.prepare(max_seq)
    nll = dec._store.prefill_nll(toks, 0)
    assert np.abs(nll.astype(np.float64) - np.asarray(want)).max() < 4e-6
    first_window = float(np.sum(nll, dtype=np.float32))
    assert abs(r["total_nll"] - first_window) < 1e-3 or r["num_windows"] > 1
    assert math.isfinite(r["perplexity"]) and r["perplexity"] > 1.0
