#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the Python side of the reference (runs only where /root/reference exists).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference's hot-path arithmetic is Rust (not buildable here), but its Python package restates several of the same operators in torch
(HF-equivalent math) and those modules import on CPU once the native extension and flashinfer are stubbed:

  * python/krasis/linear_attention.py: GatedDeltaNetAttention._forward_recurrent  (decode gated-delta-rule step, :460-591)
                                       GatedDeltaNetAttention._forward_chunked    (prefill chunked gated delta rule, :695-845)
  * python/krasis/layer.py:            TransformerLayer.compute_routing           (:526-560)
  * python/krasis/attention.py:        GQAAttention._get_rope_cos_sin/_apply_rope (:443-494)

Nothing is copied: the modules are imported from where they lie and run on seeded inputs; only inputs/outputs are stored.
The fixtures are what `tests/test_golden.py` compares the oracle (and, on the GPU, the HIP path) against.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

os.environ["KRASIS_FUSED_LINEAR_ATTN"] = "0"   # the reference's own eager switch (linear_attention.py:37): no torch.compile on CPU
REF = "/root/reference/python/krasis"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    pkg = types.ModuleType("krasis")
    pkg.__path__ = [REF]                      # submodules resolve from the reference tree; its __init__ (native ext) is NOT run
    sys.modules["krasis"] = pkg
    fi = types.ModuleType("flashinfer")       # third-party GPU kernels, unused by the functions exercised here
    fi.sampling = types.ModuleType("flashinfer.sampling")
    sys.modules["flashinfer"] = fi
    sys.modules["flashinfer.sampling"] = fi.sampling
    la = importlib.import_module("krasis.linear_attention")
    layer = importlib.import_module("krasis.layer")
    attn = importlib.import_module("krasis.attention")
    return la, layer, attn


def make_la(la, nk, nv, dk, dv, kd, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, a=1.0: (torch.rand(*s, generator=g) * 2 - 1) * a
    m = la.GatedDeltaNetAttention.__new__(la.GatedDeltaNetAttention)     # __init__ needs a CUDA stream; fields set by hand
    m.cfg = types.SimpleNamespace(rms_norm_eps=1e-6)
    m.layer_idx, m.device = 0, torch.device("cpu")
    m.num_k_heads, m.num_v_heads, m.k_head_dim, m.v_head_dim, m.kernel_dim = nk, nv, dk, dv, kd
    m.key_dim, m.value_dim = nk * dk, nv * dv
    m.conv_dim = 2 * m.key_dim + m.value_dim
    m.head_ratio = nv // nk
    m.scale = 1.0 / (dk ** 0.5)
    m.hidden_size = m.value_dim
    qkvz_dim = 2 * m.key_dim + 2 * m.value_dim
    m.in_proj_qkvz = r(qkvz_dim, m.hidden_size, a=0.08)
    m.in_proj_ba = r(2 * nv, m.hidden_size, a=0.08)
    m.out_proj = torch.eye(m.value_dim, dtype=torch.bfloat16)            # identity: result == bf16(gated-norm output)
    m.conv1d_weight = r(m.conv_dim, 1, kd, a=0.5)
    m.A_log = r(nv, a=1.0)
    m.dt_bias = r(nv, a=1.0)
    m.norm_weight = r(dv, a=1.0) + 1.0
    m._conv_state = torch.zeros(1, m.conv_dim, kd, dtype=torch.float32)  # f32 state so the fixture is not bf16-limited
    m._recurrent_state = torch.zeros(1, nv, dk, dv, dtype=torch.float32)
    return m, g


def golden_la_recurrent(la):
    nk, nv, dk, dv, kd, steps = 2, 4, 128, 128, 4, 6
    m, g = make_la(la, nk, nv, dk, dv, kd, seed=11)
    hs, qkvzs, bas, outs = [], [], [], []
    for _ in range(steps):
        h = (torch.rand(1, m.hidden_size, generator=g) * 2 - 1)
        qkvzs.append(la._linear(h, m.in_proj_qkvz)[0].numpy().copy())
        bas.append(la._linear(h, m.in_proj_ba)[0].numpy().copy())
        out = m._forward_recurrent(h)                                     # [1, v_total] bf16-rounded by the reference (:566)
        outs.append(out.float()[0].numpy().copy())
    np.savez_compressed(
        os.path.join(OUT, "la_recurrent.npz"),
        dims=np.array([nk, nv, dk, dv, kd, steps], np.int32), eps=np.float32(1e-6), scale=np.float32(m.scale),
        qkvz=np.stack(qkvzs), ba=np.stack(bas), conv_w=m.conv1d_weight.squeeze(1).numpy(), a_log=m.A_log.numpy(),
        dt_bias=m.dt_bias.numpy(), norm_w=m.norm_weight.numpy(), out=np.stack(outs),
        final_conv_state=m._conv_state[0].numpy(), final_recur_state=m._recurrent_state[0].numpy())


def golden_la_steps(la):
    """the recurrent (M = 1) path step by step: conv state and recurrent state after EVERY token (both f32 in the fixture), so the oracle is pinned per
    step on the state it carries -- not only on bf16-rounded outputs and an end-of-sequence state"""
    nk, nv, dk, dv, kd, steps = 2, 4, 128, 128, 4, 12
    m, g = make_la(la, nk, nv, dk, dv, kd, seed=31)
    qkvzs, bas, convs, recs, outs = [], [], [], [], []
    for _ in range(steps):
        h = (torch.rand(1, m.hidden_size, generator=g) * 2 - 1)
        qkvzs.append(la._linear(h, m.in_proj_qkvz)[0].numpy().copy())
        bas.append(la._linear(h, m.in_proj_ba)[0].numpy().copy())
        out = m._forward_recurrent(h)
        outs.append(out.float()[0].numpy().copy())
        convs.append(m._conv_state[0].float().numpy().copy()); recs.append(m._recurrent_state[0].float().numpy().copy())
    np.savez_compressed(
        os.path.join(OUT, "la_steps.npz"),
        dims=np.array([nk, nv, dk, dv, kd, steps], np.int32), eps=np.float32(1e-6), scale=np.float32(m.scale),
        qkvz=np.stack(qkvzs), ba=np.stack(bas), conv_w=m.conv1d_weight.squeeze(1).numpy(), a_log=m.A_log.numpy(),
        dt_bias=m.dt_bias.numpy(), norm_w=m.norm_weight.numpy(), out=np.stack(outs), conv_state=np.stack(convs), recur_state=np.stack(recs))


def golden_la_chunked(la):
    """Prefill (chunked) form on M tokens from zero state; same module, same weights as a recurrent run -> both outputs stored."""
    nk, nv, dk, dv, kd, M = 2, 4, 128, 128, 4, 150
    m, g = make_la(la, nk, nv, dk, dv, kd, seed=23)
    h = (torch.rand(M, m.hidden_size, generator=g) * 2 - 1)
    qkvz = la._linear(h, m.in_proj_qkvz).numpy().copy()
    ba = la._linear(h, m.in_proj_ba).numpy().copy()
    out = m._forward_chunked(h).float().numpy()                           # f32 throughout; result bf16-rounded at :835
    np.savez_compressed(
        os.path.join(OUT, "la_chunked.npz"),
        dims=np.array([nk, nv, dk, dv, kd, M], np.int32), eps=np.float32(1e-6), scale=np.float32(m.scale),
        qkvz=qkvz.astype(np.float32), ba=ba.astype(np.float32), conv_w=m.conv1d_weight.squeeze(1).numpy(), a_log=m.A_log.numpy(),
        dt_bias=m.dt_bias.numpy(), norm_w=m.norm_weight.numpy(), out=out,
        final_conv_state=m._conv_state[0].float().numpy(), final_recur_state=m._recurrent_state[0].float().numpy())


def golden_routing(layer):
    cases = {}
    g = torch.Generator().manual_seed(5)
    for name, (E, H, k, scoring, norm, use_corr, use_bias, swiglu) in {
        "softmax_norm": (64, 96, 6, "softmax", True, False, False, 0.0),      # Qwen3 / QCN
        "softmax_raw": (64, 96, 6, "softmax", False, False, False, 0.0),      # V2-Lite
        "sigmoid_corr": (128, 64, 8, "sigmoid", True, True, False, 0.0),      # GLM-4.7 / V3 style
        "gptoss": (32, 64, 4, "softmax", False, False, True, 7.0),            # top-k on logits then softmax over k
    }.items():
        t = layer.TransformerLayer.__new__(layer.TransformerLayer)
        t.cfg = types.SimpleNamespace(num_experts_per_tok=k, swiglu_limit=swiglu, scoring_func=scoring, norm_topk_prob=norm)
        t.gate_weight = ((torch.rand(E, H, generator=g) * 2 - 1) * 0.3).to(torch.bfloat16)
        t.gate_bias = (torch.rand(E, generator=g) - 0.5) if use_bias else None
        t.e_score_correction_bias = (torch.rand(E, generator=g) * 0.2) if use_corr else None
        hidden = ((torch.rand(5, H, generator=g) * 2 - 1)).to(torch.bfloat16)
        ids, w = t.compute_routing(hidden)
        cases[name] = dict(E=E, H=H, k=k, scoring=scoring, norm=norm, swiglu=swiglu, gate=t.gate_weight.float().numpy(),
                           bias=None if t.gate_bias is None else t.gate_bias.numpy(),
                           corr=None if t.e_score_correction_bias is None else t.e_score_correction_bias.numpy(),
                           hidden=hidden.float().numpy(), ids=ids.numpy(), w=w.numpy())
    flat = {}
    for n, c in cases.items():
        for kk, v in c.items():
            if v is not None:
                flat[f"{n}.{kk}"] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "routing.npz"), **flat)


def golden_gqa_rope(attn):
    nh, nkv, hd, rot, theta, P = 8, 2, 128, 64, 10000.0, 9
    a = attn.GQAAttention.__new__(attn.GQAAttention)
    a.device, a.rotary_dim, a.rope_theta, a._rope_cos_sin = torch.device("cpu"), rot, theta, None
    g = torch.Generator().manual_seed(3)
    q = (torch.rand(P, nh, hd, generator=g) * 2 - 1)
    k = (torch.rand(P, nkv, hd, generator=g) * 2 - 1)
    v = (torch.rand(P, nkv, hd, generator=g) * 2 - 1)
    pos = torch.arange(P)
    cos, sin = a._get_rope_cos_sin(P)                                       # bf16 tables (attention.py:451-452)
    qr, kr = a._apply_rope(q, k, pos)                                       # f32 in, bf16 tables promote to f32
    # causal GQA attention of the LAST token over the roped keys (the flashinfer call's math, stated in plain torch)
    grp = nh // nkv
    out = torch.empty(nh, hd)
    for h in range(nh):
        s = (kr[:, h // grp, :] @ qr[P - 1, h, :]) / (hd ** 0.5)
        p = torch.softmax(s, dim=0)
        out[h] = p @ v[:, h // grp, :]
    np.savez_compressed(os.path.join(OUT, "gqa_rope.npz"), dims=np.array([nh, nkv, hd, rot, P], np.int32), q=q.numpy(), k=k.numpy(),
                        v=v.numpy(), cos=cos.float().numpy(), sin=sin.float().numpy(), q_rope=qr.float().numpy(), k_rope=kr.float().numpy(),
                        attn_last=out.numpy())


def golden_mla(attn):
    """MLA absorbed decode attention: the reference's own de-interleave + RoPE (attention.py:165-211) around the flashinfer MLA math
    (scores = q_abs.ckv + q_pe.kpe, softmax, sum p.ckv, then w_vc), stated in plain torch f32."""
    nh, klr, nd, rd, vhd, theta, P, eps = 4, 512, 128, 64, 128, 10000.0, 7, 1e-6
    a = attn.MLAAttention.__new__(attn.MLAAttention)
    a.device, a.qk_rope_dim, a.rope_theta, a._rope_cos_sin = torch.device("cpu"), rd, theta, None
    a.cfg = types.SimpleNamespace(rope_scaling={"factor": 40.0, "original_max_position_embeddings": 4096, "beta_fast": 32.0, "beta_slow": 1.0,
                                                "mscale_all_dim": 0.707})            # V2-Lite's YaRN settings
    g = torch.Generator().manual_seed(17)
    r = lambda *s, amp=1.0: (torch.rand(*s, generator=g) * 2 - 1) * amp
    kv_out = r(P, klr + rd)                      # kv_a_proj outputs per position
    q_full = r(P, nh, nd + rd)                   # q_proj outputs per position
    kv_a_norm = r(klr) * 0.5 + 1.0
    w_kc = r(nh, nd, klr, amp=0.1).to(torch.bfloat16).float()
    w_vc = r(nh, vhd, klr, amp=0.1).to(torch.bfloat16).float()
    cos, sin = a._get_rope_cos_sin(P)
    pos = torch.arange(P)
    q_pe, k_pe = a._apply_rope(q_full[:, :, nd:], kv_out[:, None, klr:], pos)
    ckv = kv_out[:, :klr]
    ckv = ckv * torch.rsqrt(ckv.pow(2).mean(-1, keepdim=True) + eps) * kv_a_norm
    q_abs = torch.einsum("hi,hij->hj", q_full[P - 1, :, :nd], w_kc)                      # [nh, klr]
    sm = 1.0 / ((nd + rd) ** 0.5)
    s = (q_abs @ ckv.T + q_pe[P - 1] @ k_pe[:, 0, :].T) * sm                               # [nh, P]
    p = torch.softmax(s, dim=-1)
    ao = p @ ckv                                                                           # [nh, klr]
    vp = torch.einsum("hoj,hj->ho", w_vc, ao)                                              # [nh, vhd]
    np.savez_compressed(os.path.join(OUT, "mla.npz"), dims=np.array([nh, klr, nd, rd, vhd, P], np.int32), eps=np.float32(eps),
                        sm_scale=np.float32(sm), kv_out=kv_out.numpy(), q_full=q_full.reshape(P, -1).numpy(), kv_a_norm=kv_a_norm.numpy(),
                        w_kc=w_kc.numpy(), w_vc=w_vc.numpy(), cos=cos.float().numpy(), sin=sin.float().numpy(),
                        k_pe_rope=k_pe[:, 0, :].float().numpy(), ckv_normed=ckv.numpy(), v_projected_last=vp.numpy())


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(1)
    la, layer, attn = import_reference()
    golden_la_recurrent(la)
    golden_la_steps(la)
    golden_la_chunked(la)
    golden_routing(layer)
    golden_gqa_rope(attn)
    golden_mla(attn)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
