"""CpuDecoder -- checkpoint -> decode graph, the counterpart of the reference's decode setup (python/krasis/decode_setup.py:22-1018)
together with the part of its weight loader that feeds it (python/krasis/weight_loader.py:157-427, python/krasis/config.py:291-436).

The reference copies every non-expert tensor of an already GPU-loaded `KrasisModel` to the CPU as f32 and quantizes it into the Rust
`CpuDecodeStore` (`init_weights`, decode_setup.py:120-230), then wires the decode graph (`_configure_decode_graph`, :824-1018).  Here there
is no intermediate GPU model object: the tensors are read straight from the HF safetensors checkpoint, given the SAME treatment
(BF16 -> f32, `(1 + w)` norm folding for qwen3_next, kv_b_proj -> w_kc / w_vc, gate||up fusion of the shared expert, column padding to 128,
YaRN RoPE tables) and handed to the same builder calls in the same order -- `store_weight_f32`, `store_norm_weight`, `store_route_weight`,
`configure_decode`, `add_decode_{la,gqa,mla}_layer`, `set_decode_layer_{moe,dense}`, `set_decode_rope`, `set_moe_store`, `finalize_decode` --
on `krasis_amd.CpuDecodeStore`, whose graph runs on the MI355X.  Routed experts go through `KrasisEngine.load` (weights/mod.rs:1181).

Per request the reference copies the GPU prefill's KV / recurrent state to the CPU (`prepare`, decode_setup.py:232-278); here the prompt
pass and the decode step share one state in HBM, so `prepare` only sizes and zeroes it.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from .decode_store import CpuDecodeStore
from .engine import KrasisEngine


# --------------------------------------------------------------------------------------------------------------------------------
# config.json (config.py:291-436: only what the decode graph needs)
# --------------------------------------------------------------------------------------------------------------------------------
@dataclass
class ModelArch:
    model_path: str
    model_type: str
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    vocab_size: int
    rms_norm_eps: float
    rope_theta: float
    rope_scaling: dict
    partial_rotary_factor: float
    tie_word_embeddings: bool
    norm_bias_one: bool
    # MoE
    n_routed_experts: int
    num_experts_per_tok: int
    moe_intermediate_size: int
    n_shared_experts: int
    shared_expert_intermediate_size: int
    first_k_dense_replace: int
    routed_scaling_factor: float
    scoring_func: str
    norm_topk_prob: bool
    swiglu_limit: float
    intermediate_size: int
    # MLA (None for GQA)
    q_lora_rank: Optional[int] = None
    kv_lora_rank: Optional[int] = None
    qk_nope_head_dim: Optional[int] = None
    qk_rope_head_dim: Optional[int] = None
    v_head_dim: Optional[int] = None
    gqa_head_dim: Optional[int] = None
    # hybrid linear attention
    layer_types: Optional[List[str]] = None
    linear_conv_kernel_dim: int = 4
    linear_key_head_dim: int = 128
    linear_num_key_heads: int = 16
    linear_value_head_dim: int = 128
    linear_num_value_heads: int = 32
    layers_prefix: str = "model"

    @property
    def is_mla(self) -> bool:
        return self.kv_lora_rank is not None

    @property
    def has_q_lora(self) -> bool:
        return self.is_mla and bool(self.q_lora_rank)

    @property
    def head_dim(self) -> int:
        if self.is_mla:
            return self.qk_nope_head_dim + self.qk_rope_head_dim
        return self.gqa_head_dim or self.hidden_size // self.num_attention_heads

    @property
    def rotary_dim(self) -> int:                                   # config.py:469-473
        return self.qk_rope_head_dim if self.is_mla else int(self.head_dim * self.partial_rotary_factor)

    def layer_type(self, i: int) -> str:
        return "full_attention" if self.layer_types is None else self.layer_types[i]

    def is_moe_layer(self, i: int) -> bool:                        # config.py:525
        return self.n_routed_experts > 0 and i >= self.first_k_dense_replace

    @property
    def effective_shared_expert_intermediate(self) -> int:         # config.py:517-523
        return self.shared_expert_intermediate_size or self.n_shared_experts * self.moe_intermediate_size

    @staticmethod
    def from_model_path(model_path: str, weight_names: Optional[Sequence[str]] = None) -> "ModelArch":
        raw = json.load(open(os.path.join(model_path, "config.json")))
        cfg = raw.get("text_config", raw.get("language_config", raw))
        is_mla = "kv_lora_rank" in cfg
        if "first_k_dense_replace" in cfg:
            first_k = cfg["first_k_dense_replace"]
        elif "decoder_sparse_step" in cfg:
            first_k = 0 if cfg["decoder_sparse_step"] <= 1 else cfg["decoder_sparse_step"]
        else:
            first_k = 0
        n_layers = cfg["num_hidden_layers"]
        fai = cfg.get("full_attention_interval", 0)
        if "layer_types" in cfg:
            layer_types = list(cfg["layer_types"])
        elif fai > 0:
            layer_types = ["full_attention" if (i + 1) % fai == 0 else "linear_attention" for i in range(n_layers)]
        else:
            layer_types = None
        arch = cfg.get("model_type", "")
        n_shared = cfg.get("n_shared_experts", 0) or 0
        shared_inter = cfg.get("shared_expert_intermediate_size", 0) or 0
        if n_shared == 0 and shared_inter > 0:
            n_shared = 1
        rope_params = cfg.get("rope_parameters", {}) or {}
        names = list(weight_names or [])
        tie_default = not any("lm_head.weight" in k for k in names) if names else True
        prefix = "model"
        for k in names:                                               # config.py:167-230: everything before ".layers."
            if ".layers." in k and not k.split(".layers.")[0].endswith("mtp"):
                prefix = k.split(".layers.")[0]
                break
        return ModelArch(
            model_path=model_path, model_type=arch, hidden_size=cfg["hidden_size"], num_hidden_layers=n_layers,
            num_attention_heads=cfg["num_attention_heads"], num_key_value_heads=cfg.get("num_key_value_heads", cfg["num_attention_heads"]),
            vocab_size=cfg["vocab_size"], rms_norm_eps=cfg.get("rms_norm_eps", 1e-6),
            rope_theta=cfg.get("rope_theta", rope_params.get("rope_theta", 10000.0)), rope_scaling=cfg.get("rope_scaling") or {},
            partial_rotary_factor=cfg.get("partial_rotary_factor", rope_params.get("partial_rotary_factor", 1.0)),
            tie_word_embeddings=cfg.get("tie_word_embeddings", raw.get("tie_word_embeddings", tie_default)),
            norm_bias_one=arch in ("qwen3_next", "qwen3_5_moe_text"),
            n_routed_experts=cfg.get("n_routed_experts", cfg.get("num_experts", cfg.get("num_local_experts", 0))),
            num_experts_per_tok=cfg.get("num_experts_per_tok", cfg.get("experts_per_token", 0)),
            moe_intermediate_size=cfg.get("moe_intermediate_size", cfg.get("intermediate_size", 0)), n_shared_experts=n_shared,
            shared_expert_intermediate_size=shared_inter, first_k_dense_replace=first_k,
            routed_scaling_factor=cfg.get("routed_scaling_factor", 1.0), scoring_func=cfg.get("scoring_func", "softmax"),
            norm_topk_prob=cfg.get("norm_topk_prob", arch == "qwen3_5_moe_text"), swiglu_limit=cfg.get("swiglu_limit", 0.0) or 0.0,
            intermediate_size=cfg.get("intermediate_size", cfg.get("moe_intermediate_size", 0)),
            q_lora_rank=cfg.get("q_lora_rank") if is_mla else None, kv_lora_rank=cfg.get("kv_lora_rank") if is_mla else None,
            qk_nope_head_dim=cfg.get("qk_nope_head_dim") if is_mla else None, qk_rope_head_dim=cfg.get("qk_rope_head_dim") if is_mla else None,
            v_head_dim=cfg.get("v_head_dim") if is_mla else None, gqa_head_dim=cfg.get("head_dim") if not is_mla else None,
            layer_types=layer_types, linear_conv_kernel_dim=cfg.get("linear_conv_kernel_dim", 4),
            linear_key_head_dim=cfg.get("linear_key_head_dim", 128), linear_num_key_heads=cfg.get("linear_num_key_heads", 16),
            linear_value_head_dim=cfg.get("linear_value_head_dim", 128), linear_num_value_heads=cfg.get("linear_num_value_heads", 32),
            layers_prefix=prefix)


# --------------------------------------------------------------------------------------------------------------------------------
# safetensors access (weight_loader.py:110-155): every tensor as float32 numpy, BF16 widened exactly
# --------------------------------------------------------------------------------------------------------------------------------
class CheckpointReader:
    def __init__(self, model_dir: str):
        from .weight_store import _weight_map
        self.dir = model_dir
        self.weight_map: Dict[str, str] = _weight_map(model_dir)
        self._handles: Dict[str, object] = {}

    def has(self, name: str) -> bool:
        return name in self.weight_map

    def f32(self, name: str) -> np.ndarray:
        import torch
        from safetensors import safe_open
        shard = self.weight_map.get(name)
        if shard is None:
            raise IOError(f"tensor {name} not found in the safetensors index")
        if shard not in self._handles:
            self._handles[shard] = safe_open(os.path.join(self.dir, shard), framework="pt")
        return np.ascontiguousarray(self._handles[shard].get_tensor(name).to(torch.float32).numpy())

    def close(self) -> None:
        self._handles.clear()


def quantize_to_int8_roundtrip(w: np.ndarray) -> np.ndarray:
    """attention="int8" of the reference's GPU side: per-channel symmetric INT8 (weight_loader.py:25-43) immediately widened again the way
    CpuDecoder._to_cpu_f32 does (decode_setup.py:104-114): w_int8 * bf16(scale)."""
    import torch
    t = torch.from_numpy(w)
    amax = t.abs().amax(dim=1).clamp(min=1e-10)
    scale = amax / 127.0
    q = (t / scale.unsqueeze(1)).round().clamp(-128, 127).to(torch.int8)
    return np.ascontiguousarray((q.float() * scale.to(torch.bfloat16).float().unsqueeze(1)).numpy())


def _pad_cols(w: np.ndarray, align: int = 128) -> np.ndarray:
    """decode_setup.py:586-597: cols padded with zeros to a multiple of the group size"""
    rows, cols = w.shape
    if cols % align == 0:
        return np.ascontiguousarray(w, np.float32)
    out = np.zeros((rows, (cols + align - 1) // align * align), np.float32)
    out[:, :cols] = w
    return out


class CpuDecoder:
    """Usage (decode_setup.py:25-33, with a checkpoint directory in place of the loaded model object):

        dec = CpuDecoder(model_dir)            # optionally engine=..., decode_bits=4, expert_bits=4, kv_fp8=False
        dec.init_weights()                     # once
        dec.prepare(max_seq=4096)              # per request: fresh KV / recurrent state
        first = dec.prefill(prompt_ids)        # GPU prompt pass (replaces model.server_prefill + the state hand-off)
        out = dec._store.generate_batch(first, len(prompt_ids), 64)
    """

    def __init__(self, model_dir: str, engine: Optional[KrasisEngine] = None, decode_bits: int = 4, expert_bits: int = 4, attention_quant: str = "bf16",
                 kv_fp8: bool = False, device: int = 0, max_layers: Optional[int] = None):
        if decode_bits not in (4, 8):
            raise ValueError(f"num_bits must be 4 or 8, got {decode_bits}")
        if attention_quant not in ("bf16", "int8"):
            raise ValueError(f"attention quantization {attention_quant!r} unknown (bf16 | int8)")
        self.model_dir = model_dir
        self._reader = CheckpointReader(model_dir)
        self.cfg = ModelArch.from_model_path(model_dir, list(self._reader.weight_map))
        if max_layers:
            self.cfg.num_hidden_layers = min(self.cfg.num_hidden_layers, max_layers)
            if self.cfg.layer_types is not None:
                self.cfg.layer_types = self.cfg.layer_types[: self.cfg.num_hidden_layers]
        self.engine = engine
        self._device = device
        self._expert_bits = expert_bits
        self._attn_quant = attention_quant
        self._kv_fp8 = kv_fp8
        # norm_bias_one=False: the (1 + w) of qwen3_next is folded into the stored norm weights (decode_setup.py:45-49)
        self._store = CpuDecodeStore(group_size=128, parallel=True, norm_bias_one=False, device=device)
        self._decode_bits = decode_bits
        self._layers: List[dict] = []
        self._keep: list = []            # host arrays whose addresses were handed to the store
        self._weights_initialized = False
        self._max_kv_seq = 0
        self._max_rope_seq = 0
        self._rope_cos = self._rope_sin = self._mla_rope_cos = self._mla_rope_sin = None

    # ------------------------------------------------------------------ helpers
    def _t(self, name: str) -> np.ndarray:
        return self._reader.f32(name)

    def _proj(self, name: str) -> np.ndarray:
        w = self._t(name)
        return quantize_to_int8_roundtrip(w) if self._attn_quant == "int8" else w

    def _norm(self, name: str) -> np.ndarray:
        w = self._t(name)
        return np.ascontiguousarray(w + np.float32(1.0)) if self.cfg.norm_bias_one else w      # weight_loader.py:168-169, :258-266, :286-289

    def _qw(self, w: np.ndarray) -> int:
        """decode_setup.py:578-604: pad cols, quantize into the store, return the weight id"""
        if w.ndim == 1:
            w = w[None, :]
        w = _pad_cols(w)
        return self._store.store_weight_f32(w.ctypes.data, w.shape[0], w.shape[1], self._decode_bits)

    def _nw(self, w: np.ndarray) -> int:
        w = np.ascontiguousarray(w, np.float32)
        return self._store.store_norm_weight(w.ctypes.data, w.size)

    # ------------------------------------------------------------------ one-time initialisation (decode_setup.py:120-230)
    def init_weights(self, max_rope_seq: int = 8192) -> None:
        cfg = self.cfg
        P = cfg.layers_prefix
        R = self._reader
        # ---- routed experts: KrasisEngine.load (moe.rs:1538 -> weights/mod.rs:1181), skipped when the caller brings a loaded engine
        if self.engine is None and cfg.n_routed_experts > 0:
            n_moe = cfg.num_hidden_layers - cfg.first_k_dense_replace
            self.engine = KrasisEngine(device=self._device)
            self.engine.load(self.model_dir, num_bits=self._expert_bits, max_layers=n_moe)
        # ---- global weights
        emb = self._t(f"{P}.embed_tokens.weight"); self._embedding = emb; self._keep.append(emb)
        fin = self._norm(f"{P}.norm.weight")
        lm_name = "lm_head.weight"
        if not R.has(lm_name) and P != "model":
            lm_name = f"{P.rsplit('.', 1)[0]}.lm_head.weight"                       # weight_loader.py:178-183
        lm = emb if (cfg.tie_word_embeddings and not R.has(lm_name)) else self._proj(lm_name)
        self._lm_head_wid = self._qw(lm)
        # ---- per layer: attention + MLP tensors (decode_setup.py:150-186), quantized as they are read (:502-576)
        for li in range(cfg.num_hidden_layers):
            lt = cfg.layer_type(li)
            ld = {"type": lt, "is_moe": cfg.is_moe_layer(li),
                  "input_norm": self._norm(f"{P}.layers.{li}.input_layernorm.weight"),
                  "post_attn_norm": self._norm(f"{P}.layers.{li}.post_attention_layernorm.weight")}
            if lt == "linear_attention":
                ld["attn"] = self._init_linear_attention(li)
            elif cfg.is_mla:
                ld["attn"] = self._init_mla(li)
            else:
                ld["attn"] = self._init_gqa(li)
            if ld["is_moe"]:
                self._prepare_moe(li, ld)
            else:
                self._prepare_dense_mlp(li, ld)
            self._layers.append(ld)
        # ---- norms
        self._final_norm_id = self._nw(fin)
        for ld in self._layers:
            for k in ("input_norm", "post_attn_norm"):
                ld[f"{k}_id"] = self._nw(ld[k])
        self._max_rope_seq = max_rope_seq
        self._init_rope()
        self._configure_decode_graph()
        self._weights_initialized = True
        R.close()

    def _init_linear_attention(self, li: int) -> dict:              # weight_loader.py:369-426 + decode_setup.py:280-308
        cfg = self.cfg; p = f"{cfg.layers_prefix}.layers.{li}.linear_attn"; R = self._reader
        nk, dk, nv, dv = cfg.linear_num_key_heads, cfg.linear_key_head_dim, cfg.linear_num_value_heads, cfg.linear_value_head_dim
        hr = nv // nk
        if R.has(f"{p}.in_proj_qkvz.weight"):
            qkvz = self._proj(f"{p}.in_proj_qkvz.weight"); ba = self._proj(f"{p}.in_proj_ba.weight")
        else:   # separate format (Qwen3.5): interleave per key-head group into the fused layout
            qkv, z, b, a = (self._t(f"{p}.in_proj_{s}.weight") for s in ("qkv", "z", "b", "a"))
            kd_, parts, bap = nk * dk, [], []
            for i in range(nk):
                parts += [qkv[i * dk:(i + 1) * dk], qkv[kd_ + i * dk: kd_ + (i + 1) * dk], qkv[2 * kd_ + i * hr * dv: 2 * kd_ + (i + 1) * hr * dv],
                          z[i * hr * dv:(i + 1) * hr * dv]]
                bap += [b[i * hr:(i + 1) * hr], a[i * hr:(i + 1) * hr]]
            qkvz, ba = np.concatenate(parts, 0), np.concatenate(bap, 0)
        conv_w = self._t(f"{p}.conv1d.weight")
        if conv_w.ndim == 3:
            conv_w = conv_w[:, 0, :]
        norm_w = self._t(f"{p}.norm.weight").reshape(-1)
        if norm_w.size == dv:
            norm_w = np.tile(norm_w, nv)                                 # decode_setup.py:852-858
        a = {"in_proj_qkvz_wid": self._qw(qkvz), "in_proj_ba_wid": self._qw(ba), "out_proj_wid": self._qw(self._proj(f"{p}.out_proj.weight")),
             "conv1d_weight": np.ascontiguousarray(conv_w), "A_log": self._t(f"{p}.A_log"), "dt_bias": self._t(f"{p}.dt_bias"),
             "norm_weight": np.ascontiguousarray(norm_w), "num_k_heads": nk, "num_v_heads": nv, "k_head_dim": dk, "v_head_dim": dv, "head_ratio": hr,
             "kernel_dim": cfg.linear_conv_kernel_dim, "scale": 1.0 / math.sqrt(dk)}
        self._keep += [a["conv1d_weight"], a["A_log"], a["dt_bias"], a["norm_weight"]]
        return a

    def _init_gqa(self, li: int) -> dict:                          # weight_loader.py:235-273 + decode_setup.py:310-330
        cfg = self.cfg; p = f"{cfg.layers_prefix}.layers.{li}.self_attn"; R = self._reader
        hd, nh, nkv = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
        for proj in ("q_proj", "k_proj", "v_proj", "o_proj"):
            if R.has(f"{p}.{proj}.bias"):
                raise ValueError(f"{p}.{proj}.bias: attention biases are not wired into the decode graph")
        q = self._proj(f"{p}.q_proj.weight")
        a = {"q_proj_wid": self._qw(q), "k_proj_wid": self._qw(self._proj(f"{p}.k_proj.weight")), "v_proj_wid": self._qw(self._proj(f"{p}.v_proj.weight")),
             "o_proj_wid": self._qw(self._proj(f"{p}.o_proj.weight")),
             "q_norm": self._norm(f"{p}.q_norm.weight") if R.has(f"{p}.q_norm.weight") else None,
             "k_norm": self._norm(f"{p}.k_norm.weight") if R.has(f"{p}.k_norm.weight") else None,
             "gated": q.shape[0] == 2 * nh * hd,                         # attention.py:403: q_proj carries [q | gate] per head
             "num_heads": nh, "num_kv_heads": nkv, "head_dim": hd, "sm_scale": 1.0 / math.sqrt(hd)}
        self._keep += [x for x in (a["q_norm"], a["k_norm"]) if x is not None]
        return a

    def _init_mla(self, li: int) -> dict:                          # weight_loader.py:196-233 + decode_setup.py:332-358
        cfg = self.cfg; p = f"{cfg.layers_prefix}.layers.{li}.self_attn"
        nh, nd, rd, vhd, klr = cfg.num_attention_heads, cfg.qk_nope_head_dim, cfg.qk_rope_head_dim, cfg.v_head_dim, cfg.kv_lora_rank
        kv_b = self._t(f"{p}.kv_b_proj.weight").reshape(nh, nd + vhd, klr)
        to_bf16 = lambda x: np.ascontiguousarray((np.ascontiguousarray(x, np.float32).view(np.uint32) >> 16).astype(np.uint16))   # checkpoint values are bf16: exact
        sm = 1.0 / math.sqrt(nd + rd)
        rs = cfg.rope_scaling
        if rs and rs.get("factor", 1.0) > 1.0:                            # attention.py:80-89: YaRN mscale^2
            mscale = 0.1 * rs.get("mscale_all_dim", 0) * math.log(rs["factor"]) + 1.0
            sm *= mscale * mscale
        a = {"kv_a_proj_wid": self._qw(self._proj(f"{p}.kv_a_proj_with_mqa.weight")), "o_proj_wid": self._qw(self._proj(f"{p}.o_proj.weight")),
             "kv_a_norm": self._t(f"{p}.kv_a_layernorm.weight"), "w_kc": to_bf16(kv_b[:, :nd, :]), "w_vc": to_bf16(kv_b[:, nd:, :]),
             "num_heads": nh, "kv_lora_rank": klr, "qk_nope_dim": nd, "qk_rope_dim": rd, "v_head_dim": vhd, "sm_scale": sm}
        if cfg.has_q_lora:
            a["q_a_proj_wid"] = self._qw(self._proj(f"{p}.q_a_proj.weight")); a["q_b_proj_wid"] = self._qw(self._proj(f"{p}.q_b_proj.weight"))
            a["q_a_norm"] = self._t(f"{p}.q_a_layernorm.weight")
        else:
            a["q_proj_wid"] = self._qw(self._proj(f"{p}.q_proj.weight"))
        self._keep += [a["kv_a_norm"], a["w_kc"], a["w_vc"]] + ([a["q_a_norm"]] if "q_a_norm" in a else [])
        return a

    def _prepare_moe(self, li: int, ld: dict) -> None:             # weight_loader.py:307-367 + decode_setup.py:360-377, :541-567
        cfg = self.cfg; R = self._reader; P = cfg.layers_prefix
        gp = f"{P}.layers.{li}.mlp.gate"
        if not R.has(f"{gp}.weight"):
            gp = f"{P}.layers.{li}.mlp.router"
        gate = self._t(f"{gp}.weight")
        bias = self._t(f"{gp}.bias") if R.has(f"{gp}.bias") else None
        esc = self._t(f"{gp}.e_score_correction_bias") if R.has(f"{gp}.e_score_correction_bias") else None
        ld["_route_id"] = self._store.store_route_weight(gate.ctypes.data, gate.shape[0], gate.shape[1], bias.ctypes.data if bias is not None else None,
                                                         bias.size if bias is not None else 0, esc.ctypes.data if esc is not None else None,
                                                         esc.size if esc is not None else 0)
        sp = f"{P}.layers.{li}.mlp.shared_experts"
        if not R.has(f"{sp}.gate_proj.weight"):
            sp = f"{P}.layers.{li}.mlp.shared_expert"
        if R.has(f"{sp}.gate_proj.weight"):
            gu = np.concatenate([self._proj(f"{sp}.gate_proj.weight"), self._proj(f"{sp}.up_proj.weight")], 0)    # gate || up (decode_setup.py:478-481)
            se = {"gate_up_proj_wid": self._qw(gu), "down_proj_wid": self._qw(self._proj(f"{sp}.down_proj.weight"))}
            gname = f"{P}.layers.{li}.mlp.shared_expert_gate.weight"
            if R.has(gname):
                se["gate_wid"] = self._qw(self._proj(gname))
            ld["shared_expert"] = se

    def _prepare_dense_mlp(self, li: int, ld: dict) -> None:       # weight_loader.py:292-305 + decode_setup.py:379-387
        p = f"{self.cfg.layers_prefix}.layers.{li}.mlp"
        ld["dense_mlp"] = {"gate_proj_wid": self._qw(self._proj(f"{p}.gate_proj.weight")), "up_proj_wid": self._qw(self._proj(f"{p}.up_proj.weight")),
                           "down_proj_wid": self._qw(self._proj(f"{p}.down_proj.weight"))}

    # ------------------------------------------------------------------ RoPE tables (decode_setup.py:715-757), f32 like the reference's torch code
    def _init_rope(self) -> None:
        import torch
        cfg = self.cfg; max_pos = self._max_rope_seq
        if cfg.is_mla:
            dim = cfg.qk_rope_head_dim
            freqs = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2).float() / dim))
            rc = cfg.rope_scaling
            if rc and rc.get("factor", 1.0) > 1.0:
                factor = rc.get("factor", 1.0); original_max = rc.get("original_max_position_embeddings", 4096)
                beta_fast = rc.get("beta_fast", 32.0); beta_slow = rc.get("beta_slow", 1.0)
                low = max(0, math.floor(dim * math.log(original_max / (beta_fast * 2 * math.pi)) / (2 * math.log(cfg.rope_theta))))
                high = min(dim // 2 - 1, math.ceil(dim * math.log(original_max / (beta_slow * 2 * math.pi)) / (2 * math.log(cfg.rope_theta))))
                ramp = torch.clamp((torch.arange(dim // 2).float() - low) / max(high - low, 0.001), 0, 1)
                inv_freq_mask = 1.0 - ramp
                freqs = (freqs / factor) * (1 - inv_freq_mask) + freqs * inv_freq_mask
            f = torch.outer(torch.arange(max_pos, dtype=torch.float32), freqs)
            self._mla_rope_cos = np.ascontiguousarray(f.cos().numpy()); self._mla_rope_sin = np.ascontiguousarray(f.sin().numpy())
        else:
            dim = cfg.rotary_dim
            freqs = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2).float() / dim))
            f = torch.outer(torch.arange(max_pos, dtype=torch.float32), freqs)
            self._rope_cos = np.ascontiguousarray(f.cos().numpy()); self._rope_sin = np.ascontiguousarray(f.sin().numpy())

    # ------------------------------------------------------------------ graph (decode_setup.py:824-1018), same calls in the same order
    def _configure_decode_graph(self) -> None:
        cfg = self.cfg; store = self._store
        topk = cfg.num_experts_per_tok if cfg.num_experts_per_tok > 0 else 0
        if topk > 0:
            sf = 2 if cfg.swiglu_limit > 0 else (0 if cfg.scoring_func == "sigmoid" else 1)
        else:
            sf = 0
        store.configure_decode(cfg.hidden_size, cfg.num_hidden_layers, cfg.rms_norm_eps, self._final_norm_id, self._lm_head_wid, cfg.vocab_size, topk, sf,
                               cfg.norm_topk_prob, cfg.routed_scaling_factor, self._embedding.ctypes.data)
        first_k = cfg.first_k_dense_replace
        for li, ld in enumerate(self._layers):
            a = ld["attn"]
            if ld["type"] == "linear_attention":
                store.add_decode_la_layer(ld["input_norm_id"], ld["post_attn_norm_id"], a["in_proj_qkvz_wid"], a["in_proj_ba_wid"], a["out_proj_wid"],
                                          a["conv1d_weight"].ctypes.data, a["A_log"].ctypes.data, a["dt_bias"].ctypes.data, a["norm_weight"].ctypes.data,
                                          a["num_k_heads"], a["num_v_heads"], a["k_head_dim"], a["v_head_dim"], a["head_ratio"], a["kernel_dim"], a["scale"])
            elif cfg.is_mla:
                qn = a.get("q_a_norm")
                store.add_decode_mla_layer(ld["input_norm_id"], ld["post_attn_norm_id"], a["kv_a_proj_wid"], a["o_proj_wid"], a.get("q_proj_wid"),
                                           a.get("q_a_proj_wid"), a.get("q_b_proj_wid"), a["w_kc"].ctypes.data, a["w_kc"].size, a["w_vc"].ctypes.data,
                                           a["w_vc"].size, a["kv_a_norm"].ctypes.data, a["kv_a_norm"].size, qn.ctypes.data if qn is not None else 0,
                                           qn.size if qn is not None else 0, self._mla_rope_cos.ctypes.data, self._mla_rope_sin.ctypes.data,
                                           self._mla_rope_cos.size, self._max_rope_seq, a["num_heads"], a["kv_lora_rank"], a["qk_nope_dim"],
                                           a["qk_rope_dim"], a["v_head_dim"], a["sm_scale"])
            else:
                qn, kn = a["q_norm"], a["k_norm"]
                store.add_decode_gqa_layer(ld["input_norm_id"], ld["post_attn_norm_id"], a["q_proj_wid"], a["k_proj_wid"], a["v_proj_wid"], a["o_proj_wid"],
                                           qn.ctypes.data if qn is not None else 0, qn.size if qn is not None else 0,
                                           kn.ctypes.data if kn is not None else 0, kn.size if kn is not None else 0, a["gated"], a["num_heads"],
                                           a["num_kv_heads"], a["head_dim"], a["sm_scale"])
            moe_layer_idx = li - first_k if li >= first_k else None
            if ld["is_moe"] and "_route_id" in ld:
                se = ld.get("shared_expert", {})
                store.set_decode_layer_moe(li, ld["_route_id"], moe_layer_idx if moe_layer_idx is not None else 0, se.get("gate_up_proj_wid"),
                                           se.get("down_proj_wid"), se.get("gate_wid"))
            elif "dense_mlp" in ld:
                d = ld["dense_mlp"]
                store.set_decode_layer_dense(li, d["gate_proj_wid"], d["up_proj_wid"], d["down_proj_wid"])
        if self._rope_cos is not None:
            store.set_decode_rope(self._rope_cos.ctypes.data, self._rope_sin.ctypes.data, self._rope_cos.shape[-1], self._rope_cos.shape[0])
        if topk > 0 and self.engine is not None:
            store.set_moe_store(self.engine)
        store.finalize_decode()
        store.repack_to_tiled()
        if self._kv_fp8:
            store.set_kv_dtype(True)

    # ------------------------------------------------------------------ per request (decode_setup.py:232-278)
    def prepare(self, max_seq: int = 4096) -> None:
        assert self._weights_initialized, "Call init_weights() first (at model load time)"
        if max_seq > self._max_rope_seq:
            raise ValueError(f"max_seq {max_seq} exceeds the RoPE tables ({self._max_rope_seq}); pass max_rope_seq to init_weights")
        self._max_kv_seq = max_seq
        self._store.reset_decode_state(max_seq)

    def prefill(self, tokens: Sequence[int], start_pos: int = 0) -> int:
        return self._store.prefill(list(tokens), start_pos)

    def generate(self, prompt: Sequence[int], max_new_tokens: int, temperature: float = 0.0, top_k: int = 0, top_p: float = 1.0,
                 stop_ids: Sequence[int] = (), presence_penalty: float = 0.0, rng_seed: int = 0) -> List[int]:
        """prompt pass + generate_batch (the request path of server.py: prefill, then decode until a stop id)"""
        self.prepare(min(self._max_rope_seq, len(prompt) + max_new_tokens + 1))
        first = self.prefill(prompt)
        if max_new_tokens <= 1 or first in stop_ids:
            return [first]
        return [first] + self._store.generate_batch(first, len(prompt), max_new_tokens - 1, temperature, top_k, top_p, stop_ids, presence_penalty, rng_seed)
