"""Host loaders: checkpoint files -> resident HBM expert store.

Mirrors the reference's `WeightStore::load_from_hf` (src/weights/mod.rs:1181-1560) and `load_from_gguf` (:3251-3583) at the level the
engine API needs:

* `MoeConfig.from_json`        -- `ModelConfig::from_json` (weights/mod.rs:51-181): same key fall-backs (n_routed_experts | num_experts |
                                  num_local_experts, num_experts_per_tok | experts_per_token, first_k_dense_replace | decoder_sparse_step, ...).
* `load_from_hf`               -- BF16 safetensors: per expert `<prefix>.layers.L.mlp.experts.E.{gate,up,down}_proj.weight`, shared expert
                                  `<prefix>.layers.L.mlp.shared_experts.*` (:1506,1554); the tensors go to the GPU as BF16 and are quantized THERE
                                  with the reference's INT4 / INT8 rule (`kr_upload_expert_bf16`, csrc/kr_quant.hip == weights/marlin.rs:65,145).
                                  No host-side quantization, no on-disk cache: a 288 GB GPU holds every expert, and the GPU quantizes faster than
                                  the cache file could be read.
* `GgufFile`, `load_from_gguf` -- GGUF v2/v3 header + tensor table (src/gguf.rs:320-449), expert tensor naming `blk.L.ffn_{gate,up,down}_exps`,
                                  `blk.L.ffn_{gate,up,down}.E`, `..._shexp` (:488-526); `gguf_native=True` uploads the raw blocks
                                  (`kr_upload_expert_gguf`, the reference's GgufExpertWeights store).

The reference's KRAS disk-cache formats (weights/mod.rs:856-934) and NUMA placement are not reproduced: they exist to avoid re-quantizing on
the CPU and to place pages near cores.
"""
from __future__ import annotations

import json
import mmap
import os
import struct
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from ._lib import check

GGUF_MAGIC = 0x46554747            # "GGUF" little-endian (gguf.rs:10)
GGUF_DEFAULT_ALIGNMENT = 32
# ggml type id -> (block elements, block bytes)   (gguf.rs:42-72)
GGML_BLOCK = {0: (1, 4), 1: (1, 2), 30: (1, 2), 2: (32, 18), 3: (32, 20), 6: (32, 22), 7: (32, 24), 8: (32, 34), 9: (32, 40), 10: (256, 84),
              11: (256, 110), 12: (256, 144), 13: (256, 176), 14: (256, 210), 15: (256, 276)}
NATIVE_TYPES = {2, 6, 8, 12, 14}   # Q4_0, Q5_0, Q8_0, Q4_K, Q6_K: block kernels exist (csrc/kr_gguf.hip == gguf_kernels.rs)


@dataclass
class MoeConfig:
    hidden_size: int
    moe_intermediate_size: int
    n_routed_experts: int
    num_experts_per_tok: int
    num_hidden_layers: int
    first_k_dense_replace: int = 0
    n_shared_experts: int = 0
    routed_scaling_factor: float = 1.0
    swiglu_limit: float = 0.0
    activation_alpha: float = 0.0

    @staticmethod
    def from_json(cfg_path: str, weight_map: Optional[Dict[str, str]] = None) -> "MoeConfig":
        raw = json.load(open(cfg_path))
        # VL wrappers nest the language-model config: text_config (Kimi K2.5) or language_config (DeepSeek-VL2), weights/mod.rs:82-92
        cfg = raw.get("text_config") or raw.get("language_config") or raw
        def need(*keys):
            for k in keys:
                if cfg.get(k) is not None:
                    return cfg[k]
            raise ValueError("Missing " + " or ".join(keys))
        layers = cfg.get("num_hidden_layers")
        if layers is None:
            if not weight_map:
                raise ValueError("Missing num_hidden_layers (not in config and no index to infer from)")
            layers = 1 + max(int(k.split(".layers.")[1].split(".")[0]) for k in weight_map if ".layers." in k and ".mlp.experts." in k)
        if cfg.get("first_k_dense_replace") is not None:
            first = int(cfg["first_k_dense_replace"])
        elif cfg.get("decoder_sparse_step") is not None:
            if int(cfg["decoder_sparse_step"]) > 1:
                raise ValueError(f"decoder_sparse_step={cfg['decoder_sparse_step']} (interleaved MoE) not yet supported")
            first = 0
        else:
            first = 0
        swiglu = float(cfg.get("swiglu_limit") or 0.0)
        return MoeConfig(int(need("hidden_size")), int(need("moe_intermediate_size", "intermediate_size")),
                         int(need("n_routed_experts", "num_experts", "num_local_experts")), int(need("num_experts_per_tok", "experts_per_token")),
                         int(layers), first, int(cfg.get("n_shared_experts") or 0), float(cfg.get("routed_scaling_factor") or 1.0), swiglu,
                         1.702 if swiglu > 0 else 0.0)


def detect_expert_prefix(weight_map: Dict[str, str]) -> str:
    """weights/mod.rs:4648 -- everything before `.layers.` of an expert tensor, skipping MTP heads."""
    for key in weight_map:
        pos = key.find(".layers.")
        if pos >= 0 and ".mlp.experts." in key:
            prefix = key[:pos]
            if prefix == "mtp" or prefix.endswith(".mtp"):
                continue
            return prefix
    raise ValueError("Could not detect expert weight prefix from safetensors index")


def _weight_map(model_dir: str) -> Dict[str, str]:
    idx = os.path.join(model_dir, "model.safetensors.index.json")
    if os.path.exists(idx):
        return json.load(open(idx))["weight_map"]
    single = os.path.join(model_dir, "model.safetensors")
    if not os.path.exists(single):
        raise IOError(f"no model.safetensors.index.json / model.safetensors under {model_dir}")
    from safetensors import safe_open
    with safe_open(single, framework="pt") as f:
        return {k: "model.safetensors" for k in f.keys()}


def load_from_hf(engine, model_dir: str, num_bits: int = 4, w2_bits: Optional[int] = None, max_layers: Optional[int] = None,
                 start_layer: Optional[int] = None, use_cache: bool = True, write_cache: bool = False) -> MoeConfig:
    """KrasisEngine.load for BF16 safetensors (moe.rs:1538 -> weights/mod.rs:1181).  Configures the engine and fills layers
    [start_layer, start_layer + max_layers) of the MoE stack (MoE layer m = model layer m + first_k_dense_replace).
    use_cache: an expert cache the REFERENCE left on disk for this model (`~/.krasis/cache/<model>/experts_cpu_int{bits}_g128.bin`, else the Marlin
    file; weights/mod.rs:1226-1370 tries them in that spirit) is loaded instead of re-quantizing -- a stale or foreign file (hash / shape / size
    mismatch) is skipped with its reason kept in `engine.cache_note`.  write_cache: after quantizing a WHOLE model, write the version-4 file."""
    import torch
    from safetensors import safe_open

    from .engine import ModelConfig
    if num_bits not in (4, 8):
        raise ValueError(f"cpu_num_bits must be 4 or 8, got {num_bits}")
    w2_bits = w2_bits or num_bits
    wm = _weight_map(model_dir)
    if any(k.endswith(".weight_packed") for k in wm):
        raise ValueError("pre-quantized compressed-tensors checkpoints are not handled by this loader (BF16 or GGUF only)")
    cfg = MoeConfig.from_json(os.path.join(model_dir, "config.json"), wm)
    if cfg.hidden_size % 128 or cfg.moe_intermediate_size % 128:
        raise ValueError(f"hidden_size ({cfg.hidden_size}) must be divisible by group_size (128)")
    n_moe = cfg.num_hidden_layers - cfg.first_k_dense_replace
    start = start_layer or 0
    count = min(max_layers, n_moe - start) if max_layers else n_moe - start
    if count <= 0:
        raise ValueError(f"start_layer {start} / max_layers {max_layers} select no MoE layer (model has {n_moe})")
    engine.configure(ModelConfig(cfg.hidden_size, cfg.moe_intermediate_size, cfg.n_routed_experts, cfg.num_experts_per_tok, count,
                                 cfg.n_shared_experts, cfg.routed_scaling_factor, swiglu_limit=cfg.swiglu_limit, activation_alpha=cfg.activation_alpha))
    engine.cache_note = None
    if use_cache and w2_bits == num_bits:
        from . import expert_cache as EC
        chash = EC.config_hash(model_dir)
        for kind, path, loader in (("cpu", EC.cache_path_cpu(model_dir, num_bits, 128), EC.load_cpu_cache), ("marlin", EC.cache_path_marlin(model_dir, 128, num_bits), EC.load_marlin_cache)):
            if not os.path.exists(path):
                continue
            try:
                loader(engine, path, chash, num_bits, total_moe_layers=n_moe, start_moe_layer=start, num_layers_to_load=count)
                engine.cache_note = f"loaded {kind} cache {path}"
                return cfg
            except (RuntimeError, BufferError, ValueError) as ex:      # a bad cache never aborts the load: fall through to quantization
                engine.cache_note = f"{kind} cache {path} not used: {ex}"
    prefix = detect_expert_prefix(wm)
    handles: Dict[str, object] = {}

    def tensor(name: str):
        shard = wm.get(name)
        if shard is None:
            raise IOError(f"tensor {name} not found in the safetensors index")
        if shard not in handles:
            handles[shard] = safe_open(os.path.join(model_dir, shard), framework="pt")
        t = handles[shard].get_tensor(name)
        if t.dtype != torch.bfloat16:
            t = t.to(torch.bfloat16)                         # the reference converts F16/F32 experts to BF16 before quantizing
        return t.contiguous()

    def upload(layer_out: int, expert: int, base: str, inter: int):
        g, u, d = tensor(base + ".gate_proj.weight"), tensor(base + ".up_proj.weight"), tensor(base + ".down_proj.weight")
        if tuple(g.shape) != (inter, cfg.hidden_size) or tuple(d.shape) != (cfg.hidden_size, inter):
            raise ValueError(f"{base}: unexpected expert shape {tuple(g.shape)} / {tuple(d.shape)}")
        check(engine._lib.kr_upload_expert_bf16(engine._h, layer_out, expert, inter, g.data_ptr(), u.data_ptr(), d.data_ptr(), num_bits, w2_bits))

    for m in range(count):
        layer_idx = start + m + cfg.first_k_dense_replace
        for e in range(cfg.n_routed_experts):
            upload(m, e, f"{prefix}.layers.{layer_idx}.mlp.experts.{e}", cfg.moe_intermediate_size)
        if cfg.n_shared_experts > 0:
            base = f"{prefix}.layers.{layer_idx}.mlp.shared_experts"
            if base + ".gate_proj.weight" not in wm:
                base = f"{prefix}.layers.{layer_idx}.mlp.shared_expert"      # Qwen naming
            if base + ".gate_proj.weight" in wm:
                upload(m, -1, base, cfg.n_shared_experts * cfg.moe_intermediate_size)
    engine._cpu_bits = engine._gpu_bits = num_bits
    if write_cache and w2_bits == num_bits and start == 0 and count == n_moe:
        from . import expert_cache as EC
        EC.save_cpu_cache(engine, EC.cache_path_cpu(model_dir, num_bits, 128), EC.config_hash(model_dir), num_bits)
    return cfg


# ------------------------------------------------------------------------------------------------------------------ GGUF
@dataclass
class GgufTensor:
    name: str
    dims: Tuple[int, ...]       # GGUF order: dims[0] is the contiguous (input) dimension
    dtype: int
    offset: int
    n_elements: int

    @property
    def nbytes(self) -> int:
        be, bb = GGML_BLOCK[self.dtype]
        return self.n_elements // be * bb


class GgufFile:
    """GGUF v2/v3 reader (src/gguf.rs:320-470): header, metadata (kept: general.*), tensor table, zero-copy tensor bytes via mmap."""

    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        if len(self._mm) < 24:
            raise IOError("GGUF file too small for header")
        self._pos = 0
        magic, version = self._u32(), self._u32()
        if magic != GGUF_MAGIC:
            raise IOError(f"Bad GGUF magic: 0x{magic:08x} (expected 0x{GGUF_MAGIC:08x})")
        if version < 2 or version > 3:
            raise IOError(f"Unsupported GGUF version: {version} (supported: 2-3)")
        n_tensors, n_kv = self._u64(), self._u64()
        self.metadata: Dict[str, object] = {}
        for _ in range(n_kv):
            key = self._string(); vtype = self._u32()
            val = self._value(vtype)
            if key.startswith("general.") or not isinstance(val, (list, bytes)):
                self.metadata[key] = val
        self.tensors: Dict[str, GgufTensor] = {}
        for _ in range(n_tensors):
            name = self._string(); nd = self._u32()
            dims = tuple(self._u64() for _ in range(nd))
            dtype, off = self._u32(), self._u64()
            if dtype not in GGML_BLOCK:
                raise IOError(f"Unknown GGML type: {dtype}")
            self.tensors[name] = GgufTensor(name, dims, dtype, off, int(np.prod(dims, dtype=np.int64)))
        align = int(self.metadata.get("general.alignment", GGUF_DEFAULT_ALIGNMENT))
        self.data_offset = (self._pos + align - 1) // align * align

    # -- primitive readers
    def _take(self, n):
        b = self._mm[self._pos:self._pos + n]
        if len(b) != n:
            raise IOError("GGUF truncated")
        self._pos += n
        return b
    def _u32(self): return struct.unpack("<I", self._take(4))[0]
    def _u64(self): return struct.unpack("<Q", self._take(8))[0]
    def _string(self): return self._take(self._u64()).decode("utf-8", "replace")
    def _value(self, t):
        fixed = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<B", 10: "<Q", 11: "<q", 12: "<d"}
        if t in fixed:
            v = struct.unpack(fixed[t], self._take(struct.calcsize(fixed[t])))[0]
            return bool(v) if t == 7 else v
        if t == 8:
            return self._string()
        if t == 9:
            et, n = self._u32(), self._u64()
            return [self._value(et) for _ in range(n)]
        raise IOError(f"Unknown GGUF metadata value type {t}")

    def tensor_bytes(self, name: str, expert: Optional[int] = None, n_experts: int = 1) -> np.ndarray:
        """Raw block bytes of a tensor (gguf.rs:452), or of one expert slice of a merged `_exps` tensor (experts are the slowest dim)."""
        t = self.tensors[name]
        start, n = self.data_offset + t.offset, t.nbytes
        if expert is not None:
            n //= n_experts; start += expert * n
        return np.frombuffer(self._mm, dtype=np.uint8, count=n, offset=start)

    def find_expert_tensors(self, layer: int, expert: int):   # gguf.rs:488
        per = tuple(f"blk.{layer}.ffn_{p}.{expert}.weight" for p in ("gate", "up", "down"))
        if per[0] in self.tensors:
            return per, False
        merged = tuple(f"blk.{layer}.ffn_{p}_exps.weight" for p in ("gate", "up", "down"))
        if merged[0] in self.tensors:
            return merged, True
        return None, False

    def find_shared_expert_tensors(self, layer: int):          # gguf.rs:516
        names = tuple(f"blk.{layer}.ffn_{p}_shexp.weight" for p in ("gate", "up", "down"))
        return names if names[0] in self.tensors else None

    def close(self):
        self._mm.close(); self._f.close()


def load_from_gguf(engine, gguf_path: str, config_json: Optional[str] = None, gguf_native: bool = True, max_layers: Optional[int] = None,
                   start_layer: Optional[int] = None, cfg: Optional[MoeConfig] = None, group_size: int = 128) -> MoeConfig:
    """KrasisEngine.load(gguf_path=...) (moe.rs:1538 -> weights/mod.rs:3251).
    `gguf_native=True` keeps the file's blocks (Q4_K / Q8_0 / Q4_0 / Q5_0 / Q6_K) and runs the native block kernels.
    `gguf_native=False` is the reference's default (the only GGUF mode its CLI reaches): every expert tensor is de-quantized to f32
    (gguf.rs:872), rounded to bf16 and re-quantized to INT4 / INT8-g128 with the expert quantizer (weights/mod.rs:3592-3760, :4063-4113);
    the width is chosen once for the whole file -- the widest mapping of any gate tensor for w13, of any down tensor for w2
    (gguf_type_to_cpu_bits, weights/mod.rs:26-42, :3617-3646) -- so w13 / w2 may differ (mixed precision)."""
    from .engine import ModelConfig
    if not gguf_native:
        return _load_from_gguf_requant(engine, gguf_path, config_json, max_layers, start_layer, cfg, group_size)
    g = GgufFile(gguf_path)
    try:
        if cfg is None:
            if config_json is None:
                raise ValueError("load_from_gguf needs the model's config.json (the reference reads it from model_dir) or an explicit MoeConfig")
            cfg = MoeConfig.from_json(config_json)
        n_moe = cfg.num_hidden_layers - cfg.first_k_dense_replace
        start = start_layer or 0
        count = min(max_layers, n_moe - start) if max_layers else n_moe - start
        if count <= 0:
            raise ValueError(f"start_layer {start} / max_layers {max_layers} select no MoE layer (model has {n_moe})")
        engine.configure(ModelConfig(cfg.hidden_size, cfg.moe_intermediate_size, cfg.n_routed_experts, cfg.num_experts_per_tok, count,
                                     cfg.n_shared_experts, cfg.routed_scaling_factor, swiglu_limit=cfg.swiglu_limit,
                                     activation_alpha=cfg.activation_alpha))
        H, I, E = cfg.hidden_size, cfg.moe_intermediate_size, cfg.n_routed_experts
        for m in range(count):
            layer_idx = start + m + cfg.first_k_dense_replace
            for e in range(E):
                names, merged = g.find_expert_tensors(layer_idx, e)
                if names is None:
                    raise IOError(f"GGUF has no expert tensors for layer {layer_idx} expert {e}")
                tg, tu, td = (g.tensors[n] for n in names)
                if tg.dtype != tu.dtype:
                    raise ValueError(f"{names[0]}: gate / up ggml types differ ({tg.dtype} vs {tu.dtype})")
                for t in (tg, td):
                    if t.dtype not in NATIVE_TYPES:
                        raise ValueError(f"{t.name}: ggml type {t.dtype} has no native block kernel (supported: {sorted(NATIVE_TYPES)})")
                if tg.dims[0] != H or td.dims[0] != I:
                    raise ValueError(f"{names[0]}: unexpected dims {tg.dims} / {td.dims} for hidden {H}, intermediate {I}")
                sl = (e, E) if merged else (None, 1)
                engine.load_gguf_expert(m, e, np.ascontiguousarray(g.tensor_bytes(names[0], *sl)), np.ascontiguousarray(g.tensor_bytes(names[1], *sl)),
                                        np.ascontiguousarray(g.tensor_bytes(names[2], *sl)), tg.dtype, td.dtype, I)
            sh = g.find_shared_expert_tensors(layer_idx)
            if sh is not None and cfg.n_shared_experts > 0:
                tg, td = g.tensors[sh[0]], g.tensors[sh[2]]
                engine.load_gguf_expert(m, -1, np.ascontiguousarray(g.tensor_bytes(sh[0])), np.ascontiguousarray(g.tensor_bytes(sh[1])),
                                        np.ascontiguousarray(g.tensor_bytes(sh[2])), tg.dtype, td.dtype, cfg.n_shared_experts * I)
        return cfg
    finally:
        g.close()


def _load_from_gguf_requant(engine, gguf_path, config_json, max_layers, start_layer, cfg, group_size) -> MoeConfig:
    """the `gguf_native=False` arm of load_from_gguf: GGUF blocks -> f32 -> bf16 -> INT4 / INT8-g128 experts (quantized on the GPU)"""
    from .engine import ModelConfig
    from . import gguf_dequant as GD
    from ._lib import check
    if group_size != 128:
        raise ValueError("group_size must be 128 (marlin.rs:12)")
    g = GgufFile(gguf_path)
    try:
        if cfg is None:
            if config_json is None:
                raise ValueError("load_from_gguf needs the model's config.json (the reference reads it from model_dir) or an explicit MoeConfig")
            cfg = MoeConfig.from_json(config_json)
        n_moe = cfg.num_hidden_layers - cfg.first_k_dense_replace
        start = start_layer or 0
        count = min(max_layers, n_moe - start) if max_layers else n_moe - start
        if count <= 0:
            raise ValueError(f"start_layer {start} / max_layers {max_layers} select no MoE layer (model has {n_moe})")
        H, I, E = cfg.hidden_size, cfg.moe_intermediate_size, cfg.n_routed_experts
        # one width per file: scan EVERY MoE layer of the model (not only the selected range), as the reference's cache builder does
        w13_bits, w2_bits = 4, 4
        for li in range(cfg.first_k_dense_replace, cfg.num_hidden_layers):
            names, merged = g.find_expert_tensors(li, 0)
            if names is None:
                continue
            for e in range(1 if merged else E):
                nm, _ = g.find_expert_tensors(li, e)
                if nm is None:
                    continue
                w13_bits = max(w13_bits, GD.cpu_bits(g.tensors[nm[0]].dtype)); w2_bits = max(w2_bits, GD.cpu_bits(g.tensors[nm[2]].dtype))
        engine.configure(ModelConfig(cfg.hidden_size, cfg.moe_intermediate_size, cfg.n_routed_experts, cfg.num_experts_per_tok, count,
                                     cfg.n_shared_experts, cfg.routed_scaling_factor, swiglu_limit=cfg.swiglu_limit,
                                     activation_alpha=cfg.activation_alpha))

        def upload(layer_out, expert, inter, tn):
            tg, tu, td = (g.tensors[n] for n in tn[0])
            sl = tn[1]
            if tg.dims[0] != H or td.dims[0] != inter:
                raise ValueError(f"{tn[0][0]}: unexpected dims {tg.dims} / {td.dims} for hidden {H}, intermediate {inter}")
            parts = []
            for t, name, n_el in ((tg, tn[0][0], inter * H), (tu, tn[0][1], inter * H), (td, tn[0][2], H * inter)):
                f = GD.dequantize_raw_data(t.dtype, np.ascontiguousarray(g.tensor_bytes(name, *sl)), n_el)
                parts.append(np.ascontiguousarray(GD.f32_to_bf16(f)))
            check(engine._lib.kr_upload_expert_bf16(engine._h, layer_out, expert, inter, parts[0].ctypes.data, parts[1].ctypes.data, parts[2].ctypes.data,
                                                    w13_bits, w2_bits))

        for m in range(count):
            layer_idx = start + m + cfg.first_k_dense_replace
            for e in range(E):
                names, merged = g.find_expert_tensors(layer_idx, e)
                if names is None:
                    raise IOError(f"GGUF has no expert tensors for layer {layer_idx} expert {e}")
                upload(m, e, I, (names, (e, E) if merged else (None, 1)))
            sh = g.find_shared_expert_tensors(layer_idx)
            if sh is not None and cfg.n_shared_experts > 0:
                upload(m, -1, cfg.n_shared_experts * I, (sh, (None, 1)))
        return cfg
    finally:
        g.close()
