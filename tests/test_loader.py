"""Host loader (krasis_amd/weight_store.py): config.json fall-backs and the GGUF v3 reader on synthetic files (CPU), the GPU-side BF16 ->
INT4/INT8 quantizer against the oracle restatement of weights/marlin.rs:65,145 (bit-exact, GPU), and a safetensors round trip."""
import json
import os
import struct

import numpy as np
import pytest

from krasis_amd.weight_store import GgufFile, MoeConfig, detect_expert_prefix


def test_config_fallbacks(tmp_path):
    p = tmp_path / "config.json"
    p.write_text(json.dumps({"hidden_size": 2048, "moe_intermediate_size": 512, "num_experts": 512, "num_experts_per_tok": 10, "num_hidden_layers": 48,
                             "decoder_sparse_step": 1}))
    c = MoeConfig.from_json(str(p))
    assert (c.n_routed_experts, c.first_k_dense_replace, c.n_shared_experts, c.routed_scaling_factor, c.activation_alpha) == (512, 0, 0, 1.0, 0.0)
    p.write_text(json.dumps({"hidden_size": 2048, "moe_intermediate_size": 1408, "n_routed_experts": 64, "num_experts_per_tok": 6, "num_hidden_layers": 27,
                             "first_k_dense_replace": 1, "n_shared_experts": 2, "routed_scaling_factor": 1.0}))
    c = MoeConfig.from_json(str(p))
    assert (c.first_k_dense_replace, c.n_shared_experts) == (1, 2)
    p.write_text(json.dumps({"hidden_size": 2880, "intermediate_size": 2880, "num_local_experts": 32, "experts_per_token": 4, "num_hidden_layers": 24,
                             "swiglu_limit": 7.0}))
    c = MoeConfig.from_json(str(p))
    assert (c.n_routed_experts, c.num_experts_per_tok, c.swiglu_limit, c.activation_alpha) == (32, 4, 7.0, 1.702)
    p.write_text(json.dumps({"hidden_size": 64, "moe_intermediate_size": 64, "num_experts": 4, "num_experts_per_tok": 2, "num_hidden_layers": 4,
                             "decoder_sparse_step": 2}))
    with pytest.raises(ValueError):
        MoeConfig.from_json(str(p))
    p.write_text(json.dumps({"hidden_size": 64}))
    with pytest.raises(ValueError):
        MoeConfig.from_json(str(p))
    # num_hidden_layers inferred from the index (weights/mod.rs:117-130)
    p.write_text(json.dumps({"hidden_size": 64, "moe_intermediate_size": 64, "num_experts": 4, "num_experts_per_tok": 2}))
    wm = {f"model.layers.{l}.mlp.experts.0.gate_proj.weight": "a" for l in range(7)}
    assert MoeConfig.from_json(str(p), wm).num_hidden_layers == 7


def test_expert_prefix():
    assert detect_expert_prefix({"mtp.layers.0.mlp.experts.0.up_proj.weight": "x", "model.language_model.layers.3.mlp.experts.1.up_proj.weight": "y"}) \
        == "model.language_model"
    with pytest.raises(ValueError):
        detect_expert_prefix({"model.embed_tokens.weight": "x"})


def _gguf_string(s):
    b = s.encode()
    return struct.pack("<Q", len(b)) + b


def write_gguf(path, tensors, meta=()):
    """tensors: list of (name, dims (gguf order), ggml_type, raw bytes)."""
    hdr = struct.pack("<IIQQ", 0x46554747, 3, len(tensors), len(meta))
    for k, t, v in meta:
        hdr += _gguf_string(k) + struct.pack("<I", t) + (_gguf_string(v) if t == 8 else struct.pack("<I", v))
    off, infos = 0, b""
    for name, dims, ty, raw in tensors:
        infos += _gguf_string(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", ty, off)
        off += (len(raw) + 31) // 32 * 32
    head = hdr + infos
    pad = (-len(head)) % 32
    with open(path, "wb") as f:
        f.write(head + b"\0" * pad)
        for _, _, _, raw in tensors:
            f.write(raw + b"\0" * ((-len(raw)) % 32))


def test_gguf_reader(tmp_path):
    rng = np.random.default_rng(0)
    H, I, E = 256, 256, 3
    q4k = lambda rows, cols: rng.integers(0, 256, rows * cols // 256 * 144, dtype=np.uint8).tobytes()
    q80 = lambda rows, cols: rng.integers(0, 256, rows * cols // 32 * 34, dtype=np.uint8).tobytes()
    gate, up, down = q4k(E * I, H), q4k(E * I, H), q80(E * H, I)
    p = str(tmp_path / "m.gguf")
    write_gguf(p, [("blk.1.ffn_gate_exps.weight", (H, I, E), 12, gate), ("blk.1.ffn_up_exps.weight", (H, I, E), 12, up),
                   ("blk.1.ffn_down_exps.weight", (I, H, E), 8, down), ("blk.1.ffn_gate_shexp.weight", (H, I), 12, q4k(I, H))],
               meta=[("general.architecture", 8, "qwen3moe"), ("general.file_type", 4, 15)])
    g = GgufFile(p)
    assert g.metadata["general.architecture"] == "qwen3moe" and g.metadata["general.file_type"] == 15
    names, merged = g.find_expert_tensors(1, 2)
    assert merged and names[0] == "blk.1.ffn_gate_exps.weight"
    per = len(gate) // E
    assert g.tensor_bytes(names[0], 2, E).tobytes() == gate[2 * per:3 * per]
    assert g.tensor_bytes(names[2], 1, E).tobytes() == down[len(down) // E:2 * len(down) // E]
    assert g.find_shared_expert_tensors(1)[0] == "blk.1.ffn_gate_shexp.weight" and g.find_shared_expert_tensors(0) is None
    assert g.find_expert_tensors(0, 0)[0] is None
    g.close()
    bad = str(tmp_path / "bad.gguf")
    open(bad, "wb").write(struct.pack("<IIQQ", 0x12345678, 3, 0, 0))
    with pytest.raises(IOError):
        GgufFile(bad)
    open(bad, "wb").write(struct.pack("<IIQQ", 0x46554747, 7, 0, 0))
    with pytest.raises(IOError):
        GgufFile(bad)


# ------------------------------------------------------------------------------------------------------------ GPU
_GGML = {"F32": (0, 1, 4), "F16": (1, 1, 2), "Q4_0": (2, 32, 18), "Q5_0": (6, 32, 22), "Q8_0": (8, 32, 34), "Q4_K": (12, 256, 144), "Q5_K": (13, 256, 176),
         "Q6_K": (14, 256, 210), "BF16": (30, 1, 2)}


def _random_blocks(rng, name, n):
    """n elements of ggml type `name` as raw bytes with finite f16 scales"""
    ty, qk, bb = _GGML[name]
    nb = n // qk
    raw = rng.integers(0, 256, (nb, bb), dtype=np.uint8)
    f16 = lambda k: (rng.standard_normal((nb, k)) * 0.05).astype(np.float16).view(np.uint8).reshape(nb, 2 * k)
    if name in ("Q4_0", "Q5_0", "Q8_0"): raw[:, 0:2] = f16(1)
    elif name in ("Q4_K", "Q5_K"): raw[:, 0:4] = f16(2)
    elif name == "Q6_K": raw[:, 208:210] = f16(1)
    elif name == "F16": raw = (rng.standard_normal((nb, 1)) * 0.05).astype(np.float16).view(np.uint8).reshape(nb, 2)
    elif name == "BF16": raw = ((rng.standard_normal((nb, 1)) * 0.05).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16).view(np.uint8).reshape(nb, 2)
    elif name == "F32": raw = (rng.standard_normal((nb, 1)) * 0.05).astype(np.float32).view(np.uint8).reshape(nb, 4)
    return ty, np.ascontiguousarray(raw.reshape(-1))


@pytest.mark.parametrize("name", sorted(_GGML))
def test_gguf_dequantizers_match_oracle(name):
    """krasis_amd.gguf_dequant (the product's load-time de-quantizers, gguf_native=False) == the oracle's restatement of gguf.rs:872, bit for bit"""
    from oracle import oracle as O
    from krasis_amd import gguf_dequant as GD
    rng = np.random.default_rng(len(name) * 7 + 1)
    n = 256 * 12
    ty, raw = _random_blocks(rng, name, n)
    got = GD.dequantize_raw_data(ty, raw, n)
    ref = O.dequantize(ty, raw, n)
    assert got.dtype == np.float32 and got.shape == (n,)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), float(np.max(np.abs(got - ref)))
    assert np.array_equal(GD.f32_to_bf16(got), O.f32_to_bf16(ref))


def test_gguf_cpu_bits_mapping():
    from krasis_amd import gguf_dequant as GD     # gguf_type_to_cpu_bits (weights/mod.rs:26-42)
    assert [GD.cpu_bits(t) for t in (2, 12, 6, 13, 14, 8, 1, 30, 0)] == [4, 4, 4, 4, 8, 8, 8, 8, 8]
    with pytest.raises(ValueError):
        GD.cpu_bits(99)
    with pytest.raises(ValueError):
        GD.dequantize_raw_data(10, np.zeros(84, np.uint8), 256)          # Q2_K: no de-quantizer in the reference either


@pytest.mark.gpu
@pytest.mark.parametrize("gate_t,down_t,merged", [("Q4_K", "Q6_K", True), ("Q5_0", "Q4_0", False), ("Q8_0", "Q5_K", True)])
def test_load_from_gguf_requantized(tmp_path, gate_t, down_t, merged):
    """gguf_native=False: GGUF blocks -> f32 -> bf16 -> INT4 / INT8-g128 (file-wide widths, mixed precision allowed) == the oracle's
    dequantize + unified_from_bf16, bit for bit (weights/mod.rs:3592-3760, :4063-4113)"""
    from oracle import oracle as O
    from krasis_amd import KrasisEngine, gguf_dequant as GD
    rng = np.random.default_rng(3)
    H, I, E, L = 256, 256, 3, 2                                  # layer 0 dense, layer 1 MoE; shared expert = 1 x I
    tensors, raw = [], {}
    def add(name, dims, tname, n):
        ty, b = _random_blocks(rng, tname, n); tensors.append((name, dims, ty, b.tobytes())); raw[name] = (ty, b)
    if merged:
        add("blk.1.ffn_gate_exps.weight", (H, I, E), gate_t, E * I * H); add("blk.1.ffn_up_exps.weight", (H, I, E), gate_t, E * I * H)
        add("blk.1.ffn_down_exps.weight", (I, H, E), down_t, E * H * I)
    else:
        for e in range(E):
            add(f"blk.1.ffn_gate.{e}.weight", (H, I), gate_t, I * H); add(f"blk.1.ffn_up.{e}.weight", (H, I), gate_t, I * H); add(f"blk.1.ffn_down.{e}.weight", (I, H), down_t, H * I)
    add("blk.1.ffn_gate_shexp.weight", (H, I), gate_t, I * H); add("blk.1.ffn_up_shexp.weight", (H, I), gate_t, I * H); add("blk.1.ffn_down_shexp.weight", (I, H), down_t, H * I)
    write_gguf(str(tmp_path / "m.gguf"), tensors)
    (tmp_path / "config.json").write_text(json.dumps({"hidden_size": H, "moe_intermediate_size": I, "n_routed_experts": E, "num_experts_per_tok": 2,
                                                      "num_hidden_layers": L, "first_k_dense_replace": 1, "n_shared_experts": 1}))
    eng = KrasisEngine()
    eng.load(str(tmp_path), gguf_path=str(tmp_path / "m.gguf"), gguf_native=False)
    b13, b2 = GD.cpu_bits(_GGML[gate_t][0]), GD.cpu_bits(_GGML[down_t][0])
    def ref_expert(names, e):
        out = []
        for nm, n_el, shape in ((names[0], I * H, (I, H)), (names[1], I * H, (I, H)), (names[2], H * I, (H, I))):
            ty, b = raw[nm]
            per = b.size // (E if merged and "exps" in nm else 1)
            blk = b[e * per:(e + 1) * per] if merged and "exps" in nm else b
            out.append(O.f32_to_bf16(O.dequantize(ty, blk, n_el)).reshape(shape))
        return O.unified_from_bf16(out[0], out[1], out[2], num_bits=b13, w2_bits=b2)
    for e in range(E):
        names = ("blk.1.ffn_gate_exps.weight", "blk.1.ffn_up_exps.weight", "blk.1.ffn_down_exps.weight") if merged else (f"blk.1.ffn_gate.{e}.weight", f"blk.1.ffn_up.{e}.weight", f"blk.1.ffn_down.{e}.weight")
        ref = ref_expert(names, e)
        w13, w13s, w2, w2s = eng.download_expert(0, e, b13, b2)
        assert np.array_equal(w13, ref.w13) and np.array_equal(w13s, ref.w13_scales) and np.array_equal(w2, ref.w2) and np.array_equal(w2s, ref.w2_scales), e
    ref = ref_expert(("blk.1.ffn_gate_shexp.weight", "blk.1.ffn_up_shexp.weight", "blk.1.ffn_down_shexp.weight"), 0)
    w13, w13s, w2, w2s = eng.download_expert(0, -1, b13, b2)
    assert np.array_equal(w13, ref.w13) and np.array_equal(w2, ref.w2) and np.array_equal(w2s, ref.w2_scales)


@pytest.mark.gpu
@pytest.mark.parametrize("bits,w2_bits", [(4, 4), (8, 8), (4, 8)])
def test_gpu_quantizer_matches_reference_rule(bits, w2_bits):
    """kr_upload_expert_bf16 == quantize_int4/int8 (marlin.rs:65,145) + transposed packing (weights/mod.rs:329-470), bit for bit."""
    from oracle import oracle as O
    from krasis_amd import KrasisEngine, ModelConfig
    from krasis_amd._lib import check
    rng = np.random.default_rng(bits * 10 + w2_bits)
    H, I, E = 256, 384, 3
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, 2, 1, 1, 1.0))
    for e in (0, 2, -1):
        g = O.f32_to_bf16((rng.standard_normal((I, H)) * 0.05).astype(np.float32)); u = O.f32_to_bf16((rng.standard_normal((I, H)) * 0.05).astype(np.float32))
        d = O.f32_to_bf16((rng.standard_normal((H, I)) * 0.05).astype(np.float32))
        g[3, 128:256] = 0                                        # an all-zero group: scale 1.0, q = 0 (marlin.rs:171)
        g[5, 7] = O.f32_to_bf16(np.array([9.0], np.float32))[0]  # a dominant value: clamp at +7 exercised
        check(eng._lib.kr_upload_expert_bf16(eng._h, 0, e, I, g.ctypes.data, u.ctypes.data, d.ctypes.data, bits, w2_bits))
        ref = O.unified_from_bf16(g, u, d, num_bits=bits, w2_bits=w2_bits)
        w13, w13s, w2, w2s = eng.download_expert(0, e, bits, w2_bits)
        assert np.array_equal(w13, ref.w13) and np.array_equal(w13s, ref.w13_scales), e
        assert np.array_equal(w2, ref.w2) and np.array_equal(w2s, ref.w2_scales), e


@pytest.mark.gpu
def test_load_from_hf_roundtrip(tmp_path):
    import torch
    from safetensors.torch import save_file
    from oracle import oracle as O
    from krasis_amd import KrasisEngine
    H, I, E, L = 256, 128, 4, 3                                  # layer 0 dense (first_k_dense_replace = 1), layers 1-2 MoE
    g = torch.Generator().manual_seed(1)
    tens, wm = {}, {}
    for l in (1, 2):
        for e in range(E):
            for nm, shape in (("gate_proj", (I, H)), ("up_proj", (I, H)), ("down_proj", (H, I))):
                tens[f"model.layers.{l}.mlp.experts.{e}.{nm}.weight"] = (torch.randn(shape, generator=g) * 0.05).to(torch.bfloat16)
        for nm, shape in (("gate_proj", (2 * I, H)), ("up_proj", (2 * I, H)), ("down_proj", (H, 2 * I))):
            tens[f"model.layers.{l}.mlp.shared_experts.{nm}.weight"] = (torch.randn(shape, generator=g) * 0.05).to(torch.bfloat16)
    save_file(tens, str(tmp_path / "model-00001-of-00001.safetensors"))
    (tmp_path / "model.safetensors.index.json").write_text(json.dumps({"weight_map": {k: "model-00001-of-00001.safetensors" for k in tens}}))
    (tmp_path / "config.json").write_text(json.dumps({"hidden_size": H, "moe_intermediate_size": I, "n_routed_experts": E, "num_experts_per_tok": 2,
                                                      "num_hidden_layers": L, "first_k_dense_replace": 1, "n_shared_experts": 2,
                                                      "routed_scaling_factor": 1.5}))
    eng = KrasisEngine()
    eng.load(str(tmp_path), num_bits=4)
    assert (eng.num_moe_layers(), eng.num_experts(), eng.hidden_size()) == (2, E, H)
    bits16 = lambda t: t.view(torch.int16).numpy().view(np.uint16)
    for m, l in ((0, 1), (1, 2)):
        for e in (0, 3):
            p = f"model.layers.{l}.mlp.experts.{e}"
            ref = O.unified_from_bf16(bits16(tens[p + ".gate_proj.weight"]), bits16(tens[p + ".up_proj.weight"]), bits16(tens[p + ".down_proj.weight"]), num_bits=4, w2_bits=4)
            w13, w13s, w2, w2s = eng.download_expert(m, e, 4)
            assert np.array_equal(w13, ref.w13) and np.array_equal(w2, ref.w2) and np.array_equal(w13s, ref.w13_scales) and np.array_equal(w2s, ref.w2_scales)
        p = f"model.layers.{l}.mlp.shared_experts"
        ref = O.unified_from_bf16(bits16(tens[p + ".gate_proj.weight"]), bits16(tens[p + ".up_proj.weight"]), bits16(tens[p + ".down_proj.weight"]), num_bits=4, w2_bits=4)
        w13, w13s, w2, w2s = eng.download_expert(m, -1, 4)
        assert np.array_equal(w13, ref.w13) and np.array_equal(w2, ref.w2)
    # max_layers / start_layer select a slice of the MoE stack (moe.rs:1538)
    eng2 = KrasisEngine(); eng2.load(str(tmp_path), num_bits=8, max_layers=1, start_layer=1)
    assert eng2.num_moe_layers() == 1
    p = "model.layers.2.mlp.experts.1"
    ref = O.unified_from_bf16(bits16(tens[p + ".gate_proj.weight"]), bits16(tens[p + ".up_proj.weight"]), bits16(tens[p + ".down_proj.weight"]), num_bits=8, w2_bits=8)
    w13, w13s, w2, w2s = eng2.download_expert(0, 1, 8)
    assert np.array_equal(w13, ref.w13) and np.array_equal(w2s, ref.w2_scales)
