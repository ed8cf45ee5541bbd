"""krasis_amd.expert_cache on the GPU: an engine writes the reference's version-4 (CPU transposed) and version-3 (Marlin) cache files and another
engine loads them; the file body is compared BYTE FOR BYTE with one assembled independently from the oracle's quantized experts (CPU layout) and
from kr_marlin_repack of those (Marlin layout, itself pinned by the reference's permutation tables: tests/golden/marlin_int4.npz); partial layer
ranges (the reference's pipeline-stage load), the shared expert block, stale-cache errors, and KrasisEngine.load() picking a cache up."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import make_experts, rand_bf16, upload

pytestmark = pytest.mark.gpu


def _engine(H, I, E, k, L, ns):
    from krasis_amd import KrasisEngine, ModelConfig
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, L, ns, 1.0))
    return eng


def _fill(rng, eng, H, I, E, L, ns, bits):
    layers = []
    for li in range(L):
        ex = make_experts(rng, E, H, I, bits); sh = make_experts(rng, 1, H, ns * I, bits)[0] if ns else None
        upload(eng, li, ex, sh); layers.append((ex, sh))
    return layers


def _forward(eng, rng, H, E, k, layer):
    act = rand_bf16(rng, (1, H))[0]; ids = [int(x) for x in rng.choice(E, k, replace=False)]; w = [0.3] * k
    return np.frombuffer(eng.moe_forward(layer, act.tobytes(), ids, w), np.float32).copy(), (act, ids, w)


@pytest.mark.parametrize("bits,ns", [(4, 1), (4, 0), (8, 2)])
def test_cpu_cache_file_is_the_reference_layout_and_round_trips(tmp_path, bits, ns):
    from krasis_amd import expert_cache as EC
    H, I, E, k, L = 256, 128, 6, 2, 3
    rng = np.random.default_rng(bits * 10 + ns)
    eng = _engine(H, I, E, k, L, ns); layers = _fill(rng, eng, H, I, E, L, ns, bits)
    path = str(tmp_path / "experts_cpu.bin")
    size = EC.save_cpu_cache(eng, path, 0xABCDEF, bits)
    raw = open(path, "rb").read()
    assert size == len(raw) == EC.expected_cpu_cache_size(H, I, E, 128, bits, L, ns)
    body = b"".join(a.tobytes() for ex, _ in layers for e in ex for a in (e.w13, e.w13_scales, e.w2, e.w2_scales))
    body += b"".join(a.tobytes() for _, sh in layers if sh is not None for a in (sh.w13, sh.w13_scales, sh.w2, sh.w2_scales))
    assert raw[:64] == EC.pack_header(4, H, I, E, L, 128, 0xABCDEF, ns, bits)
    assert raw[64:] == body                                            # per (layer, expert) w13 | w13 scales | w2 | w2 scales, then the shared experts
    eng2 = _engine(H, I, E, k, L, ns)
    EC.load_cpu_cache(eng2, path, 0xABCDEF, bits)
    r1, (act, ids, w) = _forward(eng, np.random.default_rng(1), H, E, k, 2)
    r2 = np.frombuffer(eng2.moe_forward(2, act.tobytes(), ids, w), np.float32)
    assert np.array_equal(r1.view(np.uint32), r2.view(np.uint32))
    for a, b in zip(eng.download_expert(1, 3, bits), eng2.download_expert(1, 3, bits)):
        assert np.array_equal(a, b)
    # pipeline-stage load: file layers [1, 3) -> engine layers 0, 1
    eng3 = _engine(H, I, E, k, 2, ns)
    EC.load_cpu_cache(eng3, path, 0xABCDEF, bits, total_moe_layers=L, start_moe_layer=1, num_layers_to_load=2)
    r3 = np.frombuffer(eng3.moe_forward(1, act.tobytes(), ids, w), np.float32)
    assert np.array_equal(r1.view(np.uint32), r3.view(np.uint32))
    # stale / foreign files: the reference's messages
    with pytest.raises(RuntimeError, match="Config hash mismatch in CPU cache"):
        EC.load_cpu_cache(eng2, path, 1, bits)
    with pytest.raises(RuntimeError, match="num_bits mismatch"):
        EC.load_cpu_cache(eng2, path, 0xABCDEF, 12 - bits)
    open(path, "ab").write(b"\0")
    with pytest.raises(RuntimeError, match="CPU cache size mismatch: expected %d bytes, got %d" % (size, size + 1)):
        EC.load_cpu_cache(eng2, path, 0xABCDEF, bits)
    with pytest.raises(RuntimeError, match="Failed to open CPU cache"):
        EC.load_cpu_cache(eng2, str(tmp_path / "nope.bin"), 0xABCDEF, bits)


def test_marlin_cache_file_is_the_reference_layout_and_round_trips(tmp_path):
    import ctypes as C
    from krasis_amd import _lib, expert_cache as EC
    H, I, E, k, L, ns, bits = 256, 128, 5, 2, 2, 1, 4
    rng = np.random.default_rng(4)
    eng = _engine(H, I, E, k, L, ns); layers = _fill(rng, eng, H, I, E, L, ns, bits)
    path = str(tmp_path / "experts_marlin.bin")
    size = EC.save_marlin_cache(eng, path, 77, bits)
    raw = open(path, "rb").read()
    assert size == len(raw) == EC.expected_marlin_cache_size(H, I, E, 128, L, ns, bits)
    assert raw[:64] == EC.pack_header(3, H, I, E, L, 128, 77, ns)
    # the first expert's w13 block = marlin_repack of its row-major [2I, H] nibbles (weights/marlin.rs:179): rebuild from the CPU transposed layout
    e0 = layers[0][0][0]
    K, N = H, 2 * I
    nib = np.stack([((e0.w13 >> (4 * j)) & 15) for j in range(8)], axis=1).reshape(K, N).T          # [N, K] nibbles
    rowmajor = np.zeros((N, K // 8), np.uint32)
    for j in range(8):
        rowmajor |= nib[:, j::8].astype(np.uint32) << (4 * j)
    scales_rm = np.ascontiguousarray(e0.w13_scales.T)                                                # [N, K / gs]
    outp = np.empty(K // 8 * N, np.uint32); outs = np.empty(K // 128 * N, np.uint16)
    lib = _lib.load_library()
    _lib.check(lib.kr_marlin_repack(rowmajor.ctypes.data, scales_rm.ctypes.data, N, K, 128, 4, outp.ctypes.data, outs.ctypes.data))
    assert raw[64:64 + outp.nbytes] == outp.tobytes()
    assert raw[64 + outp.nbytes:64 + outp.nbytes + outs.nbytes] == outs.tobytes()
    eng2 = _engine(H, I, E, k, L, ns)
    EC.load_marlin_cache(eng2, path, 77, bits)
    r1, (act, ids, w) = _forward(eng, np.random.default_rng(2), H, E, k, 1)
    r2 = np.frombuffer(eng2.moe_forward(1, act.tobytes(), ids, w), np.float32)
    assert np.array_equal(r1.view(np.uint32), r2.view(np.uint32))
    with pytest.raises(RuntimeError, match=r"Cache version 3, expected 4 \(CPU\)"):
        EC.load_cpu_cache(eng2, path, 77, bits)


def test_engine_load_picks_up_a_reference_cache(tmp_path, monkeypatch):
    """KrasisEngine.load(model_dir): with a version-4 file in the reference's cache directory the safetensors are not touched (they are deleted here
    after the first load); a stale file is skipped and the reason kept."""
    from krasis_amd import KrasisEngine, expert_cache as EC
    from tests.tiny_checkpoints import make_qcn_tiny
    monkeypatch.setenv("HOME", str(tmp_path / "home"))
    mdir = str(tmp_path / "qcn_tiny"); make_qcn_tiny(mdir, seed=3, layers=2)
    a = KrasisEngine(); a.load(mdir, num_bits=4)
    assert a.cache_note is None
    path = EC.cache_path_cpu(mdir, 4, 128)
    EC.save_cpu_cache(a, path, EC.config_hash(mdir), 4)
    import safetensors
    real_open = safetensors.safe_open

    class NoTensors:          # the tensor NAMES may be listed (config parsing), no tensor may be read
        def __init__(self, *a, **k): self._f = real_open(*a, **k)
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def keys(self): return self._f.keys()
        def get_tensor(self, name): raise AssertionError("a cached model must not read " + name)
    monkeypatch.setattr(safetensors, "safe_open", NoTensors)
    b = KrasisEngine(); b.load(mdir, num_bits=4)
    assert b.cache_note and b.cache_note.startswith("loaded cpu cache")
    for x, y in zip(a.download_expert(1, 2, 4), b.download_expert(1, 2, 4)):
        assert np.array_equal(x, y)
    monkeypatch.setattr(safetensors, "safe_open", real_open)
    open(os.path.join(mdir, "config.json"), "a").write("\n")           # config changed -> hash mismatch -> quantize again
    c = KrasisEngine(); c.load(mdir, num_bits=4)
    assert "Config hash mismatch" in c.cache_note
    for x, y in zip(a.download_expert(0, 1, 4), c.download_expert(0, 1, 4)):
        assert np.array_equal(x, y)


def test_gguf_cpu_cache_v5_mixed_widths_round_trips(tmp_path):
    """version 5 (`experts_gguf_avx2_g128.bin`, weights/mod.rs:883,930,3908-4062,4176-4206): the CPU transposed body with gate|up and down at their OWN widths
    (a GGUF whose down projections are Q6_K / Q8_0 re-quantizes them to INT8 while Q4_K gate / up stay INT4: weights/mod.rs:26-42).  Header bytes 56..64 =
    n_shared | w13_bits << 48 | w2_bits << 56.  An engine writes the file, the body equals the independently assembled one byte for byte, a second engine
    loads it (incl. a pipeline-stage range) and computes the same bits; the reference's validation messages."""
    import struct
    from krasis_amd import expert_cache as EC
    H, I, E, k, L, ns = 256, 128, 5, 2, 3, 1
    rng = np.random.default_rng(55)
    eng = _engine(H, I, E, k, L, ns)
    layers = []
    for li in range(L):
        ex = make_experts(rng, E, H, I, 4, 8); sh = make_experts(rng, 1, H, ns * I, 4, 8)[0]
        upload(eng, li, ex, sh); layers.append((ex, sh))
    path = str(tmp_path / "experts_gguf_avx2_g128.bin")
    size = EC.save_gguf_cpu_cache(eng, path, 0xFEED, 4, 8)
    raw = open(path, "rb").read()
    assert size == len(raw) == EC.expected_gguf_cpu_cache_size(H, I, E, 128, 4, 8, L, ns)
    assert raw[:4] == b"KRAS" and struct.unpack("<I", raw[4:8])[0] == 5
    assert struct.unpack("<Q", raw[56:64])[0] == (ns | (4 << 48) | (8 << 56))
    body = b"".join(a.tobytes() for ex, _ in layers for e in ex for a in (e.w13, e.w13_scales, e.w2, e.w2_scales))
    body += b"".join(a.tobytes() for _, sh in layers for a in (sh.w13, sh.w13_scales, sh.w2, sh.w2_scales))
    assert raw[64:] == body
    eng2 = _engine(H, I, E, k, L, ns)
    assert EC.load_gguf_cpu_cache(eng2, path, 0xFEED) == (4, 8)
    r1, (act, ids, w) = _forward(eng, np.random.default_rng(1), H, E, k, 2)
    r2 = np.frombuffer(eng2.moe_forward(2, act.tobytes(), ids, w), np.float32)
    assert np.array_equal(r1.view(np.uint32), r2.view(np.uint32))
    eng3 = _engine(H, I, E, k, 2, ns)
    EC.load_gguf_cpu_cache(eng3, path, 0xFEED, total_moe_layers=L, start_moe_layer=1, num_layers_to_load=2)
    r3 = np.frombuffer(eng3.moe_forward(1, act.tobytes(), ids, w), np.float32)
    assert np.array_equal(r1.view(np.uint32), r3.view(np.uint32))
    with pytest.raises(RuntimeError, match="Config hash mismatch in GGUF CPU cache"):
        EC.load_gguf_cpu_cache(eng2, path, 1)
    v4 = str(tmp_path / "v4.bin"); open(v4, "wb").write(EC.pack_header(4, H, I, E, L, 128, 0xFEED, ns, 4))
    with pytest.raises(RuntimeError, match=r"Cache version 4, expected 5 \(GGUF CPU\)"):
        EC.load_gguf_cpu_cache(eng2, v4, 0xFEED)
    open(path, "ab").write(b"\0")
    with pytest.raises(RuntimeError, match="GGUF CPU cache size mismatch: expected %d bytes, got %d" % (size, size + 1)):
        EC.load_gguf_cpu_cache(eng2, path, 0xFEED)
