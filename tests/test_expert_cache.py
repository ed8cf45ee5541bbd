"""Format of the reference's on-disk expert caches as krasis_amd.expert_cache reads / writes them (src/weights/mod.rs:856-1131 header, paths and
sizes; :4117-4180 header writers; :2794-2870 / :2367-2430 validation order and messages).  No GPU: header bytes from the reference's comment block
(:858-867), FNV-1a test vectors, sizes from its formulas with independently computed numbers, every validation error in the reference's order."""
import struct

import pytest

from krasis_amd import expert_cache as EC


def test_fnv1a_vectors_and_paths(tmp_path, monkeypatch):
    assert EC.fnv1a(b"") == 0xCBF29CE484222325                      # offset basis
    assert EC.fnv1a(b"a") == 0xAF63DC4C8601EC8C                     # published FNV-1a 64 vectors
    assert EC.fnv1a(b"foobar") == 0x85944171F73967E8
    monkeypatch.setenv("HOME", "/home/u")
    assert EC.cache_dir_for_model("/models/Qwen3-Coder-Next/") == "/home/u/.krasis/cache/Qwen3-Coder-Next"
    assert EC.cache_path_cpu("/models/M", 4, 128) == "/home/u/.krasis/cache/M/experts_cpu_int4_g128.bin"
    assert EC.cache_path_marlin("/models/M", 128, 8) == "/home/u/.krasis/cache/M/experts_marlin_int8_g128.bin"
    monkeypatch.delenv("HOME")
    assert EC.cache_dir_for_model("/models/M") == "/models/M/.krasis_cache"


def test_sizes_match_the_reference_formulas():
    # QCN: h 2048, m 512, 512 experts, 48 layers, 1 shared expert, INT4 g128
    h, m, E, L, gs = 2048, 512, 512, 48, 128
    per = (h // 8) * 2 * m * 4 + (h // gs) * 2 * m * 2 + (m // 8) * h * 4 + (m // gs) * h * 2
    assert sum(EC.cpu_expert_byte_sizes(h, m, gs, 4)) == per == 1_622_016
    assert EC.expected_cpu_cache_size(h, m, E, gs, 4, L, 1) == 64 + L * E * per + L * per
    assert sum(EC.cpu_expert_byte_sizes(h, m, gs, 8)) == h * 2 * m + (h // gs) * 2 * m * 2 + m * h + (m // gs) * h * 2
    assert sum(EC.marlin_expert_byte_sizes(h, m, gs, 4)) == per               # INT4: same bytes, other order (weights/mod.rs:969)
    # w2 padding rule (weights/mod.rs:942-949): only when hidden == intermediate and hidden % 256 != 0, and never for the shared expert's file slot
    assert EC.marlin_w2_padded_n(2048, 512) == 2048 and EC.marlin_w2_padded_n(1024, 1024) == 1024 and EC.marlin_w2_padded_n(384, 384) == 448
    a = EC.marlin_expert_byte_sizes(384, 384, 128, 4); b = EC.marlin_expert_byte_sizes(384, 384, 128, 4, shared=True)
    assert a[2] == (384 // 8) * 448 * 4 and b[2] == (384 // 8) * 384 * 4 and a[3] == 3 * 448 * 2


def test_header_bytes_and_validation_order():
    hdr = EC.pack_header(EC.CACHE_VERSION_CPU, 2048, 512, 512, 48, 128, 0x1122334455667788, 1, 4)
    assert len(hdr) == 64 and hdr[:4] == b"KRAS"
    assert struct.unpack("<I", hdr[4:8])[0] == 4
    assert struct.unpack("<6Q", hdr[8:56]) == (2048, 512, 512, 48, 128, 0x1122334455667788)
    assert struct.unpack("<Q", hdr[56:64])[0] == 1 | (4 << 32)                # n_shared | num_bits << 32 (write_cpu_cache_header)
    hm = EC.pack_header(EC.CACHE_VERSION_MARLIN, 2048, 512, 512, 48, 128, 7, 1)
    assert struct.unpack("<Q", hm[56:64])[0] == 1
    ok = dict(kind="CPU", version=4, h=2048, m=512, n_experts=512, total_moe_layers=48, gs=128, chash=0x1122334455667788, n_shared=1, expected_bits=4)
    EC.check_header(hdr, **ok)
    with pytest.raises(RuntimeError, match="too small for header"):
        EC.check_header(hdr[:10], **ok)
    with pytest.raises(RuntimeError, match="Bad magic in CPU cache"):
        EC.check_header(b"XRAS" + hdr[4:], **ok)
    with pytest.raises(RuntimeError, match=r"Cache version 3, expected 4 \(CPU\)"):
        EC.check_header(hm, **ok)
    with pytest.raises(RuntimeError, match=r"header mismatch: file has 2048h/512m/512e/48L/g128, expected 2048h/768m/512e/48L/g128"):
        EC.check_header(hdr, **{**ok, "m": 768})
    with pytest.raises(RuntimeError, match="Config hash mismatch"):
        EC.check_header(hdr, **{**ok, "chash": 1})
    with pytest.raises(RuntimeError, match="Shared expert count mismatch: cache=1, config=0"):
        EC.check_header(hdr, **{**ok, "n_shared": 0})
    with pytest.raises(RuntimeError, match="num_bits mismatch: cache=INT4, expected INT8"):
        EC.check_header(hdr, **{**ok, "expected_bits": 8})
    # a wrong shape is reported before a wrong hash, a wrong hash before the shared count (the reference's order)
    with pytest.raises(RuntimeError, match="header mismatch"):
        EC.check_header(hdr, **{**ok, "h": 1024, "chash": 1, "n_shared": 0})


def test_v5_gguf_cpu_cache_sizes_and_header():
    """version 5 (weights/mod.rs:1000-1041, 4176-4206): per-projection widths; bytes 56..64 = n_shared | w13_bits << 48 | w2_bits << 56"""
    h, m, E, L, gs = 2048, 512, 512, 48, 128
    a = EC.cpu_expert_byte_sizes_mixed(h, m, gs, 4, 8)
    assert a == ((h // 8) * 2 * m * 4, (h // gs) * 2 * m * 2, m * h, (m // gs) * h * 2)
    assert EC.cpu_expert_byte_sizes_mixed(h, m, gs, 4, 4) == EC.cpu_expert_byte_sizes(h, m, gs, 4)
    assert EC.expected_gguf_cpu_cache_size(h, m, E, gs, 4, 8, L, 1) == 64 + L * E * sum(a) + L * sum(EC.cpu_expert_byte_sizes_mixed(h, m, gs, 4, 8))
    hdr = EC.pack_header_v5(h, m, E, L, gs, 0x99, 1, 4, 8)
    assert len(hdr) == 64 and struct.unpack("<I", hdr[4:8])[0] == 5
    assert struct.unpack("<Q", hdr[56:64])[0] == 1 | (4 << 48) | (8 << 56)
    assert EC.cache_path_gguf_avx2("/models/M", 128).endswith("/experts_gguf_avx2_g128.bin")
