"""The reference's on-disk expert caches, read and written in ITS format (src/weights/mod.rs:856-934 header and paths, :955-1131 sizes,
:2367-2540 load_marlin_cache, :2794-2966 load_cpu_cache, :4117-4180 header writers, :4318-4400 read_unified_expert_cpu).

A user of the reference has `~/.krasis/cache/<model>/experts_cpu_int4_g128.bin` (version 4: every expert in the CPU transposed layout, the wire
format of kr_upload_expert_unified) and / or `experts_marlin_int4_g128.bin` (version 3: the Marlin GPU layout, kr_upload_expert_marlin) on disk.
Both load straight into the resident HBM layout -- no safetensors pass, no quantizer -- and an engine can write either file back byte for byte as the
reference would have (the quantizers and the Marlin permutation are bit-pinned: tests/test_golden_bits.py).  On MI355X the quantizer runs on the
GPU in seconds, so nothing here is needed for speed; it is here so that an existing cache directory keeps working and a cache written here is
accepted by the reference.

Errors are the reference's `Err(String)` texts raised as RuntimeError (what PyO3 turns them into)."""
import mmap
import os
import struct
from typing import Optional, Tuple

import numpy as np

CACHE_MAGIC = b"KRAS"
CACHE_VERSION_MARLIN = 3
CACHE_VERSION_CPU = 4
CACHE_VERSION_CPU_GGUF = 5
CACHE_HEADER_SIZE = 64


def fnv1a(data: bytes) -> int:
    """weights/mod.rs:887-894: the cache key is FNV-1a over the bytes of config.json"""
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def config_hash(model_dir: str) -> int:
    with open(os.path.join(model_dir, "config.json"), "rb") as f:
        return fnv1a(f.read())


def cache_dir_for_model(model_dir: str) -> str:
    """weights/mod.rs:898-909: ~/.krasis/cache/<model folder name>/, or <model_dir>/.krasis_cache/ without HOME"""
    name = os.path.basename(os.path.normpath(model_dir)) or "unknown_model"
    home = os.environ.get("HOME")
    return os.path.join(home, ".krasis", "cache", name) if home else os.path.join(model_dir, ".krasis_cache")


def cache_path_cpu(model_dir: str, num_bits: int, group_size: int) -> str:
    return os.path.join(cache_dir_for_model(model_dir), f"experts_cpu_int{num_bits}_g{group_size}.bin")


def cache_path_gguf_avx2(model_dir: str, group_size: int) -> str:
    """weights/mod.rs:930-933: the version-5 file built from a GGUF (gguf_native=False: de-quantized and re-quantized to INT4 / INT8, w13 and w2 widths may differ)"""
    return os.path.join(cache_dir_for_model(model_dir), f"experts_gguf_avx2_g{group_size}.bin")


def cache_path_marlin(model_dir: str, group_size: int, gpu_bits: int) -> str:
    return os.path.join(cache_dir_for_model(model_dir), f"experts_marlin_int{gpu_bits}_g{group_size}.bin")


def marlin_w2_padded_n(hidden: int, intermediate: int) -> int:
    return hidden + 64 if (hidden == intermediate and hidden % 256 != 0) else hidden


def cpu_expert_byte_sizes(h: int, m: int, gs: int, bits: int) -> Tuple[int, int, int, int]:
    """weights/mod.rs:972-995 (w13_packed, w13_scales, w2_packed, w2_scales) of one expert in the CPU transposed layout"""
    two_n = 2 * m
    if bits == 4:
        return (h // 8) * two_n * 4, (h // gs) * two_n * 2, (m // 8) * h * 4, (m // gs) * h * 2
    return ((h * two_n + 3) // 4) * 4, (h // gs) * two_n * 2, ((m * h + 3) // 4) * 4, (m // gs) * h * 2


def cpu_expert_byte_sizes_mixed(h: int, m: int, gs: int, w13_bits: int, w2_bits: int) -> Tuple[int, int, int, int]:
    """weights/mod.rs:1000-1017: the version-5 file's per-expert sizes, gate|up and down each at its own width"""
    a = cpu_expert_byte_sizes(h, m, gs, w13_bits); b = cpu_expert_byte_sizes(h, m, gs, w2_bits)
    return a[0], a[1], b[2], b[3]


def expected_gguf_cpu_cache_size(h, m, n_experts, gs, w13_bits, w2_bits, num_moe_layers, n_shared) -> int:
    """weights/mod.rs:1020-1041"""
    total = CACHE_HEADER_SIZE + num_moe_layers * n_experts * sum(cpu_expert_byte_sizes_mixed(h, m, gs, w13_bits, w2_bits))
    if n_shared > 0:
        total += num_moe_layers * sum(cpu_expert_byte_sizes_mixed(h, n_shared * m, gs, w13_bits, w2_bits))
    return total


def marlin_expert_byte_sizes(h: int, m: int, gs: int, bits: int, shared: bool = False) -> Tuple[int, int, int, int]:
    """weights/mod.rs:955-966; the shared expert's w2 is NOT padded in the file (:1100-1106)"""
    div = 8 if bits == 4 else 4
    h_w2 = h if shared else marlin_w2_padded_n(h, m)
    return (h // div) * (2 * m) * 4, (h // gs) * (2 * m) * 2, (m // div) * h_w2 * 4, (m // gs) * h_w2 * 2


def expected_cpu_cache_size(h, m, n_experts, gs, bits, num_moe_layers, n_shared) -> int:
    total = CACHE_HEADER_SIZE + num_moe_layers * n_experts * sum(cpu_expert_byte_sizes(h, m, gs, bits))
    if n_shared > 0:
        total += num_moe_layers * sum(cpu_expert_byte_sizes(h, n_shared * m, gs, bits))
    return total


def expected_marlin_cache_size(h, m, n_experts, gs, num_moe_layers, n_shared, bits) -> int:
    total = CACHE_HEADER_SIZE + num_moe_layers * n_experts * sum(marlin_expert_byte_sizes(h, m, gs, bits))
    if n_shared > 0:
        total += num_moe_layers * sum(marlin_expert_byte_sizes(h, n_shared * m, gs, bits, shared=True))
    return total


def pack_header(version: int, h: int, m: int, n_experts: int, num_moe_layers: int, gs: int, chash: int, n_shared: int, num_bits: int = 0) -> bytes:
    """write_marlin_cache_header / write_cpu_cache_header (weights/mod.rs:4117-4180): bytes 56..64 hold n_shared (v3) or n_shared | num_bits << 32 (v4)"""
    meta = n_shared if version == CACHE_VERSION_MARLIN else (n_shared | (num_bits << 32))
    return CACHE_MAGIC + struct.pack("<I6Q", version, h, m, n_experts, num_moe_layers, gs, chash) + struct.pack("<Q", meta)


def check_header(buf, kind: str, version: int, h: int, m: int, n_experts: int, total_moe_layers: int, gs: int, chash: int, n_shared: int,
                 expected_bits: Optional[int]) -> None:
    """the validation sequence of load_cpu_cache / load_marlin_cache, same order, same messages"""
    if len(buf) < CACHE_HEADER_SIZE:
        raise RuntimeError(f"{kind} cache too small for header")
    if bytes(buf[0:4]) != CACHE_MAGIC:
        raise RuntimeError(f"Bad magic in {kind} cache")
    v, fh, fm, fe, fl, fg, fc, meta = struct.unpack("<I7Q", bytes(buf[4:64]))
    if v != version:
        raise RuntimeError(f"Cache version {v}, expected {version} ({'Marlin' if version == CACHE_VERSION_MARLIN else 'CPU'})")
    if (fh, fm, fe, fl, fg) != (h, m, n_experts, total_moe_layers, gs):
        raise RuntimeError(f"{kind} cache header mismatch: file has {fh}h/{fm}m/{fe}e/{fl}L/g{fg}, expected {h}h/{m}m/{n_experts}e/{total_moe_layers}L/g{gs}")
    if fc != chash:
        raise RuntimeError(f"Config hash mismatch in {kind} cache")
    f_shared = meta & 0xFFFFFFFF
    if f_shared != n_shared:
        raise RuntimeError(f"Shared expert count mismatch: cache={f_shared}, config={n_shared}")
    if version == CACHE_VERSION_CPU:
        f_bits = (meta >> 32) & 0xFF
        if f_bits != expected_bits:
            raise RuntimeError(f"CPU cache num_bits mismatch: cache=INT{f_bits}, expected INT{expected_bits}")


def _dims(engine):
    c = engine._cfg
    gs = getattr(c, "group_size", 0) or 128
    return c.hidden_size, c.moe_intermediate_size, c.n_routed_experts, c.num_moe_layers, gs, c.n_shared_experts


def _open(path: str, kind: str):
    try:
        f = open(path, "rb")
    except OSError as ex:
        raise RuntimeError(f"Failed to open {kind} cache: {ex}")
    try:
        return f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    except (OSError, ValueError) as ex:
        f.close()
        raise RuntimeError(f"Failed to mmap {kind} cache: {ex}")


# ---------------------------------------------------------------------------------------------------------------- CPU transposed cache (v4)
def save_cpu_cache(engine, path: str, chash: int, num_bits: int, total_moe_layers: Optional[int] = None) -> int:
    """Write the engine's experts as the reference's version-4 file (streaming_build_cpu_cache's output, weights/mod.rs:2544-2790): header, then per
    (layer, expert) w13_packed | w13_scales | w2_packed | w2_scales, then per layer the shared expert (intermediate = n_shared * m).  Returns the size."""
    h, m, E, L, gs, ns = _dims(engine)
    L = total_moe_layers or L
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(pack_header(CACHE_VERSION_CPU, h, m, E, L, gs, chash, ns, num_bits))
        for layer in range(L):
            for e in range(E):
                for a in engine.download_expert(layer, e, num_bits):
                    f.write(a.tobytes())
        if ns > 0:
            for layer in range(L):
                for a in engine.download_expert(layer, -1, num_bits):
                    f.write(a.tobytes())
        size = f.tell()
    os.replace(tmp, path)
    assert size == expected_cpu_cache_size(h, m, E, gs, num_bits, L, ns)
    return size


def _close_map(mm, f) -> None:
    """Close a cache mapping.  When an upload raised, the traceback still references numpy views of the map and mmap.close() answers BufferError
    ("cannot close exported pointers exist"), which would mask the loader's own error: the map is then left to the garbage collector."""
    try:
        mm.close()
    except BufferError:
        pass
    f.close()


def load_cpu_cache(engine, path: str, chash: int, expected_bits: int, total_moe_layers: Optional[int] = None, start_moe_layer: int = 0,
                   num_layers_to_load: Optional[int] = None) -> None:
    """load_cpu_cache (weights/mod.rs:2794-2966) into a configured engine: layers [start, start + n) of the file become engine layers 0..n-1
    (the reference's partial load for pipeline stages).  Validation order and messages are the reference's."""
    h, m, E, L_eng, gs, ns = _dims(engine)
    total = total_moe_layers or L_eng
    n = num_layers_to_load if num_layers_to_load is not None else min(L_eng, total - start_moe_layer)
    f, mm = _open(path, "CPU")
    try:
        check_header(mm, "CPU", CACHE_VERSION_CPU, h, m, E, total, gs, chash, ns, expected_bits)
        if start_moe_layer + n > total:
            raise RuntimeError(f"Range [{start_moe_layer}, {start_moe_layer + n}) exceeds total MoE layers {total}")
        expected = expected_cpu_cache_size(h, m, E, gs, expected_bits, total, ns)
        if len(mm) != expected:
            raise RuntimeError(f"CPU cache size mismatch: expected {expected} bytes, got {len(mm)}")
        if n > L_eng:
            raise RuntimeError(f"engine was configured for {L_eng} MoE layers, cannot hold {n}")

        def upload(layer: int, expert: int, off: int, inter: int) -> int:      # read_unified_expert_cpu (weights/mod.rs:4318-4400), then straight to HBM
            p13, s13, p2, s2 = cpu_expert_byte_sizes(h, inter, gs, expected_bits)
            if expected_bits == 4:
                w13 = np.frombuffer(mm, np.uint32, p13 // 4, off).reshape(h // 8, 2 * inter)
                w2 = np.frombuffer(mm, np.uint32, p2 // 4, off + p13 + s13).reshape(inter // 8, h)
            else:
                w13 = np.frombuffer(mm, np.int8, h * 2 * inter, off).reshape(h, 2 * inter)
                w2 = np.frombuffer(mm, np.int8, inter * h, off + p13 + s13).reshape(inter, h)
            w13s = np.frombuffer(mm, np.uint16, s13 // 2, off + p13).reshape(h // gs, 2 * inter)
            w2s = np.frombuffer(mm, np.uint16, s2 // 2, off + p13 + s13 + p2).reshape(inter // gs, h)
            engine.load_unified_expert(layer, expert, w13, w13s, w2, w2s, num_bits=expected_bits)
            return off + p13 + s13 + p2 + s2

        per_layer = E * sum(cpu_expert_byte_sizes(h, m, gs, expected_bits))
        off = CACHE_HEADER_SIZE + start_moe_layer * per_layer
        for layer in range(n):
            for e in range(E):
                off = upload(layer, e, off, m)
        if ns > 0:
            per_shared = sum(cpu_expert_byte_sizes(h, ns * m, gs, expected_bits))
            off = CACHE_HEADER_SIZE + total * per_layer + start_moe_layer * per_shared
            for layer in range(n):
                off = upload(layer, -1, off, ns * m)
        engine.synchronize()
        engine._cpu_bits = engine._gpu_bits = expected_bits
    finally:
        _close_map(mm, f)


# ---------------------------------------------------------------------------------------------------------------- Marlin GPU cache (v3)
def save_marlin_cache(engine, path: str, chash: int, gpu_bits: int, total_moe_layers: Optional[int] = None) -> int:
    """version-3 file (streaming_build_marlin_cache, weights/mod.rs:1873-2210): Marlin-tiled words and permuted scales per expert, w2 padded per
    marlin_w2_padded_n for routed experts"""
    from ._lib import check
    from .engine import _addr
    h, m, E, L, gs, ns = _dims(engine)
    L = total_moe_layers or L
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"

    def export(layer: int, expert: int, shared: bool):
        if shared and marlin_w2_padded_n(h, ns * m) != h:
            raise RuntimeError("shared expert with hidden == intermediate and hidden % 256 != 0: the reference's file holds it unpadded (not handled)")
        sizes = marlin_expert_byte_sizes(h, (ns if shared else 1) * m, gs, gpu_bits, shared)
        bufs = [np.empty(sizes[0] // 4, np.uint32), np.empty(sizes[1] // 2, np.uint16), np.empty(sizes[2] // 4, np.uint32), np.empty(sizes[3] // 2, np.uint16)]
        check(engine._lib.kr_download_expert_marlin(engine._h, layer, expert, *[_addr(b) for b in bufs]))
        return bufs

    with open(tmp, "wb") as f:
        f.write(pack_header(CACHE_VERSION_MARLIN, h, m, E, L, gs, chash, ns))
        for layer in range(L):
            for e in range(E):
                for a in export(layer, e, False):
                    f.write(a.tobytes())
        if ns > 0:
            for layer in range(L):
                for a in export(layer, -1, True):
                    f.write(a.tobytes())
        size = f.tell()
    os.replace(tmp, path)
    assert size == expected_marlin_cache_size(h, m, E, gs, L, ns, gpu_bits)
    return size


def load_marlin_cache(engine, path: str, chash: int, gpu_bits: int, total_moe_layers: Optional[int] = None, start_moe_layer: int = 0,
                      num_layers_to_load: Optional[int] = None) -> None:
    """load_marlin_cache (weights/mod.rs:2367-2540) into a configured engine: the Marlin tiles are un-permuted into the resident layout on upload
    (kr_upload_expert_marlin)"""
    from ._lib import check
    from .engine import _addr
    h, m, E, L_eng, gs, ns = _dims(engine)
    total = total_moe_layers or L_eng
    n = num_layers_to_load if num_layers_to_load is not None else min(L_eng, total - start_moe_layer)
    f, mm = _open(path, "Marlin")
    try:
        check_header(mm, "Marlin", CACHE_VERSION_MARLIN, h, m, E, total, gs, chash, ns, None)
        if start_moe_layer + n > total:
            raise RuntimeError(f"Range [{start_moe_layer}, {start_moe_layer + n}) exceeds total MoE layers {total}")
        expected = expected_marlin_cache_size(h, m, E, gs, total, ns, gpu_bits)
        if len(mm) != expected:
            raise RuntimeError(f"Marlin cache size mismatch: expected {expected} bytes, got {len(mm)}")
        if n > L_eng:
            raise RuntimeError(f"engine was configured for {L_eng} MoE layers, cannot hold {n}")

        def upload(layer: int, expert: int, off: int, inter: int, shared: bool) -> int:
            sizes = marlin_expert_byte_sizes(h, inter, gs, gpu_bits, shared)
            if shared and marlin_w2_padded_n(h, inter) != h:
                raise RuntimeError("shared expert with hidden == intermediate and hidden % 256 != 0: the file holds it unpadded, the upload path pads (not handled)")
            arrs = []
            for sz, dt in zip(sizes, (np.uint32, np.uint16, np.uint32, np.uint16)):
                arrs.append(np.frombuffer(mm, dt, sz // np.dtype(dt).itemsize, off)); off += sz
            check(engine._lib.kr_upload_expert_marlin(engine._h, layer, expert, inter, *[_addr(a) for a in arrs], gpu_bits))
            return off

        per_layer = E * sum(marlin_expert_byte_sizes(h, m, gs, gpu_bits))
        off = CACHE_HEADER_SIZE + start_moe_layer * per_layer
        for layer in range(n):
            for e in range(E):
                off = upload(layer, e, off, m, False)
        if ns > 0:
            per_shared = sum(marlin_expert_byte_sizes(h, ns * m, gs, gpu_bits, shared=True))
            off = CACHE_HEADER_SIZE + total * per_layer + start_moe_layer * per_shared
            for layer in range(n):
                off = upload(layer, -1, off, ns * m, True)
        engine.synchronize()
        engine._cpu_bits = engine._gpu_bits = gpu_bits
    finally:
        _close_map(mm, f)


# ---------------------------------------------------------------------------------------------------------------- GGUF-sourced CPU cache (v5)
def pack_header_v5(h: int, m: int, n_experts: int, num_moe_layers: int, gs: int, chash: int, n_shared: int, w13_bits: int, w2_bits: int) -> bytes:
    """write_cpu_cache_header_v5 (weights/mod.rs:4176-4206): bytes 56..64 = n_shared (low 16 bits) | w13_bits << 48 | w2_bits << 56"""
    meta = (n_shared & 0xFFFF) | (w13_bits << 48) | (w2_bits << 56)
    return CACHE_MAGIC + struct.pack("<I6Q", CACHE_VERSION_CPU_GGUF, h, m, n_experts, num_moe_layers, gs, chash) + struct.pack("<Q", meta)


def save_gguf_cpu_cache(engine, path: str, chash: int, w13_bits: int, w2_bits: int, total_moe_layers: Optional[int] = None) -> int:
    """Write the engine's experts as the reference's version-5 file (streaming_build_cpu_cache_from_gguf's output, weights/mod.rs:3700-3905): the version-4
    body with gate|up at w13_bits and down at w2_bits."""
    h, m, E, L, gs, ns = _dims(engine)
    L = total_moe_layers or L
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(pack_header_v5(h, m, E, L, gs, chash, ns, w13_bits, w2_bits))
        for layer in range(L):
            for e in range(E):
                for a in engine.download_expert(layer, e, w13_bits, w2_bits):
                    f.write(a.tobytes())
        if ns > 0:
            for layer in range(L):
                for a in engine.download_expert(layer, -1, w13_bits, w2_bits):
                    f.write(a.tobytes())
        size = f.tell()
    os.replace(tmp, path)
    assert size == expected_gguf_cpu_cache_size(h, m, E, gs, w13_bits, w2_bits, L, ns)
    return size


def load_gguf_cpu_cache(engine, path: str, chash: int, total_moe_layers: Optional[int] = None, start_moe_layer: int = 0,
                        num_layers_to_load: Optional[int] = None) -> Tuple[int, int]:
    """load_gguf_cpu_cache (weights/mod.rs:3908-4062) into a configured engine; returns (w13_bits, w2_bits) as stored in the header.  Validation order and
    messages are the reference's."""
    h, m, E, L_eng, gs, ns = _dims(engine)
    total = total_moe_layers or L_eng
    n = num_layers_to_load if num_layers_to_load is not None else min(L_eng, total - start_moe_layer)
    f, mm = _open(path, "GGUF CPU")
    try:
        if len(mm) < CACHE_HEADER_SIZE:
            raise RuntimeError("GGUF CPU cache too small for header")
        if bytes(mm[0:4]) != CACHE_MAGIC:
            raise RuntimeError("Bad magic in GGUF CPU cache")
        v, fh, fm, fe, fl, fg, fc, meta = struct.unpack("<I7Q", bytes(mm[4:64]))
        if v != CACHE_VERSION_CPU_GGUF:
            raise RuntimeError(f"Cache version {v}, expected {CACHE_VERSION_CPU_GGUF} (GGUF CPU)")
        if (fh, fm, fe, fl, fg) != (h, m, E, total, gs):
            raise RuntimeError(f"GGUF CPU cache header mismatch: file has {fh}h/{fm}m/{fe}e/{fl}L/g{fg}, expected {h}h/{m}m/{E}e/{total}L/g{gs}")
        if fc != chash:
            raise RuntimeError("Config hash mismatch in GGUF CPU cache")
        f_shared, w13_bits, w2_bits = meta & 0xFFFF, (meta >> 48) & 0xFF, (meta >> 56) & 0xFF
        if f_shared != ns:
            raise RuntimeError(f"Shared expert count mismatch: cache={f_shared}, config={ns}")
        if w13_bits not in (4, 8):
            raise RuntimeError(f"Invalid w13_bits in cache: {w13_bits}")
        if w2_bits not in (4, 8):
            raise RuntimeError(f"Invalid w2_bits in cache: {w2_bits}")
        expected = expected_gguf_cpu_cache_size(h, m, E, gs, w13_bits, w2_bits, total, ns)
        if len(mm) != expected:
            raise RuntimeError(f"GGUF CPU cache size mismatch: expected {expected} bytes, got {len(mm)}")
        if start_moe_layer + n > total:
            raise RuntimeError(f"Range [{start_moe_layer}, {start_moe_layer + n}) exceeds total MoE layers {total}")
        if n > L_eng:
            raise RuntimeError(f"engine was configured for {L_eng} MoE layers, cannot hold {n}")

        def upload(layer: int, expert: int, off: int, inter: int) -> int:
            p13, s13, p2, s2 = cpu_expert_byte_sizes_mixed(h, inter, gs, w13_bits, w2_bits)
            w13 = (np.frombuffer(mm, np.uint32, p13 // 4, off).reshape(h // 8, 2 * inter) if w13_bits == 4
                   else np.frombuffer(mm, np.int8, h * 2 * inter, off).reshape(h, 2 * inter))
            w13s = np.frombuffer(mm, np.uint16, s13 // 2, off + p13).reshape(h // gs, 2 * inter)
            w2 = (np.frombuffer(mm, np.uint32, p2 // 4, off + p13 + s13).reshape(inter // 8, h) if w2_bits == 4
                  else np.frombuffer(mm, np.int8, inter * h, off + p13 + s13).reshape(inter, h))
            w2s = np.frombuffer(mm, np.uint16, s2 // 2, off + p13 + s13 + p2).reshape(inter // gs, h)
            engine.load_unified_expert(layer, expert, w13, w13s, w2, w2s, num_bits=w13_bits, w2_bits=w2_bits)
            return off + p13 + s13 + p2 + s2

        per_layer = E * sum(cpu_expert_byte_sizes_mixed(h, m, gs, w13_bits, w2_bits))
        off = CACHE_HEADER_SIZE + start_moe_layer * per_layer
        for layer in range(n):
            for e in range(E):
                off = upload(layer, e, off, m)
        if ns > 0:
            per_shared = sum(cpu_expert_byte_sizes_mixed(h, ns * m, gs, w13_bits, w2_bits))
            off = CACHE_HEADER_SIZE + total * per_layer + start_moe_layer * per_shared
            for layer in range(n):
                off = upload(layer, -1, off, ns * m)
        engine.synchronize()
        engine._cpu_bits = engine._gpu_bits = max(w13_bits, w2_bits)
        return w13_bits, w2_bits
    finally:
        _close_map(mm, f)
