"""`krasis.krasis` -- the names the reference's native module registers (src/lib.rs:14-24), bound to libkrasis_hip.so.

    KrasisEngine            src/moe.rs         -> krasis_amd.KrasisEngine
    CpuDecodeStore          src/decode.rs:193  -> krasis_amd.CpuDecodeStore
    WeightStore             src/weights/mod.rs -> a constructor-only class, like the reference's (#[new] is its only Python method:
                                                 loading goes through KrasisEngine.load / krasis_amd.weight_store)
    bench_decode_synthetic  src/decode.rs:4618 -> krasis_amd.bench_decode_synthetic
RustServer and system_check belong to the serving stack (out of scope): not provided.
"""
from krasis_amd import CpuDecodeStore, KrasisEngine, bench_decode_synthetic  # noqa: F401


class WeightStore:
    """weights/mod.rs:1139: `WeightStore()` builds an empty store; the reference exposes no further Python methods."""

    def __init__(self) -> None:
        self.group_size = 128
        self.cpu_num_bits = 4
        self.gpu_num_bits = 4


def __getattr__(name):
    if name in ("RustServer", "system_check"):
        raise AttributeError(f"krasis.krasis.{name} is part of the reference's serving stack, which this build does not provide")
    raise AttributeError(name)
