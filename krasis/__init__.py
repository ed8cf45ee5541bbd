"""Drop-in name for the reference's Python package: `from krasis import KrasisEngine, CpuDecodeStore, ...` resolves to the MI355X-native classes.

Only the hot path's names exist: the engine, the decode store, the prefill operator, the synthetic decode benchmark.  The
reference's server, launcher, tokenizer and VRAM-budget modules are out of scope and deliberately absent -- importing them fails loudly."""
from krasis.krasis import KrasisEngine, WeightStore, CpuDecodeStore, bench_decode_synthetic  # noqa: F401
from krasis_amd import GpuPrefillManager  # noqa: F401

__version__ = "mi355x-native"
