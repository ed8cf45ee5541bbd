"""krasis_amd -- MI355X-native quantized-MoE hot path behind the Krasis engine API.

Only what the hot path needs lives here: the HIP kernels + C ABI (csrc/, libkrasis_hip.so) and
the host-side mirror of the reference's operator interface (KrasisEngine, GpuPrefillManager, CpuDecodeStore names).
There is NO CPU fallback: every operator fails loudly if libkrasis_hip.so or a GPU is missing.
"""
from ._lib import KrasisHipError, lib_path, load_library  # noqa: F401
from .engine import KrasisEngine, ModelConfig  # noqa: F401
from .decode_store import CpuDecodeStore  # noqa: F401
from .prefill import GpuPrefillManager  # noqa: F401
from .perplexity import evaluate_perplexity  # noqa: F401
from .synthetic import bench_decode_synthetic  # noqa: F401

__all__ = ["KrasisEngine", "ModelConfig", "CpuDecodeStore", "GpuPrefillManager", "evaluate_perplexity", "bench_decode_synthetic", "KrasisHipError", "load_library", "lib_path"]
