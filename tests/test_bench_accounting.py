"""bench.py's roofline accounting against the figures SURVEY.md section 8(d) states for Qwen3-Coder-Next INT4-g128 (each touched weight / state
byte counted once per decode token).  Pure arithmetic: runs without a GPU."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_layer_mix(bench):
    L = bench.QCN["layers"]
    gqa = [l for l in range(L) if bench.is_gqa(l)]
    assert L == 48 and len(gqa) == 12 and gqa[0] == 3            # 36 linear-attention + 12 gated GQA layers (decode.rs:4670-4692)


def test_algorithmic_bytes_match_survey(bench):
    b = bench.algorithmic_bytes(48)
    mb = lambda x: x / 1e6
    assert abs(mb(b["moe_w13"] + b["moe_w2"]) - (778.6 + 77.9)) < 0.2          # routed 778.6 MB + shared 77.9 MB
    assert abs(mb(b["proj_matvec"]) - 794.0) < 0.5                             # LA + GQA projections, 1.540 G weights
    assert abs(mb(b["lm_head"]) - 160.4) < 0.1
    assert abs(mb(b["route_logits"]) - 100.7) < 0.1                            # gate as bf16
    assert abs(mb(b["la_recurrent"]) - 151.0) < 0.1                            # 36 layers x 2 MiB read + write
    assert abs(b["total"] / 1e9 - 2.06) < 0.01                                 # "Total ~ 2.06 GB/token"
    assert b["total"] == sum(v for k, v in b.items() if k != "total")


def test_bytes_per_weight_constant(bench):
    assert bench.B4 == 0.515625                                               # INT4-gs128: 4 bits + a bf16 scale per 128 weights
    assert bench.HBM_PEAK_GBS == 8000.0
