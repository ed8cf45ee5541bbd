#!/usr/bin/env python3
"""bench.py -- Qwen3-Coder-Next (QCN) Q4 hot-path benchmark on MI355X.  Contract: see the task statement.

One "step" = one decode token through the hot path that exists on the GPU (see `config.scope` in the JSON line):
  scope "moe"  : per token, for each of the 48 MoE layers: router (512 experts, top-10, f32 gate) + 10 routed INT4-g128
                 experts + the shared expert.  Attention / linear-attention / lm_head are NOT included in this scope.
Inputs are resident in HBM before the timed region.  Synthetic data: GPU-generated pseudo-random INT4 words and bf16
scales in [0.005, 0.05] (the distribution of the reference's bench_decode_synthetic, src/decode.rs:4379-4392), router
gate uniform +-0.02 (decode.rs:5181), hidden uniform +-0.5 (decode.rs:5437).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Qwen3-Coder-Next dims (src/decode.rs:4670-4692)
QCN = dict(hidden=2048, inter=512, experts=512, topk=10, layers=48, n_shared=1)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
BYTES_PER_W_INT4 = 0.515625


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=QCN["layers"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(max_seconds):
    """Reference CPU decode experts (AVX2 integer kernel on tiled weights, src/kernel/avx2.rs:1066, via the oracle port),
    timed on this box's host cores on a bounded sample: one MoE layer (10 routed experts) per 'layer-token'."""
    import numpy as np
    from oracle import oracle as O
    H, I, k = QCN["hidden"], QCN["inter"], QCN["topk"]
    rng = O.Xorshift64()
    experts = []
    for _ in range(k):
        e = O.UnifiedExpert(rng.fill_u32(H // 8 * 2 * I).reshape(H // 8, 2 * I), rng.fill_scales_bf16(H // 128 * 2 * I).reshape(H // 128, 2 * I),
                            rng.fill_u32(I // 8 * H).reshape(I // 8, H), rng.fill_scales_bf16(I // 128 * H).reshape(I // 128, H), H, I)
        experts.append(O.tile_expert(e))
    act = O.f32_to_bf16(rng.fill_f32(H, 0.5))
    w = np.full(k, 1.0 / k, np.float32)
    for _ in range(3):
        O.moe_forward_unified_tiled_avx2(experts, w, act)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < max_seconds:
        for _ in range(20):
            O.moe_forward_unified_tiled_avx2(experts, w, act)
        n += 20
    dt = time.perf_counter() - t0
    layer_ms = dt / n * 1e3
    return dict(value=1e3 / (layer_ms * QCN["layers"]), unit="tok/s", cores=O.num_threads(), kind="port",
                sample=f"{n} x one QCN MoE layer (10 routed INT4 experts, weights L3-resident: optimistic for the CPU), "
                       f"scaled to {QCN['layers']} layers; routed experts only",
                layer_ms=layer_ms)


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from krasis_amd import KrasisEngine, ModelConfig, _lib
    L = args.layers
    H, I, E, k = QCN["hidden"], QCN["inter"], QCN["experts"], QCN["topk"]
    eng = KrasisEngine(device=local_rank)
    eng.configure(ModelConfig(H, I, E, k, L, QCN["n_shared"], 1.0))
    eng.fill_synthetic(4, seed=0x12345678ABCDEF01 + rank)
    eng.set_routing_config("softmax", True, k, E, H)
    g = torch.Generator().manual_seed(1234 + rank)
    for l in range(L):
        gate = ((torch.rand((E, H), generator=g) - 0.5) * 0.04).numpy()
        eng.set_route_weight_f32(l, gate)
    x32 = ((torch.rand((L, H), generator=g) - 0.5)).cuda()            # hidden +-0.5, one row per layer
    xbf = x32.to(torch.bfloat16).contiguous()                          # decode.rs:3307: experts see bf16(hidden)
    ids = torch.empty((k,), dtype=torch.int32, device="cuda")
    wts = torch.empty((k,), dtype=torch.float32, device="cuda")
    out = torch.empty((L, H), dtype=torch.float32, device="cuda")
    lib, h = eng._lib, eng._h
    st = torch.cuda.current_stream().cuda_stream

    def step():
        for l in range(L):
            _lib.check(lib.kr_route_topk(h, l, x32[l].data_ptr(), 1, _lib.KR_ROUTE_RULE_DECODE, ids.data_ptr(), wts.data_ptr(), None, st))
            _lib.check(lib.kr_moe_forward(h, l, xbf[l].data_ptr(), ids.data_ptr(), wts.data_ptr(), out[l].data_ptr(), 1, k,
                                          _lib.KR_OUT_F32, 0, st))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())

    # per-kernel durations with HIP events on the launch stream (separate pass: events serialise launches)
    _lib.check(lib.kr_set_profiling(h, 1))
    import ctypes as C
    for _ in range(max(3, min(args.steps, 20))):
        step()
    torch.cuda.synchronize()
    prof = {}
    for kind, name in enumerate(["kr_moe_w13_kernel", "kr_moe_w2_kernel", "kr_moe_combine_kernel"]):
        ms, n = C.c_double(), C.c_long()
        _lib.check(lib.kr_get_profile(h, kind, C.byref(ms), C.byref(n)))
        prof[name] = (ms.value / max(n.value, 1)) * 1e3  # us per launch
    _lib.check(lib.kr_set_profiling(h, 0))

    if rank == 0:
        n_sh = QCN["n_shared"]
        w13_bytes = (k + n_sh) * H * 2 * I * BYTES_PER_W_INT4      # algorithmic bytes of ONE w13 launch (weights once)
        w2_bytes = (k + n_sh) * I * H * BYTES_PER_W_INT4
        dom = "kr_moe_w13_kernel"
        achieved = w13_bytes / (prof[dom] * 1e-6) / 1e9
        res = {
            "metric": "decode tok/s, Qwen3-Coder-Next Q4 (INT4-g128 experts) @%d MI355X" % world,
            "value": world * args.steps / dt, "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int4 weights x int16 activations -> i32, f32 scale chain", "data": "synthetic",
            "config": {"workload": "Qwen3-Coder-Next Q4 int4gpu on 1xMI355X (512-expert top-10)", "scope": "moe",
                       "scope_note": "router + routed experts + shared expert of all %d MoE layers per token; attention, linear "
                                     "attention, norms and lm_head are not yet on the GPU path, so this is NOT the full decode step" % L,
                       "layers": L, "hidden": H, "moe_intermediate": I, "experts": E, "topk": k,
                       "parallelism": "replica x%d" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": w13_bytes, "us_per_launch": prof[dom],
                         "other_kernels_us": {n: v for n, v in prof.items() if n != dom},
                         "w2_achieved_GBs": w2_bytes / (prof["kr_moe_w2_kernel"] * 1e-6) / 1e9},
        }
        if not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            except Exception as ex:  # the baseline is a reported side number, never the product path
                res["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
