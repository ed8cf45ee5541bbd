#!/usr/bin/env python3
"""bench.py -- decode tok/s of Qwen3-Coder-Next (QCN) Q4 on MI355X, the reference's headline metric (BASELINE.json).

One "step" = one full decode token through the GPU decode graph (the reference's `decode_step`, src/decode.rs:2690):
embedding -> 48 x [fused add+RMSNorm -> gated-delta-net linear attention (36 layers) | gated GQA with FP16 KV (12 layers)
-> fused add+RMSNorm -> router (512 experts, softmax, top-10) -> 10 routed INT4-g128 experts + shared expert with sigmoid
gate] -> final norm -> lm_head (151936 x 2048 INT4) -> greedy argmax.  Same protocol as the reference's synthetic benchmark
(bench_decode_synthetic, decode.rs:4618): token 0, positions 10.., kv_max_seq 256, random weights / state with the reference's
value distributions, generated on the GPU.  Everything is resident in HBM before the timed region.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

QCN = dict(hidden=2048, inter=512, experts=512, topk=10, layers=48, shared_inter=512, vocab=151936, nk=16, nv=32, dk=128, dv=128,
           nh=16, nkv=2, hd=256, full_attn_interval=4, kv_max_seq=256, eps=1e-6)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable with a float4 copy)
B4 = 0.515625              # bytes per INT4-g128 weight incl. bf16 group scale
KINDS = ["embed", "fused_add_rmsnorm", "proj_matvec", "la_conv", "la_recurrent", "gated_rmsnorm_silu", "gqa", "route_logits",
         "route_select", "moe_w13", "moe_w2", "moe_combine", "lm_head", "argmax", "shared_gate"]
SYMBOL = {"proj_matvec": "kr_matvec_coop_kernel<float,4>", "lm_head": "kr_matvec_kernel<float,4>", "shared_gate": "kr_matvec_kernel<float,4>",
          "moe_w13": "kr_moe_w13_kernel<4>", "moe_w2": "kr_moe_w2_kernel<4,0>", "la_recurrent": "kr_la_step_kernel<128,128>",
          "route_logits": "kr_route_fused_decode_kernel<true,8>", "route_select": "kr_route_select_kernel",
          "fused_add_rmsnorm": "kr_fused_add_rmsnorm_kernel"}


def pmc_traffic(symbol):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 --pmc FETCH_SIZE pass (separate run, profiles/*pmc*.json; corrected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes).  None when no PMC summary is committed for that kernel."""
    import glob
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*pmc*.json")), reverse=True):
        try:
            d = json.load(open(f))
            if symbol in d.get("kernels", {}):
                return d["kernels"][symbol], os.path.basename(f)
        except Exception:
            pass
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=QCN["layers"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--prefill-chunk", type=int, default=0, help="tokens per chunk of the prompt pass (0 = library default)")
    ap.add_argument("--prefill-depth", type=int, default=0, help="chunks of the prompt pass in flight (0 = library default)")
    ap.add_argument("--prefill-reps", type=int, default=2, help="timed repetitions of the whole-model prompt pass")
    ap.add_argument("--no-long-context", action="store_true", help="skip the long-cache decode side measurement of the N = 1 line")
    ap.add_argument("--no-ep", action="store_true", help="skip the expert-parallel prompt-pass leg of the N > 1 lines")
    ap.add_argument("--ep-selftest", action="store_true", help="run the expert-parallel leg at N = 1 too (no collectives: checks the row path)")
    ap.add_argument("--prefill-tokens", type=int, default=8192, help="tokens per prefill chunk for the experts-only prefill side measurement (0 = skip)")
    return ap.parse_args()


def is_gqa(l):
    return (l + 1) % QCN["full_attn_interval"] == 0


def algorithmic_bytes(L):
    """Bytes a decode token must touch, each weight/state byte once, per kernel kind."""
    q = QCN; H, I, E, k = q["hidden"], q["inter"], q["experts"], q["topk"]
    n_la = sum(1 for l in range(L) if not is_gqa(l)); n_gqa = L - n_la
    group_dim = 2 * q["dk"] + 2 * q["dv"] * (q["nv"] // q["nk"])
    la_w = (q["nk"] * group_dim + q["nk"] * 2 * (q["nv"] // q["nk"])) * H + H * (q["nv"] * q["dv"])
    gqa_w = (q["nh"] * q["hd"] * 2 + 2 * q["nkv"] * q["hd"]) * H + H * (q["nh"] * q["hd"])
    b = {
        "proj_matvec": (n_la * la_w + n_gqa * gqa_w) * B4,
        "moe_w13": L * (k + 1) * H * 2 * I * B4,          # 10 routed + shared expert
        "moe_w2": L * (k + 1) * I * H * B4,
        "lm_head": q["vocab"] * H * B4,
        "route_logits": L * E * H * 2,                      # gate stored as bf16 in HBM
        "la_recurrent": n_la * 2 * q["nv"] * q["dk"] * q["dv"] * 4,   # state read + write
        "shared_gate": L * H * B4,
    }
    b["total"] = sum(b.values())
    return b


def build_qcn(rank, local_rank, L, rope_len=0):
    import numpy as np
    from krasis_amd import CpuDecodeStore, KrasisEngine, ModelConfig
    q = QCN; H, I, E, k, V = q["hidden"], q["inter"], q["experts"], q["topk"], q["vocab"]
    eng = KrasisEngine(device=local_rank)
    eng.configure(ModelConfig(H, I, E, k, L, 0, 1.0))
    bits = int(os.environ.get("KR_BENCH_BITS", "4"))            # probe hook: 8 = INT8-g128 weights everywhere (not the headline configuration)
    eng.fill_synthetic(bits, seed=0x12345678ABCDEF01 + rank)
    eng.set_routing_config("softmax", True, k, E, H)
    st = CpuDecodeStore(128, True, True)                       # norm_bias_one: qwen3_next (decode.rs:4701)
    st.set_moe_store(eng)
    rng = np.random.default_rng(1234 + rank)
    keep = []
    seed = [100 + rank * 100000]

    def W(rows, cols):
        seed[0] += 1
        return st.store_weight_synthetic(rows, cols, bits, seed[0])

    def N(n):
        w = ((rng.random(n, dtype=np.float32) - 0.5) * 0.2).astype(np.float32); keep.append(w)
        return st.store_norm_weight(w.ctypes.data, n)

    fin, lm = N(H), W(V, H)
    st.configure_decode(H, L, q["eps"], fin, lm, V, k, 1, True, 1.0, 0, synth_seed=777 + rank)
    nk, nv, dk, dv, nh, nkv, hd = q["nk"], q["nv"], q["dk"], q["dv"], q["nh"], q["nkv"], q["hd"]
    hr = nv // nk; group_dim = 2 * dk + 2 * dv * hr; conv_dim = 2 * nk * dk + nv * dv
    for l in range(L):
        n_in, n_post = N(H), N(H)
        if is_gqa(l):
            qw, kw, vw, ow = W(nh * hd * 2, H), W(nkv * hd, H), W(nkv * hd, H), W(H, nh * hd)
            qn = (rng.random(hd, dtype=np.float32) + 0.5).astype(np.float32); kn = (rng.random(hd, dtype=np.float32) + 0.5).astype(np.float32)
            keep += [qn, kn]
            st.add_decode_gqa_layer(n_in, n_post, qw, kw, vw, ow, qn.ctypes.data, hd, kn.ctypes.data, hd, True, nh, nkv, hd, 1.0 / hd ** 0.5)
        else:
            qkvz, ba, out = W(nk * group_dim, H), W(nk * 2 * hr, H), W(H, nv * dv)
            cw = ((rng.random(conv_dim * 4, dtype=np.float32) - 0.5) * 1.0).astype(np.float32)
            a_log = ((rng.random(nv, dtype=np.float32) - 0.5) * 2.0).astype(np.float32); dtb = ((rng.random(nv, dtype=np.float32) - 0.5)).astype(np.float32)
            nw = (rng.random(nv * dv, dtype=np.float32) + 0.5).astype(np.float32); keep += [cw, a_log, dtb, nw]
            st.add_decode_la_layer(n_in, n_post, qkvz, ba, out, cw.ctypes.data, a_log.ctypes.data, dtb.ctypes.data, nw.ctypes.data,
                                   nk, nv, dk, dv, 4, 1.0 / dk ** 0.5)
        # router gate +-0.02 (decode.rs:5181), rounded to bf16 like a real checkpoint -> stored as bf16 in HBM
        gate = ((rng.random((E, H), dtype=np.float32) - 0.5) * 0.04).astype(np.float32)
        gate = (gate.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        eng.set_route_weight_f32(l, gate)
        sgu, sd, sg = W(2 * q["shared_inter"], H), W(H, q["shared_inter"]), W(1, H)
        st.set_decode_layer_moe(l, l, l, sgu, sd, sg)
    half = hd // 2                                                    # decode.rs:5379: full rotary in the synthetic bench
    rope_len = max(rope_len, q["kv_max_seq"])                        # long enough for the prompt-pass measurement
    pos = np.arange(rope_len, dtype=np.float32)[:, None]
    freq = (1.0 / (10000.0 ** (2.0 * np.arange(half, dtype=np.float32) / hd))).astype(np.float32)[None, :]
    cos, sin = np.cos(pos * freq).astype(np.float32), np.sin(pos * freq).astype(np.float32); keep += [cos, sin]
    st.set_decode_rope(cos.ctypes.data, sin.ctypes.data, half, rope_len)
    st.finalize_decode()
    st.fill_state_synthetic(q["kv_max_seq"], seed=4242 + rank)
    return eng, st, keep


def prefill_experts(eng, L, M, torch):
    """Side measurement (NOT the headline value): the prefill expert path alone -- token sort + int8-MFMA grouped GEMM + combine of all
    L MoE layers for one chunk of M tokens with uniform random routing (k distinct experts per token).  Attention / linear-attention
    prefill kernels are not built yet, so this is an upper bound on prefill tok/s, reported with its MFMA roofline fraction."""
    from krasis_amd import GpuPrefillManager
    q = QCN; H, I, E, k = q["hidden"], q["inter"], q["experts"], q["topk"]
    g = torch.Generator(device="cuda").manual_seed(7)
    x = ((torch.rand((M, H), device="cuda", generator=g) - 0.5)).to(torch.bfloat16)
    ids = torch.rand((M, E), device="cuda", generator=g).topk(k, dim=1).indices.to(torch.int32)
    w = torch.softmax(torch.randn((M, k), device="cuda", generator=g), dim=1)
    mgr = GpuPrefillManager(eng, k)
    for l in range(min(L, 2)):
        mgr.forward(l, x, ids, w, routed_only=True)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for l in range(L):
        mgr.forward(l, x, ids, w, routed_only=True)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    macs = M * k * 3 * H * I * L                       # routed experts only
    tops = 2.0 * macs * 2 / (ms * 1e-3) / 1e12         # x2: two int8 MFMA passes (high / low activation digit) per MAC
    return {"tokens": M, "layers": L, "ms": ms, "tok_s_experts_only": M / (ms * 1e-3), "int8_TOPS_issued": tops,
            "mfma_i8_dense_peak_TOPS": 4400.0, "frac_of_i8_peak": tops / 4400.0,
            "effective_TFLOPs_2MAC": 2.0 * macs / (ms * 1e-3) / 1e12,
            "note": "experts only (sort + 2 grouped GEMMs + act + combine): the MFMA-bound part of the prompt pass in isolation"}


def prefill_ep(eng, L, M, world, rank, torch, dist):
    """Expert-parallel prompt-pass experts over RCCL (krasis_amd/ep.py, mode "alltoall"): every rank owns E/N experts and M
    tokens; each (token, slot) row travels once to the rank that owns its expert (all_to_all over the xGMI mesh), runs through the int8-MFMA
    expert GEMMs there, the f32 expert row comes back and the source rank combines its k rows in routing order -- bit-identical to one GPU
    (tests/test_ep_cpu.py world 2 on gloo, tests/test_ep_gpu.py).  All L MoE layers, uniform random routing, weak scaling (M tokens per rank).
    The bench engines are replicas, so a rank's slice is experts [0, E/N) of its resident synthetic set: same bytes, same arithmetic."""
    from krasis_amd.ep import ExpertParallelMoE, engine_row_ops
    q = QCN; H, I, E, k = q["hidden"], q["inter"], q["experts"], q["topk"]
    g = torch.Generator(device="cuda").manual_seed(7 + rank)
    x = ((torch.rand((M, H), device="cuda", generator=g) - 0.5)).to(torch.bfloat16)
    ids = torch.rand((M, E), device="cuda", generator=g).topk(k, dim=1).indices.to(torch.int32)
    w = torch.softmax(torch.randn((M, k), device="cuda", generator=g), dim=1)
    ops, combine = engine_row_ops(eng)
    ep = ExpertParallelMoE(ops, E, mode="alltoall")
    for l in range(min(L, 2)):
        ep.forward(l, x, ids, w, combine)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for l in range(L):
        ep.forward(l, x, ids, w, combine)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    macs = world * M * k * 3 * H * I * L
    off = (world - 1) / world                          # share of the rows that leave the GPU under uniform routing
    return {"tokens_per_gpu": M, "tokens_total": world * M, "layers": L, "ms": dt * 1e3, "tok_s_experts_only": world * M / dt, "scaling": "weak",
            "experts_per_gpu": E // world, "effective_TFLOPs_2MAC": 2.0 * macs / dt / 1e12,
            "exchange_GB_per_gpu_per_layer": {"dispatch_bf16": M * k * H * 2 * off / 1e9, "combine_f32": M * k * H * 4 * off / 1e9},
            "note": "sort by owner + all_to_all dispatch + expert GEMMs + all_to_all combine, per layer, no overlap between layers; "
                    "compare with prefill_experts_only of the N = 1 line"}


def prefill_model(st, L, P, reps, torch):
    """Whole-model prompt pass (kr_decode_prefill): P synthetic tokens through all L layers (projection + expert GEMMs on int8 MFMA, exact
    gated-delta-rule recurrence, exact causal GQA attention, router, norms), bit-identical to token-by-token decode.  tok/s = P / time."""
    import numpy as np
    q = QCN
    st.fill_state_synthetic(P + 64, 7)                      # FP16 KV caches / states sized for the prompt
    toks = [int(x) for x in np.random.default_rng(5).integers(0, q["vocab"], P)]
    st.prefill(toks, 0)                                     # warm-up: scratch arena, per-weight nibble sums
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        st.prefill(toks, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    H, I, k = q["hidden"], q["inter"], q["topk"]
    n_la = sum(1 for l in range(L) if not is_gqa(l)); n_gqa = L - n_la
    hr = q["nv"] // q["nk"]; gd = 2 * q["dk"] + 2 * q["dv"] * hr
    w_la = q["nk"] * gd * H + q["nk"] * 2 * hr * H + H * q["nv"] * q["dv"]
    w_gqa = (q["nh"] * q["hd"] * 2 + 2 * q["nkv"] * q["hd"]) * H + H * q["nh"] * q["hd"]
    w_moe = (k + 1) * 3 * H * I + H                          # routed + shared expert (+ its gate row)
    macs = P * (n_la * w_la + n_gqa * w_gqa + L * w_moe)     # GEMM MACs (lm_head runs for the last token only)
    tops = 2.0 * macs * 2 / dt / 1e12                        # two int8 MFMA passes (high / low activation digit) per MAC
    return {"value": P / dt, "unit": "tok/s", "tokens": P, "ms": dt * 1e3, "reps": reps, "layers": L, "target_tok_s": 3300,
            "roofline": {"bound": "mfma", "achieved": tops, "peak": 4400.0, "unit": "TOP/s (int8, 2 digit passes per MAC)", "frac": tops / 4400.0,
                         "effective_TFLOPs_2MAC": 2.0 * macs / dt / 1e12},
            "note": "bit-identical to decoding the prompt token by token (tests/test_prefill_model_gpu.py); attention and the gated delta rule are "
                    "evaluated in the decode order on the vector ALUs, GEMM-shaped work on the matrix cores"}


def cpu_baseline(max_seconds, L):
    """The reference's CPU decode arithmetic (AVX2 integer INT4 kernel on tiled weights, src/kernel/avx2.rs:1066, OpenMP over
    256-column tiles like rayon) timed on this host.  The sample is the weight set ONE decoded token touches -- L MoE layers (10 routed +
    shared experts each, distinct per layer), the LA / GQA projections of every layer and lm_head, ~1.9 GB, so the pass streams from DRAM
    like the real decode instead of living in the last-level cache -- run as whole-token passes for the time budget.  Norms, recurrent
    state and attention are left out (optimistic for the CPU).  Weights use the reference's xorshift generator."""
    import numpy as np
    from oracle import oracle as O
    q = QCN; H, I, k = q["hidden"], q["inter"], q["topk"]
    rng = O.Xorshift64()

    def tiled(rows, cols):
        p = rng.fill_u32(cols // 8 * rows).reshape(cols // 8, rows); s = rng.fill_scales_bf16(cols // 128 * rows).reshape(cols // 128, rows)
        return O.repack_tiled_u32(p), O.repack_tiled_u16(s), cols, rows

    def expert_set():
        out = []
        for _ in range(k + 1):
            e = O.UnifiedExpert(rng.fill_u32(H // 8 * 2 * I).reshape(H // 8, 2 * I), rng.fill_scales_bf16(H // 128 * 2 * I).reshape(H // 128, 2 * I),
                                rng.fill_u32(I // 8 * H).reshape(I // 8, H), rng.fill_scales_bf16(I // 128 * H).reshape(I // 128, H), H, I)
            out.append(O.tile_expert(e))
        return out

    group_dim = 2 * q["dk"] + 2 * q["dv"] * (q["nv"] // q["nk"])
    layers = []
    for l in range(L):
        proj = ([tiled(q["nh"] * q["hd"] * 2 + 2 * q["nkv"] * q["hd"], H), tiled(H, q["nh"] * q["hd"])] if is_gqa(l)
                else [tiled(q["nk"] * group_dim, H), tiled(H, q["nv"] * q["dv"])])
        layers.append((expert_set(), proj))
    lm = tiled(q["vocab"], H)
    act = O.f32_to_bf16(rng.fill_f32(H, 0.5)); w = np.full(k + 1, 1.0 / (k + 1), np.float32)
    x2048 = rng.fill_f32(H, 0.5); x4096 = rng.fill_f32(4096, 0.5)
    qa, sa = O.quant_act_int16_f32(x2048); qb, sb = O.quant_act_int16_f32(x4096)

    def mv(t, qx, sx):
        O.matvec_int4_tiled_avx2(t[0], t[1], qx, sx, t[2], t[3])

    def token():
        for experts, proj in layers:
            mv(proj[0], qa, sa); mv(proj[1], qb, sb)
            O.moe_forward_unified_tiled_avx2(experts, w, act)
        mv(lm, qa, sa)

    def timed(fn, budget):
        fn(); n, t0 = 0, time.perf_counter()
        while n == 0 or time.perf_counter() - t0 < budget:
            fn(); n += 1
        return (time.perf_counter() - t0) / n

    # thread count: the reference warns that SMT / too many threads hurts (CHANGELOG.md:24); a quick sweep picks the team size
    hw = O.num_threads()
    cands = sorted({c for c in (8, 16, 32, 64, hw) if c <= hw})
    best_c, best_t, sweep = hw, None, {}
    for c in cands:
        O.set_num_threads(c)
        tt = timed(token, max_seconds * 0.4 / len(cands)); sweep[c] = round(1.0 / tt, 2)
        if best_t is None or tt < best_t:
            best_c, best_t = c, tt
    O.set_num_threads(best_c)
    t_tok = timed(token, max_seconds * 0.6)
    return dict(value=1.0 / t_tok, unit="tok/s", cores=best_c, host_threads=hw, kind="port", tok_s_by_threads=sweep,
                sample="whole-token passes over one token's weight set (%d MoE layers x (10 routed + shared INT4 experts) + LA/GQA projections + lm_head, "
                       "~1.9 GB streamed from DRAM), AVX2+OpenMP port of avx2.rs:1066 on tiled weights; norms/attention/state omitted "
                       "(optimistic for the CPU)" % L, ms_per_token=t_tok * 1e3)


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=5))

    from krasis_amd import _lib
    L = args.layers
    eng, st, keep = build_qcn(rank, local_rank, L, max(args.prefill_tokens + 64, 8192))      # rope table: prompt pass and the long-cache side measurement
    st.set_use_graph(not args.no_graph)
    kvm = QCN["kv_max_seq"]

    def step(i):
        st.decode_step(0, (10 + i) % (kvm - 1))

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())

    # per-kernel durations: un-graphed steps with HIP events around every launch on the launch stream
    ms = (C.c_double * 16)(); cnt = (C.c_long * 16)()
    tot_ms = [0.0] * 16; tot_n = [0] * 16; P = 5
    for i in range(P):
        _lib.check(st._lib.kr_decode_profile_step(st._h, 0, (10 + i) % (kvm - 1), ms, cnt, 16))
        for j in range(15):
            tot_ms[j] += ms[j]; tot_n[j] += cnt[j]
    per_kind_us = {KINDS[j]: (tot_ms[j] / P) * 1e3 for j in range(15)}            # us per step
    per_launch_us = {KINDS[j]: (tot_ms[j] / max(tot_n[j], 1)) * 1e3 for j in range(15)}

    prefill = None; prefill_full = None
    if args.prefill_tokens > 0 and world == 1:       # side measurements (prompt pass, CPU baseline) belong to the N = 1 line only
        prefill = prefill_experts(eng, L, args.prefill_tokens, torch)
        if args.prefill_chunk:
            st.set_prefill_chunk(args.prefill_chunk)
        if args.prefill_depth:
            st.set_prefill_depth(args.prefill_depth)
        prefill_full = prefill_model(st, L, args.prefill_tokens, args.prefill_reps, torch)
        prefill_full["chunk"] = args.prefill_chunk or 1024; prefill_full["chunks_in_flight"] = args.prefill_depth or 3

    long_ctx = None
    if world == 1 and not args.no_long_context:          # side measurement: the same decode step late in a long cache (split attention launches)
        try:
            kv_long = 8192
            st.fill_state_synthetic(kv_long, 7)
            for i in range(3):
                st.decode_step(0, kv_long - 6 + i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(20):
                st.decode_step(0, kv_long - 2)
            torch.cuda.synchronize()
            d1 = (time.perf_counter() - t1) / 20
            long_ctx = {"kv_max_seq": kv_long, "position": kv_long - 2, "ms_per_step": d1 * 1e3, "tok_s": 1.0 / d1,
                        "note": "GQA layers read 8190 cached positions per head: scores over (heads x 256-position blocks) workgroups, then softmax + p.v "
                                "with a column-major V stage on producer / consumer waves; the headline value follows the reference protocol (positions 10..)"}
        except Exception as ex:
            long_ctx = {"error": repr(ex)}

    ep_leg = None
    if args.prefill_tokens > 0 and not args.no_ep and (world > 1 or args.ep_selftest):   # every rank takes part; same call sequence on all
        try:
            ep_leg = prefill_ep(eng, L, args.prefill_tokens, world, rank, torch, dist)
        except Exception as ex:
            ep_leg = {"error": repr(ex)}

    if rank == 0:
        ab = algorithmic_bytes(L)
        sym_us, sym_bytes, sym_n = {}, {}, {}
        for j in range(15):
            kname = KINDS[j]; sym = SYMBOL.get(kname, kname)
            sym_us[sym] = sym_us.get(sym, 0.0) + per_kind_us[kname]; sym_bytes[sym] = sym_bytes.get(sym, 0.0) + ab.get(kname, 0.0)
            sym_n[sym] = sym_n.get(sym, 0) + tot_n[j] / P
        dom = max(sym_us, key=lambda s: sym_us[s])
        achieved = sym_bytes[dom] / (sym_us[dom] * 1e-6) / 1e9 if sym_us[dom] > 0 else 0.0
        tok_s = world * args.steps / dt
        traffic, traffic_src = pmc_traffic(dom)
        res = {
            "metric": "decode tok/s, Qwen3-Coder-Next Q4 @%d MI355X" % world,
            "value": tok_s, "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int4-g128 weights x int16 activations -> i32, f32 scale chain (reference CPU-decode numerics, bit-exact)",
            "data": "synthetic",
            "config": {"workload": "Qwen3-Coder-Next Q4 int4gpu on 1xMI355X (512-expert top-10, hybrid linear+GQA)",
                       "scope": "full decode_step: embedding, %d layers (LA/GQA + MoE + shared expert), final norm, lm_head, greedy sample" % L,
                       "kv": "FP16 KV cache (reference CPU-decode numerics), kv_max_seq %d" % kvm, "layers": L,
                       "parallelism": "replica x%d (QCN fits one GPU; decode is not expert-parallel)" % world,
                       "hip_graph": not args.no_graph, "target_tok_s": 200},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "peak_measured_stream_read": 6996.0, "peak_measured_source": "tools/probes/hbm_stream.hip on this box type (8 GiB, 16-byte loads)", "traffic": traffic, "traffic_unit": "HBM fetch bytes per launch (PMC FETCH_SIZE, separate pass)",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": sym_bytes[dom] / max(sym_n[dom], 1), "us_per_launch": sym_us[dom] / max(sym_n[dom], 1),
                         "launches_per_step": sym_n[dom],
                         "step_algorithmic_bytes": ab["total"], "step_effective_GBs": ab["total"] * (args.steps / dt) / 1e9,
                         "step_frac_of_hbm_peak": ab["total"] * (args.steps / dt) / 1e9 / HBM_PEAK_GBS,
                         "per_kind_us_per_step": {k_: round(v, 2) for k_, v in per_kind_us.items()},
                         "per_kind_us_per_launch": {k_: round(v, 2) for k_, v in per_launch_us.items()}},
        }
        if prefill_full is not None:
            res["prefill"] = prefill_full
        if prefill is not None:
            res["prefill_experts_only"] = prefill
        if ep_leg is not None:
            res["prefill_experts_ep_alltoall"] = ep_leg
        if long_ctx is not None:
            res["decode_long_context"] = long_ctx
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, L)
            except Exception as ex:  # a reported side number, never the product path
                res["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
