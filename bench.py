#!/usr/bin/env python3
"""bench.py -- decode tok/s of Qwen3-Coder-Next (QCN) Q4 on MI355X, the reference's headline metric (BASELINE.json, config 3).

One "step" = one full decode token through the GPU decode graph (the reference's `decode_step`, src/decode.rs:2690):
embedding -> 48 x [fused add+RMSNorm -> gated-delta-net linear attention (36 layers) | gated GQA with an FP8-E4M3 KV cache (12 layers)
-> fused add+RMSNorm -> router (512 experts, softmax, top-10) -> 10 routed INT4-g128 experts + shared expert with sigmoid
gate] -> final norm -> lm_head (151936 x 2048 INT4) -> greedy argmax.  Same protocol as the reference's synthetic benchmark
(bench_decode_synthetic, decode.rs:4618): token 0, positions 10.., kv_max_seq 256, random weights / state with the reference's
value distributions (router gate from the reference's xorshift64 stream), generated on the GPU.  Everything is resident in HBM before
the timed region.

Side measurements on the N = 1 line (never the headline `value`): the whole-model prompt pass at the reference benchmark's prompt
lengths (--prefill-tokens, default 8192,20434,35139,49863: benchmark.py:434-505 / SURVEY 8d), the expert path alone, the same decode
step late in a long cache, the other BASELINE configurations that fit one GPU (--side-configs: V2-Lite-shaped MLA model = config 2,
QCN with INT8-g128 weights = config 5), and the CPU baseline (oracle port, incl. the V2-Lite Q4_K CPU expert pass = config 1).
On N > 1 lines: the expert-parallel prompt-pass experts over RCCL (QCN and the Qwen3-235B expert shape = config 4).
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

QCN = dict(hidden=2048, inter=512, experts=512, topk=10, layers=48, shared_inter=512, vocab=151936, nk=16, nv=32, dk=128, dv=128,
           nh=16, nkv=2, hd=256, full_attn_interval=4, kv_max_seq=256, eps=1e-6)
# DeepSeek-V2-Lite (H 2048, I 1408, 64 experts top-6, 2 shared, 27 layers = 1 dense + 26 MoE, MLA 16 heads, kv_lora 512, nope 128, rope 64, v 128)
V2L = dict(hidden=2048, inter=1408, experts=64, topk=6, layers=27, n_shared=2, vocab=102400, nh=16, klr=512, nd=128, rd=64, vhd=128, dense_inter=10944,
           kv_max_seq=256, eps=1e-6)
# Qwen3-235B-A22B (config 4; SURVEY 8: H 4096, I 1536, 128 experts top-8, 94 layers, GQA; head counts / vocab from the public model card: 64 q heads, 4 kv heads,
# head_dim 128, per-head QK-norm, no shared expert, vocab 151936).  INT4-g128: experts 117 GB + attention 3.5 GB + lm_head 0.3 GB -- the whole model fits one MI355X.
Q235 = dict(hidden=4096, inter=1536, experts=128, topk=8, layers=94, vocab=151936, nh=64, nkv=4, hd=128, kv_max_seq=256, eps=1e-6)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable with a float4 copy)
I8_PEAK_TOPS = 5000.0      # dense int8 MFMA peak the roofline is priced against: 2 x the 2.5 PFLOP/s dense bf16 peak (MI355X_MICROARCH.md: "I8 ~ 2 x bf16 rate (2 x K)"; the guide gives no
                           # spec line of its own for int8 and a measured >= 3944 TOP/s for a 16x16x64 microbenchmark, reported beside every int8 fraction as peak_measured_ubench)
F16_PEAK_TFLOPS = 2500.0            # dense f16 / bf16 MFMA peak (MI355X_MICROARCH.md)
B4 = 0.515625              # bytes per INT4-g128 weight incl. bf16 group scale
B8 = 1.015625              # INT8-g128
KINDS = ["embed", "fused_add_rmsnorm", "proj_matvec", "la_conv", "la_recurrent", "gated_rmsnorm_silu", "gqa", "route_logits",
         "route_select", "moe_w13", "moe_w2", "moe_combine", "lm_head", "argmax", "shared_gate", "out_proj_matvec"]
NK = len(KINDS)
SYMBOL = {"proj_matvec": "kr_matvec_coop_kernel<float,4>", "lm_head": "kr_matvec_kernel<float,4>", "shared_gate": "kr_matvec_kernel<float,4>",
          "moe_w13": "kr_moe_w13_kernel<4>", "moe_w2": "kr_moe_w2_kernel<4,0>", "la_recurrent": "kr_la_step_kernel<128,128>",
          "route_logits": "kr_route_fused_decode_kernel<true,8>", "route_select": "kr_route_select_kernel",
          "fused_add_rmsnorm": "kr_fused_add_rmsnorm_kernel"}
# the same kinds in KR_DECODE_FAST (kr_decode_fast.hip): the norms ride in the projection / router launches, top-k + silu*up in the gate|up launch, the combine in the down launch
SYMBOL_FAST = {"proj_matvec": "kr_fdm_kernel<4,1,8,1>", "out_proj_matvec": "kr_fdm_kernel<4,4,4,0>", "lm_head": "kr_fdm_kernel<4,1,8,1>@grid1215488",      # (final norm + vocabulary projection: the in-projection kernel on ceil(151936 / 32) workgroups of 256 threads -- its own row in the PMC summary)
               "moe_w13": "kr_fw13_kernel<4,4>", "moe_w2": "kr_fw2_kernel<4,2,false>",
               "la_recurrent": "kr_fla_kernel<128,128>", "route_logits": "kr_frt_kernel<true,4>", "fused_add_rmsnorm": "kr_fused_add_rmsnorm_kernel"}
WORKLOAD = {"qcn-q4": "Qwen3-Coder-Next Q4 int4gpu on 1×MI355X (512-expert top-10, hybrid linear+GQA, FP8 KV)",
            "qcn-q8": "Qwen3-Coder-Next Q8 int8gpu on 1×MI355X (int8 MFMA path, Q8_0 dequant)",
            "v2lite-q4": "DeepSeek-V2-Lite Q4 int4gpu on 1×MI355X (MLA + 64-expert top-6)",
            "qcn-q4k-gguf": "Qwen3-Coder-Next, routed experts as native GGUF Q4_K super-blocks (gguf_native), decode step + whole-model prompt pass on 1×MI355X",
            "v2lite-q4k-gguf": "DeepSeek-V2-Lite, routed experts as native GGUF blocks (Q4_K gate / up, Q8_0 down: the int4cpu build of BASELINE config 1) decoded on 1×MI355X -- the GPU twin of cpu_baseline.v2lite_q4k_cpu",
            "qwen3-235b-q4": "Qwen3-235B-A22B Q4 int4gpu, the WHOLE model resident on 1×MI355X (94 GQA layers, 128-expert top-8; BASELINE config 4 names expert parallelism on 8 GPUs: see prefill_experts_ep_235b on the N > 1 lines)"}


def pmc_traffic(symbol):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 --pmc FETCH_SIZE pass (separate run, profiles/*pmc*.json; corrected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes).  None when no PMC summary is committed for that kernel."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json")), reverse=True):
        try:
            d = json.load(open(f))
            ks = d.get("kernels", {})
            if symbol in ks:
                return ks[symbol], os.path.basename(f)
            if "|" in symbol:      # a kind launched in several template forms, equally often ("kr_fdm_kernel<4,1,8>|<4,4,4>"): mean over the forms
                parts = symbol.split("|"); base = parts[0].split("<")[0]
                names = [parts[0]] + [(base + x if x.startswith("<") else x) for x in parts[1:]]
                if all(n in ks for n in names):
                    return sum(ks[n] for n in names) / len(names), os.path.basename(f) + " (mean of " + ", ".join(names) + ")"
        except Exception:
            pass
    return None, None


LINE_TARGET_BYTES = 6000        # the printed line (target <= 6 KB)
LINE_HARD_CAP_BYTES = 12000     # never printed above this: fields are dropped, least important first


def _r(x, nd=4):
    """numbers of the compact line: floats to `nd` significant decimals of their magnitude, everything else as is"""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        ax = abs(x)
        if ax >= 1000:
            return round(x, 1)
        if ax >= 1:
            return round(x, 3)
        return float("%.4g" % x)
    return x


def _pick(d, *keys):
    """{key: rounded number} of the keys present in dict d (an {"error": ...} leg keeps a short error string)"""
    if not isinstance(d, dict):
        return None
    if "error" in d and len(d) == 1:
        return {"error": str(d["error"])[:120]}
    out = {}
    for k in keys:
        src, dst = (k if isinstance(k, tuple) else (k, k))
        v = d
        for part in src.split("."):
            v = v.get(part) if isinstance(v, dict) else None
        if v is not None and not isinstance(v, (dict, list, str)):
            out[dst] = _r(v)
        elif isinstance(v, str) and len(v) <= 48:
            out[dst] = v
    return out or None


def _prefill_leg(d):
    """a whole-model prompt-pass leg as numbers only: tok/s at the first length, every length, roofline fraction, validity"""
    if not isinstance(d, dict):
        return None
    if "value" not in d:
        return {"error": str(d.get("error", "no value"))[:120]}
    out = {"tok_s": _r(float(d["value"])), "ms": _r(float(d.get("ms", 0.0))), "frac": _r(float(d.get("roofline", {}).get("frac", 0.0)))}
    if d.get("by_prompt_length"):
        out["by_len"] = {k: round(float(v)) for k, v in d["by_prompt_length"].items()}
    if d.get("allocs_in_timed_region"):
        out["allocs"] = d["allocs_in_timed_region"]
    if d.get("invalid"):
        out["invalid"] = True
    return out


def _experts_leg(d):
    if not isinstance(d, dict):
        return None
    if "tok_s_experts_only" not in d:
        return {"error": str(d.get("error", "no value"))[:120]}
    return {"tok_s": _r(float(d["tok_s_experts_only"])), "ms": _r(float(d["ms"])), "layers": d.get("layers"), "frac": _r(float(d.get("roofline", {}).get("frac", 0.0))),
            "peak": d.get("roofline", {}).get("peak")}


def _side_leg(d):
    """a side configuration: numbers only"""
    if not isinstance(d, dict):
        return None
    if "error" in d and "decode_tok_s" not in d:
        return {"error": str(d["error"])[:120]}
    out = _pick(d, "decode_tok_s", "decode_fast_tok_s", ("step_frac_of_hbm_peak", "frac"), ("decode_fast_frac_of_hbm_peak", "fast_frac")) or {}
    for key, short in (("prefill", "prefill"), ("prefill_fast", "prefill_attn_fast"), ("prefill_fast_gemm", "prefill_fast_gemm")):
        if isinstance(d.get(key), dict) and "value" in d[key]:
            out[short] = round(float(d[key]["value"]))
        elif isinstance(d.get(key), dict) and "error" in d[key]:
            out[short] = "error"
    if isinstance(d.get("prefill_experts_only"), dict) and "tok_s_experts_only" in d["prefill_experts_only"]:
        out["experts_only"] = round(float(d["prefill_experts_only"]["tok_s_experts_only"]))
    for k in ("decode_tok_s", "decode_fast_tok_s"):
        if isinstance(d.get(k), dict):
            out[k] = "error"
    return out


def compact_line(res, detail_path=None):
    """The ONE JSON line the driver parses, from the full result dict: contract keys, the roofline and cpu_baseline objects, and NUMBERS ONLY for
    every side leg.  Prose notes, per-kind tables, ms_all, by-thread sweeps stay in the detail file (`detail`).  Kept under LINE_TARGET_BYTES by
    construction; if a future leg pushes it over LINE_HARD_CAP_BYTES the optional groups are dropped from the end of `order` until it fits."""
    cfg = res.get("config", {})
    line = {k: res.get(k) for k in ("metric", "value", "value_exact", "value_fast", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                      "scaling", "vs_baseline", "dtype", "data") if k in res}
    for k in ("value", "value_exact", "value_fast", "ms_per_step"):
        if isinstance(line.get(k), float):
            line[k] = _r(line[k])
    line["metric"] = str(res.get("metric", ""))[:160]
    line["dtype"] = str(res.get("dtype", ""))[:96]
    line["config"] = {k: (v if not isinstance(v, str) else v[:140]) for k, v in cfg.items()
                      if k in ("workload", "kv", "layers", "hip_graph", "decode_mode", "parallelism", "value_is")}
    rf = res.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = _pick(rf, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", ("algorithmic_bytes_per_launch", "algorithmic_bytes"),
                                 "us_per_launch", "launches_per_step", ("us_per_launch_raw_events", "us_per_launch_raw"), ("frac_raw_events", "frac_raw"),
                                 ("event_pair_overhead_us", "event_overhead_us"), ("step_algorithmic_bytes", "step_bytes"), ("step_frac_of_hbm_peak", "step_frac"),
                                 ("peak_measured_stream_read", "peak_measured"), "traffic_source",
                                 ("step_algorithmic_bytes_per_gpu", "step_bytes_per_gpu")) or {}
        if "traffic" not in line["roofline"]:
            line["roofline"]["traffic"] = None
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, "value", "unit", "cores", "kind", "host_threads", "ms_per_token") or {}
        if "error" in cb:
            c = {"error": str(cb["error"])[:120]}
        else:
            c["sample"] = "whole-token passes over one token's weight set (~1.9 GB), AVX2+OpenMP port of avx2.rs:1066; ~%gs" % res.get("_cpu_seconds", 5)
            v2 = cb.get("v2lite_q4k_cpu")
            if isinstance(v2, dict):
                c["v2lite_q4k_cpu"] = _pick(v2, "value", "cores", "dram_GBs") or {"error": str(v2.get("error"))[:80]}
        line["cpu_baseline"] = c
    opt = {}
    oth = res.get("decode_exact") or res.get("decode_fast")
    if isinstance(oth, dict):
        opt["decode_other_mode"] = _pick(oth, "tok_s", "ms_per_step", "steps")
    g = res.get("decode_generate")
    if isinstance(g, dict):
        gg = _pick(g, "tok_s", "tok_s_over_value", ("allocs_in_timed_decode_steps", "allocs")) or {}
        for k in ("exact", "fast", "fast_lookahead", "exact_lookahead"):
            if isinstance(g.get(k), dict) and "tok_s" in g[k]:
                gg[k] = _r(float(g[k]["tok_s"]))
        if "error" in g:
            gg["error"] = str(g["error"])[:120]
        opt["decode_generate"] = gg
    if isinstance(res.get("router_id_flip_rate"), dict):
        opt["router_id_flip_rate"] = _pick(res["router_id_flip_rate"], "tokens", "flips_ordered", "flips_set", "rate_ordered", "rate_set") or {"error": str(res["router_id_flip_rate"].get("error"))[:120]}
    for k in ("prefill", "prefill_fast", "prefill_fast_gemm"):
        if k in res:
            opt[k] = _prefill_leg(res[k])
    for k in ("prefill_experts_only", "prefill_experts_only_fast_gemm", "prefill_experts_only_q4k_gguf", "prefill_experts_only_q4k_gguf_fast_gemm"):
        if k in res:
            opt[k] = _experts_leg(res[k])
    if isinstance(res.get("decode_other_kv"), dict):
        opt["decode_other_kv"] = _pick(res["decode_other_kv"], "kv", "tok_s")
    lc = {}
    for k, short in (("decode_long_context", "8k_exact"), ("decode_long_context_fast", "8k_fast"), ("decode_long_context_32k", "32k_exact"), ("decode_long_context_32k_fast", "32k_fast")):
        if isinstance(res.get(k), dict):
            lc[short] = _r(float(res[k]["tok_s"])) if "tok_s" in res[k] else "error"
    if lc:
        opt["decode_long_context_tok_s"] = lc
    if isinstance(res.get("configs"), dict):
        opt["configs"] = {k: _side_leg(v) for k, v in res["configs"].items()}
    # N > 1 legs (main_multi) and the one-rank RCCL self-test
    for k in ("decode_ep_exact", "decode_ep_fast", "decode_ep_fast_graph"):
        if isinstance(res.get(k), dict):
            opt[k] = _pick(res[k], "tok_s", "ms_per_step", "hip_graph")
    for k in ("prefill_model_ep", "prefill_model_ep_attn_fast"):
        if isinstance(res.get(k), dict):
            opt[k] = _pick(res[k], ("value", "tok_s"), "ms", "tokens_total", ("roofline.frac", "frac"))
    if isinstance(res.get("prefill_experts_ep_alltoall"), dict):
        opt["prefill_experts_ep_alltoall"] = _pick(res["prefill_experts_ep_alltoall"], ("tok_s_experts_only", "tok_s"), "ms", "tokens_total", ("roofline.frac", "frac"))
    if isinstance(res.get("replicas"), dict):
        opt["replicas"] = _pick(res["replicas"], "tok_s_aggregate", "tok_s_per_replica")
    for k in ("rccl_ranks", "experts_per_gpu", "watchdog"):
        if k in res:
            opt[k] = res[k] if not isinstance(res[k], str) else res[k][:100]
    if isinstance(res.get("qwen3_235b_ep"), dict):
        q = res["qwen3_235b_ep"]
        o = _pick(q, "value", "frac_of_hbm_peak_per_gpu") or {}
        for k in ("decode_ep_exact", "decode_ep_fast", "decode_ep_fast_graph"):
            if isinstance(q.get(k), dict):
                o[k] = _r(float(q[k]["tok_s"])) if "tok_s" in q[k] else "error"
        if "error" in q:
            o["error"] = str(q["error"])[:120]
        opt["qwen3_235b_ep"] = o
    if isinstance(res.get("expert_parallel_selftest_one_rank_rccl"), dict):
        q = res["expert_parallel_selftest_one_rank_rccl"]
        o = _pick(q, "value", "rccl_ranks") or {}
        for k in ("decode_ep_exact", "decode_ep_fast", "decode_ep_fast_graph"):
            if isinstance(q.get(k), dict):
                o[k] = _r(float(q[k]["tok_s"])) if "tok_s" in q[k] else "error"
        opt["ep_selftest_one_rank_rccl"] = o
    if isinstance(res.get("expert_parallel"), dict):
        opt["expert_parallel"] = {"error": str(res["expert_parallel"].get("error"))[:160]}
    if detail_path:
        opt["detail"] = detail_path
    order = list(opt)                                    # least important last
    out = dict(line); out.update({k: v for k, v in opt.items() if v is not None})
    text = json.dumps(out, separators=(",", ":"))
    while len(text) > LINE_HARD_CAP_BYTES and order:
        out.pop(order.pop(), None)
        text = json.dumps(out, separators=(",", ":"))
    return out, text


def flush_native_stdout():
    """libc's own stdout buffer (RCCL prints its version banner there; on a pipe it stays buffered until exit) is written out NOW"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def silence_stdout():
    """nothing this process writes to fd 1 from here on reaches the driver's stdout: the compact line stays the LAST stdout line (a native library's buffered text, flushed
    at exit, or a message at communicator teardown would otherwise follow it); fd 1 becomes a copy of stderr, so such text is still seen"""
    try:
        sys.stdout.flush(); flush_native_stdout()
        os.dup2(2, 1)              # later writes to fd 1 (native teardown messages, a second emit after a watchdog emit) are mirrored to stderr, not lost
        sys.stdout = sys.stderr
    except Exception:
        pass


def emit_line(res, args):
    """write the full result dict to the detail file (and to stderr), print the compact line as the LAST stdout line"""
    detail_path = None
    full = json.dumps(res)
    for cand in (getattr(args, "detail_file", None), os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "/tmp/bench_detail.json"):
        if not cand:
            continue
        try:
            os.makedirs(os.path.dirname(cand) or ".", exist_ok=True)
            with open(cand, "w") as f:
                f.write(full + "\n")
            detail_path = cand
            break
        except OSError:
            continue
    res = dict(res); res["_cpu_seconds"] = getattr(args, "cpu_seconds", 5)
    rel = os.path.relpath(detail_path, ROOT) if detail_path and detail_path.startswith(ROOT) else detail_path
    out, text = compact_line(res, rel)
    assert len(text) <= LINE_HARD_CAP_BYTES, "bench line is %d bytes" % len(text)
    sys.stderr.write("bench.py: full detail (%d bytes) -> %s; compact line %d bytes\n" % (len(full), detail_path, len(text)))
    sys.stderr.flush()
    flush_native_stdout()      # buffered text of native libraries goes out BEFORE the line
    print(text, flush=True)
    silence_stdout()           # ... and nothing follows it
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="qcn-q4", choices=sorted(WORKLOAD), help="which BASELINE configuration the headline line measures")
    ap.add_argument("--kv", default="fp8", choices=["fp8", "fp16"], help="KV cache element type of the headline run (BASELINE config 3 names FP8 KV)")
    ap.add_argument("--layers", type=int, default=0, help="0 = the model's layer count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=5.0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--decode-mode", default="fast", choices=["fast", "exact"],
                    help="numerics of the headline decode steps: fast = KR_DECODE_FAST (the reference's products, tree reductions; logits within the tolerance of "
                         "tests/test_decode_fast_gpu.py, router ids identical for identical logits), exact = bit-identical to the reference's CPU decode; the other mode is reported as a side leg")
    ap.add_argument("--prefill-chunk", type=int, default=0, help="tokens per chunk of the prompt pass (0 = library default)")
    ap.add_argument("--prefill-depth", type=int, default=0, help="chunks of the prompt pass in flight (0 = library default)")
    ap.add_argument("--prefill-reps", type=int, default=2, help="timed repetitions (>= 2) of the whole-model prompt pass per prompt length, after one un-timed pass of the same prompt; value = median")
    ap.add_argument("--no-long-context", action="store_true", help="skip the long-cache decode side measurement of the N = 1 line")
    ap.add_argument("--no-ep", action="store_true", help="skip the expert-parallel prompt-pass leg of the N > 1 lines")
    ap.add_argument("--ep-timeout", type=int, default=420, help="seconds the multi-GPU expert-parallel side legs may take before every rank gives up (rank 0 still prints the line)")
    ap.add_argument("--ep-selftest", action="store_true", help="run the expert-parallel leg at N = 1 too (no peer traffic: checks the row path)")
    ap.add_argument("--prefill-tokens", default="8192,20434,35139,49863",
                    help="prompt lengths of the prompt-pass side measurement (benchmark.py:434-505: 20 434 / 35 139 / 49 863 tokens; 0 = skip)")
    ap.add_argument("--flip-tokens", type=int, default=2000, help="tokens of the exact-vs-KR_DECODE_FAST router-id comparison on the N = 1 line (0 = skip)")
    ap.add_argument("--detail-file", default="", help="where the full (un-abridged) result dict goes; default gpurun_out/bench_detail.json")
    ap.add_argument("--side-configs", default="v2lite-q4,v2lite-q4k-gguf,qcn-q8,qcn-q4k-gguf,qwen3-235b-q4", help="other single-GPU BASELINE configurations measured as side legs of the N = 1 line ('' = none)")
    return ap.parse_args()


def is_gqa(l):
    return (l + 1) % QCN["full_attn_interval"] == 0


def algorithmic_bytes(L, bw=B4, split_out=False):
    """Bytes a QCN decode token must touch, each weight/state byte once, per kernel kind.  split_out: the out / o projections as
    their own kind (KR_DECODE_FAST runs them on their own kernel instantiation)."""
    q = QCN; H, I, E, k = q["hidden"], q["inter"], q["experts"], q["topk"]
    n_la = sum(1 for l in range(L) if not is_gqa(l)); n_gqa = L - n_la
    group_dim = 2 * q["dk"] + 2 * q["dv"] * (q["nv"] // q["nk"])
    la_w = (q["nk"] * group_dim + q["nk"] * 2 * (q["nv"] // q["nk"])) * H + H * (q["nv"] * q["dv"])
    gqa_w = (q["nh"] * q["hd"] * 2 + 2 * q["nkv"] * q["hd"]) * H + H * (q["nh"] * q["hd"])
    b = {
        "proj_matvec": (n_la * la_w + n_gqa * gqa_w) * bw,
        "out_proj_matvec": 0.0,
        "moe_w13": L * (k + 1) * H * 2 * I * bw,          # 10 routed + shared expert
        "moe_w2": L * (k + 1) * I * H * bw,
        "lm_head": q["vocab"] * H * bw,
        "route_logits": L * E * H * 2,                      # gate stored as bf16 in HBM
        "la_recurrent": n_la * 2 * q["nv"] * q["dk"] * q["dv"] * 4,   # state read + write
        "shared_gate": L * H * bw,
    }
    if split_out:
        b["out_proj_matvec"] = (n_la * H * q["nv"] * q["dv"] + n_gqa * H * q["nh"] * q["hd"]) * bw
        b["proj_matvec"] -= b["out_proj_matvec"]
    b["total"] = sum(b.values())
    return b


def algorithmic_bytes_v2lite(L, bw=B4):
    """routed + shared experts of the 26 MoE layers, MLA projections, absorbed w_kc / w_vc (f32 in HBM like the
    reference's decode store, decode.rs:2157-2160), the dense layer-0 MLP, lm_head, router gate (bf16)."""
    v = V2L; H, I = v["hidden"], v["inter"]
    n_moe = max(L - 1, 0)
    mla_w = ((v["klr"] + v["rd"]) + v["nh"] * (v["nd"] + v["rd"])) * H + H * v["nh"] * v["vhd"]
    b = {"moe_w13": n_moe * (v["topk"] + v["n_shared"]) * H * 2 * I * bw, "moe_w2": n_moe * (v["topk"] + v["n_shared"]) * I * H * bw,
         "proj_matvec": L * mla_w * bw + 3 * H * v["dense_inter"] * bw, "gqa": L * 2 * v["nh"] * v["nd"] * v["klr"] * 4,
         "lm_head": v["vocab"] * H * bw, "route_logits": n_moe * v["experts"] * H * 2}
    b["total"] = sum(b.values())
    return b


def algorithmic_bytes_q235(L, bw=B4):
    """SURVEY 8d, Qwen3-235B: routed experts 94 * 8 * 3 * 4096 * 1536 * 0.5156 = 7.32 GB, GQA projections, lm_head, router gate (bf16)."""
    q = Q235; H, I, E, k = q["hidden"], q["inter"], q["experts"], q["topk"]
    attn_w = (q["nh"] * q["hd"] + 2 * q["nkv"] * q["hd"]) * H + H * q["nh"] * q["hd"]
    b = {"moe_w13": L * k * H * 2 * I * bw, "moe_w2": L * k * I * H * bw, "proj_matvec": L * attn_w * bw, "lm_head": q["vocab"] * H * bw, "route_logits": L * E * H * 2}
    b["total"] = sum(b.values())
    return b


def q235_gemm_macs_per_token(L):
    q = Q235; H, I, k = q["hidden"], q["inter"], q["topk"]
    return L * ((q["nh"] * q["hd"] + 2 * q["nkv"] * q["hd"]) * H + H * q["nh"] * q["hd"] + k * 3 * H * I)


def ep_local_experts(E, world, rank):
    """the reference's contiguous expert slices (gpu_prefill.py:353-359): floor(E / R) per rank, the last rank takes the remainder"""
    per = E // world
    return E - per * (world - 1) if rank == world - 1 else per


def build_q235(rank, local_rank, L, rope_len=0, bits=4, kv_fp8=False, ep_world=1):
    """Qwen3-235B-A22B-shaped decode graph (BASELINE config 4): 94 x [GQA (64 q / 4 kv heads, head_dim 128, per-head QK-norm) + 128-expert top-8 MoE, no shared expert].
    ep_world > 1: THIS RANK'S SHARD of an expert-parallel model -- E / ep_world routed experts per layer (its slice), everything else replicated with
    rank-independent seeds (the ranks route independently and must see the same router, projections, norms and state)."""
    import numpy as np
    from krasis_amd import CpuDecodeStore, KrasisEngine, ModelConfig
    q = Q235; H, I, E, k, V = q["hidden"], q["inter"], q["experts"], q["topk"], q["vocab"]
    nh, nkv, hd = q["nh"], q["nkv"], q["hd"]
    e_rank = rank; E_loc = E
    if ep_world > 1:
        E_loc = ep_local_experts(E, ep_world, rank); rank = 0
    eng = KrasisEngine(device=local_rank); eng.configure(ModelConfig(H, I, E_loc, k, L, 0, 1.0))
    eng.fill_synthetic(bits, seed=0x235 + e_rank); eng.set_routing_config("softmax", True, k, E, H)
    st = CpuDecodeStore(128, True, False); st.set_moe_store(eng)
    rng = np.random.default_rng(235 + rank); keep = []; seed = [700 + rank * 100000]

    def W(r, c):
        seed[0] += 1; return st.store_weight_synthetic(r, c, bits, seed[0])

    def N(n):
        w = (rng.random(n, dtype=np.float32) * 0.2 + 0.9).astype(np.float32); keep.append(w); return st.store_norm_weight(w.ctypes.data, n)

    fin, lm = N(H), W(V, H)
    st.configure_decode(H, L, q["eps"], fin, lm, V, k, 1, True, 1.0, 0, synth_seed=235 + rank)
    for l in range(L):
        n_in, n_post = N(H), N(H)
        qw, kw, vw, ow = W(nh * hd, H), W(nkv * hd, H), W(nkv * hd, H), W(H, nh * hd)
        qn = (rng.random(hd, dtype=np.float32) + 0.5).astype(np.float32); kn = (rng.random(hd, dtype=np.float32) + 0.5).astype(np.float32); keep += [qn, kn]
        st.add_decode_gqa_layer(n_in, n_post, qw, kw, vw, ow, qn.ctypes.data, hd, kn.ctypes.data, hd, False, nh, nkv, hd, 1.0 / hd ** 0.5)
        eng.set_route_weight_synthetic(l, 0x12345678ABCDEF01 + rank, 0.02, True)
        st.set_decode_layer_moe(l, l, l, None, None, None)
    half = hd // 2; rope_len = max(rope_len, q["kv_max_seq"])
    pos = np.arange(rope_len, dtype=np.float32)[:, None]
    freq = (1.0 / (1000000.0 ** (2.0 * np.arange(half, dtype=np.float32) / hd))).astype(np.float32)[None, :]
    cos, sin = np.cos(pos * freq).astype(np.float32), np.sin(pos * freq).astype(np.float32); keep += [cos, sin]
    st.set_decode_rope(cos.ctypes.data, sin.ctypes.data, half, rope_len)
    st.finalize_decode()
    st.set_kv_dtype(kv_fp8)
    st.fill_state_synthetic(q["kv_max_seq"], seed=235 + rank)
    return eng, st, keep


def build_qcn(rank, local_rank, L, rope_len=0, bits=4, kv_fp8=False, gguf=False, ep_world=1):
    """ep_world > 1: this rank's shard of an expert-parallel model (see build_q235)"""
    import numpy as np
    from krasis_amd import CpuDecodeStore, KrasisEngine, ModelConfig
    q = QCN; H, I, E, k, V = q["hidden"], q["inter"], q["experts"], q["topk"], q["vocab"]
    e_rank = rank; E_loc = E
    if ep_world > 1:
        E_loc = ep_local_experts(E, ep_world, rank); rank = 0
    eng = KrasisEngine(device=local_rank)
    eng.configure(ModelConfig(H, I, E_loc, k, L, 0, 1.0))
    if gguf:      # routed experts as native Q4_K super-blocks (gguf_native=True of the reference): prompt pass and decode step consume them natively
        eng.fill_synthetic_gguf(12, 12, seed=0x12345678ABCDEF01 + e_rank)
    else:
        eng.fill_synthetic(bits, seed=0x12345678ABCDEF01 + e_rank)
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
                                   nk, nv, dk, dv, nv // nk, 4, 1.0 / dk ** 0.5)
        # router gate: the reference's xorshift64 stream, uniform +-0.02 (decode.rs:5181, :4356-4376), truncated to bf16 like a real
        # checkpoint's gate tensor -> stored as bf16 in HBM
        eng.set_route_weight_synthetic(l, 0x12345678ABCDEF01 + rank, 0.02, True)
        sgu, sd, sg = W(2 * q["shared_inter"], H), W(H, q["shared_inter"]), W(1, H)
        st.set_decode_layer_moe(l, l, l, sgu, sd, sg)
    half = hd // 2                                                    # decode.rs:5379: full rotary in the synthetic bench
    rope_len = max(rope_len, q["kv_max_seq"])                        # long enough for the prompt-pass measurement
    pos = np.arange(rope_len, dtype=np.float32)[:, None]
    freq = (1.0 / (10000.0 ** (2.0 * np.arange(half, dtype=np.float32) / hd))).astype(np.float32)[None, :]
    cos, sin = np.cos(pos * freq).astype(np.float32), np.sin(pos * freq).astype(np.float32); keep += [cos, sin]
    st.set_decode_rope(cos.ctypes.data, sin.ctypes.data, half, rope_len)
    st.finalize_decode()
    st.set_kv_dtype(kv_fp8)
    st.fill_state_synthetic(q["kv_max_seq"], seed=4242 + rank)
    return eng, st, keep


def build_v2lite(rank, local_rank, L, rope_len=0, bits=4, kv_fp8=False, ep_world=1, gguf=False):
    """DeepSeek-V2-Lite-shaped decode graph (BASELINE config 2): MLA attention (direct q projection), layer 0 dense, 64 experts top-6 + 2 shared.
    ep_world > 1: this rank's shard of an expert-parallel model (see build_q235)."""
    import numpy as np
    from krasis_amd import CpuDecodeStore, KrasisEngine, ModelConfig
    v = V2L; H, I, E, k, V = v["hidden"], v["inter"], v["experts"], v["topk"], v["vocab"]
    nh, klr, nd, rd, vhd = v["nh"], v["klr"], v["nd"], v["rd"], v["vhd"]
    e_rank = rank; E_loc = E
    if ep_world > 1:
        E_loc = ep_local_experts(E, ep_world, rank); rank = 0
    eng = KrasisEngine(device=local_rank); eng.configure(ModelConfig(H, I, E_loc, k, L, v["n_shared"], 1.0))
    if gguf:      # routed experts as native GGUF blocks, the int4cpu build of BASELINE config 1: Q4_K gate / up, Q8_0 down (1408 is not a multiple of 256)
        eng.fill_synthetic_gguf(12, 8, seed=11 + e_rank)
    else:
        eng.fill_synthetic(bits, seed=11 + e_rank)
    eng.set_routing_config("softmax", False, k, E, H)
    st = CpuDecodeStore(128, True, False); st.set_moe_store(eng)
    rng = np.random.default_rng(3 + rank); keep = []; seed = [50 + rank * 100000]

    def W(r, c):
        seed[0] += 1; return st.store_weight_synthetic(r, c, bits, seed[0])

    def N(n):
        w = (rng.random(n, dtype=np.float32) * 0.2 + 0.9).astype(np.float32); keep.append(w); return st.store_norm_weight(w.ctypes.data, n)

    fin, lm = N(H), W(V, H)
    st.configure_decode(H, L, v["eps"], fin, lm, V, k, 1, False, 1.0, 0, synth_seed=5 + rank)
    half = rd // 2; rope_len = max(rope_len, v["kv_max_seq"])
    ang = np.arange(rope_len)[:, None] * (1.0 / 10000.0 ** (2 * np.arange(half) / rd))[None, :]
    cos, sin = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32); keep += [cos, sin]
    DI = (v["dense_inter"] + 127) // 128 * 128                   # non-MoE weights pad cols to a multiple of 128 (decode_setup.py:586-597)
    for l in range(L):
        n_in, n_post = N(H), N(H)
        kv_a, o, q = W(klr + rd, H), W(H, nh * vhd), W(nh * (nd + rd), H)
        w_kc = ((rng.standard_normal((nh, nd, klr)) * 0.06).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        w_vc = ((rng.standard_normal((nh, vhd, klr)) * 0.06).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        kvn = (rng.random(klr) + 0.5).astype(np.float32); keep += [w_kc, w_vc, kvn]
        st.add_decode_mla_layer(n_in, n_post, kv_a, o, q, None, None, w_kc.ctypes.data, w_kc.size, w_vc.ctypes.data, w_vc.size, kvn.ctypes.data, klr, 0, 0,
                                cos.ctypes.data, sin.ctypes.data, half, rope_len, nh, klr, nd, rd, vhd, float(1.0 / np.sqrt(nd + rd)))
        if l == 0:
            st.set_decode_layer_dense(l, W(DI, H), W(DI, H), W(H, DI))
        else:
            eng.set_route_weight_synthetic(l, 0x12345678ABCDEF01 + rank, 0.02, True)
            st.set_decode_layer_moe(l, l, l, W(2 * v["n_shared"] * I, H), W(H, v["n_shared"] * I), None)
    st.finalize_decode()
    st.set_kv_dtype(kv_fp8)
    st.fill_state_synthetic(v["kv_max_seq"], seed=9 + rank)
    return eng, st, keep


def prefill_experts(eng, dims, L, M, torch, gemm_fast=False, gemm_mode=None):
    """Side measurement (NOT the headline value): the prefill expert path alone -- token sort + int8-MFMA grouped GEMM + combine of all
    L MoE layers for one chunk of M tokens with uniform random routing (k distinct experts per token).  Roofline per SURVEY 8(d):
    achieved = 2 * T * k * 3 * H * I / t (useful MACs x 2); `int8_TOPS_issued` counts both INT16-digit passes the exact arithmetic issues."""
    from krasis_amd import GpuPrefillManager, _lib
    H, I, E, k = dims["hidden"], dims["inter"], dims["experts"], dims["topk"]
    g = torch.Generator(device="cuda").manual_seed(7)
    x = ((torch.rand((M, H), device="cuda", generator=g) - 0.5)).to(torch.bfloat16)
    ids = torch.rand((M, E), device="cuda", generator=g).topk(k, dim=1).indices.to(torch.int32)
    w = torch.softmax(torch.randn((M, k), device="cuda", generator=g), dim=1)
    mgr = GpuPrefillManager(eng, k)
    # gemm_mode 3: tolerance form on the register-staged kernels only (A/B against the default LDS-ring form, same bits)
    _lib.check(eng._lib.kr_moe_set_gemm_mode(eng._h, gemm_mode if gemm_mode is not None else (1 if gemm_fast else 0)))
    try:
        for l in range(L):      # every layer builds its derived tables on first use (per-weight nibble sums; the tolerance copy of GGUF layers) -- outside the timed region
            mgr.forward(l, x, ids, w, routed_only=True)
        torch.cuda.synchronize()
        a0 = alloc_count()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for l in range(L):
            mgr.forward(l, x, ids, w, routed_only=True)
        ev1.record(); torch.cuda.synchronize()
        allocs = alloc_count() - a0
    finally:
        if gemm_mode is not None:
            _lib.check(eng._lib.kr_moe_set_gemm_mode(eng._h, 1))       # the ring kernel back to its default dispatch
        _lib.check(eng._lib.kr_moe_set_gemm_mode(eng._h, 0))
    ms = ev0.elapsed_time(ev1)
    if allocs:
        raise RuntimeError("prefill_experts: the library allocated device memory %d time(s) inside the timed region" % allocs)
    macs = M * k * 3 * H * I * L                       # routed experts only
    useful = 2.0 * macs / (ms * 1e-3) / 1e12
    if gemm_fast:
        return {"tokens": M, "layers": L, "ms": ms, "tok_s_experts_only": M / (ms * 1e-3),
                "roofline": {"bound": "mfma", "achieved": useful, "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s (f16 MFMA; 2 x useful MACs / s)", "frac": useful / F16_PEAK_TFLOPS},
                "note": "experts only, tolerance form (kr_moe_set_gemm_mode 1): f16 rows x INT4 weights de-quantized in registers, one MFMA pass per MAC, f32 accumulation over "
                        "the whole k range; outputs within 2-4e-4 relative RMS of the exact kernel (tests/test_gemm_fast_gpu.py)"}
    return {"tokens": M, "layers": L, "ms": ms, "tok_s_experts_only": M / (ms * 1e-3),
            "roofline": {"bound": "mfma", "achieved": useful, "peak": I8_PEAK_TOPS, "unit": "TOP/s (int8; 2 x useful MACs / s, SURVEY 8d)", "frac": useful / I8_PEAK_TOPS,
                         "peak_measured_ubench": 3944.0},
            "int8_TOPS_issued": 2.0 * useful, "frac_of_i8_peak_issued": 2.0 * useful / I8_PEAK_TOPS,
            "note": "experts only (sort + 2 grouped GEMMs + act + combine): the MFMA-bound part of the prompt pass in isolation; every MAC is issued "
                    "twice (high / low INT16 activation digit) to reproduce the reference's integer arithmetic exactly"}


def prefill_experts_gguf(local_rank, torch, gate_up_type=12, down_type=12, L=8, gemm_fast=False):
    """Side measurement: the expert path of the prompt pass on NATIVE GGUF blocks (QCN shape, Q4_K gate / up / down): raw super-blocks staged
    in LDS feeding the int8 MFMA (kr_gguf_prefill.hip), L layers of 512 synthetic experts."""
    from krasis_amd import KrasisEngine, ModelConfig
    q = QCN
    eng = KrasisEngine(device=local_rank); eng.configure(ModelConfig(q["hidden"], q["inter"], q["experts"], q["topk"], L, 0, 1.0))
    eng.fill_synthetic_gguf(gate_up_type, down_type, seed=77)
    r = prefill_experts(eng, q, L, 8192, torch, gemm_fast=gemm_fast)
    if gemm_fast:
        r["note"] = ("tolerance form of the native Q4_K experts: super-blocks re-tiled once (nibbles + per-sub-block f16 scale / offset tables), f16 MFMA with the scale "
                     "folded into the de-quantization, offsets as K / 32 extra k-columns, libm SiLU; outputs within 1.5e-3 relative RMS of the exact path (tests/test_gguf_gpu.py)")
        r["weights"] = "native GGUF Q4_K blocks (0.5625 B / weight) + the tolerance GEMM's re-tiled copy (0.625 B / weight, built on first use)"
        del eng
        gc.collect(); torch.cuda.empty_cache()
        return r
    r["weights"] = "native GGUF blocks, gate/up ggml type %d, down type %d (Q4_K = 12: 0.5625 B / weight, Q8_0 = 8)" % (gate_up_type, down_type)
    r["note"] = ("sort + 3 grouped GEMMs (gate, up, down) + libm-SiLU act + combine; raw Q4_K super-blocks in LDS, one int8 MFMA per 32-wide sub-block and "
                 "activation digit, per-sub-block scale / min epilogue (one f32 chain per output: ~1e-6 relative to the streaming kernels)")
    del eng
    gc.collect(); torch.cuda.empty_cache()
    return r


def prefill_ep(eng, dims, L, M, world, rank, torch, dist, experts_local=None, ep=None):
    """Expert-parallel prompt-pass experts over RCCL inside libkrasis_hip.so (kr_ep_init / kr_moe_prefill_ep): every rank owns
    E/N experts and M tokens; each (token, slot) row travels once to the rank that owns its expert (ncclSend/ncclRecv groups over the xGMI
    mesh), runs through the int8-MFMA expert GEMMs there (the w2 GEMM writes the row in bf16 at its return slot), comes back, and the source
    rank combines its k rows in routing order.  All L MoE layers, uniform random routing, weak scaling (M tokens per rank)."""
    from krasis_amd.ep import ExpertParallel
    H, I, E, k = dims["hidden"], dims["inter"], dims["experts"], dims["topk"]
    g = torch.Generator(device="cuda").manual_seed(7 + rank)
    x = ((torch.rand((M, H), device="cuda", generator=g) - 0.5)).to(torch.bfloat16)
    ids = torch.rand((M, E), device="cuda", generator=g).topk(k, dim=1).indices.to(torch.int32)
    w = torch.softmax(torch.randn((M, k), device="cuda", generator=g), dim=1)
    if ep is None:
        ep = ExpertParallel(eng, E, world, rank, dist if world > 1 else None)
    out = torch.empty((M, H), dtype=torch.bfloat16, device="cuda")
    for l in range(min(L, 2)):
        ep.forward(l, x, ids, w, out)
    ep.synchronize(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for l in range(L):
        ep.forward(l, x, ids, w, out)
    ep.synchronize(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    macs = world * M * k * 3 * H * I * L
    off = (world - 1) / world                          # share of the rows that leave the GPU under uniform routing
    useful = 2.0 * macs / dt / 1e12
    return {"tokens_per_gpu": M, "tokens_total": world * M, "layers": L, "ms": dt * 1e3, "tok_s_experts_only": world * M / dt, "scaling": "weak", "rccl_ranks": ep.comm_ranks(),
            "experts_per_gpu": E // world, "roofline": {"bound": "mfma", "achieved": useful, "peak": I8_PEAK_TOPS * world, "unit": "TOP/s (int8, useful)", "frac": useful / (I8_PEAK_TOPS * world)},
            "exchange_GB_per_gpu_per_layer": {"dispatch_bf16": M * k * H * 2 * off / 1e9, "combine_bf16": M * k * H * 2 * off / 1e9},
            "note": "owner sort + RCCL send/recv dispatch (bf16 rows) + expert GEMMs (w2 scatters bf16 rows to their return slots) + RCCL send/recv return + combine in "
                    "routing order, per layer, inside libkrasis_hip.so (kr_moe_prefill_ep); compare with prefill_experts_only of the N = 1 line"}


def alloc_count():
    """device allocations libkrasis_hip.so has made so far (kr_alloc_count_total): read on both sides of a timed region"""
    from krasis_amd import _lib
    return int(_lib.load_library().kr_alloc_count_total())


def prefill_model(st, dims, gemm_macs_per_token, L, P, reps, torch):
    """Whole-model prompt pass (kr_decode_prefill): P synthetic tokens through all L layers.  tok/s = P / time.  Roofline per SURVEY 8(d):
    useful GEMM MACs x 2 / t against the int8 MFMA peak.  Protocol: ONE un-timed pass of the SAME prompt in the SAME
    mode first (every arena, per-weight table and context-sized buffer the timed pass needs exists afterwards), then `reps` >= 2 timed passes, each
    bracketed by a device synchronize; `value` is the MEDIAN pass, the best pass rides along; the library's allocation counter must not move inside
    the timed region (`allocs_in_timed_region`, 0 or the measurement is flagged invalid)."""
    import numpy as np
    reps = max(2, reps)
    st.fill_state_synthetic(P + 64, 7)                      # KV caches / states sized for the prompt
    toks = [int(x) for x in np.random.default_rng(5).integers(0, dims["vocab"], P)]
    st.prefill(toks, 0)                                     # warm-up: same length, same mode
    torch.cuda.synchronize()
    a0 = alloc_count()
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        st.prefill(toks, 0)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    allocs = alloc_count() - a0
    ts = sorted(times)
    dt = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
    useful = 2.0 * P * gemm_macs_per_token / dt / 1e12
    r = {"value": P / dt, "unit": "tok/s", "tokens": P, "ms": dt * 1e3, "reps": reps, "ms_best": ts[0] * 1e3, "tok_s_best": P / ts[0], "ms_all": [round(x * 1e3, 2) for x in times],
         "allocs_in_timed_region": allocs, "layers": L, "target_tok_s": 3300,
         "roofline": {"bound": "mfma", "achieved": useful, "peak": I8_PEAK_TOPS, "unit": "TOP/s (int8; 2 x useful GEMM MACs / s, SURVEY 8d)",
                      "frac": useful / I8_PEAK_TOPS, "int8_TOPS_issued": 2.0 * useful}}
    if allocs:
        r["invalid"] = "the library allocated device memory %d time(s) inside the timed region" % allocs
    return r


def qcn_gemm_macs_per_token(L):
    q = QCN; H, I, k = q["hidden"], q["inter"], q["topk"]
    n_la = sum(1 for l in range(L) if not is_gqa(l)); n_gqa = L - n_la
    hr = q["nv"] // q["nk"]; gd = 2 * q["dk"] + 2 * q["dv"] * hr
    w_la = q["nk"] * gd * H + q["nk"] * 2 * hr * H + H * q["nv"] * q["dv"]
    w_gqa = (q["nh"] * q["hd"] * 2 + 2 * q["nkv"] * q["hd"]) * H + H * q["nh"] * q["hd"]
    w_moe = (k + 1) * 3 * H * I + H                          # routed + shared expert (+ its gate row)
    return n_la * w_la + n_gqa * w_gqa + L * w_moe           # GEMM MACs per prompt token (lm_head runs for the last token only)


def v2l_gemm_macs_per_token(L):
    v = V2L; H, I = v["hidden"], v["inter"]
    mla = ((v["klr"] + v["rd"]) + v["nh"] * (v["nd"] + v["rd"])) * H + H * v["nh"] * v["vhd"]
    return L * mla + 3 * H * v["dense_inter"] + max(L - 1, 0) * (v["topk"] + v["n_shared"]) * 3 * H * I


def time_decode(st, steps, warmup, kvm, torch, dist, world):
    def step(i):
        st.decode_step(0, (10 + i) % (kvm - 1))
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    return dt


def decode_generate(st, kvm, n_tokens=64, runs=3, lookahead=False):
    """The reference's decode protocol (benchmark.py:434-505 -> generate_batch, decode.rs:3525-3600): `runs` generations of `n_tokens` tokens, every
    sampled (greedy) token read back and fed to the next step; timed by the library's own clock around the loop (last_decode_elapsed_s, the
    reference's Rust Instant).  lookahead = kr_decode_set_option("generate_lookahead"): the token feeds the next step on the device and the host reads
    it while that step runs (same tokens; stated difference in the state after an early stop id -- none is used here)."""
    st.set_option("generate_lookahead", 1 if lookahead else 0)
    try:
        st.generate_batch(0, 10, 8)                          # warm-up (graph capture, pinned ring)
        per = []
        for r in range(runs):
            toks = st.generate_batch(0, 10, n_tokens)
            per.append(len(toks) / st.last_decode_elapsed_s)
    finally:
        st.set_option("generate_lookahead", 0)
    return {"tok_s": sum(per) / len(per), "runs": [round(x, 1) for x in per], "tokens_per_run": n_tokens,
            "loop": "look-ahead (device-side token feedback, host reads token i during step i + 1)" if lookahead else "reference loop (host reads every token before queuing the next step)"}


def router_flip_rate(st, dims, kvm, n_tokens, seed=12345):
    """Cross-mode companion of "router ids identical for identical logits" (VERDICT r4 next #3c): the SAME token stream decoded in the exact mode and in
    KR_DECODE_FAST, each from the same initial state and each carrying its own state forward; per token, the ids the LAST MoE layer's router selected
    (kr_decode_read_buffer) are compared -- as ordered lists and as sets.  The inputs of that router differ in the last bits between the modes (47 layers of
    another summation order before it), so a flip is a near-tie between the k-th and the (k + 1)-th expert; on synthetic (un-trained) router weights the
    margins are what random gates give -- a lower bound for a trained router."""
    import numpy as np
    E, k = dims["experts"], dims["topk"]
    toks = [int(x) for x in np.random.default_rng(seed).integers(0, dims["vocab"], n_tokens)]
    got = {}
    for mode in ("exact", "fast"):
        st.set_attention_mode(False, decode_fast=mode == "fast")
        st.fill_state_synthetic(kvm, seed=4242)
        ids = np.empty((n_tokens, k), np.int32); mg = np.empty(n_tokens, np.float32)
        for i, t in enumerate(toks):
            st.decode_step(t, 10 + (i % (kvm - 12)))
            lg, ii, _w = st.read_router(E, k)
            ids[i] = ii
            srt = np.sort(lg)[::-1]; mg[i] = srt[k - 1] - srt[k]
        got[mode] = (ids, mg)
    a, b = got["exact"][0], got["fast"][0]
    ordered = int((a != b).any(axis=1).sum()); as_set = int(sum(1 for x, y in zip(a, b) if set(x.tolist()) != set(y.tolist())))
    return {"tokens": n_tokens, "layer": "last MoE layer", "flips_ordered": ordered, "flips_set": as_set, "rate_ordered": ordered / n_tokens, "rate_set": as_set / n_tokens,
            "min_margin_kth_vs_next_logit_exact": float(got["exact"][1].min()), "experts": E, "topk": k}


def profile_kinds(st, kvm, P=5, step_ms=None):
    from krasis_amd import _lib
    ms = (C.c_double * 16)(); cnt = (C.c_long * 16)()
    tot_ms = [0.0] * 16; tot_n = [0] * 16
    _lib.check(st._lib.kr_decode_profile_step(st._h, 0, 9, ms, cnt, 16))       # un-timed: the first un-graphed step pays one-time launch costs
    for i in range(P):
        _lib.check(st._lib.kr_decode_profile_step(st._h, 0, (10 + i) % (kvm - 1), ms, cnt, 16))
        for j in range(NK):
            tot_ms[j] += ms[j]; tot_n[j] += cnt[j]
    # an event pair adds marker-packet time to every launch it brackets.  The graph-replayed step is the sum of its kernels' durations (the gaps inside a
    # replayed graph are ~0: sum of rocprof durations = 98 % of the step, profiles/r03_decode_fast_kernel_stats.txt), so the constant per-launch
    # excess is (sum of event times - graph step time) / launches; it is taken off every launch so that a launch's figure is its duration as a kernel
    # trace reports it (profiles/r03_bench_cmd_kernel_stats.txt is the rocprofv3 trace of this same command: fw13 9.19 us, fdm<4,1,8> 7.69 us)
    n_launch = sum(tot_n[j] for j in range(16)) / P
    ovh_ms = max(0.0, (sum(tot_ms) / P - step_ms) / n_launch) if (step_ms and n_launch) else 0.0
    net_ms = [max(tot_ms[j] - ovh_ms * tot_n[j], 0.0) for j in range(16)]
    per_kind_us = {KINDS[j]: (net_ms[j] / P) * 1e3 for j in range(NK)}            # us per step
    per_launch_us = {KINDS[j]: (net_ms[j] / max(tot_n[j], 1)) * 1e3 for j in range(NK)}
    n_per_step = {KINDS[j]: tot_n[j] / P for j in range(NK)}
    profile_kinds.event_overhead_us = ovh_ms * 1e3
    profile_kinds.raw_us_per_step = {KINDS[j]: (tot_ms[j] / P) * 1e3 for j in range(NK)}      # the event times as measured (no excess taken off)
    return per_kind_us, per_launch_us, n_per_step


def long_context(st, kv_long, torch, kv_name, fast=False):
    st.set_attention_mode(fast, decode_fast=fast)
    st.fill_state_synthetic(kv_long, 7)
    for i in range(3):
        st.decode_step(0, kv_long - 6 + i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(20):
        st.decode_step(0, kv_long - 2)
    torch.cuda.synchronize()
    d1 = (time.perf_counter() - t1) / 20
    st.set_attention_mode(False)
    return {"kv_max_seq": kv_long, "position": kv_long - 2, "kv": kv_name, "ms_per_step": d1 * 1e3, "tok_s": 1.0 / d1,
            "attention": "fast: split-KV flash-decode on the f16 MFMA + log-sum-exp merge (KR_ATTN_FAST) and the tolerance-mode step kernels (KR_DECODE_FAST)" if fast else
                         "exact: the reference's sequential softmax sum / p.v order (bit-identical to the CPU decode)"}


def side_config(name, rank, local_rank, args, torch):
    """Another single-GPU BASELINE configuration as a side leg: decode tok/s (same protocol, fewer steps) + prompt pass at 8192 tokens."""
    qcn = name.startswith("qcn"); q235 = name.startswith("qwen3-235b")
    bits = 8 if name.endswith("q8") else 4
    dims = QCN if qcn else (Q235 if q235 else V2L)
    L = dims["layers"]
    build = build_qcn if qcn else (build_q235 if q235 else build_v2lite)
    if name.endswith("-gguf"):      # routed experts as native GGUF blocks: decode step (kr_gguf.hip block kernels inside the graph) + prompt pass
        eng, st, keep = (build_qcn if qcn else build_v2lite)(rank, local_rank, L, 8192 + 64, 4, kv_fp8=True, gguf=True)
        st.set_use_graph(not args.no_graph)
        H, I, k_ = dims["hidden"], dims["inter"], dims["topk"]
        bq = 0.5625; bd = 0.5625 if I % 256 == 0 else 1.0625          # Q4_K 144 / 256, Q8_0 34 / 32 bytes per weight
        res = {"workload": WORKLOAD[name], "kv": "FP8-E4M3",
               "weights": "routed experts: native GGUF blocks (gate / up Q4_K 0.5625 B / weight, down %s); projections, shared expert, lm_head: INT4-g128" % ("Q4_K" if I % 256 == 0 else "Q8_0 1.0625 B / weight")}
        try:
            steps = min(args.steps, 50)
            ab = dict(algorithmic_bytes(L, B4) if qcn else algorithmic_bytes_v2lite(L, B4))
            n_moe = L if qcn else max(L - 1, 0); n_sh = 1 if qcn else V2L["n_shared"]
            ab["moe_w13"] = n_moe * (k_ * H * 2 * I * bq + n_sh * H * 2 * I * B4); ab["moe_w2"] = n_moe * (k_ * I * H * bd + n_sh * I * H * B4)
            ab.pop("total", None); tot = sum(ab.values())
            dtg = time_decode(st, steps, args.warmup, dims["kv_max_seq"], torch, None, 1)
            res.update({"decode_tok_s": steps / dtg, "ms_per_step": dtg / steps * 1e3, "steps": steps, "step_algorithmic_bytes": tot,
                        "step_frac_of_hbm_peak": tot * (steps / dtg) / 1e9 / HBM_PEAK_GBS,
                        "decode_numerics": "exact: the routed experts through the block kernels of kr_gguf.hip (per-32 INT16 activations, exact integer dots, the AVX2 kernel's 8 lane chains + hsum per row) -- bit-equal to the oracle's moe_forward_gguf driver (tests/test_decode_gpu.py)"})
            st.set_attention_mode(False, decode_fast=True)
            dtf = time_decode(st, steps, args.warmup, dims["kv_max_seq"], torch, None, 1)
            res["decode_fast_tok_s"] = steps / dtf; res["decode_fast_frac_of_hbm_peak"] = tot * (steps / dtf) / 1e9 / HBM_PEAK_GBS
            res["decode_fast_note"] = "KR_DECODE_FAST: the routed slots of the mode's gate|up and down launches walk the native GGUF blocks (the block kernels' products on per-32 INT16 activations, a row's blocks split over two waves; select, libm SiLU and the weighted combine folded in), the shared expert keeps its transposed INT4 form"
            st.set_attention_mode(False)
        except Exception as ex:
            res["decode_tok_s"] = {"error": repr(ex)}
        if qcn:
            try:
                macs = qcn_gemm_macs_per_token(L)
                res["prefill"] = prefill_model(st, dims, macs, L, 8192, args.prefill_reps, torch)
                st.set_attention_mode(True, gemm_fast=True)
                res["prefill_fast_gemm"] = prefill_model(st, dims, macs, L, 8192, args.prefill_reps, torch)
                r0 = res["prefill_fast_gemm"]["roofline"]
                res["prefill_fast_gemm"]["roofline"] = {"bound": "mfma", "achieved": r0["achieved"], "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s (f16 MFMA; 2 x useful GEMM MACs / s)", "frac": r0["achieved"] / F16_PEAK_TFLOPS}
                st.set_attention_mode(False)
            except Exception as ex:
                res["prefill"] = {"error": repr(ex)}
        del st, eng, keep
        gc.collect(); torch.cuda.empty_cache()
        return res
    eng, st, keep = build(rank, local_rank, L, 8192 + 64, bits, kv_fp8=True)
    st.set_use_graph(not args.no_graph)
    steps = min(args.steps, 50)
    dt = time_decode(st, steps, args.warmup, dims["kv_max_seq"], torch, None, 1)
    bw = B8 if bits == 8 else B4
    ab = algorithmic_bytes(L, bw) if qcn else (algorithmic_bytes_q235(L, bw) if q235 else algorithmic_bytes_v2lite(L, bw))
    res = {"workload": WORKLOAD[name], "decode_tok_s": steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "kv": "FP8-E4M3", "weights": "INT%d-g128" % bits,
           "step_algorithmic_bytes": ab["total"], "step_frac_of_hbm_peak": ab["total"] * (steps / dt) / 1e9 / HBM_PEAK_GBS}
    try:       # the same steps in KR_DECODE_FAST (layers / geometries its kernels do not cover -- MLA projections, dense MLP -- keep the exact kernels)
        st.set_attention_mode(False, decode_fast=True)
        dtf = time_decode(st, steps, args.warmup, dims["kv_max_seq"], torch, None, 1)
        res["decode_fast_tok_s"] = steps / dtf; res["decode_fast_frac_of_hbm_peak"] = ab["total"] * (steps / dtf) / 1e9 / HBM_PEAK_GBS
    except Exception as ex:
        res["decode_fast_tok_s"] = {"error": repr(ex)}
    st.set_attention_mode(False)
    try:
        macs = qcn_gemm_macs_per_token(L) if qcn else (q235_gemm_macs_per_token(L) if q235 else v2l_gemm_macs_per_token(L))
        res["prefill"] = prefill_model(st, dims, macs, L, 8192, args.prefill_reps, torch)
        for key, gfast in (("prefill_fast", False), ("prefill_fast_gemm", True)):       # tolerance modes: attention (+ delta rule), then the GEMMs as well
            st.set_attention_mode(True, gemm_fast=gfast)
            res[key] = prefill_model(st, dims, macs, L, 8192, args.prefill_reps, torch)
            if gfast:      # one f16 MFMA per MAC in this form: priced against the f16 matrix peak
                r0 = res[key]["roofline"]
                res[key]["roofline"] = {"bound": "mfma", "achieved": r0["achieved"], "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s (f16 MFMA; 2 x useful GEMM MACs / s)", "frac": r0["achieved"] / F16_PEAK_TFLOPS}
        st.set_attention_mode(False)
        res["prefill_experts_only"] = prefill_experts(eng, dims, L if qcn else L, 8192, torch)
    except Exception as ex:
        res["prefill"] = {"error": repr(ex)}
    del st, eng, keep
    gc.collect(); torch.cuda.empty_cache()
    return res


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
    res = dict(value=1.0 / t_tok, unit="tok/s", cores=best_c, host_threads=hw, kind="port", tok_s_by_threads=sweep,
               sample="whole-token passes over one token's weight set (%d MoE layers x (10 routed + shared INT4 experts) + LA/GQA projections + lm_head, "
                      "~1.9 GB streamed from DRAM), AVX2+OpenMP port of avx2.rs:1066 on tiled weights; norms/attention/state omitted "
                      "(optimistic for the CPU)" % L, ms_per_token=t_tok * 1e3)
    try:
        res["v2lite_q4k_cpu"] = cpu_v2lite_q4k(min(8.0, max_seconds * 0.5), best_c)
    except Exception as ex:
        res["v2lite_q4k_cpu"] = {"error": repr(ex)}
    return res


def cpu_v2lite_q4k(budget, threads):
    """BASELINE config 1 (DeepSeek-V2-Lite Q4_K int4cpu pure-CPU decode, testconfigs/v2lite-4-4.conf): the quantized matvecs of ONE WHOLE TOKEN on
    this host's cores -- 27 layers of MLA projections (q, kv_a, kv_b, o), the dense MLP of layer 0, 26 MoE layers of 6 routed experts + the
    shared expert (I = 2 x 1408) through moe_forward_gguf (moe.rs:990 -> gguf_kernels.rs:690: Q4_K gate / up, Q8_0 down where the row length
    is not a multiple of 256) and lm_head -- with the AVX2 integer rows of gguf_kernels.rs:271-432 (oracle q4k_row_avx2 / q8_0_row_avx2,
    bit-identical to the scalar oracle: tests/test_oracle_avx2.py) and the output rows of every matvec split over an OpenMP team (rayon in the
    reference).  Every layer has its own weights (~1.7 GB per token: streams from DRAM).  Norms, rope, the attention itself and the router are
    left out (optimistic for the CPU).  Synthetic blocks per SURVEY 8d (d = f16((0.005 + u * 0.045) / 63), dmin = f16(8 d), raw scale / quant
    bytes; Q8_0: d = f16((0.005 + u * 0.045) / 127))."""
    import numpy as np
    from oracle import oracle as O
    v = V2L; H, I, k = v["hidden"], v["inter"], v["topk"]
    rng = np.random.default_rng(0x1234)

    def blocks(t, rows, K):
        be, bb = (256, 144) if t == O.Q4_K else (32, 34)
        nb = K // be
        raw = rng.integers(0, 256, size=(rows, nb, bb), dtype=np.uint8)
        u = rng.random((rows, nb)).astype(np.float32)
        d = ((0.005 + u * 0.045) / (63.0 if t == O.Q4_K else 127.0)).astype(np.float16)
        dv = d.view(np.uint16)
        raw[:, :, 0] = dv & 0xFF; raw[:, :, 1] = dv >> 8
        if t == O.Q4_K:
            dm = (d.astype(np.float32) * 8.0).astype(np.float16).view(np.uint16)
            raw[:, :, 2] = dm & 0xFF; raw[:, :, 3] = dm >> 8
        return np.ascontiguousarray(raw.reshape(rows, nb * bb))

    def mat(rows, K):                       # the type the int4cpu build gives a [rows, K] matrix: Q4_K when K is a multiple of 256, else Q8_0
        t = O.Q4_K if K % 256 == 0 else O.Q8_0
        return (t, blocks(t, rows, K), rows, K)

    def expert(inter):
        dt = O.Q4_K if inter % 256 == 0 else O.Q8_0
        return O.GgufExpert(blocks(O.Q4_K, inter, H), blocks(O.Q4_K, inter, H), blocks(dt, H, inter), O.Q4_K, dt, H, inter)

    n_layers, nh = 27, 16
    kv_lora, rope, nope, vd = v["klr"], v["rd"], v["nd"], v["vhd"]
    layers = []
    for l in range(n_layers):
        attn = [mat(nh * (nope + rope), H), mat(kv_lora + rope, H), mat(nh * (nope + vd), kv_lora), mat(H, nh * vd)]
        if l == 0:
            layers.append((attn, None, None, [mat(v["dense_inter"], H), mat(v["dense_inter"], H), mat(H, v["dense_inter"])]))
        else:
            layers.append((attn, [expert(I) for _ in range(k)], expert(2 * I), None))
    lm = mat(v["vocab"], H)
    act = O.f32_to_bf16(((rng.random(H) - 0.5)).astype(np.float32)); w = np.full(k, 1.0 / k, np.float32)
    qx = {}
    for K in sorted({H, kv_lora, nh * vd, v["dense_inter"]}):
        qx[K] = O.gguf_quant_f32(((rng.random(K) - 0.5)).astype(np.float32))

    def mv(m):
        q, s, sm = qx[m[3]]
        O.gguf_matvec_int(m[0], m[1], q, s, sm, m[2], m[3])

    def token():
        for attn, routed, shared, dense in layers:
            for m in attn:
                mv(m)
            if dense is not None:
                for m in dense:
                    mv(m)
            else:
                O.moe_forward_gguf(routed, w, act, shared=shared, rsf=1.0)
        mv(lm)

    n_bytes = sum(m[1].nbytes for L_ in layers for m in L_[0]) + sum(m[1].nbytes for L_ in layers if L_[3] for m in L_[3]) + lm[1].nbytes
    n_bytes += sum(e.gate.nbytes + e.up.nbytes + e.down.nbytes for L_ in layers if L_[1] for e in L_[1] + [L_[2]])

    def timed(nt, b):
        O.set_num_threads(nt)
        token(); n, t0 = 0, time.perf_counter()
        while n == 0 or time.perf_counter() - t0 < b:
            token(); n += 1
        return (time.perf_counter() - t0) / n

    O.gguf_set_avx2(True)
    try:
        hw = O.num_threads()
        cands = sorted({c for c in (8, 16, 32, 64, threads, hw) if 0 < c <= hw})
        sweep = {c: timed(c, budget * 0.5 / len(cands)) for c in cands}
        best = min(sweep, key=sweep.get)
        t = timed(best, budget * 0.5)
    finally:
        O.gguf_set_avx2(False); O.set_num_threads(threads)
    return {"value": 1.0 / t, "unit": "tok/s", "ms_per_token": t * 1e3, "cores": best, "host_threads": hw, "kind": "port",
            "tok_s_by_threads": {c: round(1.0 / x, 2) for c, x in sweep.items()}, "weight_bytes_per_token": int(n_bytes),
            "dram_GBs": n_bytes / t / 1e9,
            "workload": "DeepSeek-V2-Lite Q4_K int4cpu pure-CPU decode (testconfigs/v2lite-4-4.conf, no GPU)",
            "sample": "whole-token passes over one token's weights: 27 x MLA projections, dense MLP of layer 0, 26 x (6 routed + shared native-GGUF experts: Q4_K gate/up, "
                      "Q8_0 down at I = 1408, Q4_K at 2816) through kro_moe_forward_gguf, lm_head; AVX2 integer rows (gguf_kernels.rs:271-432) + OpenMP over "
                      "output rows; norms / rope / attention / router omitted (optimistic for the CPU)"}


def ep_bytes_per_gpu(name, L, bw, world):
    """Algorithmic bytes ONE rank of an expert-parallel decode step touches (SURVEY 8d per-unit figures): the routed experts' bytes divided by the
    ranks (uniform routing: k / world slots per rank and layer on average), the shared expert on one rank per layer, everything else replicated."""
    qcn = name.startswith("qcn"); q235 = name.startswith("qwen3-235b")
    ab = algorithmic_bytes(L, bw, split_out=True) if qcn else (algorithmic_bytes_q235(L, bw) if q235 else algorithmic_bytes_v2lite(L, bw))
    ab = dict(ab); ab.pop("total", None)
    for kname in ("moe_w13", "moe_w2"):
        ab[kname] = ab[kname] / world
    ab["total"] = sum(ab.values())
    return ab


def ep_suite(name, args, torch, dist, world, rank, local_rank, with_replicas, with_prefill, force_comm=False):
    """The expert-parallel measurements of ONE model configuration on `world` ranks (see main_multi).  Returns (legs, best) -- best = dict(value, dt, form).
    world == 1 with force_comm: the same program over a ONE-RANK RCCL communicator (`--ep-selftest`, tests/test_ep_gpu.py): every collective still goes
    through librccl, so the code path the driver's 2 / 4 / 8-GPU runs take is exercised on a single-GPU box."""
    from krasis_amd.ep import ExpertParallel
    qcn = name.startswith("qcn"); q235 = name.startswith("qwen3-235b")
    dims = QCN if qcn else (Q235 if q235 else V2L)
    bits = 8 if name.endswith("q8") else 4
    L = args.layers or dims["layers"]
    kv_fp8 = args.kv == "fp8"
    kvm = dims["kv_max_seq"]
    E = dims["experts"]
    P = getattr(args, "ep_prompt_tokens", 8192)
    build = build_qcn if qcn else (build_q235 if q235 else build_v2lite)
    legs, best = {}, {"value": None, "dt": None, "form": None}
    # ---- replicas first (no collective in the data path; the earlier rounds' N > 1 number, now a labelled side leg)
    if with_replicas:
        try:
            eng0, st0, keep0 = build(rank, local_rank, L, 0, bits, kv_fp8)
            st0.set_use_graph(not args.no_graph); st0.set_attention_mode(False, decode_fast=args.decode_mode == "fast")
            d0 = time_decode(st0, args.steps, args.warmup, kvm, torch, dist, world)
            legs["replicas"] = {"tok_s_aggregate": world * args.steps / d0, "tok_s_per_replica": args.steps / d0, "scaling": "weak",
                                "note": "%d independent whole-model replicas, no collective: NOT the line's value (it measures no xGMI traffic)" % world}
            del st0, eng0, keep0
            gc.collect(); torch.cuda.empty_cache()
        except Exception as ex:
            legs["replicas"] = {"error": repr(ex)}
    # ---- this rank's shard of the expert-parallel model
    eng, st, keep = build(rank, local_rank, L, P + 64, bits, kv_fp8, ep_world=world)
    ep = ExpertParallel(eng, E, world, rank, dist if world > 1 else None, return_bf16=False, force_comm=force_comm and world == 1)
    legs["rccl_ranks"] = ep.comm_ranks()
    legs["experts_per_gpu"] = ep_local_experts(E, world, rank)

    def run_decode(key, fast, graph):
        st.set_attention_mode(False, decode_fast=fast)
        st.set_option("ep_graph", 1 if graph else 0)
        st.set_use_graph(graph)
        st.fill_state_synthetic(kvm, seed=4242)
        d = time_decode(st, args.steps, args.warmup, kvm, torch, dist, world)
        legs[key] = {"tok_s": args.steps / d, "ms_per_step": d / args.steps * 1e3, "hip_graph": graph,
                     "form": ("KR_DECODE_FAST: partial combines, one all-reduce of [hidden] f32 per MoE layer" if fast else
                              "exact: k expert rows summed over the ranks ([k, hidden] f32 all-reduce), combine in routing order -- bit-equal to single-engine decode")}
        return d
    forms = [("decode_ep_exact", False, False, "expert-parallel decode, exact form, eager launches"),
             ("decode_ep_fast", True, False, "expert-parallel decode, KR_DECODE_FAST, eager launches")]
    if not args.no_graph:
        forms.append(("decode_ep_fast_graph", True, True, "expert-parallel decode, KR_DECODE_FAST, hipGraph replay with the RCCL all-reduces captured"))
    ok = True
    for key, fast, graph, label in forms:
        try:
            d = run_decode(key, fast, graph)
            if best["value"] is None or args.steps / d > best["value"]:
                best.update(value=args.steps / d, dt=d, form=label)
        except Exception as ex:
            legs[key] = {"error": repr(ex)}
            ok = False
            break                      # a failed collective leaves the communicator unusable: stop here, report what finished
    try:
        st.set_option("ep_graph", 0); st.set_use_graph(False)
    except Exception:
        pass
    # ---- whole-model prompt pass on the expert-parallel stores: one prompt per rank
    if ok and with_prefill and not args.no_ep:
        import numpy as np
        macs = qcn_gemm_macs_per_token(L) if qcn else (q235_gemm_macs_per_token(L) if q235 else v2l_gemm_macs_per_token(L))
        for key, afast in (("prefill_model_ep", False), ("prefill_model_ep_attn_fast", True)):
            try:
                st.set_attention_mode(afast)
                st.fill_state_synthetic(P + 64, 7)
                toks = [int(x) for x in np.random.default_rng(5 + rank).integers(0, dims["vocab"], P)]
                st.prefill(toks, 0)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st.prefill(toks, 0)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                dtp = time.perf_counter() - t0
                if world > 1:
                    t = torch.tensor([dtp], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dtp = float(t.item())
                useful = 2.0 * world * P * macs / dtp / 1e12
                legs[key] = {"value": world * P / dtp, "unit": "tok/s", "tokens_per_gpu": P, "tokens_total": world * P, "ms": dtp * 1e3, "scaling": "weak", "layers": L,
                             "attention": "KR_ATTN_FAST" if afast else "exact", "gemm": "exact (the expert-parallel row path runs the exact int8-MFMA kernels)",
                             "roofline": {"bound": "mfma", "achieved": useful, "peak": I8_PEAK_TOPS * world, "unit": "TOP/s (int8, useful)", "frac": useful / (I8_PEAK_TOPS * world)}}
            except Exception as ex:
                legs[key] = {"error": repr(ex)}
                ok = False
                break
        if ok:
            try:
                st.set_attention_mode(False)
                legs["prefill_experts_ep_alltoall"] = prefill_ep(eng, dims, L, P, world, rank, torch, dist, ep=ep)
            except Exception as ex:
                legs["prefill_experts_ep_alltoall"] = {"error": repr(ex)}
                ok = False
    legs["_ok"] = ok
    try:
        ep.close()
    except Exception:
        pass
    del st, eng, keep
    gc.collect(); torch.cuda.empty_cache()
    return legs, best


def main_multi(args, torch, dist, world, rank, local_rank):
    """N > 1: the SHARDED workload.  Every rank holds E / N routed experts of every layer (its contiguous slice) and a replica of
    everything else.  Legs, every one a collective program that all ranks walk in the same order:
      decode_ep_*      expert-parallel decode (kr_decode_step on expert-parallel stores): router / attention / norms replicated, a rank evaluates the slots its
                       slice owns, one all-reduce per MoE layer over RCCL -- exact form ([k, hidden] f32 rows, bit-equal to one engine), KR_DECODE_FAST form
                       ([hidden] f32 partial combines), and the FAST form replayed from a hipGraph that captured the all-reduces (kr_decode_set_option ep_graph).
                       ONE token stream over N GPUs: strong scaling.  The best finished form is the line's `value`.
      prefill_model_ep whole-model prompt pass (kr_decode_prefill on the same stores), one prompt of P tokens per rank, (token, slot) rows exchanged
                       with the owners of their experts by grouped RCCL send / recv, several chunks in flight: weak scaling, tok/s = N * P / t.
      prefill_experts_ep_alltoall   the expert path alone (as in earlier rounds).
      replicas         N independent whole-model replicas (no collective): a labelled side leg, never the value.
      qwen3_235b_ep    (default configuration only) the same decode legs for BASELINE config 4: Qwen3-235B-A22B, 128 / N experts per GPU.
    A watchdog prints the line with whatever finished if a collective does not come back."""
    name = args.config
    qcn = name.startswith("qcn"); q235 = name.startswith("qwen3-235b")
    dims = QCN if qcn else (Q235 if q235 else V2L)
    bits = 8 if name.endswith("q8") else 4
    bw = B8 if bits == 8 else B4
    L = args.layers or dims["layers"]
    kv_fp8 = args.kv == "fp8"
    kvm = dims["kv_max_seq"]
    E = dims["experts"]
    legs, state = {}, {"value": None, "dt": None, "form": None}
    emitted = []

    def emit():
        if emitted or rank != 0:
            emitted.append(1); return
        emitted.append(1)
        ab = ep_bytes_per_gpu(name, L, bw, world)
        model = "Qwen3-Coder-Next Q%d" % bits if qcn else ("Qwen3-235B-A22B Q4" if q235 else "DeepSeek-V2-Lite Q4")
        value = state["value"]
        if value is None and isinstance(legs.get("replicas"), dict) and "tok_s_aggregate" in legs["replicas"]:
            value = legs["replicas"]["tok_s_aggregate"]; form = "FALLBACK: no expert-parallel decode form finished -- %d independent replicas (weak scaling, no collective)" % world
            scaling = "weak"
        else:
            form = state["form"] or "no decode form finished"; scaling = "strong"
        res = {"metric": "decode tok/s, %s expert-parallel @%d MI355X" % (model, world), "value": value, "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": (state["dt"] / args.steps * 1e3) if state["dt"] else None, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
               "dtype": "int%d-g128 w x int16 act -> i32, f32 scales" % bits, "dtype_note": form,
               "data": "synthetic",
               "config": {"workload": WORKLOAD[name], "layers": L, "kv": ("FP8-E4M3" if kv_fp8 else "FP16") + " KV cache, kv_max_seq %d" % kvm,
                          "parallelism": "ep%d: %d of %d routed experts per GPU and layer (contiguous slices, gpu_prefill.py:353-359); attention, router, norms, shared expert, lm_head replicated; "
                                         "decode: one f32 all-reduce per MoE layer over RCCL; prompt pass: grouped ncclSend / ncclRecv of (token, slot) rows" % (world, E // world, E),
                          "scope": "ONE token stream decoded by %d GPUs (strong scaling): full decode_step -- embedding, %d layers, final norm, lm_head, greedy sample" % (world, L),
                          "value_is": form},
               "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU", "step_algorithmic_bytes_per_gpu": ab["total"],
                            "achieved": (ab["total"] * state["value"] / 1e9) if state["value"] else None,
                            "frac": (ab["total"] * state["value"] / 1e9 / HBM_PEAK_GBS) if state["value"] else None,
                            "kernel": "whole step (per GPU: routed experts / %d + replicated attention, router, lm_head)" % world, "traffic": None}}
        res.update({k_: v for k_, v in legs.items() if not k_.startswith("_")})
        emit_line(res, args)

    import threading

    def bail():
        legs.setdefault("watchdog", "timeout after %d s: a rank did not come back from a collective; the line holds what had finished" % args.ep_timeout)
        try:
            emit()
        finally:
            os._exit(0)
    watchdog = threading.Timer(args.ep_timeout, bail); watchdog.daemon = True; watchdog.start()
    t_start = time.perf_counter()
    try:
        lg, best = ep_suite(name, args, torch, dist, world, rank, local_rank, with_replicas=not q235, with_prefill=True)
        legs.update(lg); state.update(best)
    except Exception as ex:
        legs["expert_parallel"] = {"error": repr(ex)}
    # BASELINE config 4 as a side leg of the default line: Qwen3-235B-A22B, 128 / N experts per GPU, decode only (time permitting: every rank takes the same decision)
    if qcn and bits == 4 and legs.get("_ok") and not args.no_ep:
        go = torch.tensor([1 if time.perf_counter() - t_start < args.ep_timeout * 0.45 else 0], device="cuda")
        dist.all_reduce(go, op=dist.ReduceOp.MIN)
        if int(go.item()):
            try:
                lg4, best4 = ep_suite("qwen3-235b-q4", args, torch, dist, world, rank, local_rank, with_replicas=False, with_prefill=False)
                ab4 = ep_bytes_per_gpu("qwen3-235b-q4", Q235["layers"], B4, world)
                lg4 = {k_: v for k_, v in lg4.items() if not k_.startswith("_")}
                lg4.update({"workload": "Qwen3-235B-A22B Q4 expert-parallel on %d×MI355X via RCCL over xGMI (BASELINE config 4)" % world, "value": best4["value"], "unit": "tok/s", "value_is": best4["form"],
                            "step_algorithmic_bytes_per_gpu": ab4["total"], "frac_of_hbm_peak_per_gpu": (ab4["total"] * best4["value"] / 1e9 / HBM_PEAK_GBS) if best4["value"] else None})
                legs["qwen3_235b_ep"] = lg4
            except Exception as ex:
                legs["qwen3_235b_ep"] = {"error": repr(ex)}
    watchdog.cancel()
    # the ranks share one stdout under torch.distributed.run: every rank writes out what its native libraries buffered, ranks != 0 close their stdout, THEN rank 0 prints
    flush_native_stdout()
    if rank != 0:
        silence_stdout()
    try:
        dist.barrier()
    except Exception:
        pass
    try:
        emit()
    except BaseException as ex:      # rank 0 must leave SOME last line: the driver otherwise gets no hint at all (ranks != 0 closed their stdout before the barrier)
        if rank == 0:
            try:
                print(json.dumps({"metric": "decode tok/s", "value": None, "unit": "tok/s", "n_gpus": world, "error": "emit failed: %r" % (ex,)}), flush=True)
            except Exception:
                pass
        raise
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: spawn the N ranks (one per GPU, RCCL) -- the same launch line the driver uses
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit("bench.py --gpus %d needs %d devices, this box has %d" % (args.gpus, args.gpus, have))
        import subprocess
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was launched with WORLD_SIZE=%d: the two must agree (n_gpus in the JSON line is the number of ranks that ran)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=5))

    if world > 1:
        return main_multi(args, torch, dist, world, rank, local_rank)

    name = args.config
    qcn = name.startswith("qcn"); q235 = name.startswith("qwen3-235b")
    dims = QCN if qcn else (Q235 if q235 else V2L)
    bits = 8 if name.endswith("q8") else 4
    bw = B8 if bits == 8 else B4
    L = args.layers or dims["layers"]
    pf_list = [int(x) for x in str(args.prefill_tokens).split(",") if x.strip() and int(x) > 0]
    rope_len = max([32768] + pf_list) + 64              # long-cache side measurements run to position 32 766
    kv_fp8 = args.kv == "fp8"
    build = build_qcn if qcn else (build_q235 if q235 else build_v2lite)
    eng, st, keep = build(rank, local_rank, L, rope_len, bits, kv_fp8)      # rope table: prompt pass and the long-cache side measurement
    st.set_use_graph(not args.no_graph)
    kvm = dims["kv_max_seq"]
    fast_mode = args.decode_mode == "fast"
    # the mode that is NOT the headline first (side leg, fewer steps), then the headline: K timed steps with the barrier / max-over-ranks protocol
    other, gen = {}, {}
    try:
        st.set_attention_mode(False, decode_fast=not fast_mode)
        n_o = min(args.steps, 50)
        d_o = time_decode(st, n_o, args.warmup, kvm, torch, None, 1)
        other = {"tok_s": n_o / d_o, "ms_per_step": d_o / n_o * 1e3, "steps": n_o,
                 "numerics": "KR_DECODE_FAST (tolerance mode)" if not fast_mode else "exact: bit-identical to the reference's CPU decode (tests/test_decode_gpu.py compares logits and every state tensor with array_equal)"}
        st.fill_state_synthetic(kvm, seed=4242)
        gen["exact" if fast_mode else "fast"] = decode_generate(st, kvm)
    except Exception as ex:
        other = {"error": repr(ex)}
    st.set_attention_mode(False, decode_fast=fast_mode)
    st.fill_state_synthetic(kvm, seed=4242 + rank)
    a0 = alloc_count()
    dt = time_decode(st, args.steps, args.warmup, kvm, torch, dist, world)
    decode_allocs = alloc_count() - a0
    # the reference's own decode protocol next to the back-to-back number: generate_batch, 3 x 64 tokens, the sampled token fed back
    try:
        st.fill_state_synthetic(kvm, seed=4242)
        gen["fast" if fast_mode else "exact"] = decode_generate(st, kvm)
        st.fill_state_synthetic(kvm, seed=4242)
        gen["fast_lookahead" if fast_mode else "exact_lookahead"] = decode_generate(st, kvm, lookahead=True)
        g0 = gen["fast" if fast_mode else "exact"]["tok_s"]
        gen["tok_s"] = g0; gen["tok_s_over_value"] = g0 / (args.steps / dt)
        gen["allocs_in_timed_decode_steps"] = decode_allocs
    except Exception as ex:
        gen["error"] = repr(ex)
    st.fill_state_synthetic(kvm, seed=4242 + rank)

    # per-kernel durations: un-graphed steps with HIP events around every launch on the launch stream
    per_kind_us, per_launch_us, n_per_step = profile_kinds(st, kvm, step_ms=dt / args.steps * 1e3)

    side = {}
    if world == 1 and args.flip_tokens > 0 and dims.get("experts"):
        try:
            side["router_id_flip_rate"] = router_flip_rate(st, dims, kvm, args.flip_tokens)
        except Exception as ex:
            side["router_id_flip_rate"] = {"error": repr(ex)}
        st.set_attention_mode(False, decode_fast=fast_mode); st.fill_state_synthetic(kvm, seed=4242 + rank)
    if world == 1:       # side measurements belong to the N = 1 line only
        # the same decode step with the other KV element type (the headline follows BASELINE config 3: FP8 KV)
        try:
            st.set_kv_dtype(not kv_fp8); st.set_attention_mode(False, decode_fast=fast_mode); st.fill_state_synthetic(kvm, seed=4242)
            d2 = time_decode(st, min(args.steps, 50), 3, kvm, torch, None, 1)
            side["decode_other_kv"] = {"kv": "FP16" if kv_fp8 else "FP8-E4M3", "tok_s": min(args.steps, 50) / d2, "ms_per_step": d2 / min(args.steps, 50) * 1e3}
            st.set_kv_dtype(kv_fp8); st.set_attention_mode(False, decode_fast=fast_mode); st.fill_state_synthetic(kvm, seed=4242)
        except Exception as ex:
            side["decode_other_kv"] = {"error": repr(ex)}
        if pf_list:
            try:
                side["prefill_experts_only"] = prefill_experts(eng, dims, L, 8192, torch)
                side["prefill_experts_only_fast_gemm"] = prefill_experts(eng, dims, L, 8192, torch, gemm_fast=True)
            except Exception as ex:
                side["prefill_experts_only"] = {"error": repr(ex)}
            if args.prefill_chunk:
                st.set_prefill_chunk(args.prefill_chunk)
            if args.prefill_depth:
                st.set_prefill_depth(args.prefill_depth)
            macs = qcn_gemm_macs_per_token(L) if qcn else (q235_gemm_macs_per_token(L) if q235 else v2l_gemm_macs_per_token(L))
            for key, fast, gfast in (("prefill", False, False), ("prefill_fast", True, False), ("prefill_fast_gemm", True, True)):
                st.set_attention_mode(fast, gemm_fast=gfast)
                runs = []
                for P in pf_list:
                    try:
                        r = prefill_model(st, dims, macs, L, P, args.prefill_reps, torch)
                        r["chunk"] = args.prefill_chunk or 1024; r["chunks_in_flight"] = args.prefill_depth or 3
                    except Exception as ex:
                        r = {"tokens": P, "error": repr(ex)}
                    runs.append(r)
                ok = [r for r in runs if "value" in r]
                if ok:
                    side[key] = dict(ok[0]); side[key]["by_prompt_length"] = {str(r["tokens"]): round(r["value"], 1) for r in ok}
                    side[key]["attention"] = ("fast: causal flash attention on f16 MFMA (f32 online softmax; logits within ~1e-3 relative of the exact pass)" if fast else
                                              "exact: the CPU decode's operation order per query (bit-identical to token-by-token decode); scores and P.V as f32 fma chains on the matrix cores, "
                                              "sequential softmax sum, per-token delta rule")
                    side[key]["gemm"] = ("tolerance form (KR_GEMM_FAST): f16 activation rows x weights de-quantized in registers on the f16 MFMA, f32 accumulation over the "
                                         "whole k range -- the dataflow of the reference's GPU prompt pass; expert outputs within 2-4e-4 relative RMS of the exact kernel, "
                                         "PPL within 1e-3 (tests/test_gemm_fast_gpu.py)" if gfast else
                                         "exact: INT16 activation digits on the int8 MFMA, one f32 fma per 128-group -- bit-identical to the reference's CPU engine")
                    if gfast:     # the useful-flop rate of the same GEMM MACs against the f16 matrix peak (one pass per MAC in this form)
                        r0 = side[key]["roofline"]
                        side[key]["roofline"] = {"bound": "mfma", "achieved": r0["achieved"], "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s (f16 MFMA; 2 x useful GEMM MACs / s)",
                                                 "frac": r0["achieved"] / F16_PEAK_TFLOPS}
                    side[key]["note"] = ("whole-model prompt pass; `value` is the first listed length, by_prompt_length holds every length of "
                                         "--prefill-tokens (the reference benchmark's 20 434 / 35 139 / 49 863-token prompts, benchmark.py:434-505)")
                else:
                    side[key] = runs[0]
            st.set_attention_mode(False)
        if not args.no_long_context:          # the same decode step late in a long cache
            try:
                kvn = "FP8-E4M3" if kv_fp8 else "FP16"
                side["decode_long_context"] = long_context(st, 8192, torch, kvn)
                side["decode_long_context_32k"] = long_context(st, 32768, torch, kvn)
                side["decode_long_context_fast"] = long_context(st, 8192, torch, kvn, fast=True)
                side["decode_long_context_32k_fast"] = long_context(st, 32768, torch, kvn, fast=True)
            except Exception as ex:
                side["decode_long_context"] = {"error": repr(ex)}

    ab = (algorithmic_bytes(L, bw, split_out=fast_mode) if qcn else (algorithmic_bytes_q235(L, bw) if q235 else algorithmic_bytes_v2lite(L, bw)))
    ep_legs = {}

    def emit():
        """rank 0's ONE JSON line (this function is the N = 1 line; N > 1 lines come from main_multi)"""
        sym_us, sym_bytes, sym_n, sym_raw = {}, {}, {}, {}
        raw_us = getattr(profile_kinds, "raw_us_per_step", {})
        for j in range(NK):
            kname = KINDS[j]; sym = (SYMBOL_FAST if fast_mode and qcn and bits == 4 else SYMBOL).get(kname, kname)
            sym_us[sym] = sym_us.get(sym, 0.0) + per_kind_us[kname]; sym_bytes[sym] = sym_bytes.get(sym, 0.0) + ab.get(kname, 0.0)
            sym_n[sym] = sym_n.get(sym, 0) + n_per_step[kname]; sym_raw[sym] = sym_raw.get(sym, 0.0) + raw_us.get(kname, 0.0)
        dom = max(sym_us, key=lambda s: sym_us[s])
        achieved = sym_bytes[dom] / (sym_us[dom] * 1e-6) / 1e9 if sym_us[dom] > 0 else 0.0
        achieved_raw = sym_bytes[dom] / (sym_raw[dom] * 1e-6) / 1e9 if sym_raw.get(dom, 0.0) > 0 else None
        tok_s = args.steps / dt
        traffic, traffic_src = pmc_traffic(dom)
        model = "Qwen3-Coder-Next Q%d" % bits if qcn else ("Qwen3-235B-A22B Q4" if q235 else "DeepSeek-V2-Lite Q4")
        mode_tag = "KR_DECODE_FAST tolerance mode" if fast_mode else "bit-exact mode"
        exact_tok_s = other.get("tok_s") if fast_mode else tok_s
        fast_tok_s = tok_s if fast_mode else other.get("tok_s")
        res = {
            "metric": "decode tok/s, %s @1 MI355X (%s; both modes: value_exact / value_fast)" % (model, mode_tag),
            "value": tok_s, "value_exact": exact_tok_s, "value_fast": fast_tok_s, "unit": "tok/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int%d-g128 w x int16 act -> i32, f32 scales" % bits,
            "dtype_note": ("int%d-g128 weights x int16 activations -> i32 group sums, f32 scales" % bits) +
                     (" (KR_DECODE_FAST: the reference's products, f32 sums as lane / wave / workgroup trees -- logits within 2e-3 of the bit-exact mode, "
                      "router ids identical for identical logits; tests/test_decode_fast_gpu.py)" if fast_mode else " (reference CPU-decode numerics, bit-exact)"),
            "data": "synthetic",
            "config": {"workload": WORKLOAD[name],
                       "scope": "full decode_step: embedding, %d layers (attention + MoE + shared expert), final norm, lm_head, greedy sample" % L,
                       "kv": ("FP8-E4M3" if kv_fp8 else "FP16") + " KV cache, kv_max_seq %d" % kvm, "layers": L,
                       "parallelism": "one GPU holds the whole model (N > 1 lines: expert parallelism, see main_multi)",
                       "hip_graph": not args.no_graph, "target_tok_s": 200,
                       "decode_mode": "`value` = fast (KR_DECODE_FAST, opt-in tolerance mode; the library default is the bit-exact graph = `value_exact`)" if fast_mode else "`value` = exact (library default); `value_fast` = KR_DECODE_FAST",
                       "prefill_modes": "`prefill` = the library default (exact: bit-identical to token-by-token decode); `prefill_fast` (KR_ATTN_FAST) and "
                                        "`prefill_fast_gemm` (KR_ATTN_FAST | KR_GEMM_FAST) are the opt-in throughput modes a serving deployment would run -- "
                                        "the one to compare with the reference's GPU prompt pass (bf16 flash attention + Marlin GEMM) is prefill_fast_gemm; protocol: one un-timed pass of the "
                                        "same prompt in the same mode, then >= 2 timed passes (median), allocation counter checked",
                       "decode_token_protocol": "`value`: token 0 at positions 10.. like bench_decode_synthetic (decode.rs:5450), K graph replays back to back; `decode_generate`: the "
                                                "reference's generate_batch protocol (benchmark.py:434-505): 3 x 64 tokens, the sampled token fed back"},
            ("decode_exact" if fast_mode else "decode_fast"): other,
            "decode_generate": gen,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "achieved_raw_events": achieved_raw, "frac_raw_events": (achieved_raw / HBM_PEAK_GBS) if achieved_raw else None,
                         "peak_measured_stream_read": 6996.0, "peak_measured_source": "tools/probes/hbm_stream.hip on this box type (8 GiB, 16-byte loads)", "traffic": traffic, "traffic_unit": "HBM fetch bytes per launch (PMC FETCH_SIZE, separate pass)",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": sym_bytes[dom] / max(sym_n[dom], 1), "us_per_launch": sym_us[dom] / max(sym_n[dom], 1),
                         "us_per_launch_raw_events": (sym_raw[dom] / max(sym_n[dom], 1)) if sym_raw.get(dom) else None,
                         "launches_per_step": sym_n[dom],
                         "timing": "HIP events around every launch of un-graphed steps on the launch stream; `us_per_launch` / `achieved` / `frac` take off the constant per-launch excess "
                                   "of the event pairs ((sum of event times - graph-replayed step time) / launches), the *_raw_events fields are the event times as measured",
                         "event_pair_overhead_us": round(getattr(profile_kinds, "event_overhead_us", 0.0), 2),
                         "step_algorithmic_bytes": ab["total"], "step_effective_GBs": ab["total"] * (args.steps / dt) / 1e9,
                         "step_frac_of_hbm_peak": ab["total"] * (args.steps / dt) / 1e9 / HBM_PEAK_GBS,
                         "per_kind_us_per_step": {k_: round(v, 2) for k_, v in per_kind_us.items()},
                         "per_kind_us_per_step_raw_events": {k_: round(v, 2) for k_, v in raw_us.items()},
                         "per_kind_us_per_launch": {k_: round(v, 2) for k_, v in per_launch_us.items()}},
        }
        res.update(side)
        res.update(ep_legs)
        if not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, QCN["layers"])
            except Exception as ex:  # a reported side number, never the product path
                res["cpu_baseline"] = {"error": repr(ex)}
        emit_line(res, args)

    del st, eng, keep
    gc.collect(); torch.cuda.empty_cache()

    if args.ep_selftest:      # the N > 1 program of main_multi over a ONE-RANK RCCL communicator: every collective goes through librccl on this single GPU
        try:
            lg, best = ep_suite(name, args, torch, None, 1, 0, local_rank, with_replicas=False, with_prefill=True, force_comm=True)
            lg = {k_: v for k_, v in lg.items() if not k_.startswith("_")}
            lg["value"] = best["value"]; lg["value_is"] = best["form"]
            ep_legs["expert_parallel_selftest_one_rank_rccl"] = lg
        except Exception as ex:
            ep_legs["expert_parallel_selftest_one_rank_rccl"] = {"error": repr(ex)}

    if pf_list:
        try:
            side["prefill_experts_only_q4k_gguf"] = prefill_experts_gguf(local_rank, torch)
            side["prefill_experts_only_q4k_gguf_fast_gemm"] = prefill_experts_gguf(local_rank, torch, gemm_fast=True)
        except Exception as ex:
            side["prefill_experts_only_q4k_gguf"] = {"error": repr(ex)}
    if args.side_configs:
        side["configs"] = {}
        for sc in [s for s in args.side_configs.split(",") if s.strip() and s.strip() != name]:
            try:
                side["configs"][sc] = side_config(sc.strip(), rank, local_rank, args, torch)
            except Exception as ex:
                side["configs"][sc] = {"error": repr(ex)}

    emit()


if __name__ == "__main__":
    main()
