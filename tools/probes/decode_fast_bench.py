#!/usr/bin/env python3
"""Decode step of the QCN-shaped synthetic model (bench.py's build) in the exact mode and in KR_DECODE_FAST: tok/s, per-kind launch times
(HIP events, un-graphed), logits difference over the first steps from the same state, and the router check (FAST ids vs the oracle's
topk_indices on the same logits).  Writes gpurun_out/r03_decode_fast_bench.{txt,json}.
    python tools/probes/decode_fast_bench.py [--steps 100] [--layers 48] [--route-tokens 300] [--only fast|exact]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--layers", type=int, default=48)
    ap.add_argument("--route-tokens", type=int, default=300)
    ap.add_argument("--only", default="")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--out", default="gpurun_out/r03_decode_fast_bench")
    ap.add_argument("--opt", action="append", default=[], help="name=value for kr_decode_set_option (A/B hooks), e.g. --opt lm_fused=0")
    args = ap.parse_args()
    import torch
    q = bench.QCN; kvm = q["kv_max_seq"]; L = args.layers
    eng, st, keep = bench.build_qcn(0, 0, L, 0, args.bits, kv_fp8=True)
    res = {"layers": L, "steps": args.steps, "bits": args.bits}
    for o in args.opt:
        name, val = o.split("="); st.set_option(name, int(val)); res.setdefault("options", {})[name] = int(val)
    lines = []
    modes = [m for m in ("exact", "fast") if not args.only or args.only == m]
    logits = {}
    for mode in modes:
        st.set_attention_mode(False, decode_fast=(mode == "fast"))
        st.fill_state_synthetic(kvm, seed=4242)
        lg = []
        for i in range(6):
            out = np.empty(q["vocab"], np.float32); st.decode_step(0 if i == 0 else 17 * i, 10 + i, out.ctypes.data); lg.append(out)
        logits[mode] = lg
        st.fill_state_synthetic(kvm, seed=4242)
        dt = bench.time_decode(st, args.steps, 5, kvm, torch, None, 1)
        per_kind_us, per_launch_us, n_per_step = bench.profile_kinds(st, kvm, step_ms=dt / args.steps * 1e3)
        res[mode] = {"tok_s": args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                     "per_kind_us_per_step": {k: round(v, 2) for k, v in per_kind_us.items() if v > 0},
                     "per_kind_us_per_launch": {k: round(v, 2) for k, v in per_launch_us.items() if n_per_step[k] > 0},
                     "launches_per_step": {k: v for k, v in n_per_step.items() if v > 0}}
        ab = bench.algorithmic_bytes(L, bench.B8 if args.bits == 8 else bench.B4)
        res[mode]["step_frac_of_hbm_peak"] = ab["total"] * (args.steps / dt) / 1e9 / bench.HBM_PEAK_GBS
        lines.append("%s: %.1f tok/s  %.3f ms/step  frac of HBM peak %.4f  launches/step %d" % (mode, args.steps / dt, dt / args.steps * 1e3, res[mode]["step_frac_of_hbm_peak"],
                                                                                               int(sum(n_per_step.values()))))
        for k in per_kind_us:
            if n_per_step[k] > 0:
                lines.append("    %-20s %6.1f launches  %8.2f us/launch  %9.1f us/step" % (k, n_per_step[k], per_launch_us[k], per_kind_us[k]))
    if len(modes) == 2:
        rel = [float(np.abs(a - b).max() / np.abs(a).max()) for a, b in zip(logits["exact"], logits["fast"])]
        same = [int(np.argmax(a)) == int(np.argmax(b)) for a, b in zip(logits["exact"], logits["fast"])]
        res["logits_rel_err_first_steps"] = rel; res["greedy_same"] = same
        lines.append("logits max|fast-exact|/max|exact| over the first 6 steps (same state, same tokens): " + " ".join("%.2e" % r for r in rel) + "  greedy same: %s" % same)
    if "fast" in modes and args.route_tokens > 0:
        from oracle import oracle as O
        st.set_attention_mode(False, decode_fast=True)
        rng = np.random.default_rng(3)
        bad = 0; wworst = 0.0; margins = []
        out = np.empty(q["vocab"], np.float32)
        for i in range(args.route_tokens):
            st.decode_step(int(rng.integers(0, q["vocab"])), 10 + (i % 200), out.ctypes.data)
            lgts, ids, w = st.read_router(q["experts"], q["topk"])
            rid, rw = O.route_score_topk(lgts, q["topk"], 1, True, None)[:2]
            bad += int(not np.array_equal(np.asarray(rid, np.int32), ids))
            wworst = max(wworst, float(np.abs(np.asarray(rw, np.float32) - w).max() / np.abs(rw).max()))
            srt = np.sort(lgts)[::-1]; margins.append(float(srt[q["topk"] - 1] - srt[q["topk"]]))
        res["router"] = {"tokens": args.route_tokens, "ids_identical_to_oracle_on_same_logits": args.route_tokens - bad, "weights_rel_err": wworst,
                         "min_margin_kth_vs_next_logit": min(margins)}
        lines.append("router (last MoE layer, E=512 k=10): %d/%d tokens with ids identical to the oracle's topk on the same logits; weights rel err %.2e; smallest k-th/next margin %.3e"
                     % (args.route_tokens - bad, args.route_tokens, wworst, min(margins)))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out + ".txt", "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(args.out + ".json", "w") as f:
        json.dump(res, f, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
