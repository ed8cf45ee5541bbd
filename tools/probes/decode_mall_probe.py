#!/usr/bin/env python3
"""Upper bound of what a prefetch of the ROUTING-INDEPENDENT weights of a decode layer (in / out projections, router gate, shared expert: ~25 of the
43 MB a QCN layer touches) into the 256 MB Infinity Cache could buy: the QCN-shaped synthetic model of bench.py with those weights SHARED by all 48
layers (one linear-attention set, one GQA set, one router gate, one shared expert: ~30 MB in total, resident in the last-level cache after the first
layer) against the normal model (every layer its own weights, 2.06 GB per token from HBM).  The routed experts stay distinct per layer in both arms.
    python tools/probes/decode_mall_probe.py [--steps 100]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402


def build(share):
    from krasis_amd import CpuDecodeStore, KrasisEngine, ModelConfig
    q = bench.QCN; H, I, E, k, V, L = q["hidden"], q["inter"], q["experts"], q["topk"], q["vocab"], q["layers"]
    eng = KrasisEngine(device=0)
    eng.configure(ModelConfig(H, I, E, k, L, 0, 1.0))
    eng.fill_synthetic(4, seed=0x12345678ABCDEF01)
    eng.set_routing_config("softmax", True, k, E, H)
    st = CpuDecodeStore(128, True, True); st.set_moe_store(eng)
    rng = np.random.default_rng(1234); keep = []; seed = [100]

    def W(rows, cols):
        seed[0] += 1
        return st.store_weight_synthetic(rows, cols, 4, seed[0])

    def N(n):
        w = ((rng.random(n, dtype=np.float32) - 0.5) * 0.2).astype(np.float32); keep.append(w)
        return st.store_norm_weight(w.ctypes.data, n)

    fin, lm = N(H), W(V, H)
    st.configure_decode(H, L, q["eps"], fin, lm, V, k, 1, True, 1.0, 0, synth_seed=777)
    nk, nv, dk, dv, nh, nkv, hd = q["nk"], q["nv"], q["dk"], q["dv"], q["nh"], q["nkv"], q["hd"]
    hr = nv // nk; group_dim = 2 * dk + 2 * dv * hr; conv_dim = 2 * nk * dk + nv * dv
    cache = {}

    def once(key, fn):
        if not share:
            return fn()
        if key not in cache:
            cache[key] = fn()
        return cache[key]

    for l in range(L):
        n_in, n_post = N(H), N(H)
        if bench.is_gqa(l):
            qw, kw, vw, ow = once("gqa", lambda: (W(nh * hd * 2, H), W(nkv * hd, H), W(nkv * hd, H), W(H, nh * hd)))
            qn = (rng.random(hd, dtype=np.float32) + 0.5).astype(np.float32); kn = (rng.random(hd, dtype=np.float32) + 0.5).astype(np.float32); keep += [qn, kn]
            st.add_decode_gqa_layer(n_in, n_post, qw, kw, vw, ow, qn.ctypes.data, hd, kn.ctypes.data, hd, True, nh, nkv, hd, 1.0 / hd ** 0.5)
        else:
            qkvz, ba, out = once("la", lambda: (W(nk * group_dim, H), W(nk * 2 * hr, H), W(H, nv * dv)))
            cw = ((rng.random(conv_dim * 4, dtype=np.float32) - 0.5) * 1.0).astype(np.float32)
            a_log = ((rng.random(nv, dtype=np.float32) - 0.5) * 2.0).astype(np.float32); dtb = ((rng.random(nv, dtype=np.float32) - 0.5)).astype(np.float32)
            nw = (rng.random(nv * dv, dtype=np.float32) + 0.5).astype(np.float32); keep += [cw, a_log, dtb, nw]
            st.add_decode_la_layer(n_in, n_post, qkvz, ba, out, cw.ctypes.data, a_log.ctypes.data, dtb.ctypes.data, nw.ctypes.data, nk, nv, dk, dv, nv // nk, 4, 1.0 / dk ** 0.5)
        eng.set_route_weight_synthetic(l, 0x12345678ABCDEF01, 0.02, True)
        sgu, sd, sg = once("shared", lambda: (W(2 * q["shared_inter"], H), W(H, q["shared_inter"]), W(1, H)))
        st.set_decode_layer_moe(l, l, l, sgu, sd, sg)
    half = hd // 2; rope_len = q["kv_max_seq"]
    pos = np.arange(rope_len, dtype=np.float32)[:, None]
    freq = (1.0 / (10000.0 ** (2.0 * np.arange(half, dtype=np.float32) / hd))).astype(np.float32)[None, :]
    cos, sin = np.cos(pos * freq).astype(np.float32), np.sin(pos * freq).astype(np.float32); keep += [cos, sin]
    st.set_decode_rope(cos.ctypes.data, sin.ctypes.data, half, rope_len)
    st.finalize_decode(); st.set_kv_dtype(True); st.fill_state_synthetic(q["kv_max_seq"], seed=4242)
    return eng, st, keep


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--out", default="gpurun_out/r05_decode_mall_probe.txt")
    args = ap.parse_args()
    import gc
    import torch
    lines = []
    for share in (False, True):
        eng, st, keep = build(share)
        st.set_attention_mode(False, decode_fast=True)
        kvm = bench.QCN["kv_max_seq"]
        dts = [bench.time_decode(st, args.steps, 5, kvm, torch, None, 1) for _ in range(3)]
        dt = sorted(dts)[1]
        per_kind_us, per_launch_us, n_per_step = bench.profile_kinds(st, kvm, step_ms=dt / args.steps * 1e3)
        lines.append("%s: %.1f tok/s, %.3f ms/step (3 runs: %s)" % ("projections / router gate / shared expert SHARED by all layers (cache-resident)" if share else "every layer its own weights (normal model)",
                                                                    args.steps / dt, dt / args.steps * 1e3, " ".join("%.3f" % (x / args.steps * 1e3) for x in dts)))
        for kk in per_kind_us:
            if n_per_step[kk] > 0:
                lines.append("    %-20s %6.1f launches  %8.2f us/launch" % (kk, n_per_step[kk], per_launch_us[kk]))
        del st, eng, keep
        gc.collect(); torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
