"""probe: the tolerance GEMM (kr_prefill_h.hip) against the exact one on the QCN shape.
  part 1: experts only (sort + 2 grouped GEMMs + act + combine), 48 layers, M tokens per call, exact vs fast
  part 2: whole-model FAST prompt pass, attention-only FAST vs attention + GEMM FAST, swept over chunk / depth
argv: [tokens=8192]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from krasis_amd import GpuPrefillManager
from krasis_amd._lib import check
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = 48
eng, st, keep = bench.build_qcn(0, 0, L, P + 64, 4, True)
q = bench.QCN; H, I, E, k = q["hidden"], q["inter"], q["experts"], q["topk"]
g = torch.Generator(device="cuda").manual_seed(7)
mgr = GpuPrefillManager(eng, k)
for M in (1024, 4096, 8192):
    x = ((torch.rand((M, H), device="cuda", generator=g) - 0.5)).to(torch.bfloat16)
    ids = torch.rand((M, E), device="cuda", generator=g).topk(k, dim=1).indices.to(torch.int32)
    w = torch.softmax(torch.randn((M, k), device="cuda", generator=g), dim=1)
    for fast in (0, 1):
        check(eng._lib.kr_moe_set_gemm_mode(eng._h, fast))
        for l in range(2): mgr.forward(l, x, ids, w, routed_only=True)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 8192 // M
        ev0.record()
        for _ in range(reps):
            for l in range(L): mgr.forward(l, x, ids, w, routed_only=True)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1)
        macs = 8192 * k * 3 * H * I * L
        print("experts-only M=%d x %d fast=%d: %.1f ms per 8192 tokens, %.0f tok/s, useful %.1f TFLOP/s" % (M, reps, fast, ms, 8192 / (ms * 1e-3), 2.0 * macs / (ms * 1e-3) / 1e12), flush=True)
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, 0))
toks = [int(t) for t in np.random.default_rng(5).integers(0, q["vocab"], P)]
lg = {}
for gf in (False, True):     # last-position logits of the two FAST forms on the QCN shape (dense GEMMs of the 128 x 256 form included)
    st.set_attention_mode(True, gemm_fast=gf); st.set_prefill_chunk(2048); st.set_prefill_depth(2)
    st.fill_state_synthetic(P + 64, 7)
    out = np.empty(q["vocab"], np.float32); st.prefill(toks[:4096], 0, out.ctypes.data); torch.cuda.synchronize(); lg[gf] = out.copy()
print("QCN 4096-token prompt, last-position logits, gemm_fast vs exact GEMMs (both attention FAST): max rel %.3e, argmax %d / %d, finite %s" % (
      float(np.abs(lg[True] - lg[False]).max() / np.abs(lg[False]).max()), int(lg[False].argmax()), int(lg[True].argmax()), bool(np.isfinite(lg[True]).all())), flush=True)
for gf in (False, True):
    st.set_attention_mode(True, gemm_fast=gf)
    for chunk, depth in ((1024, 3), (2048, 2), (2048, 3), (4096, 2), (8192, 1)):
        if chunk > P: continue
        st.set_prefill_chunk(chunk); st.set_prefill_depth(depth)
        st.fill_state_synthetic(P + 64, 7)
        st.prefill(toks, 0); torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            st.fill_state_synthetic(P + 64, 7); torch.cuda.synchronize()
            t0 = time.perf_counter(); st.prefill(toks, 0); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("prompt pass %d tokens attn FAST gemm_fast=%d chunk %d depth %d: %.1f ms  %.0f tok/s" % (P, gf, chunk, depth, best * 1e3, P / best), flush=True)
