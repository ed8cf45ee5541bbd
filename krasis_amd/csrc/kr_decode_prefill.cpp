// kr_decode_prefill.cpp -- the whole-model prompt pass on the GPU.
//
// The reference runs the prompt through third-party GPU kernels (flashinfer attention, sglang fused_marlin_moe, torch matmuls:
// python/krasis/model.py forward_prefill_layer_grouped, layer.py:242-461, attention.py:496-687, linear_attention.py:695-845) and
// then hands KV / recurrent state to the CPU decoder (decode_setup.py:232-278).  Here the prompt pass is defined as "what the
// decode graph would have produced token by token": kr_decode_prefill(tokens[0..n)) leaves logits, FP16 KV caches, conv and
// recurrent state bit-identical to n successive kr_decode_step calls (src/decode.rs:2690-3520), so there is one numerics story for
// prefill and decode and no state hand-off.  GEMM-shaped work rides the int8-MFMA grouped GEMM (kr_prefill.hip); everything else
// is the batched form of the decode operators (kr_prefill_ops.hip).  Tokens are processed in chunks through all layers.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "../../include/krasis_hip.h"
#include "kr_decode_internal.h"
#include "kr_prefill.h"
#include "kr_prefill_ops.h"
#include "kr_router.h"

#define KR_PFM_CHUNK 1024     // tokens per chunk; with KR_PFM_DEPTH chunks in flight (round-1 sweep, docs/design/08-measurement.md: 1024 x 3 beats 2048 x 2 at 4k / 8k / 20k prompts)
#define KR_PFM_DEPTH 3

namespace {
struct Scratch {   // carved from one allocation per arena (grown on demand)
    float* lac;
    float *res, *hid, *normed, *pa, *pb, *pc, *q, *k, *v, *z, *gexp, *beta, *recur, *attn, *gate, *moe, *sh, *gv, *sgu, *logits;
    int8_t *xh, *xl, *yh, *yl; float *xs, *ys;
    uint16_t *xf, *yf; float *xfm, *yfm;     // KR_GEMM_FAST: the same two activation slots as f16 rows + row multipliers (kr_prefill_h.hip)
    uint16_t* xb; int32_t* ids; float* w;
};
struct Act { const int8_t* h; const int8_t* l; const float* s; const uint16_t* f; const float* fm; };   // one GEMM input: INT16 digits or f16 rows
Act X(const Scratch& B) { return Act{B.xh, B.xl, B.xs, B.xf, B.xfm}; }
Act Y(const Scratch& B) { return Act{B.yh, B.yl, B.ys, B.yf, B.yfm}; }
struct Chunk {      // one chunk of the prompt in flight: its arena, stream and running flags
    Scratch B; float* scores; const int* tok; int Cc, pos0, set; bool first, add_is_emb; hipStream_t st;
    KrPfSync sy;      // this (chunk, layer)'s hand-overs with the previous / next chunk (set by the scheduler before every run_layer)
};
size_t al(size_t n) { return (n + 255) & ~(size_t)255; }
}  // namespace

static int pf_gemm(kr_decode_store* s, int wid, const Act& A, int C, float* out, int ld, hipStream_t st) {
    DWeight& W = *s->weights[wid];
    if (s->gemm_fast) { kr_launch_pfh_gemm(W.ms.view(), A.f, A.fm, nullptr, 1, 0, 0, C, out, ld, st); return KR_OK; }
    if (!W.ms.wsum.p) return kr_fail(KR_ERR_STATE, "internal: nibble sums of weight %d were not prepared", wid);
    kr_launch_pf_gemm(W.ms.view(), (const uint32_t*)W.ms.wsum.p, A.h, A.l, A.s, nullptr, 1, 0, 0, C, out, ld, st);
    return KR_OK;
}
// up to three projections of the same input in ONE launch (q | k | v, qkvz | ba, shared gate_up | shared gate); falls back to one launch each
// when the weight widths or K differ
static int pf_gemm_multi(kr_decode_store* s, const int* wids, float* const* outs, const int* lds, int n, const Act& A, int C, hipStream_t st) {
    KrMatDev mats[3]; const uint32_t* ws[3];
    bool same = n <= 3;
    for (int i = 0; i < n; i++) {
        DWeight& W = *s->weights[wids[i]];
        if (!s->gemm_fast && !W.ms.wsum.p) return kr_fail(KR_ERR_STATE, "internal: nibble sums of weight %d were not prepared", wids[i]);
        if (i < 3) { mats[i] = W.ms.view(); ws[i] = (const uint32_t*)W.ms.wsum.p; same = same && mats[i].bits == mats[0].bits && mats[i].ng == mats[0].ng; }
    }
    if (same && n > 1) {
        if (s->gemm_fast) kr_launch_pfh_gemm_multi(mats, outs, lds, n, A.f, A.fm, C, st);
        else kr_launch_pf_gemm_multi(mats, ws, outs, lds, n, A.h, A.l, A.s, C, st);
        return KR_OK;
    }
    for (int i = 0; i < n; i++) if (int rc = pf_gemm(s, wids[i], A, C, outs[i], lds[i], st)) return rc;
    return KR_OK;
}
// the producers of a GEMM input: INT16 digits (exact) or f16 rows (KR_GEMM_FAST)
static void pf_rows(kr_decode_store* s, const float* x, int C, int ld, int K, Scratch& B, bool slot_y, hipStream_t st) {
    if (s->gemm_fast) kr_launch_pfh_rows_f32(x, C, ld, K, slot_y ? B.yf : B.xf, slot_y ? B.yfm : B.xfm, st);
    else kr_launch_pfm_quant_f32(x, C, ld, K, slot_y ? B.yh : B.xh, slot_y ? B.yl : B.xl, slot_y ? B.ys : B.xs, st);
}
static void pf_act(kr_decode_store* s, const float* gu, int C, int n, int gu_ld, Scratch& B, hipStream_t st) {
    if (s->gemm_fast) kr_launch_pfh_act(gu, C, n, gu_ld, KR_ACT_SILU_MUL, 0.0f, 0.0f, B.yf, B.yfm, st);
    else kr_launch_pf_act(gu, C, n, gu_ld, KR_ACT_SILU_MUL, 0.0f, 0.0f, B.yh, B.yl, B.ys, st);
}

// everything a layer launches, for one chunk, on the chunk's stream
static int run_layer(kr_decode_store* s, Chunk& cx, size_t li) {
    kr_engine* e = s->eng;
    const int H = s->hidden, k = s->topk, Cc = cx.Cc, pos0 = cx.pos0;
    Scratch& B = cx.B; hipStream_t st = cx.st;
    DLayer& L = s->layers[li];
    // ---- input norm (+ digits for the projections)
    KrPfmNormArgs na{};
    na.mode = cx.add_is_emb ? 1 : 0; na.add_in = B.hid; na.emb = (const float*)s->embedding.p; na.tokens = cx.tok; na.res = B.res;
    na.w = (const float*)s->norms[L.input_norm]->p; na.out = B.normed; na.xh = B.xh; na.xl = B.xl; na.xs = B.xs; na.H = H; na.first = cx.first ? 1 : 0;
    na.bias_one = s->norm_bias_one; na.eps = s->eps;
    if (s->gemm_fast) { na.xh = nullptr; na.xl = nullptr; na.xs = nullptr; }      // tolerance GEMMs take f16 rows of the normalised value instead of the digits
    if (s->gemm_fast && s->opt_norm_rows) { na.xf = B.xf; na.xfm = B.xfm; }      // ... and the norm launch writes that row image itself (round 6: one launch per norm less)
    kr_launch_pfm_norm(na, Cc, st);
    if (s->gemm_fast && !s->opt_norm_rows) kr_launch_pfh_rows_f32(B.normed, Cc, H, H, B.xf, B.xfm, st);
    cx.first = false; cx.add_is_emb = false;
    if (L.attn == ATTN_LA) {
        const int nq = s->weights[L.qkvz_wid]->rows, nb = s->weights[L.ba_wid]->rows, oc = s->weights[L.out_wid]->cols;
        { const int wids[2] = {L.qkvz_wid, L.ba_wid}; float* outs[2] = {B.pa, B.pb}; const int lds[2] = {nq, nb};
          if (int rc = pf_gemm_multi(s, wids, outs, lds, 2, X(B), Cc, st)) return rc; }
        KrPfmLaArgs a{};
        a.qkvz = B.pa; a.ld_qkvz = nq; a.ba = B.pb; a.ld_ba = nb; a.conv_state = (float*)L.conv_state.p; a.conv_w = (const float*)L.conv_w.p;
        a.a_log = (const float*)L.a_log.p; a.dt_bias = (const float*)L.dt_bias.p; a.scale = L.la_scale; a.q = B.q; a.k = B.k; a.v = B.v; a.z = B.z;
        a.gexp = B.gexp; a.beta = B.beta; a.nk = L.nk; a.nv = L.nv; a.dk = L.dk; a.dv = L.dv; a.hr = L.nv / L.nk;
        a.lac = B.lac; a.fast = s->attn_fast; a.conv_fused = s->opt_la_conv_fused;
        if (kr_launch_pfm_la(a, (float*)L.recur_state.p, B.recur, (const float*)L.la_norm_w.p, B.attn, Cc, s->eps, st, &cx.sy))
            return kr_fail(KR_ERR_VALUE, "unsupported linear-attention geometry");
        if (oc != L.nv * L.dv) return kr_fail(KR_ERR_VALUE, "out_proj cols %d != nv*dv", oc);
        pf_rows(s, B.attn, Cc, oc, oc, B, true, st);
        if (int rc = pf_gemm(s, L.out_wid, Y(B), Cc, B.hid, H, st)) return rc;
    } else if (L.attn == ATTN_GQA) {
        if (!L.kv_k.p) return kr_fail(KR_ERR_STATE, "set_decode_state was not called (no KV cache for layer %zu)", li);
        const int nq = s->weights[L.q_wid]->rows, nk_ = s->weights[L.k_wid]->rows, nv_ = s->weights[L.v_wid]->rows, oc = s->weights[L.o_wid]->cols;
        { const int wids[3] = {L.q_wid, L.k_wid, L.v_wid}; float* outs[3] = {B.pa, B.pb, B.pc}; const int lds[3] = {nq, nk_, nv_};
          if (int rc = pf_gemm_multi(s, wids, outs, lds, 3, X(B), Cc, st)) return rc; }
        KrPfmGqaArgs a{};
        a.q_in = B.pa; a.k_in = B.pb; a.v_in = B.pc; a.ld_q = nq; a.ld_k = nk_; a.ld_v = nv_;
        a.q_norm = L.q_norm_len ? (const float*)L.q_norm.p : nullptr; a.k_norm = L.k_norm_len ? (const float*)L.k_norm.p : nullptr;
        a.q_norm_per_head = L.q_norm_len == L.nh * L.hd; a.k_norm_per_head = L.k_norm_len == L.nkv * L.hd;
        a.rope_cos = (const float*)s->rope_cos.p; a.rope_sin = (const float*)s->rope_sin.p; a.rope_half = s->rope_half;
        a.k_cache = L.kv_k.p; a.v_cache = L.kv_v.p; a.kv_fp8 = s->kv_fp8; a.q_out = B.q; a.gate = B.gate; a.attn_out = B.attn;
        a.gated = L.gated; a.nh = L.nh; a.nkv = L.nkv; a.hd = L.hd; a.pos0 = pos0; a.eps = s->eps; a.sm_scale = L.sm_scale;
        if (s->max_rope_seq > 0 && pos0 + Cc > s->max_rope_seq) return kr_fail(KR_ERR_VALUE, "prompt exceeds the rope table (%d)", s->max_rope_seq);
        bool flash = false, flash_tried = false;
        if (s->attn_fast && kr_pfm_gqa_flash_ok(L.nh, L.nkv, L.hd)) {      // tolerance mode: flash attention on the matrix cores (same prep launch)
            kr_launch_pfm_gqa_prep(a, Cc, st);
            kr_pf_wait(st, cx.sy.wait_b); kr_pf_rec(st, cx.sy.rec_b);      // own rows appended; the previous chunks' rows are read from here on
            flash = 0 == kr_launch_pfm_gqa_flash(a, Cc, st);
            flash_tried = true;
        }
        if (!flash) {
            if (flash_tried) return kr_fail(KR_ERR_HIP, "KR_ATTN_FAST: the flash-attention launch was refused (LDS window of head_dim %d)", L.hd);
            if (!cx.scores) return kr_fail(KR_ERR_STATE, "internal: no score scratch for the exact attention passes");
            const int sc_ld = (pos0 + Cc + 63) & ~63;
            float* scp = cx.scores;   // this chunk's arena (chunks in flight on other streams have their own)
            if (kr_launch_pfm_gqa(a, Cc, scp, sc_ld, scp + (size_t)Cc * L.nh * sc_ld, st, &cx.sy)) return kr_fail(KR_ERR_VALUE, "unsupported GQA geometry for the prompt pass");
        }
        if (oc != L.nh * L.hd) return kr_fail(KR_ERR_VALUE, "o_proj cols %d != nh*hd", oc);
        pf_rows(s, B.attn, Cc, oc, oc, B, true, st);
        if (int rc = pf_gemm(s, L.o_wid, Y(B), Cc, B.hid, H, st)) return rc;
    } else if (L.attn == ATTN_MLA) {
        // MLA (decode.rs:2993-3252): batched projections on the GEMM, then the three decode launches with a token dimension.  The
        // prep launch appends every token's latent / rope rows before the attention launch reads them, so token t of the chunk
        // sees exactly the cache decode_step would have built.
        if (!L.kv_k.p) return kr_fail(KR_ERR_STATE, "set_decode_state was not called (no MLA cache for layer %zu)", li);
        if (L.mla_rope_seq < pos0 + Cc) return kr_fail(KR_ERR_VALUE, "prompt exceeds the MLA rope table (%d)", L.mla_rope_seq);
        const int nkv = s->weights[L.kva_wid]->rows, nq = L.nh * (L.nd + L.rd), oc = s->weights[L.o_wid]->cols;
        if (int rc = pf_gemm(s, L.kva_wid, X(B), Cc, B.pb, nkv, st)) return rc;
        if (L.mq_wid >= 0) {
            if (int rc = pf_gemm(s, L.mq_wid, X(B), Cc, B.pa, nq, st)) return rc;
        } else {   // LoRA query path: q_a_proj -> sequential RMSNorm -> q_b_proj (decode.rs:3036-3079)
            const int qlr = s->weights[L.mqa_wid]->rows, qc = s->weights[L.mqb_wid]->cols;
            if (qc != qlr || qlr % 128) return kr_fail(KR_ERR_VALUE, "q_b_proj cols %d != q_a_proj rows %d (multiple of 128)", qc, qlr);
            if (int rc = pf_gemm(s, L.mqa_wid, X(B), Cc, B.pc, qlr, st)) return rc;
            if (L.q_a_norm_len) kr_launch_rmsnorm_seq(B.pc, (const float*)L.q_a_norm.p, qlr, s->eps, st, Cc, qlr);
            pf_rows(s, B.pc, Cc, qlr, qlr, B, true, st);
            if (int rc = pf_gemm(s, L.mqb_wid, Y(B), Cc, B.pa, nq, st)) return rc;
        }
        KrMlaArgs a{};
        a.step = nullptr; a.pos0 = pos0; a.kv_out = B.pb; a.ld_kv = nkv; a.q_full = B.pa; a.ld_q = nq;
        a.kv_a_norm = (const float*)L.kv_a_norm.p; a.w_kc = (const float*)L.w_kc.p; a.w_vc = (const float*)L.w_vc.p;
        a.rope_cos = (const float*)L.mla_cos.p; a.rope_sin = (const float*)L.mla_sin.p;
        a.ckv_cache = L.kv_k.p; a.kpe_cache = L.kv_v.p; a.kv_fp8 = s->kv_fp8; a.q_abs = B.q; a.q_pe = B.z; a.attn_lat = B.recur; a.v_proj = B.attn;
        a.nh = L.nh; a.klr = L.klr; a.nd = L.nd; a.rd = L.rd; a.vhd = L.vhd; a.eps = s->eps; a.sm_scale = L.sm_scale; a.fast = s->attn_fast;
        if (!s->attn_fast && cx.scores) { a.pf_sc = cx.scores; a.pf_sc_ld = (pos0 + Cc + 63) & ~63; }      // exact mode: this chunk's score scratch for the matrix-core passes
        kr_pf_wait(st, cx.sy.wait_b);            // the latent / rope rows of the earlier chunks (the three MLA launches append and read in one go)
        kr_launch_mla(a, s->kv_max_seq, st, Cc);
        kr_pf_rec(st, cx.sy.rec_b);
        if (oc != L.nh * L.vhd) return kr_fail(KR_ERR_VALUE, "o_proj cols %d != nh*v_head_dim", oc);
        pf_rows(s, B.attn, Cc, oc, oc, B, true, st);
        if (int rc = pf_gemm(s, L.o_wid, Y(B), Cc, B.hid, H, st)) return rc;
    }
    // ---- post-attention norm: f32 hidden, digits (shared expert / dense MLP), bf16 copy (routed experts)
    na.mode = 0; na.add_in = B.hid; na.first = 0; na.w = (const float*)s->norms[L.post_norm]->p; na.out_bf16 = L.mlp == MLP_MOE ? B.xb : nullptr;
    const bool post_rows = s->gemm_fast && (L.mlp == MLP_DENSE || (L.mlp == MLP_MOE && L.sgu_wid >= 0));      // the shared expert / dense MLP take the f16 rows of the normalised value
    na.xf = post_rows && s->opt_norm_rows ? B.xf : nullptr; na.xfm = post_rows && s->opt_norm_rows ? B.xfm : nullptr;
    kr_launch_pfm_norm(na, Cc, st);
    if (post_rows && !s->opt_norm_rows) kr_launch_pfh_rows_f32(B.normed, Cc, H, H, B.xf, B.xfm, st);
    if (L.mlp == MLP_MOE) {
        Layer& EL = e->layers[L.moe_layer];
        if (!EL.routing_present) return kr_fail(KR_ERR_STATE, "Routing weights not set for layer %d", L.moe_layer);
        if (!EL.w13.allocated() && !EL.gguf) return kr_fail(KR_ERR_STATE, "Model not loaded (MoE layer %d has no experts)", L.moe_layer);   // native GGUF layers: kr_moe_prefill_set walks the block kernels
        const int E = e->r_ne;
        const float* rbias = EL.has_bias ? (const float*)EL.bias.p : nullptr;
        // KR_GEMM_FAST: the router's logits in the tolerance form too (bf16 MFMA on x = hi + lo; its input already carries the mode's f16 operand rounding)
        if (!(s->gemm_fast && Cc >= 32 && EL.gate_row.p && 0 == kr_launch_route_logits_fast(EL.gate_row.p, EL.gate_bf16_exact, B.normed, rbias, B.logits, Cc, E, H, st)))
        if (!(Cc >= 32 && EL.gate_row.p && 0 == kr_launch_route_logits_mfma(EL.gate_row.p, EL.gate_bf16_exact, B.normed, rbias, B.logits, Cc, E, H, st)))
            kr_launch_route_logits_decode(EL.gate_cm.p, EL.gate_bf16_exact, B.normed, rbias, B.logits, Cc, E, H, st);
        kr_launch_route_select(B.logits, EL.has_esc ? (const float*)EL.esc.p : nullptr, B.ids, B.w, Cc, E, k, s->scoring, s->norm_topk, KR_ROUTE_RULE_DECODE, 0, st);
        // routed experts: exact CPU-engine arithmetic on the matrix cores, f32 weighted sum in routing order
        // expert parallelism (kr_ep_init on the engine): this rank's chunk exchanges its (token, slot) rows with the owners over RCCL.  A collective:
        // prefill_impl pads ranks that have fewer chunks with empty-shard calls.  Every chunk in flight has its own exchange-buffer set (cx.set) and stream; the
        // collectives of all chunks are ISSUED in one host order that is the same on every rank (the loop structure below depends only on the agreed chunk count).
        if (e->ep) { if (int rc = kr_moe_prefill_ep_set(e, L.moe_layer, B.xb, B.ids, B.w, B.moe, Cc, k, KR_OUT_F32, 1, cx.set, st ? (void*)st : (void*)1)) return rc; }
        else if (int rc = kr_moe_prefill_set(e, L.moe_layer, B.xb, B.ids, B.w, B.moe, Cc, k, KR_OUT_F32, 1, cx.set | (s->gemm_fast ? KR_PF_SET_FAST : 0), st)) return rc;
        const bool has_shared = L.sgu_wid >= 0, has_gate = has_shared && L.sg_wid >= 0;
        if (has_shared) {   // decode-store numerics: f32 input digits, fast_silu_mul + f32::round digits (decode.rs:3356-3378)
            const int si2 = s->weights[L.sgu_wid]->rows, SI = si2 / 2;
            if (SI % 128) return kr_fail(KR_ERR_VALUE, "shared expert intermediate %d not a multiple of 128", SI);
            { const int wids[2] = {L.sgu_wid, L.sg_wid}; float* outs[2] = {B.sgu, B.gv}; const int lds[2] = {si2, 1};       // gate_up (| the 1-column gate)
              if (int rc = pf_gemm_multi(s, wids, outs, lds, has_gate ? 2 : 1, X(B), Cc, st)) return rc; }
            pf_act(s, B.sgu, Cc, SI, si2, B, st);
            if (int rc = pf_gemm(s, L.sd_wid, Y(B), Cc, B.sh, H, st)) return rc;
        }
        kr_launch_pfm_moe_epilogue(B.moe, has_shared ? B.sh : nullptr, has_gate ? B.gv : nullptr, 1, s->rsf, B.hid, Cc, H, st);
    } else if (L.mlp == MLP_DENSE) {
        const int K = s->weights[L.down_wid]->cols, ng = s->weights[L.gate_wid]->rows, nu = s->weights[L.up_wid]->rows;
        if (K % 128) return kr_fail(KR_ERR_VALUE, "dense MLP intermediate %d not a multiple of 128", K);
        KR_HIP(hipMemsetAsync(B.sgu, 0, (size_t)Cc * 2 * K * 4, st));   // padding of gate | up stays 0 (decode.rs dense path)
        {   // gate -> [0,K), up -> [K,2K) of each row
            DWeight& Wg = *s->weights[L.gate_wid]; DWeight& Wu = *s->weights[L.up_wid];
            if (int rc = kr_ensure_wsum(e, Wg.ms, st)) return rc;
            if (int rc = kr_ensure_wsum(e, Wu.ms, st)) return rc;
            if (int rc = pf_gemm(s, L.gate_wid, X(B), Cc, B.sgu, 2 * K, st)) return rc;
            if (int rc = pf_gemm(s, L.up_wid, X(B), Cc, B.sgu + K, 2 * K, st)) return rc;
            (void)ng; (void)nu;
        }
        pf_act(s, B.sgu, Cc, K, 2 * K, B, st);
        if (int rc = pf_gemm(s, L.down_wid, Y(B), Cc, B.hid, H, st)) return rc;
    } else {
        KR_HIP(hipMemcpyAsync(B.hid, B.normed, (size_t)Cc * H * 4, hipMemcpyDeviceToDevice, st));   // no MLP: hidden stays the normalised value
    }
    return KR_OK;
}

// final norm + lm_head + greedy sample for the LAST token only (the other positions' logits are never consumed)
static void run_final(kr_decode_store* s, Chunk& cx) {
    Scratch& B = cx.B; const int H = s->hidden;
    KrPfmNormArgs na{};
    na.mode = cx.add_is_emb ? 1 : 0; na.add_in = B.hid; na.emb = (const float*)s->embedding.p; na.tokens = cx.tok; na.res = B.res;
    na.w = (const float*)s->norms[s->final_norm]->p; na.out = B.normed; na.H = H; na.first = cx.first ? 1 : 0; na.bias_one = s->norm_bias_one; na.eps = s->eps;
    kr_launch_pfm_norm(na, cx.Cc, cx.st);
    kr_launch_matvec(mv(s, s->lm_head), B.normed + (size_t)(cx.Cc - 1) * H, 1, (float*)s->logits.p, cx.st);
    kr_launch_argmax((const float*)s->logits.p, s->vocab, (int*)s->tok.p, (float*)s->argmax_scratch.p, cx.st);
}

// scoring mode (kr_decode_prefill_nll): final norm + lm_head GEMM for EVERY token of the chunk, then the next-token negative log-likelihood
// per row; the last chunk also leaves the last row's logits and greedy sample where run_final would
static int run_final_all(kr_decode_store* s, Chunk& cx, float* vlogits, int first_tok, int n_tokens, bool last) {
    Scratch& B = cx.B; const int H = s->hidden; const size_t V = (size_t)s->vocab;
    KrPfmNormArgs na{};
    na.mode = cx.add_is_emb ? 1 : 0; na.add_in = B.hid; na.emb = (const float*)s->embedding.p; na.tokens = cx.tok; na.res = B.res;
    na.w = (const float*)s->norms[s->final_norm]->p; na.out = B.normed; na.xh = B.xh; na.xl = B.xl; na.xs = B.xs; na.H = H; na.first = cx.first ? 1 : 0;
    na.bias_one = s->norm_bias_one; na.eps = s->eps;
    if (s->gemm_fast) { na.xh = nullptr; na.xl = nullptr; na.xs = nullptr; }
    if (s->gemm_fast && s->opt_norm_rows) { na.xf = B.xf; na.xfm = B.xfm; }
    kr_launch_pfm_norm(na, cx.Cc, cx.st);
    if (s->gemm_fast && !s->opt_norm_rows) kr_launch_pfh_rows_f32(B.normed, cx.Cc, H, H, B.xf, B.xfm, cx.st);
    if (int rc = pf_gemm(s, s->lm_head, X(B), cx.Cc, vlogits, (int)V, cx.st)) return rc;
    const int scored = std::min(cx.Cc, n_tokens - 1 - first_tok);      // the last prompt token has no label
    kr_launch_pfm_nll(vlogits, V, cx.tok + 1, (float*)s->pf_nll.p + first_tok, scored, (int)V, cx.st);
    if (last) {
        KR_HIP(hipMemcpyAsync(s->logits.p, vlogits + (size_t)(cx.Cc - 1) * V, V * 4, hipMemcpyDeviceToDevice, cx.st));
        kr_launch_argmax((const float*)s->logits.p, s->vocab, (int*)s->tok.p, (float*)s->argmax_scratch.p, cx.st);
    }
    return KR_OK;
}

static int prefill_impl(kr_decode_store* s, const int32_t* tokens, int n_tokens, int start_pos, float* logits_out, float* nll_out, void* stream) {
    if (!s) return kr_fail(KR_ERR_VALUE, "null decode store");
    if (!s->configured) return kr_fail(KR_ERR_STATE, "Call configure_decode first");
    if (!tokens || n_tokens <= 0) return kr_fail(KR_ERR_VALUE, "kr_decode_prefill: empty prompt");
    if ((int)s->layers.size() != s->n_layers) return kr_fail(KR_ERR_STATE, "finalize_decode was not called");
    if (start_pos < 0 || (s->kv_max_seq > 0 && start_pos + n_tokens > s->kv_max_seq))
        return kr_fail(KR_ERR_VALUE, "prompt [%d, %d) does not fit kv_max_seq %d", start_pos, start_pos + n_tokens, s->kv_max_seq);
    for (int i = 0; i < n_tokens; i++) if (tokens[i] < 0 || tokens[i] >= s->vocab) return kr_fail(KR_ERR_VALUE, "token id %d out of range (vocab %d)", tokens[i], s->vocab);
    kr_engine* e = s->eng;
    KR_HIP(hipSetDevice(e->device));
    hipStream_t st = kr_pick_stream(e, stream);
    const int H = s->hidden;
    if (H % 128) return kr_fail(KR_ERR_VALUE, "kr_decode_prefill needs hidden %% 128 == 0");
    // defaults: 1024 tokens x 3 chunks in flight for the exact pass (its serial per-token kernels need the overlap); with both tolerance bits set the
    // kernels are chunk-parallel and fatter chunks feed the expert GEMMs better (rows per expert grow with the chunk): 4096 x 2 measured best at
    // 8192 tokens (tools/probes/prefill_sweep.py: 1024 x 3 171 ms, 2048 x 2 168, 4096 x 2 164, 8192 x 1 193)
    // (second sweep: three chunks in flight beat two once the prompt has three chunks to give -- 8192 tokens: 2752 x 3 148.6 ms vs 4096 x 2 151.4;
    // 20 434 tokens: 4096 x 3 407 ms vs 4096 x 2 423 -- so: a third of the prompt per chunk, between 1024 and 4096 tokens, three in flight; 3000 tokens: 1024 x 3 56.8 ms vs 2048 + 952 60.9)
    const int third = ((n_tokens + 2) / 3 + 63) / 64 * 64;
    const int tol_chunk = std::min(4096, std::max(KR_PFM_CHUNK, third));
    // (round 6: the same rule with KR_ATTN_FAST alone -- its attention / delta-rule kernels are chunk-parallel too and the exact expert GEMM likes the fatter chunks:
    // 8192 tokens 173.8 -> 162.0 ms, 20 434: 448.5 -> 428.1, 49 863: 1221.6 -> 1185.4; the exact pass keeps 1024 x 3: its per-token recurrences need the overlap)
    const int depth = s->pf_depth >= 1 && s->pf_depth <= KR_PF_MAX_DEPTH ? s->pf_depth : KR_PFM_DEPTH;   // chunks in flight (streams / arenas)
    // exact pass: chunks of ~1024 tokens, their COUNT a multiple of the depth when the prompt has that many -- the last round of chunks then fills every stream
    // (8192 tokens: 8 x 1024 = 3 + 3 + 2 chunks 382.9 ms, 6 x 1366 355.8; 20 434: 20 x 1024 1207.9, 18 x 1136 1180.1; 49 863: unchanged, attention-bound)
    const int q1k = (n_tokens + KR_PFM_CHUNK - 1) / KR_PFM_CHUNK, nc_exact = q1k >= depth ? (q1k / depth) * depth : q1k;
    const int exact_chunk = (n_tokens + nc_exact - 1) / nc_exact;
    const int CH = std::min(n_tokens, s->pf_chunk > 0 ? s->pf_chunk : (s->attn_fast ? tol_chunk : exact_chunk));
    // expert parallelism: every rank must walk the SAME (chunk, layer) schedule -- the exchanges are collectives -- so the schedule is built from the chunk
    // count of the longest prompt shard (agreed below, before the first exchange); chunks a rank does not have run as empty shards
    int n_chunks_max = (n_tokens + CH - 1) / CH;
    if (e->ep) if (int rc = kr_ep_max_int(e, n_chunks_max, &n_chunks_max, st ? (void*)st : (void*)1)) return rc;
    const int n_chunks = (n_tokens + CH - 1) / CH, n_arenas = std::min(e->ep ? n_chunks_max : n_chunks, depth), D = n_arenas;
    const int L = (int)s->layers.size();

    // ---- geometry of the widest layer -> scratch sizes (floats per token); nibble sums of every weight the GEMMs will touch
    size_t pa = H, pb = 64, pc = 64, qd = 64, kd = 64, vd = 64, zd = 64, nvmax = 1, ad = H, sid = 0, kmax = H, sc_rows = 0;
    std::vector<int> wids;
    for (auto& Ly : s->layers) {
        if (Ly.attn == ATTN_LA) {
            pa = std::max(pa, (size_t)s->weights[Ly.qkvz_wid]->rows); pb = std::max(pb, (size_t)s->weights[Ly.ba_wid]->rows);
            qd = std::max(qd, (size_t)Ly.nv * Ly.dk); kd = std::max(kd, (size_t)Ly.nv * Ly.dk); vd = std::max(vd, (size_t)Ly.nv * Ly.dv); zd = std::max(zd, (size_t)Ly.nv * Ly.dv);
            nvmax = std::max(nvmax, (size_t)Ly.nv); ad = std::max(ad, (size_t)s->weights[Ly.out_wid]->cols);
            for (int w : {Ly.qkvz_wid, Ly.ba_wid, Ly.out_wid}) wids.push_back(w);
        } else if (Ly.attn == ATTN_GQA) {
            pa = std::max(pa, (size_t)s->weights[Ly.q_wid]->rows); pb = std::max(pb, (size_t)s->weights[Ly.k_wid]->rows); pc = std::max(pc, (size_t)s->weights[Ly.v_wid]->rows);
            qd = std::max(qd, (size_t)Ly.nh * Ly.hd); zd = std::max(zd, (size_t)Ly.nh * Ly.hd); ad = std::max(ad, (size_t)s->weights[Ly.o_wid]->cols);
            // the score scratch (rows x context f32, per arena) serves the EXACT attention passes only: under KR_ATTN_FAST the flash kernel has none.  Sizing it
            // regardless made a 49 863-token tolerance pass allocate 3 x 13.5 GB it never touched (16.7 k instead of ~45 k tok/s)
            if (!(s->attn_fast && kr_pfm_gqa_flash_ok(Ly.nh, Ly.nkv, Ly.hd))) sc_rows = std::max(sc_rows, (size_t)Ly.nh);
            for (int w : {Ly.q_wid, Ly.k_wid, Ly.v_wid, Ly.o_wid}) wids.push_back(w);
        } else if (Ly.attn == ATTN_MLA) {
            const size_t nq = (size_t)Ly.nh * (Ly.nd + Ly.rd);
            pa = std::max(pa, nq); pb = std::max(pb, (size_t)s->weights[Ly.kva_wid]->rows);
            qd = std::max(qd, (size_t)Ly.nh * Ly.klr); vd = std::max(vd, (size_t)Ly.nh * Ly.klr); zd = std::max(zd, (size_t)Ly.nh * Ly.rd);
            ad = std::max(ad, (size_t)s->weights[Ly.o_wid]->cols);
            if (!s->attn_fast && kr_mla_exact_mfma_ok(Ly.nh, Ly.klr, Ly.rd)) sc_rows = std::max(sc_rows, (size_t)Ly.nh);     // score scratch of the exact matrix-core passes
            wids.push_back(Ly.kva_wid); wids.push_back(Ly.o_wid);
            if (Ly.mq_wid >= 0) wids.push_back(Ly.mq_wid);
            else { pc = std::max(pc, (size_t)s->weights[Ly.mqa_wid]->rows); kmax = std::max(kmax, (size_t)s->weights[Ly.mqa_wid]->rows); wids.push_back(Ly.mqa_wid); wids.push_back(Ly.mqb_wid); }
        }
        if (Ly.mlp == MLP_MOE) {
            if (Ly.sgu_wid >= 0) { sid = std::max(sid, (size_t)s->weights[Ly.sgu_wid]->rows); kmax = std::max(kmax, (size_t)s->weights[Ly.sd_wid]->cols); wids.push_back(Ly.sgu_wid); wids.push_back(Ly.sd_wid); }
            if (Ly.sgu_wid >= 0 && Ly.sg_wid >= 0) wids.push_back(Ly.sg_wid);
            if (s->own_eng || Ly.moe_layer >= (int)e->layers.size()) return kr_fail(KR_ERR_STATE, "set_moe_store was not called (MoE layer %d has no engine)", Ly.moe_layer);
            Layer& EL = e->layers[Ly.moe_layer];
            if (int rc = kr_ensure_gate_row(e, Ly.moe_layer)) return rc;
            if (int rc = kr_ensure_wsum(e, EL.w13, st)) return rc;
            if (int rc = kr_ensure_wsum(e, EL.w2, st)) return rc;
            if (int rc = kr_moe_prefill_prepare(e, Ly.moe_layer, s->gemm_fast ? 1 : 0, 1, st)) return rc;     // native-GGUF layers: block sums / tolerance copies, before any chunk stream runs
        } else if (Ly.mlp == MLP_DENSE) {
            sid = std::max(sid, 2 * (size_t)s->weights[Ly.down_wid]->cols); kmax = std::max(kmax, (size_t)s->weights[Ly.down_wid]->cols);
            for (int w : {Ly.gate_wid, Ly.up_wid, Ly.down_wid}) wids.push_back(w);
        }
    }
    if (nll_out) wids.push_back(s->lm_head);
    for (int w : wids) {
        DWeight& W = *s->weights[w];
        if (int rc = kr_ensure_wsum(e, W.ms, st)) return rc;
    }
    kmax = std::max(kmax, ad);
    const size_t C = CH;
    if (s->gemm_fast && C * std::max(kmax, (size_t)H) * 2 >= (1ull << 32)) return kr_fail(KR_ERR_VALUE, "KR_GEMM_FAST: chunk of %d tokens x %zu values exceeds 4 GiB of f16 activations; lower the chunk", CH, std::max(kmax, (size_t)H));
    size_t total = 0, lac_floats = 0;
    if (s->attn_fast)
        for (auto& Ly : s->layers) if (Ly.attn == ATTN_LA && kr_pfm_la_chunk_ok(Ly.dk, Ly.dv, CH)) lac_floats = std::max(lac_floats, kr_pfm_la_chunk_scratch_floats(CH, Ly.nv));
    auto take = [&](size_t bytes) { const size_t o = total; total += al(bytes); return o; };
    const size_t o_res = take(C * H * 4), o_hid = take(C * H * 4), o_nrm = take(C * H * 4), o_pa = take(C * pa * 4), o_pb = take(C * pb * 4), o_pc = take(C * pc * 4),
                 o_q = take(C * qd * 4), o_k = take(C * kd * 4), o_v = take(C * vd * 4), o_z = take(C * zd * 4), o_ge = take(C * nvmax * 4), o_be = take(C * nvmax * 4),
                 o_rec = take(C * vd * 4), o_att = take(C * ad * 4), o_gate = take(C * zd * 4), o_moe = take(C * H * 4), o_sh = take(C * H * 4), o_gv = take(C * 4),
                 o_sgu = take(C * std::max(sid, (size_t)64) * 4), o_xh = take(C * H), o_xl = take(C * H), o_xs = take(C * (H / 128) * 4), o_yh = take(C * kmax),
                 o_yl = take(C * kmax), o_ys = take(C * (kmax / 128 + 1) * 4), o_xb = take(C * H * 2), o_ids = take(C * 32 * 4), o_w = take(C * 32 * 4),
                 o_lg = take(C * (size_t)std::max(e->r_ne, 64) * 4), o_lac = take(lac_floats * 4),
                 o_xf = take(C * H * 2), o_xfm = take(C * 4), o_yf = take(C * kmax * 2), o_yfm = take(C * 4);
    const size_t sc_ld_max = (size_t)((start_pos + n_tokens + 63) & ~63), sc_bytes = al(C * sc_rows * (sc_ld_max + 1 + sc_ld_max / 32) * 4);     // scores + 1 / sum per row + row maxima per 32 positions
    if (s->pf_scratch.ensure(total * n_arenas) || s->pf_tokens.ensure((size_t)n_tokens * 4) || (sc_rows && s->pf_scores.ensure(sc_bytes * n_arenas)))
        return kr_fail(KR_ERR_HIP, "hipMalloc of prefill scratch (%zu MiB) failed", (total * n_arenas + sc_bytes * n_arenas) >> 20);
    const size_t vl_bytes = al(C * (size_t)s->vocab * 4);                 // scoring mode: logits of a whole chunk, one buffer per arena
    if (nll_out && (s->pf_vlogits.ensure(vl_bytes * n_arenas) || s->pf_nll.ensure((size_t)n_tokens * 4)))
        return kr_fail(KR_ERR_HIP, "hipMalloc of the all-position logits (%zu MiB) failed", (vl_bytes * n_arenas) >> 20);
    KR_HIP(hipMemcpyAsync(s->pf_tokens.p, tokens, (size_t)n_tokens * 4, hipMemcpyHostToDevice, st));
    auto carve = [&](int arena) {
        char* base = (char*)s->pf_scratch.p + (size_t)arena * total;
        Scratch B{};
        B.res = (float*)(base + o_res); B.hid = (float*)(base + o_hid); B.normed = (float*)(base + o_nrm); B.pa = (float*)(base + o_pa); B.pb = (float*)(base + o_pb);
        B.pc = (float*)(base + o_pc); B.q = (float*)(base + o_q); B.k = (float*)(base + o_k); B.v = (float*)(base + o_v); B.z = (float*)(base + o_z);
        B.gexp = (float*)(base + o_ge); B.beta = (float*)(base + o_be); B.recur = (float*)(base + o_rec); B.attn = (float*)(base + o_att); B.gate = (float*)(base + o_gate);
        B.moe = (float*)(base + o_moe); B.sh = (float*)(base + o_sh); B.gv = (float*)(base + o_gv); B.sgu = (float*)(base + o_sgu); B.xh = (int8_t*)(base + o_xh);
        B.xl = (int8_t*)(base + o_xl); B.xs = (float*)(base + o_xs); B.yh = (int8_t*)(base + o_yh); B.yl = (int8_t*)(base + o_yl); B.ys = (float*)(base + o_ys);
        B.xb = (uint16_t*)(base + o_xb); B.ids = (int32_t*)(base + o_ids); B.w = (float*)(base + o_w); B.logits = (float*)(base + o_lg);
        B.lac = lac_floats ? (float*)(base + o_lac) : nullptr;
        B.xf = (uint16_t*)(base + o_xf); B.xfm = (float*)(base + o_xfm); B.yf = (uint16_t*)(base + o_yf); B.yfm = (float*)(base + o_yfm);
        return B;
    };

    // ---- streams: chunk c runs on stream c % D with arena c % D.  Cell (chunk c, layer l) needs (c, l-1) [same stream] and, from (c-1, l) on another
    // stream, exactly two things: the carried conv slots before its conv launch, and the recurrent state before its scan (linear attention) or the
    // appended KV / latent rows before its attention launch -> two events per (arena, layer), waited for AT THOSE LAUNCHES (KrPfSync), not at the top of the
    // layer.  (Round 3 recorded one event at the END of a layer and waited for it at the start of the next chunk's layer: the kernel trace showed the state
    // scan of a chunk running alone for 62 % of its time -- 8.5 % of the wall clock -- with both neighbours parked behind whole-layer waits, and two kernels
    // in flight 78 % of the time at a depth of three.)  The serial, low-occupancy kernels of one chunk (delta-rule scan, softmax sums, router top-k) overlap
    // with the GEMMs and attention passes of its neighbours.
    hipStream_t streams[KR_PF_MAX_DEPTH];
    for (auto& x : streams) x = st;
    const size_t ev_start = (size_t)2 * D * L, ev_end = ev_start + 1;     // [arena][layer][a | b], + one start event, + one end event per side stream
    if (D > 1) {
        while ((int)s->pf_side.size() < D - 1) { hipStream_t ns; KR_HIP(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking)); s->pf_side.push_back(ns); }
        for (int i = 1; i < D; i++) streams[i] = s->pf_side[i - 1];
        while (s->pf_events.size() < ev_end + D) { hipEvent_t ev; KR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); s->pf_events.push_back(ev); }
        KR_HIP(hipEventRecord(s->pf_events[ev_start], st));               // side streams start after everything already queued on the main one
        for (int i = 1; i < D; i++) KR_HIP(hipStreamWaitEvent(streams[i], s->pf_events[ev_start], 0));
    }
    const auto t_enqueue = std::chrono::steady_clock::now();
    std::vector<Chunk> chunks(n_chunks);
    for (int c = 0; c < n_chunks; c++) {
        Chunk& cx = chunks[c];
        cx.set = c % D; cx.B = carve(cx.set); cx.scores = sc_rows ? (float*)((char*)s->pf_scores.p + (size_t)cx.set * sc_bytes) : nullptr;
        cx.Cc = std::min(CH, n_tokens - c * CH); cx.pos0 = start_pos + c * CH; cx.tok = (const int*)s->pf_tokens.p + (size_t)c * CH;
        cx.first = true; cx.add_is_emb = true; cx.st = streams[cx.set];
    }
    // D chunks in flight (one per stream / arena): groups of D chunks are enqueued layer-interleaved; chunk c + D follows chunk c on the same
    // stream, so its arena is free, and it waits layer by layer for chunk c + D - 1 -- the pipeline never drains between groups
    const int n_sched = e->ep ? n_chunks_max : n_chunks;      // expert parallelism: the schedule of the longest shard; this rank's missing chunks are phantoms
    for (int p0 = 0; p0 < n_sched; p0 += D) {
        const int nb = std::min(D, n_sched - p0);
        for (int d = 0; d < L + nb - 1; d++) {
            for (int j = 0; j < nb; j++) {
                const int l = d - j, c = p0 + j;
                if (l < 0 || l >= L) continue;
                if (c >= n_chunks) {
                    // phantom chunk: kr_moe_prefill_ep is a collective, one call per (chunk, MoE layer) on every rank IN THE SAME ORDER.  A rank whose prompt
                    // shard has fewer chunks answers with an empty shard at the position the chunk would have had (its experts still serve the peers' rows).
                    const DLayer& Ly = s->layers[(size_t)l];
                    if (Ly.mlp == MLP_MOE) {
                        hipStream_t ps = streams[c % D];
                        if (int rc = kr_moe_prefill_ep_set(e, Ly.moe_layer, nullptr, nullptr, nullptr, nullptr, 0, s->topk, KR_OUT_F32, 1, c % D, ps ? (void*)ps : (void*)1)) return rc;
                    }
                    continue;
                }
                Chunk& cx = chunks[c];
                cx.sy = KrPfSync{};
                if (D > 1) {
                    hipEvent_t* own = &s->pf_events[((size_t)(c % D) * L + l) * 2];
                    cx.sy.rec_a = own[0]; cx.sy.rec_b = own[1];
                    if (c > 0) { hipEvent_t* prev = &s->pf_events[((size_t)((c - 1) % D) * L + l) * 2]; cx.sy.wait_a = prev[0]; cx.sy.wait_b = prev[1]; }
                }
                if (int rc = run_layer(s, cx, (size_t)l)) return rc;
                if (nll_out && l == L - 1)
                    if (int rc = run_final_all(s, cx, (float*)((char*)s->pf_vlogits.p + (size_t)cx.set * vl_bytes), c * CH, n_tokens, c == n_chunks - 1)) return rc;
            }
        }
    }
    Chunk& last = chunks[n_chunks - 1];
    if (!nll_out) run_final(s, last);
    for (int i = 1; i < D; i++) {                                          // results become visible on the caller's stream
        KR_HIP(hipEventRecord(s->pf_events[ev_end + i], streams[i]));
        KR_HIP(hipStreamWaitEvent(st, s->pf_events[ev_end + i], 0));
    }
    KR_HIP(hipGetLastError());
    s->last_stream = st;
    if (s->opt_pfm_timing) {      // tuning aid: how long the host took to enqueue the pass vs how long the GPU needs to drain it
        const double enq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enqueue).count();
        (void)hipStreamSynchronize(st);
        const double tot = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enqueue).count();
        fprintf(stderr, "[kr_decode_prefill] %d tokens: host enqueue %.1f ms, GPU drained after %.1f ms\n", n_tokens, enq, tot);
    }
    if (logits_out) {
        if (is_device_ptr(logits_out)) KR_HIP(hipMemcpyAsync(logits_out, s->logits.p, (size_t)s->vocab * 4, hipMemcpyDeviceToDevice, st));
        else { KR_HIP(hipMemcpyAsync(logits_out, s->logits.p, (size_t)s->vocab * 4, hipMemcpyDeviceToHost, st)); KR_HIP(hipStreamSynchronize(st)); }
    }
    if (nll_out && n_tokens > 1) {
        if (is_device_ptr(nll_out)) KR_HIP(hipMemcpyAsync(nll_out, s->pf_nll.p, (size_t)(n_tokens - 1) * 4, hipMemcpyDeviceToDevice, st));
        else { KR_HIP(hipMemcpyAsync(nll_out, s->pf_nll.p, (size_t)(n_tokens - 1) * 4, hipMemcpyDeviceToHost, st)); KR_HIP(hipStreamSynchronize(st)); }
    }
    return KR_OK;
}

extern "C" int kr_decode_prefill(kr_decode_store* s, const int32_t* tokens, int n_tokens, int start_pos, float* logits_out, void* stream) {
    return prefill_impl(s, tokens, n_tokens, start_pos, logits_out, nullptr, stream);
}
// prompt pass that also scores the prompt: nll_out[i] = -log softmax(logits at position i)[tokens[i + 1]], i in [0, n_tokens - 1)
extern "C" int kr_decode_prefill_nll(kr_decode_store* s, const int32_t* tokens, int n_tokens, int start_pos, float* nll_out, float* logits_out, void* stream) {
    if (!nll_out) return kr_fail(KR_ERR_VALUE, "kr_decode_prefill_nll: null nll_out");
    if (n_tokens < 2) return kr_fail(KR_ERR_VALUE, "Need at least 2 tokens, got %d", n_tokens);
    return prefill_impl(s, tokens, n_tokens, start_pos, logits_out, nll_out, stream);
}

// tuning hook: chunks in flight (1..8, 0 = default 3)
extern "C" int kr_decode_set_prefill_depth(kr_decode_store* s, int depth) {
    if (!s) return kr_fail(KR_ERR_VALUE, "null decode store");
    if (depth < 0 || depth > KR_PF_MAX_DEPTH) return kr_fail(KR_ERR_VALUE, "prefill depth %d out of range [0, %d]", depth, KR_PF_MAX_DEPTH);
    s->pf_depth = depth;
    return KR_OK;
}

// test / tuning hook: tokens per chunk of the prompt pass (0 = default 1024)
extern "C" int kr_decode_set_prefill_chunk(kr_decode_store* s, int chunk) {
    if (!s) return kr_fail(KR_ERR_VALUE, "null decode store");
    if (chunk < 0 || chunk > 8192) return kr_fail(KR_ERR_VALUE, "prefill chunk %d out of range [0, 8192]", chunk);
    s->pf_chunk = chunk;
    return KR_OK;
}
