// kr_prefill_ops.h -- launch wrappers of kr_prefill_ops.hip (batched decode-graph operators for kr_decode_prefill)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct KrPfmNormArgs {
    int mode;                     // 0: add_in[t]; 1: embedding row of tokens[t]
    const float* add_in; const float* emb; const int* tokens;
    float* res;                   // [C,H] residual stream, updated in place
    const float* w; float* out;   // norm weight [H]; normalised hidden f32 [C,H]
    int8_t *xh, *xl; float* xs;   // optional INT16 digits of the normalised hidden (quantize_activation_int16_f32)
    uint16_t* out_bf16;           // optional bf16 copy (input of the routed experts)
    uint16_t* xf; float* xfm;     // optional f16 row image of the normalised hidden + row multipliers: the tolerance GEMMs' A operand (kr_pfh_dev.h; what kr_pfh_rows_kernel<0> makes of `out`)
    int H, first, bias_one; float eps;
};
// Cross-chunk ordering INSIDE a layer (kr_decode_prefill.cpp runs several chunks of a prompt on their own streams): what chunk c needs from chunk c - 1 is
// not "layer l finished" but two narrow hand-overs -- (a) the carried conv state, (b) the recurrent state (linear attention) or the KV / latent rows
// (GQA, MLA) of layer l.  wait_* are the previous chunk's events (null: first chunk / one chunk in flight), rec_* this chunk's.  Every record is issued AFTER
// the matching wait on the same stream, so an event implies the whole chain of earlier chunks (a chunk two back has no event of its own to wait for).
struct KrPfSync {
    hipEvent_t wait_a = nullptr, rec_a = nullptr, wait_b = nullptr, rec_b = nullptr;
};
static inline void kr_pf_wait(hipStream_t st, hipEvent_t ev) { if (ev) (void)hipStreamWaitEvent(st, ev, 0); }
static inline void kr_pf_rec(hipStream_t st, hipEvent_t ev) { if (ev) (void)hipEventRecord(ev, st); }
struct KrPfmLaArgs {
    const float* qkvz; int ld_qkvz; const float* ba; int ld_ba;
    float* conv_state; const float* conv_w; const float* a_log; const float* dt_bias; float scale;
    float *q, *k, *v, *z, *gexp, *beta;   // [C, nv*dk] x2, [C, nv*dv] x2, [C, nv] x2
    int nk, nv, dk, dv, hr;
    float* lac; int fast;                 // FAST mode: scratch of the 64-token closed form (kr_la_chunk.hip), kr_pfm_la_chunk_scratch_floats(C, nv) floats
    int conv_fused;                       // FAST mode (round 6): the conv / norms / gates inside the prep launch, z read by the gated norm where the in-projection left it (no conv launch)
};
struct KrPfmGqaArgs {
    const float *q_in, *k_in, *v_in; int ld_q, ld_k, ld_v;
    const float *q_norm, *k_norm; int q_norm_per_head, k_norm_per_head;
    const float *rope_cos, *rope_sin; int rope_half;
    void *k_cache, *v_cache; int kv_fp8;   // FP16 (reference CPU decode) or FP8-E4M3 (reference GPU cache dtype) elements
    float *q_out, *gate, *attn_out;   // [C, nh*hd]
    int gated, nh, nkv, hd, pos0; float eps, sm_scale;
};
void kr_launch_pfm_norm(const KrPfmNormArgs& a, int C, hipStream_t st);
void kr_launch_pfm_nll(const float* logits, size_t ld, const int* labels, float* nll, int rows, int V, hipStream_t st);
void kr_launch_pfm_quant_f32(const float* x, int rows, int ld, int K, int8_t* xh, int8_t* xl, float* xs, hipStream_t st);
// conv + conv-state update + gated delta rule over the chunk + gated RMSNorm; non-zero = unsupported geometry
int kr_launch_pfm_la(const KrPfmLaArgs& a, float* recur_state, float* recur_out, const float* norm_w, float* gated_out, int C, float eps, hipStream_t st, const KrPfSync* sy = nullptr);
// prep (norm, RoPE, KV append) + scores -> softmax -> P.V over the score scratch sc[C*nh rows][sc_ld] (sc_ld >= pos0 + C, multiple of 64),
// inv[C*nh * (1 + sc_ld/32)] (1 / sum per row, then the per-32-position row maxima of the matrix-core passes);
// non-zero = unsupported geometry
int kr_launch_pfm_gqa(const KrPfmGqaArgs& a, int C, float* sc, int sc_ld, float* inv, hipStream_t st, const KrPfSync* sy = nullptr);
// passes A and C of the above on the f32 matrix cores, bit-identical (kr_attn_exact_mfma.hip); *_ok: the geometry is covered (group divides 32, head_dim 64 / 128 / 256)
int kr_pfm_gqa_exact_mfma_ok(const KrPfmGqaArgs& a);
void kr_launch_pfm_gqa_scores_mfma(const KrPfmGqaArgs& a, int C, float* sc, int sc_ld, float* tmax /* [C*nh][sc_ld/32] row maxima per 32 positions, or null */, hipStream_t st);
void kr_launch_pfm_gqa_pv_mfma(const KrPfmGqaArgs& a, int C, const float* sc, int sc_ld, const float* inv /* null: sc holds probabilities; else exponentials, scaled here */, hipStream_t st);
// pass B alone (exact softmax of score rows: max -> libm exp -> sum in position order -> 1 / sum in inv[row]); tmax != null: row maxima per 32 positions are
// given and the rows stay as exponentials (the matrix-core pass C scales them)
void kr_launch_pfm_softmax_rows(float* sc, int sc_ld, float* inv, int nh, int pos0, int rows, const float* tmax, hipStream_t st);
// MLA prompt pass, exact scores on the f32 matrix cores (kr_attn_exact_mfma.hip)
int kr_mla_exact_mfma_ok(int nh, int klr, int rd);
int kr_launch_mla_scores_mfma(const float* q_abs, const float* q_pe, const void* ckv, const void* kpe, int kv_fp8, int nh, int klr, int rd, int pos0, int n_tok,
                              float sm_scale, float* sc, int sc_ld, float* tmax, hipStream_t st);
void kr_launch_pfm_moe_epilogue(const float* moe, const float* shared, const float* gate_val, int gate_ld, float rsf, float* hidden, int C, int H, hipStream_t st);
// FAST mode (kr_attn_flash.hip): causal flash attention on f16 MFMA after the same prep launch; non-zero = geometry not covered
int kr_launch_pfm_gqa_flash(const KrPfmGqaArgs& a, int C, hipStream_t st);
bool kr_pfm_gqa_flash_ok(int nh, int nkv, int hd);      // geometries the flash kernel covers (the others take the exact passes and need their score scratch)
void kr_launch_pfm_gqa_prep(const KrPfmGqaArgs& a, int C, hipStream_t st);
int kr_launch_pfm_la_recur(float* state, const float* q, const float* k, const float* v, const float* gexp, const float* beta, float* out, int nv, int dk, int dv, int C, hipStream_t st);
// FAST mode (kr_la_chunk.hip): the gated delta rule over the chunk in sub-chunks of 64 tokens on the f32 MFMA; non-zero = geometry not covered
int kr_launch_pfm_la_chunked(const KrPfmLaArgs& a, float* recur_state, float* recur_out, float* scratch, int C, hipStream_t st, const KrPfSync* sy = nullptr, int fused = 0,
                             void (*between)(const KrPfmLaArgs&, int, hipStream_t, const KrPfSync*) = nullptr);
size_t kr_pfm_la_chunk_scratch_floats(int C, int nv);
bool kr_pfm_la_chunk_ok(int dk, int dv, int C);
