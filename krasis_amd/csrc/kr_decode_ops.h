// kr_decode_ops.h -- launch wrappers of kr_decode_ops.hip (decode-graph operators)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct KrStep { int token; int pos; };  // lives in device memory so a captured graph can be replayed per token

struct KrLaArgs {
    const float* qkvz; const float* ba; float* conv_state; const float* conv_w; const float* a_log; const float* dt_bias;
    float scale; float *q, *k, *v, *z, *g, *beta; int nk, nv, dk, dv, hr;
};
struct KrGqaArgs {
    const KrStep* step;
    const float *q_in, *k_in, *v_in;
    const float *q_norm, *k_norm; int q_norm_per_head, k_norm_per_head;
    const float *rope_cos, *rope_sin; int rope_half;
    void *k_cache, *v_cache; int kv_fp8;   // FP16 (reference CPU decode) or FP8-E4M3 (reference GPU cache dtype) elements
    float *q_out, *gate, *attn_out;
    int gated, nh, nkv, hd; float eps, sm_scale;
    void* img_out;   // optional: INT16 image of attn_out for the o-projection launch (hd % 128 == 0)
    float* sc_g;     // long caches: [nh][max_seq] score scratch -- the scores are computed by (nh x max_seq/256) workgroups in their own launch
    float *fd_o, *fd_ml;   // fast (tolerance) mode: split-KV partials [nkv][chunks][G][hd] and (max, sum) [nh][chunks][2]; null = exact order
    int tree_norm;         // KR_DECODE_FAST: the QK-norm sums of squares as workgroup trees instead of the reference's element-order chain (decode.rs:2893)
    int force_stream;      // test hook (kr_decode_set_option "gqa_stream"): take the HBM-streamed score row even when it would fit LDS
};

// FAST decode attention over long caches, second generation (kr_attn_flash.hip): split-KV flash-decode on the f16 MFMA + log-sum-exp merge
struct KrFdFlashArgs {
    const KrStep* step; const float* q;          // q [nh][hd] after QK-norm / RoPE (kr_gqa_prep_kernel)
    const void *k_cache, *v_cache;               // [max_seq][nkv * hd], FP16 or E4M3 elements
    float *fd_o, *fd_ml;                         // partials [nkv][chunks][G][hd], (max in log2 units, sum) [nh][chunks][2]
    int nh, nkv; float sm_scale;
    const float* gate; int gated; float* out; void* img_out;
};
int kr_fd_flash_chunk(int max_seq);
size_t kr_fd_flash_chunks(int max_seq);
int kr_fd_flash_prepare(int hd, int fp8);        // outside graph capture
int kr_launch_fd_flash(const KrFdFlashArgs& a, int hd, int fp8, int max_seq, hipStream_t st);   // non-zero = geometry not covered
int kr_gqa_attn_prepare(int max_seq, int hd, int fp8);   // 0, -1 (scores + stage exceed 160 KiB of LDS), -2 (HIP refused)

struct KrMlaArgs {   // decode.rs:2993-3252
    const KrStep* step;
    const float* kv_out;      // kv_a_proj output [klr + rd]
    const float* q_full;      // q (or q_b) projection output [nh * (nd + rd)]
    const float *kv_a_norm, *w_kc, *w_vc, *rope_cos, *rope_sin;
    void *ckv_cache, *kpe_cache;       // [max_seq, klr] / [max_seq, rd], FP16 or (kv_fp8) E4M3 elements
    int kv_fp8;
    float* sc_g;                       // decode, long caches: [nh][max_seq] score scratch (scores in their own, head-shared launch)
    float *fd_o, *fd_ml; int fast;     // FAST (tolerance) mode: split-KV partials (decode, long caches) / flash attention (prompt pass); fast = 0: exact order
    float *q_abs, *q_pe, *attn_lat, *v_proj;
    int nh, klr, nd, rd, vhd; float eps, sm_scale;
    // prompt pass (step == nullptr): token t = blockIdx.y (blockIdx.z for the w_vc launch) sits at position pos0 + t; row t of the
    // per-token buffers starts at t * ld_* floats (kv_out, q_full) or t * their natural size (q_abs, q_pe, attn_lat, v_proj)
    int pos0; int ld_kv, ld_q;
    int decode_fast;   // KR_DECODE_FAST (decode, step != nullptr): the absorption, the latent RMSNorm and the w_vc projection as tree sums (same products)
    int decode_fused;  // ... and, over a short cache, the attention launch itself as kr_fmla_kernel (kr_decode_set_option "gqa_fused" 0 keeps the exact-order launch)
    int absorb_done;   // prompt pass: q_abs of the chunk was produced by kr_launch_mla_absorb_mfma -- the prep launch skips its absorption loop
    float* pf_sc; int pf_sc_ld;   // prompt pass, exact mode: score scratch [n_tok * nh][pf_sc_ld] + n_tok * nh * (1 + pf_sc_ld / 32) floats (1 / sum, row maxima) for the matrix-core passes; null: per-token launches
};
void kr_launch_mla(const KrMlaArgs& a, int max_seq, hipStream_t s, int n_tok = 1);
size_t kr_mla_flash_decode_chunks(int max_seq);   // chunks of the FAST split-KV decode (sizes the partial buffers)
void kr_mla_attn_prepare(const KrMlaArgs& a, int max_seq);   // outside graph capture: LDS window of the staged attention kernel
void kr_launch_rmsnorm_seq(float* x, const float* w, int n, float eps, hipStream_t s, int rows = 1, int ld = 0);

void kr_launch_set_step_dev(KrStep* dst, const int* token_dev, int pos, hipStream_t s);
void kr_launch_set_step(KrStep* dst, int token, int pos, hipStream_t s);   // (token, pos) by value: safe with many steps queued
void kr_launch_embed(const float* emb, const KrStep* st, float* hidden, int H, hipStream_t s);
struct KrNormSrc {   // where the value added to the residual comes from (see kr_fused_add_rmsnorm_kernel)
    int mode;        // 0 hidden buffer, 1 embedding row of the current token, 2 MoE epilogue of the previous layer
    const float* emb; const KrStep* step;
    const float* eo; const int32_t* ids; const float* wts; int topk; int has_shared; const float* gate_val; float rsf;
};
void kr_launch_fused_add_rmsnorm(const KrNormSrc& src, float* hidden, const float* res_in, float* residual, const float* w, int n, float eps, int first, int bias_one, hipStream_t s,
                                 void* img_out = nullptr);   // img_out: optional INT16 image of the normalised hidden (n % 128 == 0)
void kr_launch_la_conv(const KrLaArgs& a, hipStream_t s);
int kr_launch_la_recurrent_gnorm(float* state, const float* q, const float* k, const float* v, const float* g, const float* beta, const float* z,
                                 const float* w, float* out, int nv, int dk, int dv, float eps, hipStream_t s, void* img_out = nullptr);
int kr_launch_la_step(const KrLaArgs& a, float* state, const float* w, float* out, float eps, hipStream_t s, void* img_out = nullptr);
int kr_launch_la_step_heads(const KrLaArgs& a, float* state, const float* w, float* out, float eps, hipStream_t s, void* img_out = nullptr);
void kr_launch_gated_rmsnorm_silu(const float* recur, const float* z, const float* w, float* out, int nv, int dv, float eps, hipStream_t s);
void kr_launch_gqa(const KrGqaArgs& a, int max_seq, hipStream_t s);   // a.sc_g != nullptr: prep, scores (many workgroups), softmax + p.v
void kr_launch_moe_combine_decode(const float* eo, const int32_t* ids, const float* wts, int topk, int has_shared, const float* gate_val,
                                  float rsf, float* hidden, int H, hipStream_t s);
void kr_launch_argmax(const float* x, int n, int* out, float* scratch, hipStream_t s);
