/*
 * krasis_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's (brontoguana/krasis v0.1.62) Rust
 * algorithms for the quantized-MoE hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (krasis_amd/, libkrasis_hip.so) never links or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout, e.g. src/kernel/avx2.rs:1066).  Where the reference is AVX2
 * code, the restatement emulates the vector lanes ("virtual lanes") so that the
 * floating-point evaluation order -- and therefore every output bit -- is the same.
 *
 * Parity pinning: see oracle/README.md.  The Rust crate cannot be built here (no
 * rustc/cargo), so the oracle is pinned against the reference's own synthetic unit
 * tests (tests/test_oracle_kat.py) -- not against binaries of the reference.
 */
#ifndef KRASIS_ORACLE_H
#define KRASIS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* GGML type ids, src/gguf.rs:15-31 */
enum {
    KRO_F32 = 0, KRO_F16 = 1, KRO_Q4_0 = 2, KRO_Q5_0 = 6, KRO_Q8_0 = 8,
    KRO_Q4_K = 12, KRO_Q5_K = 13, KRO_Q6_K = 14, KRO_BF16 = 30
};

/* sigmoid flavour used inside silu*up (+quant) kernels */
enum {
    KRO_SIG_POLY5_DIV = 0, /* degree-5 poly exp (fma Horner) + IEEE divide; the form the HIP kernels implement */
    KRO_SIG_POLY5_RCPNR = 1, /* degree-5 poly exp + rcpps + 1 Newton step (avx2.rs:2277-2291); host-CPU dependent */
    KRO_SIG_LIBM = 2,      /* 1/(1+expf(-x)) */
    KRO_SIG_POLY5_SCALAR = 3 /* scalar twin fast_sigmoid, non-fused Horner + divide (moe.rs:1216-1227) */
};

/* ---- scalar conversions ---- */
float    kro_bf16_to_f32(uint16_t v);              /* marlin.rs:19 */
uint16_t kro_f32_to_bf16(float v);                 /* marlin.rs:25 (RNE, no NaN special-case) */
float    kro_f16_to_f32(uint16_t v);               /* half::f16::to_f32 (exact) */
uint16_t kro_f32_to_f16(float v);                  /* VCVTPS2PH RNE, decode.rs:4464-4478 */

/* ---- A: codecs / weight quantizers / layouts ---- */
size_t kro_ggml_block_size(int dtype);             /* gguf.rs:56 */
size_t kro_ggml_block_bytes(int dtype);            /* gguf.rs:68 */
void   kro_get_scale_min_k4(int j, const uint8_t* scales, uint8_t* sc, uint8_t* mn); /* gguf.rs:666 */
int    kro_dequantize(int dtype, const uint8_t* data, size_t n, float* out);         /* gguf.rs:872 */

void kro_quantize_int4(const uint16_t* w_bf16, int rows, int cols, int gs,
                       uint32_t* packed /*[rows,cols/8]*/, uint16_t* scales /*[rows,cols/gs]*/); /* marlin.rs:145 */
void kro_quantize_int8(const uint16_t* w_bf16, int rows, int cols, int gs,
                       int8_t* data /*[rows,cols]*/, uint16_t* scales);                         /* marlin.rs:65 */
void kro_dequantize_int4(const uint32_t* packed, const uint16_t* scales, int rows, int cols, int gs, float* out); /* marlin.rs:210 */
void kro_dequantize_int8(const int8_t* data, const uint16_t* scales, int rows, int cols, int gs, float* out);     /* marlin.rs:117 */
/* decode.rs:46 / :115 -- f32 weights -> transposed [K/8,N] (int4) or [K,N] (int8) + scales [K/gs,N] */
void kro_quantize_f32_to_transposed_int4(const float* w, int rows, int cols, int gs, uint32_t* packed_t, uint16_t* scales_t);
void kro_quantize_f32_to_transposed_int8(const float* w, int rows, int cols, int gs, int8_t* data_t, uint16_t* scales_t);
/* weights/mod.rs:329 / avx2.rs:2195 -- [N,K/8] -> [K/8,N] */
void kro_transpose_int4(const uint32_t* packed, const uint16_t* scales, int n, int k, int gs, uint32_t* packed_t, uint16_t* scales_t);
void kro_transpose_int8(const int8_t* data, const uint16_t* scales, int n, int k, int gs, int8_t* data_t, uint16_t* scales_t);
/* avx2.rs:1318 / :1340 -- [K/8,N] -> [N/256,K/8,256] zero padded; returns elements written */
size_t kro_repack_tiled_u32(const uint32_t* src, int k_rows, int n, uint32_t* dst);
size_t kro_repack_tiled_u16(const uint16_t* src, int k_rows, int n, uint16_t* dst);

/* ---- B: activation quantizers ---- */
void kro_quant_act_int16_bf16(const uint16_t* x, int k, int gs, int16_t* q, float* scales); /* avx2.rs:234 */
void kro_quant_act_int16_f32(const float* x, int k, int gs, int16_t* q, float* scales);     /* avx2.rs:274 */
void kro_gguf_quant_bf16(const uint16_t* x, int k, int16_t* q, float* scales, int32_t* sums); /* gguf_kernels.rs:110 */
void kro_gguf_quant_f32(const float* x, int k, int16_t* q, float* scales, int32_t* sums);     /* gguf_kernels.rs:143 */
float kro_sigmoid(float x, int mode);
/* avx2.rs:2310: gate_up = [gate(n) | up(n)]; hidden_f32 (optional out) = silu(gate)*up */
void kro_silu_quant_int16(const float* gate, const float* up, int n, int gs, int sig_mode,
                          float* hidden_f32, int16_t* q, float* scales);
void kro_fast_silu_mul(const float* gate, const float* up, int n, int sig_mode, float* out); /* decode.rs:1693 */

/* ---- C: matvecs (M=1) ---- */
/* avx2.rs:1066 (+wrappers :1211): packed [K/8,N], scales [K/gs,N] */
void kro_matvec_int4_t(const uint32_t* packed, const uint16_t* scales, const int16_t* a, const float* a_s,
                       int k, int n, int gs, float* out);
/* avx2.rs:1477: data [K,N] i8 */
void kro_matvec_int8_t(const int8_t* data, const uint16_t* scales, const int16_t* a, const float* a_s,
                       int k, int n, int gs, float* out);
/* avx2.rs:309 row-major scalar integer reference ([N,K/8]) */
void kro_matvec_int4_rowmajor(const uint32_t* packed, const uint16_t* scales, const int16_t* a, const float* a_s,
                              int k, int n, int gs, float* out);
/* gguf_kernels.rs:271/379/436 (int16 act, AVX2 lane order) ; weights [N, K/blk] raw blocks */
void kro_gguf_matvec_int(int dtype, const uint8_t* w, const int16_t* a, const float* a_s, const int32_t* a_sum,
                         int n, int k, float* out);
/* Q4_K / Q8_0 rows of kro_gguf_matvec_int through AVX2 intrinsics with the output rows split over the OpenMP team (bit-identical results;
 * bench.py's config-1 CPU leg).  Off by default: the scalar lane-by-lane form is the oracle the tests read. */
void kro_gguf_set_avx2(int on);
/* gguf_kernels.rs:495-635 scalar f32 activation fallbacks */
void kro_gguf_matvec_f32(int dtype, const uint8_t* w, const float* x, int n, int k, float* out);

/* ---- D/E: expert + MoE forward (single token) ---- */
typedef struct {
    /* unified transposed expert, weights/mod.rs:287-322 */
    const void* w13;            /* int4: u32 [K/8,2N]; int8: i8 [K,2N] */
    const uint16_t* w13_scales; /* [K/gs,2N] */
    const void* w2;             /* int4: u32 [N/8,K]; int8: i8 [N,K] */
    const uint16_t* w2_scales;  /* [N/gs,K] */
    int hidden, inter, gs, num_bits, w2_bits;
    const float* gate_bias; const float* up_bias; const float* down_bias; /* optional (GPT-OSS) */
} kro_unified_expert;

typedef struct {
    /* raw GGUF expert, weights/mod.rs:252-266 */
    const uint8_t* gate; const uint8_t* up; const uint8_t* down;
    int gate_up_type, down_type, hidden, inter;
} kro_gguf_expert;

/* moe.rs:184 ; act already INT16-quantized per gs; out f32[hidden]; scratch sized by callee */
void kro_expert_forward_unified(const kro_unified_expert* e, const int16_t* a, const float* a_s,
                                float swiglu_limit, float alpha, int sig_mode, float* out);
/* gguf_kernels.rs:690 */
void kro_expert_forward_gguf(const kro_gguf_expert* e, const uint16_t* act_bf16, float* out);
/* moe.rs:572 ; experts[i] already resolved for ids; shared may be NULL */
void kro_moe_forward_unified(const kro_unified_expert* const* experts, const float* weights, int n_sel,
                             const kro_unified_expert* shared, float rsf,
                             const uint16_t* act_bf16, float swiglu_limit, float alpha, int sig_mode, float* out);
/* moe.rs:990 */
void kro_moe_forward_gguf(const kro_gguf_expert* const* experts, const float* weights, int n_sel,
                          const kro_gguf_expert* shared, float rsf, const uint16_t* act_bf16, float* out);

/* ---- F: routers ---- */
/* decode.rs:1495 */
void kro_topk_indices(const float* values, int n, int k, int32_t* out);
/* decode.rs:1385 */
void kro_route_matmul(const float* gate, const float* hidden, int ne, int hd, float* logits);
/* decode.rs:4088 ; scoring 0 sigmoid(poly4) 1 softmax 2 topk-then-softmax; logits modified like the reference */
void kro_route_score_topk(float* logits, int ne, const float* e_score_corr, int scoring, int norm_topk, int topk,
                          float* scores, int32_t* ids, float* w);
/* moe.rs:3081-3246 engine router: bf16 gate x bf16 act, sequential f32; scoring "sigmoid"(1)/softmax(0);
 * swiglu_limit>0 selects the GPT-OSS branch */
void kro_route_engine(const uint16_t* gate_bf16, const uint16_t* act_bf16, int ne, int hd, const float* corr_bias,
                      int sigmoid, int norm_topk, int topk, float swiglu_limit, int32_t* ids, float* w);
/* python/krasis/layer.py:526-560 prefill routing semantics on f32 logits (torch.topk tie rule: see oracle README) */

/* ---- G: decode-graph ops ---- */
void kro_fused_add_rmsnorm(float* hidden, float* residual, const float* w, int n, float eps, int first, int bias_one); /* decode.rs:1199 */
void kro_la_conv(const float* qkvz, const float* ba, float* conv_state, const float* conv_w, const float* a_log,
                 const float* dt_bias, float scale, float* q, float* k, float* v, float* z, float* g, float* beta,
                 int nk, int nv, int dk, int dv, int kernel_dim, int sig_mode);                                       /* decode.rs:3815 */
void kro_la_recurrent(float* state, const float* q, const float* k, const float* v, const float* g, const float* beta,
                      float* out, int nv, int dk, int dv);                                                            /* decode.rs:1293 */
void kro_gated_rmsnorm_silu(const float* recur, const float* z, const float* w, float* out, int nv, int dv, float eps,
                            int sig_mode);                                                                            /* decode.rs:3979 */
/* stand-alone CpuDecodeStore operators (decode.rs:473-890): scalar loops + libm exp, not the decode graph's AVX2 forms */
void kro_op_rmsnorm(const float* x, const float* w, float* out, int n, float eps, int bias_one);
void kro_op_silu_mul(const float* gate, const float* up, float* out, int n);
void kro_op_gated_rmsnorm_silu(const float* x, const float* z, const float* w, float* out, float eps, int nv, int dv);
void kro_op_la_conv(const float* qkvz, const float* ba, float* conv_state, const float* conv_w, const float* a_log, const float* dt_bias, float scale,
                    float* q, float* k, float* v, float* z, float* g, float* beta, int nk, int nv, int dk, int dv, int hr, int kd);
/* decode.rs:2873-2975 + :4194 ; q_in = q_proj output ([nh*hd] or [nh*2*hd] if gated); writes k/v fp16 at position */
void kro_gqa_step(const float* q_in, float* k, float* v, const float* q_norm, int q_norm_len, const float* k_norm,
                  int k_norm_len, int gated, int nh, int nkv, int hd, float eps, const float* rope_cos,
                  const float* rope_sin, int rope_half, uint16_t* k_cache, uint16_t* v_cache, int max_seq, int position,
                  float sm_scale, float* attn_out);
uint8_t kro_f32_to_e4m3(float f);   /* torch.float8_e4m3fn conversion (kv_cache.py:38-135 cache dtype) */
float   kro_e4m3_to_f32(uint8_t x);
void    kro_set_kv_fp8(int on);     /* kro_gqa_step cache element type: 0 FP16 (default), 1 FP8-E4M3 (one byte per uint16 slot) */
/* G5 MLA (decode.rs:2993-3252 driver; :4286 dot, :4326 weighted sum, :4508 absorb, :4555 w_vc) */
void kro_rmsnorm_seq(float* x, const float* w, int n, float eps);                                                     /* decode.rs:3023-3032 */
void kro_mla_step(float* kv_out, float* q_full, const float* kv_a_norm, const float* w_kc, const float* w_vc,
                  const float* rope_cos, const float* rope_sin, int nh, int klr, int nd, int rd, int vhd, float eps, float sm_scale,
                  uint16_t* ckv_cache, uint16_t* kpe_cache, int position, float* v_projected);
int  kro_sample_greedy(const float* logits, int n);
uint64_t kro_xorshift64_next(uint64_t* st);                                                                          /* decode.rs:3556-3561 */
int  kro_sample_from_logits(float* logits, int vocab, float temperature, int top_k, float top_p, uint64_t* rng_state); /* decode.rs:3718 */                                                                   /* decode.rs:3718 */

/* ---- H: GPU prefill semantics (third-party sglang fused_marlin_moe 0.5.9; parity unpinned) ---- */
/* python/krasis/gpu_prefill.py:64-239 dataflow: bf16 act x dequant(INT4/8) weights, fp32 accumulate,
 * bf16 intermediate after each stage, topk weight on GEMM2, sum over k, *rsf */
void kro_moe_prefill_bf16(const kro_unified_expert* const* experts /*[E]*/, int n_experts,
                          const uint16_t* x_bf16, const int32_t* ids, const float* w, int m, int topk, float rsf,
                          uint16_t* out_bf16);

/* ---- J ---- */
void kro_reduce_sum_bf16(const uint16_t* const* inputs, int n_inputs, size_t n, uint16_t* out); /* moe.rs:2505 */

/* ---- synthetic data generator, decode.rs:4356-4411 ---- */
typedef struct { uint64_t state; } kro_xorshift64;
uint64_t kro_xs_next(kro_xorshift64* r);
uint32_t kro_xs_next_u32(kro_xorshift64* r);
void kro_xs_fill_u32(kro_xorshift64* r, uint32_t* dst, size_t n);
void kro_xs_fill_bf16_scales(kro_xorshift64* r, uint16_t* dst, size_t n);
void kro_xs_fill_f32_uniform(kro_xorshift64* r, float* dst, size_t n, float amp); /* (u32/2^32*2-1)*amp */

/* ---- CPU baseline (AVX2 + OpenMP twin of kro_matvec_int4_t on the tiled layout; bit-identical) ---- */
void kro_matvec_int4_tiled_avx2(const uint32_t* packed_tiled, const uint16_t* scales_tiled, const int16_t* a,
                                const float* a_s, int k, int n, int gs, float* out, int parallel);
/* experts' w13/w2 pointers must reference TILED arrays ([N/256][K/8][256] u32, scales [N/256][K/gs][256]) */
void kro_moe_forward_unified_tiled_avx2(const kro_unified_expert* const* experts_tiled, const float* weights, int n_sel,
                                        const uint16_t* act_bf16, int sig_mode, float* out);
int  kro_num_threads(void);
void kro_set_num_threads(int n);   /* OpenMP team size of the AVX2 twins (rayon pool size in the reference, numa.rs:420) */

#ifdef __cplusplus
}
#endif
#endif
