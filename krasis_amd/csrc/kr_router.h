// kr_router.h -- launch wrappers of kr_router.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
void kr_launch_route_logits_decode(const void* gate_cm, int gate_bf16, const float* x, const float* bias, float* logits,
                                   int m, int E, int H, hipStream_t st);
// prompt pass (m >= 32 tokens): the same logits, bit for bit, on the f32 MFMA (kr_route_mfma.hip); gate_row = row-major [E][H] bf16 or f32.
// non-zero = geometry not covered
int kr_launch_route_logits_fast(const void* gate_row, int gate_bf16, const float* x, const float* bias, float* logits, int T, int E, int H, hipStream_t st);   // tolerance form (bf16 MFMA, x split hi + lo); non-zero = not covered
int kr_launch_route_logits_mfma(const void* gate_row, int gate_bf16, const float* x, const float* bias, float* logits, int T, int E, int H, hipStream_t st);
int kr_launch_mla_wvc_mfma(const float* w_vc, const float* attn_lat, float* v_proj, int T, int nh, int vhd, int klr, hipStream_t st);   // the same chains: MLA w_vc projection of a chunk
int kr_launch_mla_absorb_mfma(const float* q_full, int ld_q, int hd, int nd, const float* w_kc, int klr, float* q_abs, int T, int nh, hipStream_t st);   // MLA w_kc absorption of a chunk (one fma chain per output)
void kr_launch_route_logits_engine(const void* gate_rm, const uint16_t* act, float* logits, int m, int E, int H, hipStream_t st);
void kr_launch_route_select(const float* logits, const float* esc, int32_t* ids, float* w, int m, int E, int topk, int scoring,
                            int norm, int rule, int gptoss, hipStream_t st);
/* decode graph, M = 1: [fused add+RMSNorm ->] gate GEMV -> scoring + top-k in one launch (norm_w == nullptr: x is already normalised).
 * returns non-zero when the geometry is not covered (caller falls back to the separate launches) */
int kr_launch_route_fused_decode(const void* gate_cm, int gate_bf16, const float* bias, float* logits, unsigned* counter, const float* esc,
                                 int32_t* ids, float* w, int E, int H, int topk, int scoring, int norm_topk, const float* x,
                                 const float* hid_in, const float* res_in, const float* norm_w, float* hid_out, float* res_out, float eps,
                                 int bias_one, hipStream_t st, void* img_f32 = nullptr, void* img_bf16 = nullptr);
