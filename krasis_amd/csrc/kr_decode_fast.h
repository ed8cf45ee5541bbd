// kr_decode_fast.h -- launch wrappers of kr_decode_fast.hip: the decode step in TOLERANCE mode (KR_DECODE_FAST).
//
// The default decode graph reproduces the reference's CPU decode (src/decode.rs:2690-3520) bit for bit, which pins every reduction to the
// reference's sequential order: a 256-fma norm chain, a 512-add softmax sum, 128-step state chains -- the serial path of each launch.  The
// kernels here keep the reference's PRODUCTS (INT16 activation digits, exact integer group sums, the same scale products, the same
// polynomial / libm functions) and give up only the SUMMATION ORDER (wave / workgroup trees) and the launch structure that order forced:
//   kr_launch_fdm     dequant-matvec of up to 4 matrices sharing an input; the input is a pre-built INT16 image, or the fused add+RMSNorm
//                     (decode.rs:1199) folded in (every workgroup rebuilds the normalised vector with a tree sum); optional epilogue:
//                     depthwise conv1d + SiLU + conv-state shift of the linear-attention channels (decode.rs:3815-3890) on the owning lane
//   kr_launch_fla     gated delta-rule step per VALUE head (decode.rs:1293 + 3909-3945 + 3979): state rows split over 16 slices of a workgroup
//   kr_launch_frt     post-attention add+RMSNorm + router gate GEMV (decode.rs:1385), expert images for the expert launches
//   kr_launch_fw13    scoring + top-k (decode.rs:4088, wave-parallel; ids identical to the exact kernel for identical logits unless two leaders
//                     tie, then the reference's heap order is emulated) in the prologue of the gate|up matvec; silu(g) * u in its epilogue
//   kr_launch_fw2     INT16 quantisation of the expert hidden, down matvec of all k (+ shared) experts of one column tile in ONE workgroup
//                     and the weighted combine (moe.rs:661-667, decode.rs:3343-3402) -- the MoE output lands in the hidden buffer
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kr_decode_ops.h"
#include "kr_kernels.h"
#include "kr_gguf.h"

struct KrFdmArgs {
    KrMultiMat mm;
    int mode;                 // 0: `img` is the INT16 image of the input vector; 1: input = RMSNorm(hid_in (+ res_in)) built by every workgroup; 2: input = the f32 vector hid_in as is
    const void* img;
    const float *hid_in, *res_in, *norm_w; float* res_out;
    const float* emb; const KrStep* step;   // mode 1, first layer: the added value is the embedding row of the current token
    int first; float eps; int bias_one;
    // linear-attention epilogues (null conv_state: plain stores): matrix conv_mi = in_proj_qkvz (conv1d + SiLU + state shift per channel, z copied),
    // matrix gate_mi = in_proj_ba (beta = sigmoid(b), e^g with g = -e^{A_log} softplus(a + dt_bias); decode.rs:3891-3901)
    float* conv_state; const float* conv_w; float *qk_out, *v_out, *z_out; int nk, dk, hr, dv; int conv_mi, gate_mi;
    const float *a_log, *dt_bias; float *ge_out, *beta_out;
};
int kr_launch_fdm(const KrFdmArgs& a, hipStream_t st);    // non-zero: geometry not covered (caller takes the exact kernels)

struct KrFlaArgs {
    const float *qk, *v, *z, *ge, *beta; float scale;     // ge = e^g, beta: per value head, from the projection launch
    float* state; const float* norm_w; float* out; void* img_out; int img_k;
    int nk, nv, hr, dk, dv; float eps;
};
int kr_launch_fla(const KrFlaArgs& a, hipStream_t st);

struct KrFrtArgs {
    const void* gate_cm; int gate_bf16; const float* bias; float* logits; int E, H;
    const float *hid_in, *res_in, *norm_w; float *hid_out, *res_out; float eps; int bias_one;
    void *img_f32, *img_bf16;
};
int kr_launch_frt(const KrFrtArgs& a, hipStream_t st);

struct KrFmoeArgs {
    KrMoeArgs m;              // B == 1, act_img / act_img_bf16 set, ids / wts = OUTPUT of the w13 launch (read by the w2 launch), gu = expert hidden [n_slots][gu_ld]
    const float* logits; const float* esc; int scoring, norm_topk;
    float* hid_out;           // w2 launch: combined MoE output [H]
    int gguf;                 // the ROUTED experts are native GGUF blocks (ggate / gup / gdown: Q4_K, Q8_0 or Q4_0 with K % 32 == 0): m.w13 / m.w2 are unset, the shared expert
    GgMat ggate, gup, gdown;  // (m.sw13 / m.sw2, transposed INT4 / INT8) keeps the path above.  Per-32 INT16 activations of bf16(act_f32) (gguf_kernels.rs:110-172), the
    const float* act_f32;     // block kernels' products (kr_gguf_dev.h) with a row's blocks split over two waves, libm SiLU (gguf_kernels.rs:733-737)
    int shared_skip;          // expert-parallel decode (m.e_hi > 0): the shared expert of this layer is evaluated by another rank; this rank's hid_out is then its PARTIAL
                              // rsf * sum over its own slots (+ the shared term on the one rank that evaluates it) and the ranks' partials are summed by one all-reduce of [H]
};
int kr_fmoe_check(const KrFmoeArgs& a);      // 0 when both launches below cover the geometry
int kr_launch_fw13(const KrFmoeArgs& a, hipStream_t st);
int kr_launch_fw2(const KrFmoeArgs& a, hipStream_t st);

// GQA layers over a short cache (kv_max_seq <= 1024) in KR_DECODE_FAST: prep (gated split, QK-norm, RoPE, KV append) + attention of one query head per workgroup in ONE
// launch (the exact path: kr_gqa_prep_kernel + kr_gqa_attn_kernel).  Non-zero: geometry not covered (caller takes kr_launch_gqa).
int kr_launch_fgqa(const KrGqaArgs& a, int max_seq, hipStream_t st);
// MLA layers over a short cache in KR_DECODE_FAST: scores + softmax + weighted sum of the latent rows of one head per workgroup with tree sums (the exact path:
// kr_mla_attn_staged_kernel).  Non-zero: geometry not covered (caller takes the exact-order launch).
int kr_launch_fmla(const KrMlaArgs& a, int max_seq, hipStream_t st);
