// kr_gguf.h -- native GGUF block experts (launch wrappers of kr_gguf.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

enum { GG_Q4_0 = 2, GG_Q5_0 = 6, GG_Q8_0 = 8, GG_Q4_K = 12, GG_Q6_K = 14 };   // GGML type ids (gguf.rs:15-31)

struct GgMat {          // one projection [K -> N rows] of one (or all) expert(s) in the lane-tiled GGUF layout
    const void* q; const void* h; int type, K, N; size_t q_stride, h_stride;
};
struct GgMoeArgs {
    const uint16_t* act; const int32_t* ids; int B, topk, n_slots, H, I_max; int E;   // ids outside [0, E) are skipped
    const float* act_f32;        // decode graph: f32 hidden [B, H], rounded to bf16 (RNE) on the way in (decode.rs:3307-3309); when set `act` is unused
    GgMat gate, up, down;        // routed experts: expert e at q + e*q_stride
    GgMat sgate, sup, sdown;     // shared expert (n_slots > topk)
    float* gu; float* eo; int gu_ld;
};
void kr_launch_gguf_moe(const GgMoeArgs& a, hipStream_t st);
size_t gg_q_bytes(int type, int K, int N);
size_t gg_h_bytes(int type, int K, int N);

// ---- prompt pass: GGUF blocks on the int8-MFMA grouped GEMM (kr_gguf_prefill.hip) ----
struct KrPfSort;
bool kr_gpf_type_supported(int type, int K);
size_t kr_gpf_ws_bytes(int type, int K, int N);
void kr_launch_gpf_wsum(const GgMat& m, int n_experts, void* ws, size_t ws_stride, hipStream_t st);
void kr_launch_gpf_quant_x(const uint16_t* x, int M, int K, int8_t* xh, int8_t* xl, float* xs, float* xm, hipStream_t st);
void kr_launch_gpf_act(const float* gu, int rows, int n, int gu_ld, int8_t* hh, int8_t* hl, float* hs, float* hm, hipStream_t st);
void kr_launch_gpf_gemm(const GgMat& m, const void* ws, size_t ws_stride, const int8_t* a_hi, const int8_t* a_lo, const float* a_scale, const float* a_sum,
                        const KrPfSort* sort, int topk, int gather_tokens, int max_tiles, int single_expert_rows, float* out, int out_ld, int col_off,
                        hipStream_t st);
// Q4_K -> the tolerance GEMM's operand form (INT4 lane tiles + per-sub-block f16 scale / offset tables); tile_off: first output tile (gate | up share one matrix)
void kr_launch_gq_repack(const GgMat& m, int n_experts, void* q_out, size_t q_stride, void* qs_out, void* qo_out, size_t qs_stride, int tile_off, int tiles_out, hipStream_t st);
// Q8_0 -> the tolerance GEMM's INT8 operand form (INT8 lane tiles + one f16 scale per 32-wide block)
void kr_launch_gq8_repack(const GgMat& m, int n_experts, void* q_out, size_t q_stride, void* qs_out, size_t qs_stride, int tile_off, hipStream_t st);
void kr_launch_gpf_fill_synth(void* q, size_t q_bytes, void* h, size_t h_bytes, int type, uint64_t seed, hipStream_t st);
