// kr_prefill.h -- launch wrappers of kr_prefill.hip (token sort, activation digits, int8-MFMA grouped GEMM, combine)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kr_kernels.h"

// dense GEMMs: shape of the (row tile x column block) super-tile one XCD works on at a time (see kr_prefill_h.hip): up to 8 x 8, shrunk until there
// are at least 3 super-tiles per XCD to balance
static inline void kr_pf_super_tile(int nrt, int ncb, int* sr, int* sc, int* n_super) {
    int r = nrt < 8 ? nrt : 8, c = ncb < 8 ? ncb : 8;
    auto count = [&]() { return ((nrt + r - 1) / r) * ((ncb + c - 1) / c); };
    while (count() < 24 && (r > 1 || c > 1)) { if (r >= c && r > 1) r = (r + 1) / 2; else c = (c + 1) / 2; }
    *sr = r; *sc = c; *n_super = count();
}

struct KrPfSort {
    int* counts; int* offsets; int* cursor;        // [E]
    int* tile_expert; int* tile_row0; int* tile_rows;  // [max_tiles]
    int* n_tiles;                                   // [1]
    int* row_pair;                                  // [n_pairs]  GEMM row -> (token*topk + slot)
    int* pair_row;                                  // [n_pairs]  inverse (-1 for skipped ids)
};
void kr_launch_pf_sort(const int32_t* ids, int M, int topk, int E, KrPfSort s, hipStream_t st, int bm = 64 /* rows per tile of the tile table */);
void kr_launch_pf_quant_x(const uint16_t* x, int M, int K, int8_t* xh, int8_t* xl, float* xs, hipStream_t st);
void kr_launch_pf_act(const float* gu, int rows, int n, int gu_ld, int act_mode, float swiglu_limit, float alpha, int8_t* hh, int8_t* hl, float* hs, hipStream_t st);
void kr_launch_pf_wsum(const KrMatDev& m, int n_experts, uint32_t* wsum, hipStream_t st);
void kr_launch_pf_gemm(const KrMatDev& m, const uint32_t* wsum, const int8_t* a_hi, const int8_t* a_lo, const float* a_scale, const KrPfSort* sort, int topk,
                       int gather_tokens, int max_tiles, int single_expert_rows, float* out, int out_ld, hipStream_t st, int scatter_rows = 0, int out_bf16 = 0,
                       int var_rows = 0 /* expert tiles mostly <= 32 rows: kernel variant that skips empty 32-row blocks */);
void kr_launch_pf_gemm_multi(const KrMatDev* mats, const uint32_t* const* wsums, float* const* outs, const int* out_lds, int n, const int8_t* a_hi, const int8_t* a_lo,
                             const float* a_scale, int M, hipStream_t st);
void kr_launch_pf_combine(const float* eo, const int* pair_row, const float* wts, int M, int topk, int H, const float* shared_eo, float rsf, void* out,
                          int out_bf16, hipStream_t st);
void kr_launch_pf_combine_f16rows(const uint16_t* eo, const float* row_mul, const int* pair_row, const float* wts, int M, int topk, int H, const float* shared_eo, float rsf,
                                  void* out, int out_bf16, hipStream_t st);
void kr_launch_pf_combine_bf16rows(const uint16_t* eo, const int* pair_row, const float* wts, int M, int topk, int H, const float* shared_eo, float rsf, void* out,
                                   int out_bf16, hipStream_t st);
// expert parallelism helpers (kr_ep.cpp)
void kr_launch_ep_sort(const int32_t* dest, int n, int W, KrPfSort s, hipStream_t st);   // owner sort, world <= 64 destinations
void kr_launch_ep_dest(const int32_t* ids, int n, int E_total, int per, int world, int full, int32_t* dest, int32_t* lid, hipStream_t st);
void kr_launch_ep_gather(const uint16_t* x, const int* row_pair, const int32_t* lid, int topk, int H, const int* n_rows, int max_rows, uint16_t* rows, int32_t* row_lid,
                         hipStream_t st);
void kr_launch_ep_rows_bf16(const float* in, uint16_t* out, size_t n, hipStream_t st);
void kr_launch_ep_sum_f32(const float* parts, int W, size_t n, float* out, hipStream_t st);

// FAST (tolerance) form, kr_prefill_h.hip: f16 rows with a power-of-two row multiplier x weights de-quantized in registers, f32 accumulation
void kr_launch_pfh_rows_f32(const float* x, int rows, int ld, int K, uint16_t* out, float* mul, hipStream_t st);
void kr_launch_pfh_rows_bf16(const uint16_t* x, int rows, int ld, int K, uint16_t* out, float* mul, hipStream_t st, uint16_t* sums32 = nullptr);   // sums32: f16 sums per 32 values (Q4_K copy)
void kr_launch_pfh_act(const float* gu, int rows, int n, int gu_ld, int act_mode, float swiglu_limit, float alpha, uint16_t* out, float* mul, hipStream_t st,
                       uint16_t* sums32 = nullptr);     // act_mode 3 = libm SiLU of expert_forward_gguf (rows of at most 2048 values)
void kr_launch_pfh_gemm(const KrMatDev& m, const uint16_t* a_h, const float* a_mul, const KrPfSort* sort, int topk, int gather_tokens, int max_tiles,
                        int single_expert_rows, float* out, int out_ld, hipStream_t st, int scatter_rows = 0, int out_bf16 = 0, int run = 1,
                        const uint16_t* a_sum32 = nullptr);     // a_sum32: required when m.qs is set (Q4_K copy)
void kr_launch_pfh_w13_act(const KrMatDev& m, const uint16_t* a_h, const float* a_mul, const KrPfSort* sort, int topk, int gather_tokens, int max_tiles,
                           int single_expert_rows, float* gu, int rows, int act_mode, float swiglu_limit, float alpha, uint16_t* h_out, float* h_mul,
                           hipStream_t st, int run = 1, const uint16_t* a_sum32 = nullptr, uint16_t* h_sums32 = nullptr);
void kr_launch_pfh_gemm_multi(const KrMatDev* mats, float* const* outs, const int* out_lds, int n, const uint16_t* a_h, const float* a_mul, int M, hipStream_t st);
// the LDS-ring form of the INT4 tolerance GEMM (kr_prefill_ring.hip) is on by default; 0 keeps the register-staged kernels (process-wide A/B and test hook)
void kr_pfr_set_enabled(int on);
