// kr_kernels.h -- host-visible launch wrappers for the gfx950 kernels (implemented in *.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// One quantized [K -> N] matrix in the lane-tiled HBM layout:
//   INT4-g128: q = [N/8 tiles][ngp group-pairs][64 lanes] x 16 B, lane = col*8 + l holds packed words
//              {W(g0,2l), W(g0,2l+1), W(g1,2l), W(g1,2l+1)}, W(g,i) = nibbles k = g*128+8i..+8 of the column
//   INT8-g128: q = [N/8 tiles][ng groups][64 lanes] x 16 B, lane holds k = g*128+16l..+16 of the column
//   scales   : s = [N/8 tiles][ngp][8 cols] u32 = bf16(even group) | bf16(odd group) << 16
struct KrMatDev {
    const void* q;
    const uint32_t* s;
    int K, N;        // logical dims (N = outputs)
    int ng, ngp;     // groups of 128, group pairs (ceil)
    int bits;        // 4 or 8
    int n_fma;       // columns < n_fma accumulate with fma (avx2.rs:1175), the tail with mul+add (avx2.rs:1201)
    size_t q_stride; // bytes between consecutive experts (0 for a single matrix)
    size_t s_stride;
    // tolerance-GEMM copy of a native Q4_K matrix (kr_prefill_h.hip, G = 1): the nibbles n of the super-blocks in the INT4 layout above and, per
    // (column, 32-wide sub-block j), f16 tables  qs = (d * sc_j) / 4  and  qo = 16 * (8 * d * sc_j - dmin * mn_j)  so that  w = d sc_j (n - 8) + qo / 16:
    // [N/8 tiles][K/256 blocks][8 cols][8 sub-blocks] (16 B per (column, block)); null for INT4 / INT8-g128 matrices
    const uint16_t* qs; const uint16_t* qo; size_t qs_stride;
};

static inline size_t kr_mat_q_bytes(int K, int N, int bits) {
    const int ng = (K + 127) / 128, ngp = (ng + 1) / 2, nt = (N + 7) / 8;
    return (size_t)nt * (bits == 4 ? ngp : ng) * 64 * 16;
}
static inline size_t kr_mat_s_bytes(int K, int N) {
    const int ng = (K + 127) / 128, ngp = (ng + 1) / 2, nt = (N + 7) / 8;
    return (size_t)nt * ngp * 8 * 4;
}

// bytes of the INT16 activation image of a K-vector for INT4 weights (== kr_lds_bytes(K, false) in kr_device.h)
static inline size_t kr_act_image_bytes(int K) {
    const int tail_words = K / 16 + ((K / 128 + 3) & ~3);
    return (size_t)(K / 8) * 16 + (size_t)((tail_words + 3) / 4) * 16;
}

// activation transform feeding the down projection
enum { KR_ACT_SILU_FUSED = 0,  // silu_quantize_int16_avx2 (avx2.rs:2310): poly sigmoid, RNE quant
       KR_ACT_GPTOSS = 1,      // moe.rs:268-287: clamp, gate*sigmoid(alpha*gate)*(up+1), round-half-away quant
       KR_ACT_SILU_MUL = 2 };  // fast_silu_mul_avx2 + quantize_activation_int16_f32 (decode.rs:3364-3374)

struct KrMoeArgs {
    const uint16_t* act;   // bf16 [B,H]
    const void* act_img;      // optional pre-built INT16 image of the f32 activation (B == 1, INT4 weights): shared slot with decode numerics
    const void* act_img_bf16; // optional pre-built image of bf16(activation): routed slots
    const float* act_f32;  // decode graph: f32 hidden [B,H]; routed experts see bf16(hidden) (decode.rs:3307), when set `act` is unused
    int shared_decode;     // shared slot follows the decode-store numerics: f32 input quant, fast_silu_mul + f32::round quant (decode.rs:3356-3378)
    const int32_t* ids;    // [B,topk]
    const float* wts;      // [B,topk]
    int B, topk, n_slots;  // n_slots = topk (+1 when the shared expert runs in the same launches)
    int E;                 // routed experts of the layer: ids outside [0, E) are skipped like -1 (never an out-of-bounds read)
    int e_lo, e_hi, e_sub; // expert-parallel decode (e_hi > 0): only ids in [e_lo, e_hi) are computed here, as local expert id - e_sub; the other slots' rows stay 0 for the all-reduce
    int H, I, I_shared;
    KrMatDev w13, w2;      // routed experts of the layer: expert e at q + e*q_stride
    KrMatDev sw13, sw2;    // shared expert (valid when n_slots > topk)
    float* gu;             // scratch [B][n_slots][gu_ld]   (gate | up)
    float* eo;             // scratch [B][n_slots][H]
    int gu_ld;
    void* out;             // [B,H] f32 or bf16
    int out_bf16;
    float rsf, swiglu_limit, alpha;
    int act_mode;
    KrMatDev sgate;        // optional [H -> 1] sigmoid-gate row of the shared expert, evaluated by the shared slot's w13 launch
    float* gate_out;       // [B]
};

#define KR_MAX_MULTI 4
struct KrMultiMat { KrMatDev m[KR_MAX_MULTI]; float* y[KR_MAX_MULTI]; int tile_end[KR_MAX_MULTI]; int n; };

void kr_launch_moe_decode(const KrMoeArgs& a, hipStream_t st);
void kr_launch_moe_w13(const KrMoeArgs& a, hipStream_t st);
void kr_launch_moe_w2(const KrMoeArgs& a, hipStream_t st);
void kr_launch_moe_combine(const KrMoeArgs& a, hipStream_t st);
int kr_launch_moe_w2c(const KrMoeArgs& a, const float* gate_val, float* out, hipStream_t st);      // decode step: stage 2 + the routing-order combine in one launch

// generic single-matrix matvec: y[N] = W . quant(x[K]); x f32 or bf16; used for projections / lm_head
// act_mode < 0: x[K] is quantized as is; otherwise x = [gate(K) | up(K)] and the kernel applies KR_ACT_* first (dense MLP)
void kr_launch_matvec(const KrMatDev& m, const void* x, int x_is_f32, float* y, hipStream_t st, int act_mode = -1);
// several matrices sharing one input vector (same K, same bits) in ONE launch
void kr_launch_multi_matvec(const KrMatDev* mats, float* const* ys, int n, const void* x, int x_is_f32, hipStream_t st, int act_mode = -1);
// linear-attention in-projection with the conv1d + SiLU + conv-state shift and the gates as its epilogue (exact decode step; kr_moe_decode.hip)
struct KrCoLa {
    float* conv_state; const float* conv_w; float *qk_out, *v_out, *z_out;
    int nk, dk, hr, dv, conv_mi;
};
int kr_launch_multi_matvec_la(const KrMatDev* mats, float* const* ys, int n, const void* x_img, const KrCoLa& la, hipStream_t st);

void kr_launch_fill_synth(void* q, size_t q_bytes, uint32_t* s, size_t s_words, uint64_t seed, hipStream_t st);
void kr_launch_fill_uniform_f32(float* x, size_t n, float amp, uint64_t seed, hipStream_t st);
void kr_launch_fill_fp16_kv(uint16_t* x, size_t n, uint64_t seed, hipStream_t st);
void kr_launch_fill_e4m3_kv(uint8_t* x, size_t n, uint64_t seed, hipStream_t st);
void kr_launch_reduce_sum_bf16(const uint16_t* const* dev_ptr_table, int n_inputs, uint16_t* out, size_t n, hipStream_t st);

// marlin.rs:65,145 quantizers on the GPU: bf16 W[rows][K] (HF row-major) -> lane-tiled records of column tiles [tile0, tile0 + rows/8)
void kr_launch_quant_bf16(const uint16_t* w_dev, int rows, int K, int bits, void* qdst, uint32_t* sdst, int tile0, hipStream_t st);
