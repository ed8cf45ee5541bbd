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
    GgMat gate, up, down;        // routed experts: expert e at q + e*q_stride
    GgMat sgate, sup, sdown;     // shared expert (n_slots > topk)
    float* gu; float* eo; int gu_ld;
};
void kr_launch_gguf_moe(const GgMoeArgs& a, hipStream_t st);
size_t gg_q_bytes(int type, int K, int N);
size_t gg_h_bytes(int type, int K, int N);
