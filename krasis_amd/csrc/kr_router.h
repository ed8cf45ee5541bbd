// kr_router.h -- launch wrappers of kr_router.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
void kr_launch_route_logits_decode(const void* gate_cm, int gate_bf16, const float* x, const float* bias, float* logits,
                                   int m, int E, int H, hipStream_t st);
void kr_launch_route_logits_engine(const void* gate_rm, const uint16_t* act, float* logits, int m, int E, int H, hipStream_t st);
void kr_launch_route_select(const float* logits, const float* esc, int32_t* ids, float* w, int m, int E, int topk, int scoring,
                            int norm, int rule, int gptoss, hipStream_t st);
