// kr_standalone.h -- launch wrappers of kr_standalone.hip (the reference's stand-alone CpuDecodeStore operators, decode.rs:328-1086)
#pragma once
#include <hip/hip_runtime.h>
struct KrOpLaConvArgs {
    const float *qkvz, *ba; float* conv_state; const float *conv_w, *a_log, *dt_bias; float scale;
    float *q, *k, *v, *z, *g, *beta; float* conv_out;   // conv_out: [2*nk*dk + nv*dv] scratch
    int nk, nv, dk, dv, hr, kernel_dim;
};
void kr_launch_op_rmsnorm(const float* x, const float* w, float* out, int n, float eps, int bias_one, hipStream_t st);
void kr_launch_op_silu_mul(const float* gate, const float* up, float* out, int n, hipStream_t st);
void kr_launch_op_gated_rmsnorm_silu(const float* x, const float* z, const float* w, float* out, int nv, int dv, float eps, hipStream_t st);
void kr_launch_op_la_conv(const KrOpLaConvArgs& a, hipStream_t st);
void kr_launch_op_exp(const float* g, float* out, int n, hipStream_t st);
