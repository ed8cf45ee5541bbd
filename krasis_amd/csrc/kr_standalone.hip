// kr_standalone.hip -- the stand-alone CpuDecodeStore operators of the reference (src/decode.rs:328-1086): the per-op entry points the Python
// decode path called before decode_step existed, still exported by the PyO3 class.  Several of them are NOT the decode graph's arithmetic:
// the graph uses the AVX2 forms (8-lane sums, polynomial sigmoid), these use plain scalar loops and libm's exp.  Each kernel below follows the
// scalar loop it cites, operation for operation, so the results are bit-identical to the reference method (the parity tests check them against the
// CPU restatement kro_op_*).  They are convenience / parity operators, launched with one workgroup where the loop is a serial sum; the hot
// path is the graph (kr_decode_ops.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kr_libm.h"
#include "kr_standalone.h"

// decode.rs:473-507 rmsnorm: sum_sq += x*x in index order (mul, then add -- the build keeps contraction off), rms = 1/sqrt(sum/n + eps),
// out = (x * rms) * (1 + w) or (x * rms) * w
__global__ void __launch_bounds__(256) kr_op_rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int n, float eps, int bias_one) {
    __shared__ float rms_s;
    if (threadIdx.x == 0) {
        float ss = 0.0f;
        for (int i = 0; i < n; i++) ss += x[i] * x[i];
        rms_s = 1.0f / sqrtf(ss / (float)n + eps);
    }
    __syncthreads();
    const float rms = rms_s;
    for (int i = threadIdx.x; i < n; i += 256) out[i] = bias_one ? x[i] * rms * (1.0f + w[i]) : x[i] * rms * w[i];
}

// decode.rs:511-538 silu_mul: sigmoid = 1/(1+exp(-x)); out = (x * sigmoid) * up
__global__ void __launch_bounds__(256) kr_op_silu_mul_kernel(const float* __restrict__ gate, const float* __restrict__ up, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = gate[i];
    const float sg = 1.0f / (1.0f + kr_expf(-x));
    out[i] = x * sg * up[i];
}

// decode.rs:650-695 gated_rmsnorm_silu: per head a sequential sum of squares over dv, normed = (x * rms) * w, silu_z = z / (1 + exp(-z)), out = silu_z * normed
__global__ void __launch_bounds__(256) kr_op_gated_rmsnorm_silu_kernel(const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ w,
                                                                      float* __restrict__ out, int dv, float eps) {
    __shared__ float rms_s;
    const int base = blockIdx.x * dv;
    if (threadIdx.x == 0) {
        float ss = 0.0f;
        for (int j = 0; j < dv; j++) ss += x[base + j] * x[base + j];
        rms_s = 1.0f / sqrtf(ss / (float)dv + eps);
    }
    __syncthreads();
    const float rms = rms_s;
    for (int j = threadIdx.x; j < dv; j += 256) {
        const float normed = x[base + j] * rms * w[base + j];
        const float zv = z[base + j];
        const float silu_z = zv / (1.0f + kr_expf(-zv));
        out[base + j] = silu_z * normed;
    }
}

// decode.rs:713-890 linear_attention_conv, any kernel_dim.  Pass 1 (grid over channels): un-interleave, shift the conv state, dot product in tap
// order starting from 0.0 (mul, add), exact SiLU; z copy; gates.  Pass 2 (one workgroup per value head): sequential L2 norms, q = (c * inv) * scale.
__global__ void __launch_bounds__(256) kr_op_la_conv1_kernel(KrOpLaConvArgs a) {
    const int nk = a.nk, nv = a.nv, dk = a.dk, dv = a.dv, hr = a.hr, kd = a.kernel_dim;
    const int key_dim = nk * dk, conv_dim = 2 * key_dim + nv * dv, group_dim = 2 * dk + 2 * dv * hr;
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch < conv_dim) {
        // channel ch of mixed_qkv = [q_flat | k_flat | v_flat] comes from the interleaved projection
        int src;
        if (ch < key_dim) { const int h = ch / dk, i = ch % dk; src = h * group_dim + i; }
        else if (ch < 2 * key_dim) { const int c = ch - key_dim, h = c / dk, i = c % dk; src = h * group_dim + dk + i; }
        else { const int c = ch - 2 * key_dim, vh = c / dv, i = c % dv, h = vh / hr, r = vh % hr; src = h * group_dim + 2 * dk + r * dv + i; }
        float* cs = a.conv_state + (size_t)ch * kd; const float* cw = a.conv_w + (size_t)ch * kd;
        for (int t = 0; t < kd - 1; t++) cs[t] = cs[t + 1];
        cs[kd - 1] = a.qkvz[src];
        float dot = 0.0f;
        for (int t = 0; t < kd; t++) dot += cs[t] * cw[t];
        const float sg = 1.0f / (1.0f + kr_expf(-dot));
        a.conv_out[ch] = dot * sg;
    }
    if (ch < nv * dv) {   // z: plain copy
        const int vh = ch / dv, i = ch % dv, h = vh / hr, r = vh % hr;
        a.z[ch] = a.qkvz[h * group_dim + 2 * dk + hr * dv + r * dv + i];
    }
    if (ch < nv) {        // gates (decode.rs:878-885)
        const int h = ch / hr, r = ch % hr;
        const float b_raw = a.ba[h * 2 * hr + r], a_p = a.ba[h * 2 * hr + hr + r];
        a.beta[ch] = 1.0f / (1.0f + kr_expf(-b_raw));
        const float ap_dt = a_p + a.dt_bias[ch];
        const float softplus = ap_dt > 20.0f ? ap_dt : kr_logf(1.0f + kr_expf(ap_dt));
        a.g[ch] = -(kr_expf(a.a_log[ch])) * softplus;
    }
}
__global__ void __launch_bounds__(256) kr_op_la_conv2_kernel(KrOpLaConvArgs a) {
    __shared__ float inv_s[2];
    const int vh = blockIdx.x, kh = vh / a.hr, dk = a.dk, dv = a.dv, key_dim = a.nk * dk;
    const float* qs = a.conv_out + kh * dk; const float* ks = a.conv_out + key_dim + kh * dk;
    if (threadIdx.x < 2) {
        const float* s = threadIdx.x ? ks : qs;
        float ss = 0.0f;
        for (int i = 0; i < dk; i++) { const float v = s[i]; ss += v * v; }
        inv_s[threadIdx.x] = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
    }
    __syncthreads();
    const float iq = inv_s[0], ik = inv_s[1];
    for (int i = threadIdx.x; i < dk; i += 256) { a.q[vh * dk + i] = qs[i] * iq * a.scale; a.k[vh * dk + i] = ks[i] * ik; }
    for (int i = threadIdx.x; i < dv; i += 256) a.v[vh * dv + i] = a.conv_out[2 * key_dim + vh * dv + i];
}

// e^g per head for the recurrence launch (decode.rs:1313: g_exp = g[h].exp())
__global__ void kr_op_exp_kernel(const float* __restrict__ g, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) out[i] = kr_expf(g[i]);
}

void kr_launch_op_rmsnorm(const float* x, const float* w, float* out, int n, float eps, int bias_one, hipStream_t st) {
    hipLaunchKernelGGL(kr_op_rmsnorm_kernel, dim3(1), dim3(256), 0, st, x, w, out, n, eps, bias_one);
}
void kr_launch_op_silu_mul(const float* gate, const float* up, float* out, int n, hipStream_t st) {
    hipLaunchKernelGGL(kr_op_silu_mul_kernel, dim3((n + 255) / 256), dim3(256), 0, st, gate, up, out, n);
}
void kr_launch_op_gated_rmsnorm_silu(const float* x, const float* z, const float* w, float* out, int nv, int dv, float eps, hipStream_t st) {
    hipLaunchKernelGGL(kr_op_gated_rmsnorm_silu_kernel, dim3(nv), dim3(256), 0, st, x, z, w, out, dv, eps);
}
void kr_launch_op_la_conv(const KrOpLaConvArgs& a, hipStream_t st) {
    const int conv_dim = 2 * a.nk * a.dk + a.nv * a.dv;
    hipLaunchKernelGGL(kr_op_la_conv1_kernel, dim3((conv_dim + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(kr_op_la_conv2_kernel, dim3(a.nv), dim3(256), 0, st, a);
}
void kr_launch_op_exp(const float* g, float* out, int n, hipStream_t st) {
    hipLaunchKernelGGL(kr_op_exp_kernel, dim3((n + 63) / 64), dim3(64), 0, st, g, out, n);
}
