// kr_prefill_ops.hip -- batched (M tokens) forms of the decode-graph operators, for the whole-model prompt pass kr_decode_prefill.
//
// Contract: processing tokens t = 0..C-1 of a chunk with these kernels leaves every buffer (logits, FP16 KV, conv / recurrent state)
// BIT-IDENTICAL to C successive decode steps (src/decode.rs:2690-3520), i.e. prefill == decode == oracle.  Every per-token reduction
// keeps the decode kernel's order (8 fma lanes + hsum for the norms, one sequential chain per state column for the gated delta rule,
// sequential softmax sums); tokens only add independent parallel work.  The GEMM-shaped work (projections, experts) runs on the
// int8-MFMA grouped GEMM of kr_prefill.hip with the exact INT16-digit arithmetic.
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_prefill_ops.h"
#include <hip/hip_fp16.h>

__device__ __forceinline__ float kr_pfm_hsum8(float v) { v = v + __shfl_xor(v, 4); v = v + __shfl_xor(v, 1); v = v + __shfl_xor(v, 2); return v; }

// sum of squares of x[0..n) with 8 fma lanes (lane l owns elements 8b + l, b ascending), call with lanes 0..7 of a wave
__device__ __forceinline__ float kr_pfm_sumsq8(const float* x, int n, int l) {
    float acc = 0.0f; const int nb = n / 8; int b = 0;
    for (; b + 8 <= nb; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = x[(b + u) * 8 + l];
#pragma unroll
        for (int u = 0; u < 8; u++) acc = __builtin_fmaf(v[u], v[u], acc);
    }
    for (; b < nb; b++) { const float v = x[b * 8 + l]; acc = __builtin_fmaf(v, v, acc); }
    return kr_pfm_hsum8(acc);
}

__device__ __forceinline__ void kr_pfm_store_digits(int8_t* hi, int8_t* lo, const int (&q)[8]) {
    u32x2 h, l;
    h.x = kr_pack4(q[0] >> 8, q[1] >> 8, q[2] >> 8, q[3] >> 8); h.y = kr_pack4(q[4] >> 8, q[5] >> 8, q[6] >> 8, q[7] >> 8);
    l.x = kr_pack4((q[0] & 255) - 128, (q[1] & 255) - 128, (q[2] & 255) - 128, (q[3] & 255) - 128);
    l.y = kr_pack4((q[4] & 255) - 128, (q[5] & 255) - 128, (q[6] & 255) - 128, (q[7] & 255) - 128);
    *reinterpret_cast<u32x2*>(hi) = h; *reinterpret_cast<u32x2*>(lo) = l;
}

// quantize_activation_int16_f32 (avx2.rs:274) of one 8-chunk; 16 consecutive lanes = one group of 128
__device__ __forceinline__ void kr_pfm_quant_chunk(const float (&v)[8], int8_t* hi, int8_t* lo, float* scale_out, bool write_scale) {
    float mx = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v[i]));
    mx = kr_red16_max_f32(mx);
    const float scale = mx > 0.0f ? mx / 32767.0f : 1.0f, inv = mx > 0.0f ? 32767.0f / mx : 0.0f;
    int q[8];
    kr_quant8<false>(v, inv, q);
    kr_pfm_store_digits(hi, lo, q);
    if (write_scale) *scale_out = scale;
}

// ---- fused add + RMSNorm, one workgroup per token (decode.rs:1199) ------------------------------------------------------
// value added: mode 0 = add_in[t] (attention / MLP output), mode 1 = embedding row of tokens[t]; first: residual = value.
// outputs: residual (in place), normalised hidden f32, optional INT16 digits of the f32 value, optional bf16 copy (routed experts
// quantise bf16(hidden), decode.rs:3307-3309).
__global__ void __launch_bounds__(256) kr_pfm_norm_kernel(const KrPfmNormArgs a) {
    extern __shared__ __attribute__((aligned(16))) float r[];   // [H + 4]
    const int t = blockIdx.x, H = a.H, tid = threadIdx.x;
    const float* add = a.mode == 1 ? a.emb + (size_t)a.tokens[t] * H : a.add_in + (size_t)t * H;
    float* res = a.res + (size_t)t * H;
    for (int i = tid; i < H; i += 256) { const float v = a.first ? add[i] : (add[i] + res[i]); r[i] = v; res[i] = v; }
    __syncthreads();
    if (tid < 8) {
        float ss = kr_pfm_sumsq8(r, H, tid);
        if (tid == 0) { for (int q = (H / 8) * 8; q < H; q++) ss += r[q] * r[q]; r[H] = 1.0f / sqrtf(ss / (float)H + a.eps); }
    }
    __syncthreads();
    const float rms = r[H];
    float* out = a.out + (size_t)t * H;
    for (int c = tid; c < H / 8; c += 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = c * 8 + u; v[u] = (r[i] * rms) * (a.bias_one ? (a.w[i] + 1.0f) : a.w[i]); }
        *reinterpret_cast<float4*>(out + c * 8) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(out + c * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        if (a.xh) kr_pfm_quant_chunk(v, a.xh + (size_t)t * H + c * 8, a.xl + (size_t)t * H + c * 8, a.xs + (size_t)t * (H / 128) + (c >> 4), (c & 15) == 0);
        if (a.out_bf16) {
            u32x4 p;
            p.x = (uint32_t)kr_f32_to_bf16(v[0]) | ((uint32_t)kr_f32_to_bf16(v[1]) << 16); p.y = (uint32_t)kr_f32_to_bf16(v[2]) | ((uint32_t)kr_f32_to_bf16(v[3]) << 16);
            p.z = (uint32_t)kr_f32_to_bf16(v[4]) | ((uint32_t)kr_f32_to_bf16(v[5]) << 16); p.w = (uint32_t)kr_f32_to_bf16(v[6]) | ((uint32_t)kr_f32_to_bf16(v[7]) << 16);
            *reinterpret_cast<u32x4*>(a.out_bf16 + (size_t)t * H + c * 8) = p;
        }
    }
}

// ---- INT16 digits of f32 rows (quantize_activation_int16_f32, avx2.rs:274); grid rows, K % 128 == 0 ----------------------
__global__ void kr_pfm_quant_f32_kernel(const float* __restrict__ x, int ld, int K, int8_t* __restrict__ xh, int8_t* __restrict__ xl, float* __restrict__ xs) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < K / 8; c += blockDim.x) {
        const float4 p0 = *reinterpret_cast<const float4*>(x + (size_t)t * ld + c * 8), p1 = *reinterpret_cast<const float4*>(x + (size_t)t * ld + c * 8 + 4);
        const float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        kr_pfm_quant_chunk(v, xh + (size_t)t * K + c * 8, xl + (size_t)t * K + c * 8, xs + (size_t)t * (K / 128) + (c >> 4), (c & 15) == 0);
    }
}

// ---- linear attention: causal conv + SiLU + L2 norms + gates for every token (decode.rs:3815-3945) -------------------------
// grid (nk, C).  Tap j of token t is X(t-3+j): a chunk row for >= 0, the carried conv state slot 4+i for i < 0.
__global__ void __launch_bounds__(256) kr_pfm_la_conv_kernel(const KrPfmLaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int kh = blockIdx.x, t = blockIdx.y, dk = a.dk, dv = a.dv, hr = a.hr, nk = a.nk;
    const int group_dim = 2 * dk + 2 * dv * hr, key_dim = nk * dk, nvdk = a.nv * dk, nvdv = a.nv * dv;
    float* qc = sm; float* kc = sm + dk; float* nrm = sm + 2 * dk;
    const float* src = a.qkvz + (size_t)t * a.ld_qkvz + (size_t)kh * group_dim;
    const int nch = 2 * dk + hr * dv;
    for (int c = threadIdx.x; c < nch; c += 256) {
        int ch, off;
        if (c < dk) { ch = kh * dk + c; off = c; }
        else if (c < 2 * dk) { ch = key_dim + kh * dk + (c - dk); off = c; }
        else { const int r = (c - 2 * dk) / dv, i = (c - 2 * dk) % dv; ch = 2 * key_dim + (kh * hr + r) * dv + i; off = 2 * dk + r * dv + i; }
        const float* cs = a.conv_state + (size_t)ch * 4;
        const float* cw = a.conv_w + (size_t)ch * 4;
        float s[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int i = t - 3 + j; s[j] = i >= 0 ? a.qkvz[(size_t)i * a.ld_qkvz + (size_t)kh * group_dim + off] : cs[4 + i]; }
        float co = s[0] * cw[0] + s[1] * cw[1] + s[2] * cw[2] + s[3] * cw[3];
        co = co * kr_sigmoid_poly5(co);
        if (c < dk) qc[c] = co;
        else if (c < 2 * dk) kc[c - dk] = co;
        else a.v[(size_t)t * nvdv + (ch - 2 * key_dim)] = co;
    }
    for (int c = threadIdx.x; c < hr * dv; c += 256) {
        const int r = c / dv, i = c % dv;
        a.z[(size_t)t * nvdv + (size_t)(kh * hr + r) * dv + i] = src[2 * dk + hr * dv + r * dv + i];
    }
    if (threadIdx.x < hr) {   // decode.rs:3891-3901
        const int r = threadIdx.x, vh = kh * hr + r;
        const float* ba = a.ba + (size_t)t * a.ld_ba;
        const float b_raw = ba[kh * 2 * hr + r], a_p = ba[kh * 2 * hr + hr + r];
        a.beta[(size_t)t * a.nv + vh] = 1.0f / (1.0f + kr_expf(-b_raw));
        const float ap_dt = a_p + a.dt_bias[vh];
        const float softplus = ap_dt > 20.0f ? ap_dt : kr_logf(1.0f + kr_expf(ap_dt));
        const float g = -(kr_expf(a.a_log[vh])) * softplus;
        a.gexp[(size_t)t * a.nv + vh] = kr_expf(g);   // decode.rs:1293 decays the state by exp(g)
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int which = threadIdx.x >> 3, l = threadIdx.x & 7;
        const float ss = kr_pfm_sumsq8(which ? kc : qc, dk, l);
        if (l == 0) nrm[which] = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
    }
    __syncthreads();
    const float inv_q = nrm[0] * a.scale, inv_k = nrm[1] * 1.0f;
    for (int c = threadIdx.x; c < hr * dk; c += 256) {
        const int r = c / dk, i = c % dk, vh = kh * hr + r;
        a.q[(size_t)t * nvdk + (size_t)vh * dk + i] = qc[i] * inv_q;
        a.k[(size_t)t * nvdk + (size_t)vh * dk + i] = kc[i] * inv_k;
    }
}

// carried conv state after the chunk: slot j = X(C-4+j).  Launched AFTER the conv kernel (it reads the old slots).  one thread per channel
__global__ void kr_pfm_la_conv_state_kernel(const KrPfmLaArgs a, int C) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    const int key_dim = a.nk * a.dk, conv_dim = 2 * key_dim + a.nv * a.dv;
    if (ch >= conv_dim) return;
    const int group_dim = 2 * a.dk + 2 * a.dv * a.hr;
    int kh, off;
    if (ch < key_dim) { kh = ch / a.dk; off = ch % a.dk; }
    else if (ch < 2 * key_dim) { kh = (ch - key_dim) / a.dk; off = a.dk + (ch - key_dim) % a.dk; }
    else { const int vh = (ch - 2 * key_dim) / a.dv, i = (ch - 2 * key_dim) % a.dv; kh = vh / a.hr; off = 2 * a.dk + (vh % a.hr) * a.dv + i; }
    float* cs = a.conv_state + (size_t)ch * 4;
    float n[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const int i = C - 4 + j; n[j] = i >= 0 ? a.qkvz[(size_t)i * a.ld_qkvz + (size_t)kh * group_dim + off] : cs[4 + i]; }
#pragma unroll
    for (int j = 0; j < 4; j++) cs[j] = n[j];
}

// ---- gated delta rule over the chunk (decode.rs:1293): one thread per state column, the column lives in registers ------------
// grid nv, dv threads.  Per token: kv = chain_i fma(S[i]*e^g, k[i]); delta = (v - kv)*beta; S[i] = fma(k[i], delta, S[i]*e^g); o = chain_i fma(S[i], q[i]).
template <int DK>
__global__ void __launch_bounds__(256) kr_pfm_la_recur_kernel(float* __restrict__ state, const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, const float* __restrict__ gexp, const float* __restrict__ beta,
                                                             float* __restrict__ out, int nv, int dv, int C) {
    __shared__ __attribute__((aligned(16))) float ks[2][DK], qs[2][DK];
    const int h = blockIdx.x, j = threadIdx.x, nvdk = nv * DK, nvdv = nv * dv;
    float S[DK];
    float* Sg = state + (size_t)h * DK * dv + j;
#pragma unroll
    for (int i = 0; i < DK; i++) S[i] = Sg[(size_t)i * dv];
    for (int i = j; i < DK; i += blockDim.x) { ks[0][i] = k[(size_t)h * DK + i]; qs[0][i] = q[(size_t)h * DK + i]; }
    __syncthreads();
    for (int t = 0; t < C; t++) {
        const int cur = t & 1;
        float kn = 0.0f, qn = 0.0f;
        const bool pre = t + 1 < C && j < DK;     // DK <= blockDim (dv >= DK is checked by the launcher)
        if (pre) { kn = k[(size_t)(t + 1) * nvdk + (size_t)h * DK + j]; qn = q[(size_t)(t + 1) * nvdk + (size_t)h * DK + j]; }
        const float ge = gexp[(size_t)t * nv + h], bt = beta[(size_t)t * nv + h], vj = v[(size_t)t * nvdv + (size_t)h * dv + j];
        float kv = 0.0f;
#pragma unroll
        for (int i = 0; i < DK; i++) { S[i] = S[i] * ge; kv = __builtin_fmaf(S[i], ks[cur][i], kv); }
        const float delta = (vj - kv) * bt;
        float ob = 0.0f;
#pragma unroll
        for (int i = 0; i < DK; i++) { S[i] = __builtin_fmaf(ks[cur][i], delta, S[i]); ob = __builtin_fmaf(S[i], qs[cur][i], ob); }
        out[(size_t)t * nvdv + (size_t)h * dv + j] = ob;
        if (pre) { ks[cur ^ 1][j] = kn; qs[cur ^ 1][j] = qn; }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < DK; i++) Sg[(size_t)i * dv] = S[i];
}

// ---- gated RMSNorm + SiLU gate per (token, head) (decode.rs:3979); grid (nv, C), dv threads ----------------------------------
__global__ void __launch_bounds__(256) kr_pfm_gated_norm_kernel(const float* __restrict__ recur, const float* __restrict__ z, const float* __restrict__ w,
                                                               float* __restrict__ out, int nv, int dv, float eps) {
    __shared__ float r[256]; __shared__ float rms_s;
    const int h = blockIdx.x, t = blockIdx.y, i = threadIdx.x;
    const size_t o = (size_t)t * nv * dv + (size_t)h * dv + i;
    if (i < dv) r[i] = recur[o];
    __syncthreads();
    if (i < 8) { const float ss = kr_pfm_sumsq8(r, dv, i); if (i == 0) rms_s = 1.0f / sqrtf(ss / (float)dv + eps); }
    __syncthreads();
    if (i < dv) {
        const float normed = (r[i] * rms_s) * w[(size_t)h * dv + i];
        const float zz = z[o];
        out[o] = (zz * kr_sigmoid_poly5(zz)) * normed;
    }
}

// ---- GQA: gated split, per-head RMS norm, RoPE, FP16 KV append for every token (decode.rs:2873-2966); grid (nh + nkv, C) ---------
__global__ void __launch_bounds__(256) kr_pfm_gqa_prep_kernel(const KrPfmGqaArgs a) {
    __shared__ float x[256]; __shared__ float rms_s;
    const int b = blockIdx.x, t = blockIdx.y, d = threadIdx.x, hd = a.hd, pos = a.pos0 + t;
    const bool is_q = b < a.nh;
    const int h = is_q ? b : b - a.nh;
    const float* q_in = a.q_in + (size_t)t * a.ld_q; const float* k_in = a.k_in + (size_t)t * a.ld_k; const float* v_in = a.v_in + (size_t)t * a.ld_v;
    if (is_q) {
        if (a.gated) { if (d < hd) { x[d] = q_in[(size_t)h * hd * 2 + d]; a.gate[(size_t)t * a.nh * hd + (size_t)h * hd + d] = q_in[(size_t)h * hd * 2 + hd + d]; } }
        else if (d < hd) x[d] = q_in[(size_t)h * hd + d];
    } else if (d < hd) x[d] = k_in[(size_t)h * hd + d];
    __syncthreads();
    const float* nw = is_q ? a.q_norm : a.k_norm;
    if (nw) {
        if (d == 0) {
            float ss = 0.0f;
            for (int i = 0; i < hd; i++) ss += x[i] * x[i];
            rms_s = 1.0f / sqrtf(ss / (float)hd + a.eps);
        }
        __syncthreads();
        const int per_head = is_q ? a.q_norm_per_head : a.k_norm_per_head;
        if (d < hd) x[d] = x[d] * (rms_s * nw[(per_head ? h * hd : 0) + d]);
        __syncthreads();
    }
    const int d2 = a.rope_half;
    float val = d < hd ? x[d] : 0.0f;
    if (d < 2 * d2) {
        const float c = a.rope_cos[(size_t)pos * d2 + (d % d2)], s = a.rope_sin[(size_t)pos * d2 + (d % d2)];
        if (d < d2) val = x[d] * c - x[d2 + d] * s;
        else val = x[d] * c + x[d - d2] * s;
    }
    if (d < hd) {
        if (is_q) a.q_out[(size_t)t * a.nh * hd + (size_t)h * hd + d] = val;
        else {
            const size_t o = (size_t)pos * a.nkv * hd + (size_t)h * hd + d;
            a.k_cache[o] = __half_as_ushort(__float2half_rn(val));
            a.v_cache[o] = __half_as_ushort(__float2half_rn(v_in[(size_t)h * hd + d]));
        }
    }
}

// causal attention of token t over cache positions 0..pos0+t (decode.rs:4194); grid (nh, C), 256 threads, LDS (max_seq + 8) floats
__global__ void __launch_bounds__(256) kr_pfm_gqa_attn_kernel(const KrPfmGqaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sc[];
    __shared__ float qs[256]; __shared__ float red[8];
    const int h = blockIdx.x, t = blockIdx.y, hd = a.hd, kvs = a.nkv * hd, seq = a.pos0 + t + 1;
    const int kvh = h / (a.nh / a.nkv);
    if (threadIdx.x < hd) qs[threadIdx.x] = a.q_out[(size_t)t * a.nh * hd + (size_t)h * hd + threadIdx.x];
    __syncthreads();
    const int l = threadIdx.x & 7;
    for (int s = threadIdx.x >> 3; s < seq; s += 32) {
        const uint16_t* kr = a.k_cache + (size_t)s * kvs + (size_t)kvh * hd;
        float acc = 0.0f;
        for (int b = 0; b < hd / 8; b++) acc = __builtin_fmaf(qs[b * 8 + l], __half2float(__ushort_as_half(kr[b * 8 + l])), acc);
        acc = kr_pfm_hsum8(acc);
        if (l == 0) sc[s] = acc * a.sm_scale;
    }
    __syncthreads();
    float mx = -__builtin_inff();
    for (int s = threadIdx.x; s < seq; s += 256) mx = fmaxf(mx, sc[s]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int s = threadIdx.x; s < seq; s += 256) sc[s] = kr_expf(sc[s] - mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        float se = 0.0f; int s = 0;
        for (; s + 8 <= seq; s += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = sc[s + u];
#pragma unroll
            for (int u = 0; u < 8; u++) se += v[u];
        }
        for (; s < seq; s++) se += sc[s];
        red[4] = 1.0f / se;
    }
    __syncthreads();
    const float inv = red[4];
    for (int s = threadIdx.x; s < seq; s += 256) sc[s] *= inv;
    __syncthreads();
    const int d = threadIdx.x;
    if (d < hd) {
        const uint16_t* vc = a.v_cache + (size_t)kvh * hd + d;
        float o = 0.0f;
        for (int s = 0; s < seq; s++) o = __builtin_fmaf(sc[s], __half2float(__ushort_as_half(vc[(size_t)s * kvs])), o);
        if (a.gated) { const float gt = a.gate[(size_t)t * a.nh * hd + (size_t)h * hd + d]; o *= 1.0f / (1.0f + kr_expf(-gt)); }
        a.attn_out[(size_t)t * a.nh * hd + (size_t)h * hd + d] = o;
    }
}

// ---- MoE block epilogue (decode.rs:3343-3402): hidden = moe (*rsf) + shared (*sigmoid(gate)) -------------------------------------
__global__ void __launch_bounds__(256) kr_pfm_moe_epilogue_kernel(const float* __restrict__ moe, const float* __restrict__ shared, const float* __restrict__ gate_val,
                                                                 int gate_ld, float rsf, float* __restrict__ hidden, int H) {
    const int t = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= H) return;
    float acc = moe[(size_t)t * H + j];
    if (rsf != 1.0f) acc *= rsf;
    if (shared) {
        float sh = shared[(size_t)t * H + j];
        if (gate_val) sh *= 1.0f / (1.0f + kr_expf(-gate_val[(size_t)t * gate_ld]));
        acc = acc + sh;
    }
    hidden[(size_t)t * H + j] = acc;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void kr_launch_pfm_norm(const KrPfmNormArgs& a, int C, hipStream_t st) {
    hipLaunchKernelGGL(kr_pfm_norm_kernel, dim3(C), dim3(256), (size_t)(a.H + 4) * 4, st, a);
}
void kr_launch_pfm_quant_f32(const float* x, int rows, int ld, int K, int8_t* xh, int8_t* xl, float* xs, hipStream_t st) {
    const int thr = K / 8 < 1024 ? ((K / 8 + 15) / 16) * 16 : 1024;
    hipLaunchKernelGGL(kr_pfm_quant_f32_kernel, dim3(rows), dim3(thr), 0, st, x, ld, K, xh, xl, xs);
}
int kr_launch_pfm_la(const KrPfmLaArgs& a, float* recur_state, float* recur_out, const float* norm_w, float* gated_out, int C, float eps, hipStream_t st) {
    if (a.dv > 256 || a.dv % 8 || a.dv < a.dk || a.dk % 8) return 1;
    hipLaunchKernelGGL(kr_pfm_la_conv_kernel, dim3(a.nk, C), dim3(256), (size_t)(2 * a.dk + 4) * 4, st, a);
    const int conv_dim = 2 * a.nk * a.dk + a.nv * a.dv;
    hipLaunchKernelGGL(kr_pfm_la_conv_state_kernel, dim3((conv_dim + 255) / 256), dim3(256), 0, st, a, C);
    if (a.dk == 128) hipLaunchKernelGGL(kr_pfm_la_recur_kernel<128>, dim3(a.nv), dim3(a.dv), 0, st, recur_state, a.q, a.k, a.v, a.gexp, a.beta, recur_out, a.nv, a.dv, C);
    else if (a.dk == 64) hipLaunchKernelGGL(kr_pfm_la_recur_kernel<64>, dim3(a.nv), dim3(a.dv), 0, st, recur_state, a.q, a.k, a.v, a.gexp, a.beta, recur_out, a.nv, a.dv, C);
    else return 1;
    hipLaunchKernelGGL(kr_pfm_gated_norm_kernel, dim3(a.nv, C), dim3(a.dv), 0, st, recur_out, a.z, norm_w, gated_out, a.nv, a.dv, eps);
    return 0;
}
void kr_launch_pfm_gqa(const KrPfmGqaArgs& a, int C, hipStream_t st) {
    hipLaunchKernelGGL(kr_pfm_gqa_prep_kernel, dim3(a.nh + a.nkv, C), dim3(256), 0, st, a);
    hipLaunchKernelGGL(kr_pfm_gqa_attn_kernel, dim3(a.nh, C), dim3(256), (size_t)(a.pos0 + C + 8) * 4, st, a);
}
void kr_launch_pfm_moe_epilogue(const float* moe, const float* shared, const float* gate_val, int gate_ld, float rsf, float* hidden, int C, int H, hipStream_t st) {
    hipLaunchKernelGGL(kr_pfm_moe_epilogue_kernel, dim3((H + 255) / 256, C), dim3(256), 0, st, moe, shared, gate_val, gate_ld, rsf, hidden, H);
}
