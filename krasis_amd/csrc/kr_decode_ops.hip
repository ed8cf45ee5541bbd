// kr_decode_ops.hip -- the non-GEMV operators of the decode graph for gfx950, bit-exact restatements of the
// reference's AVX2 helpers in src/decode.rs (fused_add_rmsnorm_avx2 :1199, decode_la_conv :3815,
// l2_normalize_expand_avx2 :3909, linear_attention_recurrent_avx2 :1293, gated_rmsnorm_silu_avx2 :3979,
// GQA prep :2873-2966 + gqa_attention_compute_fp16_avx2 :4194, sample_from_logits greedy :3718).
//
// Every reduction that the reference performs with 8 AVX lanes + hsum is performed here by 8 GPU lanes walking the
// same fma chains and combined with the same tree, so each f32 result carries the same bits.
#include "kr_lds_optin.h"
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_decode_ops.h"
#include <hip/hip_fp16.h>
#include <cstdlib>

#ifdef KR_TIMING   // tools/probes/gqa_timing.hip: wall-clock stamps (10 ns units) written by thread 0 of workgroup 0, no-op in the product build
__device__ unsigned long long kr_dstamps[32];
#define KR_DSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) kr_dstamps[i] = wall_clock64(); } while (0)
#else
#define KR_DSTAMP(i) do { } while (0)
#endif

// hsum over 8 consecutive lanes in the order of the reference's hsum (lo+hi, movehdup, movehl)
__device__ __forceinline__ float kr_hsum8(float v) {
    v = v + __shfl_xor(v, 4);
    v = v + __shfl_xor(v, 1);
    v = v + __shfl_xor(v, 2);
    return v;
}

// sum of squares of x[0..n) (n % 8 == 0) with 8 fma lanes; call with the first 8 lanes of a wave (others idle)
__device__ __forceinline__ float kr_sumsq_chain8(const float* x, int n, int l) {
    float acc = 0.0f;
    const int nb = n / 8;
    int b = 0;
    for (; b + 32 <= nb; b += 32) {          // 32 LDS values in flight per lane, then the lane's fma chain
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; u++) v[u] = x[(b + u) * 8 + l];
#pragma unroll
        for (int u = 0; u < 32; u++) acc = __builtin_fmaf(v[u], v[u], acc);
    }
    for (; b + 8 <= nb; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = x[(b + u) * 8 + l];
#pragma unroll
        for (int u = 0; u < 8; u++) acc = __builtin_fmaf(v[u], v[u], acc);
    }
    for (; b < nb; b++) { const float v = x[b * 8 + l]; acc = __builtin_fmaf(v, v, acc); }
    return kr_hsum8(acc);
}

__global__ void kr_embed_kernel(const float* __restrict__ emb, const KrStep* __restrict__ st, float* __restrict__ hidden, int H) {
    const size_t base = (size_t)st->token * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H; i += gridDim.x * blockDim.x) hidden[i] = emb[base + i];
}

// decode.rs:1199.  One workgroup.  hidden/residual are updated in place.  The value added to the residual comes from
//   src.mode 0: the hidden buffer (output of the attention out-projection),
//   src.mode 1: the embedding row of the current token (decode.rs:2713-2717; first layer),
//   src.mode 2: the MoE epilogue of the previous layer, hidden = moe (*rsf) + shared * sigmoid(gate) (decode.rs:3343-3402),
// so the embedding copy and the MoE combine need no launch of their own.
#define KR_NORM_THREADS 1024
__global__ void __launch_bounds__(KR_NORM_THREADS) kr_fused_add_rmsnorm_kernel(const KrNormSrc src, float* hidden, const float* res_in, float* residual, const float* __restrict__ w,
                                                                              int n, float eps, int first, int bias_one, void* img_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    KR_DSTAMP(10);
    float* r = sm;                       // [n + 4]
    const int ldt = n / 8 + 4;
    float* rt = sm + n + 4;              // [8][ldt] lane-major copy for the sum-of-squares chain (n % 8 == 0)
    const bool tr = (n & 7) == 0;
    if (src.mode == 2 && src.topk <= 16 && src.topk >= 1 && (n & 7) == 0 && n <= 4 * KR_NORM_THREADS) {
        // MoE epilogue, four consecutive elements per thread: this ONE workgroup pulls (topk + 1) rows of n floats through one CU, and 16-byte requests move about
        // twice what 4-byte requests do there (kr_fla vs the 4-byte column loads of kr_la_step).  Requests are never masked (slots past topk re-read the last slot:
        // the compiler's load count stays exact); the sums are the same per-element chains in routing order (moe.rs:661-667), *rsf, + shared * sigmoid(gate).
        const int i4 = threadIdx.x, n4 = n >> 2;
        if (i4 < n4) {
            const float4* eo4 = reinterpret_cast<const float4*>(src.eo);
            float4 e[16]; float wv[16]; int idv[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const int uc = u < src.topk ? u : src.topk - 1; e[u] = eo4[(size_t)uc * n4 + i4]; wv[u] = src.wts[uc]; idv[u] = src.ids[uc]; }
            float4 sh = {0.0f, 0.0f, 0.0f, 0.0f}; float gv = 0.0f;
            if (src.has_shared) { sh = eo4[(size_t)src.topk * n4 + i4]; if (src.gate_val) gv = src.gate_val[0]; }
            float4 rv = {0.0f, 0.0f, 0.0f, 0.0f};
            if (!first) rv = reinterpret_cast<const float4*>(res_in)[i4];
            const float sig = (src.has_shared && src.gate_val) ? 1.0f / (1.0f + kr_expf(-gv)) : 1.0f;
            float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int u = 0; u < 16; u++) if (u < src.topk && idv[u] >= 0) { a[0] += wv[u] * e[u].x; a[1] += wv[u] * e[u].y; a[2] += wv[u] * e[u].z; a[3] += wv[u] * e[u].w; }
            const float shv[4] = {sh.x, sh.y, sh.z, sh.w}, rr4[4] = {rv.x, rv.y, rv.z, rv.w};
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float ac = a[c];
                if (src.rsf != 1.0f) ac *= src.rsf;
                if (src.has_shared) { float s1 = shv[c]; if (src.gate_val) s1 *= sig; ac = ac + s1; }
                v[c] = first ? ac : (ac + rr4[c]);
            }
            const int i0 = i4 * 4;
            reinterpret_cast<float4*>(r)[i4] = float4{v[0], v[1], v[2], v[3]};
            reinterpret_cast<float4*>(residual)[i4] = float4{v[0], v[1], v[2], v[3]};
#pragma unroll
            for (int c = 0; c < 4; c++) rt[((i0 + c) & 7) * ldt + ((i0 + c) >> 3)] = v[c];
        }
    } else if (src.mode == 2 && src.topk <= 16) {
        // (general widths) MoE epilogue: sum_i w_i * eo_i in routing order (moe.rs:661-667), *rsf, + shared * sigmoid(gate).  Every load of the
        // thread (slot rows, weights, ids, residual) is issued before the first use: one memory latency for the whole gather.
        for (int i0 = threadIdx.x; i0 < n; i0 += 2 * KR_NORM_THREADS) {
            const int i1 = i0 + KR_NORM_THREADS; const bool two = i1 < n;
            float e0[16], e1[16], wv[16]; int idv[16];
#pragma unroll
            for (int u = 0; u < 16; u++) if (u < src.topk) {
                e0[u] = src.eo[(size_t)u * n + i0]; if (two) e1[u] = src.eo[(size_t)u * n + i1];
                wv[u] = src.wts[u]; idv[u] = src.ids[u];
            }
            float sh0 = 0.0f, sh1 = 0.0f, gv = 0.0f;
            if (src.has_shared) { sh0 = src.eo[(size_t)src.topk * n + i0]; if (two) sh1 = src.eo[(size_t)src.topk * n + i1]; if (src.gate_val) gv = src.gate_val[0]; }
            const float r0 = first ? 0.0f : res_in[i0], r1 = (first || !two) ? 0.0f : res_in[i1];
            const float sig = (src.has_shared && src.gate_val) ? 1.0f / (1.0f + kr_expf(-gv)) : 1.0f;
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int u = 0; u < 16; u++) if (u < src.topk && idv[u] >= 0) { a0 += wv[u] * e0[u]; if (two) a1 += wv[u] * e1[u]; }
            if (src.rsf != 1.0f) { a0 *= src.rsf; a1 *= src.rsf; }
            if (src.has_shared) {
                if (src.gate_val) { sh0 *= sig; sh1 *= sig; }
                a0 = a0 + sh0; a1 = a1 + sh1;
            }
            const float v0 = first ? a0 : (a0 + r0);
            r[i0] = v0; residual[i0] = v0;
            if (tr) rt[(i0 & 7) * ldt + (i0 >> 3)] = v0;
            if (two) { const float v1 = first ? a1 : (a1 + r1); r[i1] = v1; residual[i1] = v1; if (tr) rt[(i1 & 7) * ldt + (i1 >> 3)] = v1; }
        }
    } else {
        __shared__ float s_w[32]; __shared__ int s_id[32]; __shared__ float s_sig;
        if (src.mode == 2) {
            if ((int)threadIdx.x < src.topk) { s_w[threadIdx.x] = src.wts[threadIdx.x]; s_id[threadIdx.x] = src.ids[threadIdx.x]; }
            if (threadIdx.x == 32) s_sig = (src.has_shared && src.gate_val) ? 1.0f / (1.0f + kr_expf(-src.gate_val[0])) : 1.0f;
            __syncthreads();
        }
        for (int i = threadIdx.x; i < n; i += KR_NORM_THREADS) {
            float hv;
            if (src.mode == 0) hv = hidden[i];
            else if (src.mode == 1) hv = src.emb[(size_t)src.step->token * n + i];
            else {
                float acc = 0.0f;
                for (int s0 = 0; s0 < src.topk; s0 += 8) {
                    float e[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) e[u] = (s0 + u < src.topk) ? src.eo[(size_t)(s0 + u) * n + i] : 0.0f;
#pragma unroll
                    for (int u = 0; u < 8; u++) if (s0 + u < src.topk && s_id[s0 + u] >= 0) acc += s_w[s0 + u] * e[u];
                }
                if (src.rsf != 1.0f) acc *= src.rsf;
                if (src.has_shared) {
                    float sh = src.eo[(size_t)src.topk * n + i];
                    if (src.gate_val) sh *= s_sig;
                    acc = acc + sh;
                }
                hv = acc;
            }
            const float v = first ? hv : (hv + res_in[i]);
            r[i] = v; residual[i] = v;
            if (tr) rt[(i & 7) * ldt + (i >> 3)] = v;
        }
    }
    KR_DSTAMP(11);
    __syncthreads();
    KR_DSTAMP(12);
    if (threadIdx.x < 8) {
        float ss = tr ? kr_hsum8(kr_sumsq_lane_t(rt, ldt, n, threadIdx.x)) : kr_sumsq_chain8(r, n, threadIdx.x);
        if (threadIdx.x == 0) {
            for (int t = (n / 8) * 8; t < n; t++) ss += r[t] * r[t];
            sm[n] = 1.0f / sqrtf(ss / (float)n + eps);
        }
    }
    KR_DSTAMP(13);
    __syncthreads();
    const float rms = sm[n];
    for (int i = threadIdx.x; i < n; i += KR_NORM_THREADS) { const float hv = (r[i] * rms) * (bias_one ? (w[i] + 1.0f) : w[i]); hidden[i] = hv; r[i] = hv; }
    KR_DSTAMP(14);
    if (img_out) {   // the INT16 image the next projection launches would otherwise each rebuild (quantize_activation_int16_f32, avx2.rs:274)
        __syncthreads();
        const KrActLds Lg = kr_carve_lds(reinterpret_cast<u32x4*>(img_out), n, false);
        kr_quant_range_f32<false>(r, 0, n / 8, Lg, false);
    }
    KR_DSTAMP(15);
}

// decode.rs:3815-3903 for kernel_dim == 4; one workgroup per key head.
__global__ void __launch_bounds__(256) kr_la_conv_kernel(const KrLaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int kh = blockIdx.x, dk = a.dk, dv = a.dv, hr = a.hr, nk = a.nk;
    const int group_dim = 2 * dk + 2 * dv * hr, key_dim = nk * dk;
    float* qc = sm;            // [dk] conv+silu output of the q channels
    float* kc = sm + dk;       // [dk]
    float* nrm = sm + 2 * dk;  // [2] inverse norms
    const float* src = a.qkvz + (size_t)kh * group_dim;
    const int nch = 2 * dk + hr * dv;
    for (int c = threadIdx.x; c < nch; c += 256) {
        int ch; float s4;
        if (c < dk) { ch = kh * dk + c; s4 = src[c]; }
        else if (c < 2 * dk) { ch = key_dim + kh * dk + (c - dk); s4 = src[c]; }
        else { const int r = (c - 2 * dk) / dv, i = (c - 2 * dk) % dv; ch = 2 * key_dim + (kh * hr + r) * dv + i; s4 = src[2 * dk + r * dv + i]; }
        float* cs = a.conv_state + (size_t)ch * 4;
        const float* cw = a.conv_w + (size_t)ch * 4;
        const float s1 = cs[1], s2 = cs[2], s3 = cs[3];
        cs[0] = s1; cs[1] = s2; cs[2] = s3; cs[3] = s4;
        float co = s1 * cw[0] + s2 * cw[1] + s3 * cw[2] + s4 * cw[3];
        co = co * kr_sigmoid_poly5(co);  // fast_silu_avx2 (conv_dim % 8 == 0)
        if (c < dk) qc[c] = co;
        else if (c < 2 * dk) kc[c - dk] = co;
        else a.v[(size_t)(ch - 2 * key_dim)] = co;
    }
    // z: plain copy
    for (int c = threadIdx.x; c < hr * dv; c += 256) {
        const int r = c / dv, i = c % dv;
        a.z[(size_t)(kh * hr + r) * dv + i] = src[2 * dk + hr * dv + r * dv + i];
    }
    // gates (decode.rs:3891-3901)
    if ((int)threadIdx.x < hr) {
        const int r = threadIdx.x, vh = kh * hr + r;
        const float b_raw = a.ba[kh * 2 * hr + r], a_p = a.ba[kh * 2 * hr + hr + r];
        a.beta[vh] = 1.0f / (1.0f + kr_expf(-b_raw));
        const float ap_dt = a_p + a.dt_bias[vh];
        const float softplus = ap_dt > 20.0f ? ap_dt : kr_logf(1.0f + kr_expf(ap_dt));
        a.g[vh] = -(kr_expf(a.a_log[vh])) * softplus;
    }
    __syncthreads();
    // L2 norms: lanes 0-7 -> q, lanes 8-15 -> k (decode.rs:3909-3945)
    if (threadIdx.x < 16) {
        const int which = threadIdx.x >> 3, l = threadIdx.x & 7;
        const float ss = kr_sumsq_chain8(which ? kc : qc, dk, l);
        if (l == 0) nrm[which] = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
    }
    __syncthreads();
    const float inv_q = nrm[0] * a.scale, inv_k = nrm[1] * 1.0f;
    for (int c = threadIdx.x; c < hr * dk; c += 256) {
        const int r = c / dk, i = c % dk, vh = kh * hr + r;
        a.q[(size_t)vh * dk + i] = qc[i] * inv_q;
        a.k[(size_t)vh * dk + i] = kc[i] * inv_k;
    }
}

// decode.rs:1293 + decode.rs:3979 fused.  grid nv, dv threads (one thread per state column, one workgroup per value head):
//   pass 1: kv[j] = sum_i fma(S[i][j]*e^g, k[i])  (chain over i), delta = (v - kv) * beta
//   pass 2: S[i][j] = fma(k[i], delta, S[i][j]*e^g); o[j] = sum_i fma(S[i][j], q[i])
// The decayed value is recomputed in pass 2 from the same operands (bit-identical), so a column never has to live in
// registers; the second read of the 64 KiB head slice hits L2.  Then the head's gated RMSNorm runs in the same workgroup.
#define KR_RB 32
template <int DK>
__global__ void __launch_bounds__(256) kr_la_recurrent_gnorm_kernel(float* __restrict__ state, const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ beta,
                                                                   const float* __restrict__ z, const float* __restrict__ w, float* __restrict__ out,
                                                                   int dv, float eps, void* img_out, int img_k) {
    __shared__ float ks[DK], qs[DK], r[256]; __shared__ float rms_s;
    const int h = blockIdx.x, j = threadIdx.x;
    for (int i = threadIdx.x; i < DK; i += blockDim.x) { ks[i] = k[(size_t)h * DK + i]; qs[i] = q[(size_t)h * DK + i]; }
    __syncthreads();
    const float g_exp = kr_expf(g[h]), beta_h = beta[h];
    float* S = state + (size_t)h * DK * dv + j;
    float kv = 0.0f;
    for (int i0 = 0; i0 < DK; i0 += KR_RB) {
        float c[KR_RB];
#pragma unroll
        for (int u = 0; u < KR_RB; u++) c[u] = S[(size_t)(i0 + u) * dv];
#pragma unroll
        for (int u = 0; u < KR_RB; u++) kv = __builtin_fmaf(c[u] * g_exp, ks[i0 + u], kv);
    }
    const float delta = (v[(size_t)h * dv + j] - kv) * beta_h;
    float ob = 0.0f;
    for (int i0 = 0; i0 < DK; i0 += KR_RB) {
        float c[KR_RB];
#pragma unroll
        for (int u = 0; u < KR_RB; u++) c[u] = S[(size_t)(i0 + u) * dv];
#pragma unroll
        for (int u = 0; u < KR_RB; u++) {
            const float sn = __builtin_fmaf(ks[i0 + u], delta, c[u] * g_exp);
            __builtin_nontemporal_store(sn, S + (size_t)(i0 + u) * dv);
            ob = __builtin_fmaf(sn, qs[i0 + u], ob);
        }
    }
    r[j] = ob;
    __syncthreads();
    if (j < 8) { const float ss = kr_sumsq_chain8(r, dv, j); if (j == 0) rms_s = 1.0f / sqrtf(ss / (float)dv + eps); }
    __syncthreads();
    const size_t o = (size_t)h * dv + j;
    const float normed = (ob * rms_s) * w[o];
    const float zz = z[o];
    const float ov = (zz * kr_sigmoid_poly5(zz)) * normed;
    out[o] = ov;
    if (img_out) {   // dv == 128: this head is exactly one quantization group of the out-projection's input
        __syncthreads();
        r[j] = ov;
        __syncthreads();
        const KrActLds Lg = kr_carve_lds(reinterpret_cast<u32x4*>(img_out), img_k, false);
        if (j < 16) {
            float v8[8];
            kr_load8(r, j, v8);
            float mx = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v8[i]));
            float scale, inv;
            kr_group_scale(mx, scale, inv);
            int q8[8];
            kr_quant8<false>(v8, inv, q8);
            kr_store_chunk<false>(Lg, h * 16 + j, q8);
            if (j == 0) Lg.ascale[h] = scale;
        }
    }
}

// Linear-attention decode step in ONE launch: conv1d + SiLU + gates + L2 norms (decode.rs:3815-3945), gated delta-rule
// recurrence (decode.rs:1293) and the head's gated RMSNorm (decode.rs:3979).  One workgroup per KEY head, hr*dv threads: thread
// (r, j) owns state column j of value head kh*hr + r.  A key head's conv channels (its q / k rows and the v rows of its hr value
// heads) are read and shifted by this workgroup only, so the in-place conv-state update needs no cross-workgroup ordering.
// The thread's whole state column (DK values) is requested before anything else and stays in registers for both passes: the
// conv / gate / norm work runs under that one memory latency, and the second pass re-reads nothing (k / q are broadcast LDS reads).
// CONV = false (the in-projection launch already ran the conv and the gates: kr_matvec_coop_kernel<.., LA = true>): one workgroup per VALUE head, dv threads -- twice
// (hr times) the CUs pull the state, which is what bounds this launch (64 KB per value head through one CU's ~20 - 35 GB/s).  a.q = the key heads' conv + SiLU outputs
// [nk][2 DK] (q | k, not yet normalised), a.v / a.z [nv][dv], a.ba = the raw in_proj_ba rows.  The head's vectors are requested BEFORE the state column: a wave's memory
// counter is in-order.
template <int DK, int DV, bool CONV = true>
__global__ void __launch_bounds__(256) kr_la_step_kernel(const KrLaArgs a, float* __restrict__ state, const float* __restrict__ w, float* __restrict__ out,
                                                        float eps, void* img_out, int img_k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int dv = DV;       // compile-time row pitch: the column's 2 * DK loads / stores use immediate offsets from a few bases
    const int hr = CONV ? a.hr : 1, t = threadIdx.x, nt = hr * dv;      // hr: value heads of THIS workgroup
    const int kh = CONV ? blockIdx.x : blockIdx.x / a.hr, vh0 = CONV ? kh * hr : blockIdx.x;
    float* qc = sm; float* kc = qc + DK; float* vs = kc + DK; float* rr = vs + nt;
    float* nrm = rr + nt; float* ge = nrm + 2; float* bt = ge + hr; float* rms = bt + hr;
    const int r = t / dv, j = t - r * dv, vh = vh0 + r;
    // buffer addressing: descriptor = this workgroup's hr state slices (workgroup-uniform), voffset = the thread's column, rows by
    // immediate offset (8 rows of DV floats per 4 KiB window) + a scalar window offset -- no per-row address registers
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(state + (size_t)vh0 * DK * dv, 0, hr * DK * dv * 4, 0x00020000);
    const int voff = (r * DK * dv + j) * 4;
    constexpr int RW = 4096 / (dv * 4);     // rows per immediate-offset window
    const size_t o = (size_t)vh * dv + j;
    float zz = 0.0f, wn = 0.0f, pq = 0.0f, pk = 0.0f, pv = 0.0f, pg = 0.0f, pb = 0.0f, pdt = 0.0f, pal = 0.0f;
    if constexpr (!CONV) {
        const int tq = t < DK ? t : 0;
        const int rh = blockIdx.x - kh * a.hr;
        zz = a.z[o]; wn = w[o]; pv = a.v[o]; pq = a.q[(size_t)kh * 2 * DK + tq]; pk = a.q[(size_t)kh * 2 * DK + DK + tq];
        pb = a.ba[kh * 2 * a.hr + rh]; pg = a.ba[kh * 2 * a.hr + a.hr + rh]; pdt = a.dt_bias[vh0]; pal = a.a_log[vh0];      // the head's raw gate inputs (decode.rs:3891-3901)
        asm volatile("" ::: "memory");      // (keeps the compiler from sinking these requests below the state's)
    }
    float c[DK];
#pragma unroll
    for (int u = 0; u < DK; u++)
        c[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srd, voff + (u % RW) * dv * 4, (u / RW) * 4096, 0));
    const int group_dim = 2 * DK + 2 * nt, key_dim = a.nk * DK;
    const float* src = a.qkvz + (size_t)kh * group_dim;
    if constexpr (CONV) { zz = src[2 * DK + nt + t]; wn = w[o]; }
    // ---- conv1d (kernel 4) + SiLU; channel c of this key head's group: q [0,DK), k [DK,2DK), v [2DK, 2DK + hr*dv)
    const int nch = 2 * DK + nt;
    auto chan = [&](int cc) { return cc < DK ? kh * DK + cc : (cc < 2 * DK ? key_dim + kh * DK + (cc - DK) : 2 * key_dim + kh * nt + (cc - 2 * DK)); };
    auto put = [&](int cc, float co) { if (cc < DK) qc[cc] = co; else if (cc < 2 * DK) kc[cc - DK] = co; else vs[cc - 2 * DK] = co; };
    if constexpr (!CONV) {
        if (t < DK) { qc[t] = pq; kc[t] = pk; }
        for (int i = t + nt; i < DK; i += nt) { qc[i] = a.q[(size_t)kh * 2 * DK + i]; kc[i] = a.q[(size_t)kh * 2 * DK + DK + i]; }      // DK > dv only
        vs[t] = pv;
        if (t == 0) {      // gates, the arithmetic of the CONV form below
            bt[0] = 1.0f / (1.0f + kr_expf(-pb));
            const float ap_dt = pg + pdt;
            const float softplus = ap_dt > 20.0f ? ap_dt : kr_logf(1.0f + kr_expf(ap_dt));
            const float g = -(kr_expf(pal)) * softplus;
            ge[0] = kr_expf(g);
        }
    }
    for (int c0 = t; CONV && c0 < nch; c0 += 2 * nt) {
        const int c1 = c0 + nt; const bool two = c1 < nch;
        const int ch0 = chan(c0), ch1 = two ? chan(c1) : ch0;
        float4* cs0 = reinterpret_cast<float4*>(a.conv_state) + ch0; float4* cs1 = reinterpret_cast<float4*>(a.conv_state) + ch1;
        const float4 s0 = *cs0, w0 = reinterpret_cast<const float4*>(a.conv_w)[ch0]; const float x0 = src[c0];
        float4 s1 = s0, w1 = w0; float x1 = 0.0f;
        if (two) { s1 = *cs1; w1 = reinterpret_cast<const float4*>(a.conv_w)[ch1]; x1 = src[c1]; }
        *cs0 = float4{s0.y, s0.z, s0.w, x0};
        float co = s0.y * w0.x + s0.z * w0.y + s0.w * w0.z + x0 * w0.w;
        put(c0, co * kr_sigmoid_poly5(co));      // fast_silu_avx2 (conv_dim % 8 == 0)
        if (two) {
            *cs1 = float4{s1.y, s1.z, s1.w, x1};
            co = s1.y * w1.x + s1.z * w1.y + s1.w * w1.z + x1 * w1.w;
            put(c1, co * kr_sigmoid_poly5(co));
        }
    }
    if (CONV && t < hr) {   // gates (decode.rs:3891-3901)
        const int vg = kh * hr + t;
        const float b_raw = a.ba[kh * 2 * hr + t], a_p = a.ba[kh * 2 * hr + hr + t];
        bt[t] = 1.0f / (1.0f + kr_expf(-b_raw));
        const float ap_dt = a_p + a.dt_bias[vg];
        const float softplus = ap_dt > 20.0f ? ap_dt : kr_logf(1.0f + kr_expf(ap_dt));
        const float g = -(kr_expf(a.a_log[vg])) * softplus;
        ge[t] = kr_expf(g);
    }
    __syncthreads();
    if (t < 16) {   // L2 norms: lanes 0-7 -> q, lanes 8-15 -> k (decode.rs:3909-3945)
        const int which = t >> 3, l = t & 7;
        const float ss = kr_sumsq_chain8(which ? kc : qc, DK, l);
        if (l == 0) nrm[which] = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
    }
    __syncthreads();
    {
        const float inv_q = nrm[0] * a.scale, inv_k = nrm[1] * 1.0f;
        for (int i = t; i < DK; i += nt) { qc[i] = qc[i] * inv_q; kc[i] = kc[i] * inv_k; }
    }
    __syncthreads();
    // ---- recurrence: kv[j] = sum_i fma(S[i][j]*e^g, k[i]); delta = (v - kv) * beta; S' = fma(k, delta, S*e^g); o = sum_i fma(S', q)
    // k and q are uniform across the workgroup and come out of LDS 4 at a time (broadcast ds_read_b128), one 16-element block ahead
    // of the chain that consumes them; a lone wave issues about one instruction per 9 cycles whatever the dependencies, so the count
    // matters: per 4 elements 1 read + 2 packed mul + 4 fma (pass 1), 2 reads + 2 packed fma + 4 stores + 4 fma (pass 2).  The
    // scheduling barriers keep the compiler from hoisting all the reads (the column already holds DK registers).
    const float g_exp = ge[r], beta_h = bt[r];
    const float4* k4 = reinterpret_cast<const float4*>(kc); const float4* q4 = reinterpret_cast<const float4*>(qc);
    float kv = 0.0f;
    float4 ka[4] = {k4[0], k4[1], k4[2], k4[3]};
#pragma unroll
    for (int b = 0; b < DK / 16; b++) {
        float4 kb[4] = {ka[0], ka[1], ka[2], ka[3]};
        if (b + 1 < DK / 16) { kb[0] = k4[4 * b + 4]; kb[1] = k4[4 * b + 5]; kb[2] = k4[4 * b + 6]; kb[3] = k4[4 * b + 7]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float* cb = c + 16 * b + 4 * u;
            cb[0] = cb[0] * g_exp; kv = __builtin_fmaf(cb[0], ka[u].x, kv);
            cb[1] = cb[1] * g_exp; kv = __builtin_fmaf(cb[1], ka[u].y, kv);
            cb[2] = cb[2] * g_exp; kv = __builtin_fmaf(cb[2], ka[u].z, kv);
            cb[3] = cb[3] * g_exp; kv = __builtin_fmaf(cb[3], ka[u].w, kv);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) ka[u] = kb[u];
    }
    const float delta = (vs[t] - kv) * beta_h;
    float ob = 0.0f;
    float4 qa[4] = {q4[0], q4[1], q4[2], q4[3]};
#pragma unroll
    for (int u = 0; u < 4; u++) ka[u] = k4[u];
#pragma unroll
    for (int b = 0; b < DK / 16; b++) {
        float4 kb[4] = {ka[0], ka[1], ka[2], ka[3]}, qb[4] = {qa[0], qa[1], qa[2], qa[3]};
        if (b + 1 < DK / 16) {
#pragma unroll
            for (int u = 0; u < 4; u++) { kb[u] = k4[4 * b + 4 + u]; qb[u] = q4[4 * b + 4 + u]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float kk[4] = {ka[u].x, ka[u].y, ka[u].z, ka[u].w}, qq[4] = {qa[u].x, qa[u].y, qa[u].z, qa[u].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = 16 * b + 4 * u + e;
                const float sn = __builtin_fmaf(kk[e], delta, c[i]);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sn), srd, voff + (i % RW) * dv * 4, (i / RW) * 4096, 2 /* nt */);
                ob = __builtin_fmaf(sn, qq[e], ob);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) { ka[u] = kb[u]; qa[u] = qb[u]; }
    }
    rr[t] = ob;
    __syncthreads();
    if (t < 8 * hr) { const int hh = t >> 3, l = t & 7; const float ss = kr_sumsq_chain8(rr + hh * dv, dv, l); if (l == 0) rms[hh] = 1.0f / sqrtf(ss / (float)dv + eps); }
    __syncthreads();
    const float normed = (ob * rms[r]) * wn;
    const float ov = (zz * kr_sigmoid_poly5(zz)) * normed;
    out[o] = ov;
    if (img_out) {   // dv == 128: each value head is exactly one quantization group of the out-projection's input
        __syncthreads();
        rr[t] = ov;
        __syncthreads();
        const KrActLds Lg = kr_carve_lds(reinterpret_cast<u32x4*>(img_out), img_k, false);
        if (t < 16 * hr) {
            const int hh = t >> 4, jj = t & 15, vg = vh0 + hh;
            float v8[8];
            kr_load8(rr + hh * dv, jj, v8);
            float mx = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v8[i]));
            float scale, inv;
            kr_group_scale(mx, scale, inv);
            int q8[8];
            kr_quant8<false>(v8, inv, q8);
            kr_store_chunk<false>(Lg, vg * 16 + jj, q8);
            if (jj == 0) Lg.ascale[vg] = scale;
        }
    }
}

// decode.rs:3979 (stand-alone form, kept for the per-op API gated_rmsnorm_silu, decode.rs:1062).  grid nv, dv threads
__global__ void __launch_bounds__(256) kr_gated_rmsnorm_silu_kernel(const float* __restrict__ recur, const float* __restrict__ z,
                                                                   const float* __restrict__ w, float* __restrict__ out, int dv, float eps) {
    __shared__ float r[256]; __shared__ float rms_s;
    const int h = blockIdx.x, i = threadIdx.x;
    if (i < dv) r[i] = recur[(size_t)h * dv + i];
    __syncthreads();
    if (i < 8) { const float ss = kr_sumsq_chain8(r, dv, i); if (i == 0) rms_s = 1.0f / sqrtf(ss / (float)dv + eps); }
    __syncthreads();
    if (i < dv) {
        const size_t o = (size_t)h * dv + i;
        const float normed = (r[i] * rms_s) * w[o];
        const float zz = z[o];
        out[o] = (zz * kr_sigmoid_poly5(zz)) * normed;
    }
}

// decode.rs:2873-2966: gated split, per-head RMS norm (scalar sequential sum), half-split RoPE, FP16 KV write.
// grid nh + nkv; block hd threads (hd <= 256).
__global__ void __launch_bounds__(256) kr_gqa_prep_kernel(const KrGqaArgs a) {
    __shared__ float x[256]; __shared__ float rms_s;
    const int b = blockIdx.x, d = threadIdx.x, hd = a.hd, pos = a.step->pos;
    const bool is_q = b < a.nh;
    const int h = is_q ? b : b - a.nh;
    if (is_q) {
        if (a.gated) { if (d < hd) { x[d] = a.q_in[(size_t)h * hd * 2 + d]; a.gate[(size_t)h * hd + d] = a.q_in[(size_t)h * hd * 2 + hd + d]; } }
        else if (d < hd) x[d] = a.q_in[(size_t)h * hd + d];
    } else if (d < hd) x[d] = a.k_in[(size_t)h * hd + d];
    __syncthreads();
    const float* nw = is_q ? a.q_norm : a.k_norm;
    if (nw && a.tree_norm) {      // tolerance mode: lane / wave / workgroup tree (the exact chain below is 2 x hd dependent adds on one thread)
        __shared__ float part[4];
        float v = d < hd ? x[d] * x[d] : 0.0f;
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if ((d & 63) == 0) part[d >> 6] = v;
        __syncthreads();
        const float ss = (part[0] + part[1]) + (part[2] + part[3]);
        const float rms = 1.0f / sqrtf(ss / (float)hd + a.eps);
        const int per_head = is_q ? a.q_norm_per_head : a.k_norm_per_head;
        const float xv = d < hd ? x[d] * (rms * nw[(per_head ? h * hd : 0) + d]) : 0.0f;
        __syncthreads();
        if (d < hd) x[d] = xv;
        __syncthreads();
    } else if (nw) {
        if (d == 0) {
            float ss = 0.0f; int i = 0;
            for (; i + 32 <= hd; i += 32) {      // 32 LDS values in flight, then the scalar chain in element order
                float v[32];
#pragma unroll
                for (int u = 0; u < 32; u++) { v[u] = x[i + u]; v[u] = v[u] * v[u]; }
#pragma unroll
                for (int u = 0; u < 32; u++) ss += v[u];
            }
            for (; i < hd; i++) ss += x[i] * x[i];
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
        if (d < d2) val = x[d] * c - x[d2 + d] * s;        // x1*cos - x2*sin
        else val = x[d] * c + x[d - d2] * s;               // x2*cos + x1*sin
    }
    if (d < hd) {
        if (is_q) a.q_out[(size_t)h * hd + d] = val;
        else {
            const size_t o = (size_t)pos * a.nkv * hd + (size_t)h * hd + d;
            kr_kv_store(a.k_cache, o, val, a.kv_fp8);
            kr_kv_store(a.v_cache, o, a.v_in[(size_t)h * hd + d], a.kv_fp8);
        }
    }
}

// decode.rs:4194.  grid nh; 256 threads; dynamic LDS = (max_seq + 40) floats of scores + one stage of KR_GQA_ROWS cache rows.
//
// Both passes over the cache (q.k scores, then the p.v chain) are latency problems: 16 workgroups, and every output is a
// sequential fma chain in the reference's order.  The rows are therefore staged: all 256 threads fetch KR_GQA_ROWS rows of the
// head's K (then V) slice with 16-byte loads (up to 16 in flight per lane), the next stage's loads are issued before the current
// stage is consumed from LDS, and the chains read 2-byte (FP16) / 1-byte (E4M3) elements from LDS.  One HBM/L2 latency per 128
// positions instead of one per element.
#define KR_GQA_ROWS 128
#define KR_GQA_NL 16      // 16-byte loads per thread per stage at hd 256 / FP16 (fewer for smaller rows)
template <bool FP8> __device__ __forceinline__ float kr_stage_elem(const unsigned char* row, int i) {
    if (FP8) return kr_e4m3_to_f32(row[i]);
    _Float16 hv; __builtin_memcpy(&hv, row + 2 * i, 2);
    return (float)hv;
}
// NB = head_dim / 8 as a compile-time constant (8 / 16 / 32): the per-lane loops unroll without the uniform guards that would put
// every LDS read into its own basic block behind an s_waitcnt.  NB == 0: any head_dim % 8 == 0 up to 256, guarded (slow) form.
// PHASE 0: the whole attention of one head in one workgroup (short caches).  Long caches split it: PHASE 1 = scores only, workgroup
// (h, y) covers positions [256 y, 256 y + 256) and writes a.sc_g (the q.k work is independent per position, so it spreads over
// nh x max_seq / 256 workgroups instead of crawling through one); PHASE 2 = softmax + p.v of head h reading those scores -- the part
// whose order is sequential by definition (the reference's position-ordered sum and fma chain).  `lds_seq` sizes the LDS score window.
template <bool FP8, int NB, int PHASE>
__global__ void __launch_bounds__(256) kr_gqa_attn_kernel(const KrGqaArgs a, int max_seq, int lds_seq) {
    extern __shared__ __attribute__((aligned(16))) float sc[];
    __shared__ float qs[256]; __shared__ float red[8];
    const int h = blockIdx.x, hd = a.hd, kvs = a.nkv * hd, seq = a.step->pos + 1, t = threadIdx.x;
    if (PHASE == 1 && (int)blockIdx.y * 256 >= seq) return;
    const int kvh = h / (a.nh / a.nkv);
    constexpr int esz = FP8 ? 1 : 2;
    const int row_bytes = hd * esz, pitch = row_bytes + 16, cpr = row_bytes >> 4;      // 16-byte chunks per row
    unsigned char* stage = reinterpret_cast<unsigned char*>(sc) + ((((size_t)lds_seq + 40) * 4 + 15) & ~(size_t)15);
    KR_DSTAMP(0);
    if (t < hd) qs[t] = a.q_out[(size_t)h * hd + t];
    // A stage is KR_GQA_ROWS rows; thread t fetches 16-byte column t % cprp (cprp = cpr rounded up to a power of two, <= 32) of rows
    // t / cprp + i * (256 / cprp): a fixed row step per load, so one voffset + a uniform row term addresses every load.  Buffer
    // addressing with num_records = seq rows: rows at or beyond the current length read as zero without a branch.
    int lg = 2; while ((1 << lg) < cpr) lg++;
    const int col = t & ((1 << lg) - 1), r0 = t >> lg, rstep = 256 >> lg, nl = KR_GQA_ROWS / rstep;   // nl = cprp / 2 <= KR_GQA_NL
    const int grow = kvs * esz;                                                          // bytes per cache row
    const int voff = col < cpr ? r0 * grow + kvh * hd * esz + col * 16 : 0x7FFFFFF0;     // idle columns: out of range -> zero
    const int loff = r0 * pitch + (col < cpr ? col : 0) * 16;
    const __amdgpu_buffer_rsrc_t srd_k = __builtin_amdgcn_make_buffer_rsrc(a.k_cache, 0, seq * grow, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_v = __builtin_amdgcn_make_buffer_rsrc(a.v_cache, 0, seq * grow, 0x00020000);
    u32x4 rg[KR_GQA_NL];
    int rg_s0 = 0;
    auto issue = [&](const __amdgpu_buffer_rsrc_t& srd, int s0) {
        rg_s0 = s0;
#pragma unroll
        for (int i = 0; i < KR_GQA_NL; i++)
            if (i < nl && s0 + i * rstep < seq) rg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, voff + (s0 + i * rstep) * grow, 0, 0);
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < KR_GQA_NL; i++)
            if (i < nl && rg_s0 + i * rstep < seq && col < cpr) *reinterpret_cast<u32x4*>(stage + loff + i * rstep * pitch) = rg[i];
    };
    const __amdgpu_buffer_rsrc_t& kbase = srd_k;
    const int nst = (seq + KR_GQA_ROWS - 1) / KR_GQA_ROWS;
    // ---- V stages, FP16 with head_dim 128 / 256: COLUMN-major in LDS.  The p.v chain of output d is one fma per position in position
    // order, and a lone wave pays per instruction, so the instructions around that fma are what can be saved: with the stage stored as
    // [d][128 positions] one 16-byte LDS read hands thread d its next 8 positions (the row-major stage costs a 2-byte read and a convert
    // per position), the probabilities come four per broadcast read, and the half -> float widening rides inside the fma (v_fma_mix).
    // A thread fetches 8 CONSECUTIVE rows of its 16-byte column (two such blocks at head_dim 256), transposes the 8 x 8 halves in
    // registers (one v_perm per output dword) and writes eight 16-byte position groups.  Row d starts at d * 272 bytes (8 consecutive d
    // on distinct bank groups for the readers) and its 16 position groups are XOR-swizzled by (d / 8) % 8 (the 8 writers of one
    // instruction, d = 8 col + j, land on distinct bank groups).
    constexpr bool VT = !FP8 && NB >= 16;
    constexpr int pitchT = KR_GQA_ROWS * 2 + 16;
    constexpr int LGT = NB == 32 ? 5 : 4, RST = 256 >> LGT, NLT = VT ? KR_GQA_ROWS / RST : 0;     // the staging geometry above as constants
    const int voffT = r0 * 8 * grow + kvh * hd * esz + col * 16;
    auto issue_v = [&](int s0) {
        if constexpr (VT) {      // unguarded: rows at or past the current length are outside the descriptor and read as zero
#pragma unroll
            for (int i = 0; i < NLT; i++) rg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_v, voffT + (s0 + (i >> 3) * RST * 8 + (i & 7)) * grow, 0, 0);
        } else issue(srd_v, s0);
    };
    auto commit_v = [&]() {
        if constexpr (VT) {
#pragma unroll
            for (int blk = 0; blk < NLT / 8; blk++) {
                const int pb = blk * RST + r0;                                        // position group of these 8 rows inside the stage
                unsigned char* base = stage + (size_t)(col * 8) * pitchT + ((pb ^ (col & 7)) << 4);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
                    u32x4 o4;
                    o4.x = __builtin_amdgcn_perm(rg[blk * 8 + 1][j >> 1], rg[blk * 8 + 0][j >> 1], sel);
                    o4.y = __builtin_amdgcn_perm(rg[blk * 8 + 3][j >> 1], rg[blk * 8 + 2][j >> 1], sel);
                    o4.z = __builtin_amdgcn_perm(rg[blk * 8 + 5][j >> 1], rg[blk * 8 + 4][j >> 1], sel);
                    o4.w = __builtin_amdgcn_perm(rg[blk * 8 + 7][j >> 1], rg[blk * 8 + 6][j >> 1], sel);
                    *reinterpret_cast<u32x4*>(base + j * pitchT) = o4;
                }
            }
        } else commit();
    };
    // one stage of the p.v chain for output t: n positions, probabilities P[0..n)
    const unsigned char* rowT = stage + (size_t)t * pitchT;
    const int swz = (t >> 3) & 7;
    auto chain8 = [&](float o, const u32x4 v, const float4 pa, const float4 pb) {
        auto lo = [](uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu)); };
        auto hi = [](uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); };
        o = __builtin_fmaf(pa.x, lo(v.x), o); o = __builtin_fmaf(pa.y, hi(v.x), o);
        o = __builtin_fmaf(pa.z, lo(v.y), o); o = __builtin_fmaf(pa.w, hi(v.y), o);
        o = __builtin_fmaf(pb.x, lo(v.z), o); o = __builtin_fmaf(pb.y, hi(v.z), o);
        o = __builtin_fmaf(pb.z, lo(v.w), o); o = __builtin_fmaf(pb.w, hi(v.w), o);
        return o;
    };
    auto pv_stage_t = [&](float o, const float* P, int n) {
        const int nfull = n >> 3;
        u32x4 va[4], vb[4]; float4 pa[8], pbv[8];
        auto loadg = [&](u32x4 (&V)[4], float4 (&Pq)[8], int g0) {
#pragma unroll
            for (int u = 0; u < 4; u++) V[u] = *reinterpret_cast<const u32x4*>(rowT + (((g0 + u) ^ swz) << 4));
#pragma unroll
            for (int u = 0; u < 8; u++) Pq[u] = *reinterpret_cast<const float4*>(P + g0 * 8 + u * 4);
        };
        auto chain32 = [&](const u32x4 (&V)[4], const float4 (&Pq)[8]) {
#pragma unroll
            for (int u = 0; u < 4; u++) o = chain8(o, V[u], Pq[2 * u], Pq[2 * u + 1]);
        };
        int g0 = 0;
        if (nfull >= 8) {                        // two register sets: the next 32 positions are read from LDS under the current chain
            loadg(va, pa, 0);
            for (; g0 + 16 <= nfull; g0 += 8) {
                loadg(vb, pbv, g0 + 4);
                __builtin_amdgcn_sched_barrier(0);
                chain32(va, pa);
                loadg(va, pa, g0 + 8);
                __builtin_amdgcn_sched_barrier(0);
                chain32(vb, pbv);
            }
            loadg(vb, pbv, g0 + 4);
            chain32(va, pa);
            chain32(vb, pbv);
            g0 += 8;
        }
        for (; g0 < nfull; g0++) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(rowT + ((g0 ^ swz) << 4));
            o = chain8(o, v, *reinterpret_cast<const float4*>(P + g0 * 8), *reinterpret_cast<const float4*>(P + g0 * 8 + 4));
        }
        const int rem = n & 7;
        if (rem) {                               // last, partial group of the cache (once per launch)
            const u32x4 v = *reinterpret_cast<const u32x4*>(rowT + ((nfull ^ swz) << 4));
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int k2 = 0; k2 < rem; k2++) {
                const uint16_t hb = (uint16_t)((k2 & 1) ? (w[k2 >> 1] >> 16) : (w[k2 >> 1] & 0xFFFFu));
                o = __builtin_fmaf(P[nfull * 8 + k2], (float)__builtin_bit_cast(_Float16, hb), o);
            }
        }
        return o;
    };
    // ---- scores: 8 lanes per position, lane l owns elements b*8 + l (the AVX2 lane), ascending b, then the 8-lane hsum
    const int l = t & 7, g = t >> 3, nb = NB ? NB : (hd >> 3);
    constexpr int NBM = NB ? NB : 32;
    if (PHASE == 3) {
        // ---- caches too long for an LDS-resident score row: the row stays in a.sc_g and is streamed in tiles.  max -> exp (in place) ->
        // position-ordered sum over 4096-value tiles (one thread, the running sum carried across tiles) -> p.v with the stage's 128
        // probabilities scaled into a small LDS window.  Same operations, same order as the resident form.
        constexpr int TILE = 4096;
        float* tile = sc;                                    // [TILE + 32]; lds_seq == TILE for this phase
        float* row = a.sc_g + (size_t)h * max_seq;
        float mx = -__builtin_inff();
        for (int s = t; s < seq; s += 256) mx = fmaxf(mx, row[s]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if ((t & 63) == 0) red[t >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        for (int s = t; s < seq; s += 256) row[s] = kr_expf(row[s] - mx);
        issue_v(0);
        __syncthreads();
        float se = 0.0f;
        for (int s0 = 0; s0 < seq; s0 += TILE) {
            const int n = min(TILE, seq - s0), n32 = (n + 31) & ~31;
            for (int i = t; i < n32; i += 256) tile[i] = i < n ? row[s0 + i] : 0.0f;   // zero padding: s + 0.0f == s for sums of exponentials
            __syncthreads();
            if (t == 0) { se = kr_seq_sum(tile, n32, se); red[5] = se; }
            __syncthreads();
        }
        const float inv3 = 1.0f / red[5];
        float o = 0.0f;
        for (int st = 0; st < nst; st++) {
            __syncthreads();
            commit_v();
            if (st + 1 < nst) issue_v((st + 1) * KR_GQA_ROWS);
            const int s0 = st * KR_GQA_ROWS, n = min(KR_GQA_ROWS, seq - s0);
            if (t < KR_GQA_ROWS) tile[t] = t < n ? row[s0 + t] * inv3 : 0.0f;          // sc[s] *= inv (decode.rs:4260)
            __syncthreads();
            if (VT) { if (t < hd) o = pv_stage_t(o, tile, n); }
            else if (t < hd) {
                for (int r = 0; r < n; r += 16) {
                    float vv[16], pp[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) { vv[u] = kr_stage_elem<FP8>(stage + (r + u) * pitch, t); pp[u] = tile[r + u]; }
#pragma unroll
                    for (int u = 0; u < 16; u++) if (r + u < n) o = __builtin_fmaf(pp[u], vv[u], o);
                }
            }
        }
        if (t < hd) {
            if (a.gated) { const float gt = a.gate[(size_t)h * hd + t]; o *= 1.0f / (1.0f + kr_expf(-gt)); }
            a.attn_out[(size_t)h * hd + t] = o;
            if (a.img_out) qs[t] = o;
        }
        if (a.img_out) {
            __syncthreads();
            const KrActLds Lg = kr_carve_lds(reinterpret_cast<u32x4*>(a.img_out), a.nh * hd, false);
            const int nch = hd / 8, c = t;
            if (c < nch) {
                float v8[8];
                kr_load8(qs, c, v8);
                float mx8 = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; i++) mx8 = fmaxf(mx8, fabsf(v8[i]));
                float scale, inv;
                kr_group_scale(mx8, scale, inv);
                int q8[8];
                kr_quant8<false>(v8, inv, q8);
                const int gc = h * nch + c;
                kr_store_chunk<false>(Lg, gc, q8);
                if ((gc & 15) == 0) Lg.ascale[gc >> 4] = scale;
            }
        }
        return;
    }
    const int st_lo = PHASE == 1 ? (int)blockIdx.y * 2 : 0, st_hi = PHASE == 1 ? min(nst, st_lo + 2) : (PHASE == 2 ? 0 : nst);
    if (PHASE == 2) {                            // scores of this head come from the scores launch
        for (int s = t; s < seq; s += 256) sc[s] = a.sc_g[(size_t)h * max_seq + s];
        issue_v(0);
    } else issue(kbase, st_lo * KR_GQA_ROWS);
    __syncthreads();
    KR_DSTAMP(1);
    float qr[NBM];                               // the lane's query elements (hd <= 256)
#pragma unroll
    for (int b = 0; b < NBM; b++) qr[b] = (NB || b < nb) ? qs[b * 8 + l] : 0.0f;
    for (int st = st_lo; st < st_hi; st++) {
        if (st > st_lo) __syncthreads();         // the previous stage's readers are done
        commit();
        if (st + 1 < st_hi) issue(kbase, (st + 1) * KR_GQA_ROWS); else if (PHASE == 0) issue_v(0);   // V stage 0 rides under the softmax
        __syncthreads();
        KR_DSTAMP(2);
        const int s0 = st * KR_GQA_ROWS;
        // four passes of 32 rows (8 lanes per row); the next pass's elements are read from LDS while the current chain runs
        float ka[NBM], kb[NBM];
        auto loadk = [&](float (&X)[NBM], int r) {
            const unsigned char* row = stage + r * pitch;
#pragma unroll
            for (int b2 = 0; b2 < NBM; b2++) if (NB || b2 < nb) X[b2] = kr_stage_elem<FP8>(row, b2 * 8 + l);
        };
        auto chaink = [&](const float (&X)[NBM], int r) {
            float acc = 0.0f;
#pragma unroll
            for (int b2 = 0; b2 < NBM; b2++) if (NB || b2 < nb) acc = __builtin_fmaf(qr[b2], X[b2], acc);
            acc = kr_hsum8(acc);
            if (l == 0) { if (PHASE == 1) a.sc_g[(size_t)h * max_seq + s0 + r] = acc * a.sm_scale; else sc[s0 + r] = acc * a.sm_scale; }
        };
        if (s0 + g < seq) loadk(ka, g);
#pragma unroll
        for (int k2 = 0; k2 < KR_GQA_ROWS / 32; k2 += 2) {
            const int ra = g + 32 * k2, rb = ra + 32, rc = ra + 64;
            if (s0 + rb < seq) loadk(kb, rb);
            __builtin_amdgcn_sched_barrier(0);
            if (s0 + ra < seq) chaink(ka, ra);
            if (rc < KR_GQA_ROWS && s0 + rc < seq) loadk(ka, rc);
            __builtin_amdgcn_sched_barrier(0);
            if (s0 + rb < seq) chaink(kb, rb);
        }
    }
    KR_DSTAMP(3);
    if (PHASE == 1) return;
    __syncthreads();
    float mx = -__builtin_inff();
    for (int s = t; s < seq; s += 256) mx = fmaxf(mx, sc[s]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int s = t; s < seq; s += 256) sc[s] = kr_expf(sc[s] - mx);
    const int seq32 = (seq + 31) & ~31;          // zero padding: the exponentials are >= +0, so s + 0.0f == s bit for bit
    if (t < seq32 - seq) sc[seq + t] = 0.0f;
    __syncthreads();
    KR_DSTAMP(4);
    if (t == 0) red[4] = 1.0f / kr_seq_sum(sc, seq32);
    KR_DSTAMP(5);   // position order (decode.rs:4194), LDS reads double-buffered under the add chain
    __syncthreads();
    const float inv = red[4];
    for (int s = t; s < seq; s += 256) sc[s] *= inv;
    // ---- p.v: thread d owns output d, one fma per position in ascending order
    float o = 0.0f;
    for (int st = 0; st < nst; st++) {
        __syncthreads();                         // previous stage consumed; the scaled scores are visible
        commit_v();
        if (st + 1 < nst) issue_v((st + 1) * KR_GQA_ROWS);
        __syncthreads();
        KR_DSTAMP(6);
        if (VT) { if (t < hd) o = pv_stage_t(o, sc + st * KR_GQA_ROWS, min(KR_GQA_ROWS, seq - st * KR_GQA_ROWS)); }
        else if (t < hd) {
            const int s0 = st * KR_GQA_ROWS, n = min(KR_GQA_ROWS, seq - s0);
            const unsigned char* col = stage;
            // batches of 16 rows, two register sets: the next batch's LDS reads are in flight under the current fma chain
            float va[16], pa[16], vb[16], pb[16];
            auto loadv = [&](float (&V)[16], float (&P)[16], int r) {
#pragma unroll
                for (int u = 0; u < 16; u++) { V[u] = kr_stage_elem<FP8>(col + (r + u) * pitch, t); P[u] = sc[s0 + r + u]; }
            };
            int r = 0;
            if (n >= 32) {
                loadv(va, pa, 0);
                for (; r + 64 <= n; r += 32) {
                    loadv(vb, pb, r + 16);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 16; u++) o = __builtin_fmaf(pa[u], va[u], o);
                    loadv(va, pa, r + 32);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 16; u++) o = __builtin_fmaf(pb[u], vb[u], o);
                }
                loadv(vb, pb, r + 16);          // set a = rows r.., at least 32 and fewer than 64 rows remain
#pragma unroll
                for (int u = 0; u < 16; u++) o = __builtin_fmaf(pa[u], va[u], o);
#pragma unroll
                for (int u = 0; u < 16; u++) o = __builtin_fmaf(pb[u], vb[u], o);
                r += 32;
            }
            for (; r < n; r += 16) {             // n <= 128 and r % 16 == 0: rows r..r+15 are inside the stage
                loadv(va, pa, r);
#pragma unroll
                for (int u = 0; u < 16; u++) if (r + u < n) o = __builtin_fmaf(pa[u], va[u], o);
            }
        }
    }
    KR_DSTAMP(7);
    const int d = t;
    if (d < hd) {
        if (a.gated) { const float gt = a.gate[(size_t)h * hd + d]; o *= 1.0f / (1.0f + kr_expf(-gt)); }
        a.attn_out[(size_t)h * hd + d] = o;
        if (a.img_out) qs[d] = o;
    }
    if (a.img_out) {   // hd % 128 == 0: the head's output is hd/128 whole quantization groups of the o-projection's input
        __syncthreads();
        const KrActLds Lg = kr_carve_lds(reinterpret_cast<u32x4*>(a.img_out), a.nh * hd, false);
        const int nch = hd / 8, c = threadIdx.x;
        if (c < nch) {
            float v8[8];
            kr_load8(qs, c, v8);
            float mx = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v8[i]));
            float scale, inv;
            kr_group_scale(mx, scale, inv);
            int q8[8];
            kr_quant8<false>(v8, inv, q8);
            const int gc = h * nch + c;
            kr_store_chunk<false>(Lg, gc, q8);
            if ((gc & 15) == 0) Lg.ascale[gc >> 4] = scale;
        }
    }
}

// ---- long caches, head_dim 64 / 128 / 256 (FP16 or E4M3): softmax + p.v of one head with PRODUCER and CONSUMER waves -------------------------------------
// The p.v chain is one fma per position in position order, and a wave issues roughly one instruction per 9 cycles whatever it is doing,
// so everything that is not that fma is moved OFF the chain's waves: waves 0-3 (thread d owns output d) only read the staged values and run
// the chain; waves 4-7 fetch the next 64 cache rows, transpose them to column-major in registers and write them to the other half of a
// double-buffered LDS stage (and, when the score row is streamed, scale the next 64 probabilities into a small window).  Two waves share
// a SIMD, so the producer's instructions fill issue slots the chain leaves empty.  One workgroup barrier per 64 positions.
// STREAM = the score row stays in a.sc_g (caches past ~22 k positions): max / exp / position-ordered sum over 4096-value tiles.
// Same operations in the same order as kr_gqa_attn_kernel PHASE 2 / 3 (decode.rs:4194-4281).
#define KR_PV_ROWS 64
#define KR_PV_DEPTH 4
template <int NB, bool STREAM, bool FP8>
__global__ void __launch_bounds__(512) kr_gqa_pv_kernel(const KrGqaArgs a, int max_seq, int lds_seq) {
    extern __shared__ __attribute__((aligned(16))) float sc[];
    __shared__ float qs[256]; __shared__ float red[12]; __shared__ __attribute__((aligned(16))) float pw[2][KR_PV_ROWS];
    constexpr int hd = NB * 8, esz = FP8 ? 1 : 2, pitchT = KR_PV_ROWS * esz + 16, stage_bytes = hd * pitchT;
    const int h = blockIdx.x, kvs = a.nkv * hd, seq = a.step->pos + 1, t = threadIdx.x, kvh = h / (a.nh / a.nkv);
    unsigned char* stage = reinterpret_cast<unsigned char*>(sc) + ((((size_t)lds_seq + 40) * 4 + 15) & ~(size_t)15);
    const int nst = (seq + KR_PV_ROWS - 1) / KR_PV_ROWS;
    float* row = a.sc_g + (size_t)h * max_seq;
    // ---- producer state.  FP16: thread pt fetches rows rb * 8 .. + 7 of 16-byte column `col` (NB columns x 8 row blocks = NB * 8 pieces per
    // stage).  E4M3: rows rb * 16 .. + 15 of the 4-byte column `col` (hd / 4 columns x 4 row blocks): a 16 x 4 byte block becomes four
    // 16-position groups with 8 v_perm per four rows; row d of the stage is 64 bytes + 16, position groups swizzled by (d / 8) % 4.
    const bool producer = t >= 256;
    constexpr int NCOL = FP8 ? hd / 4 : NB, NRB = FP8 ? 4 : 8, RPB = KR_PV_ROWS / NRB, CB = FP8 ? 4 : 16;   // columns, row blocks, rows per block, column bytes
    const int pt = t - 256, col = pt & (NCOL - 1), rb = pt / NCOL;
    const bool pactive = producer && rb < NRB;
    const int grow = kvs * esz;
    const __amdgpu_buffer_rsrc_t srd_v = __builtin_amdgcn_make_buffer_rsrc(a.v_cache, 0, seq * grow, 0x00020000);
    const int voff = pactive ? rb * RPB * grow + kvh * hd * esz + col * CB : 0x7FFFFFF0;
    // KR_PV_DEPTH register sets: the rows of stage k are requested KR_PV_DEPTH stages (~2 us) before they are transposed into LDS
    constexpr int NRG = FP8 ? 4 : 8;
    u32x4 rg[KR_PV_DEPTH][NRG];
    auto issue_v = [&](u32x4 (&R)[NRG], int s0) {      // unguarded: rows at or past the current length are outside the descriptor and read as zero
        if constexpr (FP8) {
#pragma unroll
            for (int i = 0; i < 16; i++) R[i >> 2][i & 3] = __builtin_amdgcn_raw_buffer_load_b32(srd_v, voff + (s0 + i) * grow, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) R[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_v, voff + (s0 + i) * grow, 0, 0);
        }
    };
    auto commit_v = [&](const u32x4 (&R)[NRG], int buf) {
        if (!pactive) return;
        if constexpr (FP8) {
            u32x4 o4[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {              // rows 4m .. 4m+3 of the block: byte j of each -> dword m of output j
                const uint32_t ta = __builtin_amdgcn_perm(R[m].y, R[m].x, 0x05010400u), tb = __builtin_amdgcn_perm(R[m].y, R[m].x, 0x07030602u);
                const uint32_t tc = __builtin_amdgcn_perm(R[m].w, R[m].z, 0x05010400u), td = __builtin_amdgcn_perm(R[m].w, R[m].z, 0x07030602u);
                o4[0][m] = __builtin_amdgcn_perm(tc, ta, 0x05040100u); o4[1][m] = __builtin_amdgcn_perm(tc, ta, 0x07060302u);
                o4[2][m] = __builtin_amdgcn_perm(td, tb, 0x05040100u); o4[3][m] = __builtin_amdgcn_perm(td, tb, 0x07060302u);
            }
            unsigned char* base = stage + buf * stage_bytes + (size_t)(col * 4) * pitchT + ((rb ^ ((col >> 1) & 3)) << 4);
#pragma unroll
            for (int j = 0; j < 4; j++) *reinterpret_cast<u32x4*>(base + j * pitchT) = o4[j];
        } else {
            unsigned char* base = stage + buf * stage_bytes + (size_t)(col * 8) * pitchT + ((rb ^ (col & 7)) << 4);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
                u32x4 o4;
                o4.x = __builtin_amdgcn_perm(R[1][j >> 1], R[0][j >> 1], sel);
                o4.y = __builtin_amdgcn_perm(R[3][j >> 1], R[2][j >> 1], sel);
                o4.z = __builtin_amdgcn_perm(R[5][j >> 1], R[4][j >> 1], sel);
                o4.w = __builtin_amdgcn_perm(R[7][j >> 1], R[6][j >> 1], sel);
                *reinterpret_cast<u32x4*>(base + j * pitchT) = o4;
            }
        }
    };
    if (producer) {
#pragma unroll
        for (int u = 0; u < KR_PV_DEPTH; u++) issue_v(rg[u], u * KR_PV_ROWS);
    }
    // ---- softmax: maximum, exponentials, position-ordered sum (one lane), scale
    float mx = -__builtin_inff();
    if (STREAM) { for (int s = t; s < seq; s += 512) mx = fmaxf(mx, row[s]); }
    else { for (int s = t; s < seq; s += 512) { const float v = row[s]; sc[s] = v; mx = fmaxf(mx, v); } }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 8; w++) mx = fmaxf(mx, red[w]);
    float inv;
    if (STREAM) {
        constexpr int TILE = 4096;
        for (int s = t; s < seq; s += 512) row[s] = kr_expf(row[s] - mx);
        if (producer) { commit_v(rg[0], 0); issue_v(rg[0], KR_PV_DEPTH * KR_PV_ROWS); }
        __syncthreads();
        float se = 0.0f;
        for (int s0 = 0; s0 < seq; s0 += TILE) {
            const int n = min(TILE, seq - s0), n32 = (n + 31) & ~31;
            for (int i = t; i < n32; i += 512) sc[i] = i < n ? row[s0 + i] : 0.0f;     // zero padding: s + 0.0f == s for sums of exponentials
            __syncthreads();
            if (t == 0) { se = kr_seq_sum(sc, n32, se); red[9] = se; }
            __syncthreads();
        }
        inv = 1.0f / red[9];
        if (t < KR_PV_ROWS) pw[0][t] = t < seq ? row[t] * inv : 0.0f;                  // sc[s] *= inv (decode.rs:4260), stage 0
    } else {
        for (int s = t; s < seq; s += 512) sc[s] = kr_expf(sc[s] - mx);
        const int seq32 = (seq + 31) & ~31;
        if (t < seq32 - seq) sc[seq + t] = 0.0f;
        if (producer) { commit_v(rg[0], 0); issue_v(rg[0], KR_PV_DEPTH * KR_PV_ROWS); }
        __syncthreads();
        if (t == 0) red[8] = 1.0f / kr_seq_sum(sc, seq32);
        __syncthreads();
        inv = red[8];
        for (int s = t; s < seq; s += 512) sc[s] *= inv;
    }
    __syncthreads();
    // ---- p.v
    const unsigned char* rowT = stage + (size_t)(t & 255) * pitchT;
    const int swz = FP8 ? (t >> 3) & 3 : (t >> 3) & 7;
    float o = 0.0f;
    auto chain8 = [&](float acc, const u32x4 v, const float4 pa, const float4 pb) {
        auto lo = [](uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu)); };
        auto hi = [](uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); };
        acc = __builtin_fmaf(pa.x, lo(v.x), acc); acc = __builtin_fmaf(pa.y, hi(v.x), acc);
        acc = __builtin_fmaf(pa.z, lo(v.y), acc); acc = __builtin_fmaf(pa.w, hi(v.y), acc);
        acc = __builtin_fmaf(pb.x, lo(v.z), acc); acc = __builtin_fmaf(pb.y, hi(v.z), acc);
        acc = __builtin_fmaf(pb.z, lo(v.w), acc); acc = __builtin_fmaf(pb.w, hi(v.w), acc);
        return acc;
    };
    // E4M3: one dword = 4 positions; the hardware widening (v_cvt_f32_fp8, OCP E4M3 on gfx950 == torch.float8_e4m3fn for every finite code)
    auto chain4f8 = [&](float acc, const uint32_t w, const float4 p4) {
        acc = __builtin_fmaf(p4.x, __builtin_amdgcn_cvt_f32_fp8((int)w, 0), acc); acc = __builtin_fmaf(p4.y, __builtin_amdgcn_cvt_f32_fp8((int)w, 1), acc);
        acc = __builtin_fmaf(p4.z, __builtin_amdgcn_cvt_f32_fp8((int)w, 2), acc); acc = __builtin_fmaf(p4.w, __builtin_amdgcn_cvt_f32_fp8((int)w, 3), acc);
        return acc;
    };
    for (int st0 = 0; st0 < nst; st0 += KR_PV_DEPTH) {
#pragma unroll
      for (int u = 0; u < KR_PV_DEPTH; u++) {           // stage st0 + u lives in register set u (st0 % KR_PV_DEPTH == 0)
        const int st = st0 + u;
        if (st >= nst) break;
        const int buf = st & 1, s0 = st * KR_PV_ROWS;
        if (producer) {
            if (st + 1 < nst) {
                constexpr int DN = KR_PV_DEPTH;
                commit_v(rg[(u + 1) % DN], buf ^ 1);
                issue_v(rg[(u + 1) % DN], (st + 1 + DN) * KR_PV_ROWS);
                if (STREAM && pt < KR_PV_ROWS) { const int sn = s0 + KR_PV_ROWS + pt; pw[buf ^ 1][pt] = sn < seq ? row[sn] * inv : 0.0f; }
            }
        } else if (t < hd) {
            const float* P = STREAM ? pw[buf] : sc + s0;
            const unsigned char* rT = rowT + buf * stage_bytes;
            const int n = min(KR_PV_ROWS, seq - s0);
            if (FP8) {
                if (n == KR_PV_ROWS) {
                    u32x4 v[4]; float4 pq[16];
#pragma unroll
                    for (int u2 = 0; u2 < 4; u2++) v[u2] = *reinterpret_cast<const u32x4*>(rT + ((u2 ^ swz) << 4));
#pragma unroll
                    for (int u2 = 0; u2 < 16; u2++) pq[u2] = *reinterpret_cast<const float4*>(P + u2 * 4);
#pragma unroll
                    for (int u2 = 0; u2 < 16; u2++) o = chain4f8(o, v[u2 >> 2][u2 & 3], pq[u2]);
                } else {                         // last, partial stage of the cache (once per launch)
                    for (int k2 = 0; k2 < n; k2++) {
                        const uint32_t w = *reinterpret_cast<const uint32_t*>(rT + (((k2 >> 4) ^ swz) << 4) + ((k2 >> 2) & 3) * 4);
                        const int bs = k2 & 3;
                        const float vv = bs == 0 ? __builtin_amdgcn_cvt_f32_fp8((int)w, 0) : bs == 1 ? __builtin_amdgcn_cvt_f32_fp8((int)w, 1)
                                       : bs == 2 ? __builtin_amdgcn_cvt_f32_fp8((int)w, 2) : __builtin_amdgcn_cvt_f32_fp8((int)w, 3);
                        o = __builtin_fmaf(P[k2], vv, o);
                    }
                }
            } else if (n == KR_PV_ROWS) {        // full stage: both halves' reads are issued before the first chain
                u32x4 v[8]; float4 pq[16];
#pragma unroll
                for (int u2 = 0; u2 < 8; u2++) v[u2] = *reinterpret_cast<const u32x4*>(rT + ((u2 ^ swz) << 4));
#pragma unroll
                for (int u2 = 0; u2 < 16; u2++) pq[u2] = *reinterpret_cast<const float4*>(P + u2 * 4);
#pragma unroll
                for (int u2 = 0; u2 < 8; u2++) o = chain8(o, v[u2], pq[2 * u2], pq[2 * u2 + 1]);
            } else {                             // last, partial stage of the cache (once per launch)
                const int nfull = n >> 3, rem = n & 7;
                for (int g0 = 0; g0 < nfull; g0++)
                    o = chain8(o, *reinterpret_cast<const u32x4*>(rT + ((g0 ^ swz) << 4)), *reinterpret_cast<const float4*>(P + g0 * 8), *reinterpret_cast<const float4*>(P + g0 * 8 + 4));
                if (rem) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(rT + ((nfull ^ swz) << 4));
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                    for (int k2 = 0; k2 < rem; k2++) {
                        const uint16_t hb = (uint16_t)((k2 & 1) ? (w[k2 >> 1] >> 16) : (w[k2 >> 1] & 0xFFFFu));
                        o = __builtin_fmaf(P[nfull * 8 + k2], (float)__builtin_bit_cast(_Float16, hb), o);
                    }
                }
            }
        }
        __syncthreads();
      }
    }
    if (t < hd) {
        if (a.gated) { const float gt = a.gate[(size_t)h * hd + t]; o *= 1.0f / (1.0f + kr_expf(-gt)); }
        a.attn_out[(size_t)h * hd + t] = o;
        if (a.img_out) qs[t] = o;
    }
    if (a.img_out) {   // hd % 128 == 0: the head's output is hd/128 whole quantization groups of the o-projection's input
        __syncthreads();
        const KrActLds Lg = kr_carve_lds(reinterpret_cast<u32x4*>(a.img_out), a.nh * hd, false);
        constexpr int nch = hd / 8;
        if (t < nch) {
            float v8[8];
            kr_load8(qs, t, v8);
            float mx8 = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) mx8 = fmaxf(mx8, fabsf(v8[i]));
            float scale, qinv;
            kr_group_scale(mx8, scale, qinv);
            int q8[8];
            kr_quant8<false>(v8, qinv, q8);
            const int gc = h * nch + t;
            kr_store_chunk<false>(Lg, gc, q8);
            if ((gc & 15) == 0) Lg.ascale[gc >> 4] = scale;
        }
    }
}
static size_t kr_gqa_pv_lds(int lds_seq, int hd, int fp8) { return ((((size_t)lds_seq + 40) * 4 + 15) & ~(size_t)15) + 2 * (size_t)hd * (KR_PV_ROWS * (fp8 ? 1 : 2) + 16); }

// FAST (tolerance) mode for long caches: kr_launch_fd_flash (kr_attn_flash.hip), dispatched at the top of kr_launch_gqa
// decode-step MoE epilogue (decode.rs:3343-3345, 3391-3402): hidden = moe (*rsf) + shared (*sigmoid(gate))
__global__ void __launch_bounds__(256) kr_moe_combine_decode_kernel(const float* __restrict__ eo, const int32_t* __restrict__ ids,
                                                                   const float* __restrict__ wts, int topk, int has_shared,
                                                                   const float* __restrict__ gate_val, float rsf, float* __restrict__ hidden, int H) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= H) return;
    float acc = 0.0f;
    for (int s = 0; s < topk; s++) {
        if (ids[s] < 0) continue;
        acc += wts[s] * eo[(size_t)s * H + j];
    }
    if (rsf != 1.0f) acc *= rsf;
    if (has_shared) {
        float sh = eo[(size_t)topk * H + j];
        if (gate_val) sh *= 1.0f / (1.0f + kr_expf(-gate_val[0]));
        acc = acc + sh;
    }
    hidden[j] = acc;
}

// hidden = act(gu[0..n), gu[n..2n)) for the dense-MLP path is handled by the matvec prologue (KR_ACT_SILU_MUL).

// greedy sampling: first maximum wins (decode.rs:3718).  64 workgroups scan slices; the one that finishes last (device-scope
// counter, reset for the next replay) reduces the 64 partials.  (value desc, index asc) is associative, so the split is exact.
#define KR_ARGMAX_BLOCKS 64
__global__ void __launch_bounds__(1024) kr_argmax_kernel(const float* __restrict__ x, int n, int* __restrict__ out, float* part_v, int* part_i,
                                                         unsigned* counter) {
    __shared__ float bv[16]; __shared__ int bi[16]; __shared__ int s_last;
    float v = -__builtin_inff(); int idx = 0x7FFFFFFF;
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += KR_ARGMAX_BLOCKS * 1024) { const float t = x[i]; if (t > v || (t == v && i < idx)) { v = t; idx = i; } }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off); const int oi = __shfl_xor(idx, off);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = v; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) if (bv[w] > v || (bv[w] == v && bi[w] < idx)) { v = bv[w]; idx = bi[w]; }
        __hip_atomic_store(part_v + blockIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part_i + blockIdx.x, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == KR_ARGMAX_BLOCKS - 1;
        if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    v = __hip_atomic_load(part_v + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    idx = __hip_atomic_load(part_i + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off); const int oi = __shfl_xor(idx, off);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    if (threadIdx.x == 0) out[0] = idx;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// (token, position) of a step are KERNEL ARGUMENTS of this one-thread launch: they are captured by value when the call is queued, so any
// number of steps may be in flight without a shared host-side slot (a pinned slot + async copy let a later step overwrite an earlier one's
// parameters before its DMA ran)
__global__ void kr_set_step_kernel(KrStep* dst, int token, int pos) { dst->token = token; dst->pos = pos; }
void kr_launch_set_step(KrStep* dst, int token, int pos, hipStream_t s) { hipLaunchKernelGGL(kr_set_step_kernel, dim3(1), dim3(1), 0, s, dst, token, pos); }
// the same with the token taken from device memory (the previous step's sample): generate_batch's look-ahead loop queues step i + 1 without a host round trip
__global__ void kr_set_step_dev_kernel(KrStep* dst, const int* token, int pos) { dst->token = *token; dst->pos = pos; }
void kr_launch_set_step_dev(KrStep* dst, const int* token_dev, int pos, hipStream_t s) { hipLaunchKernelGGL(kr_set_step_dev_kernel, dim3(1), dim3(1), 0, s, dst, token_dev, pos); }

void kr_launch_embed(const float* emb, const KrStep* st, float* hidden, int H, hipStream_t s) {
    hipLaunchKernelGGL(kr_embed_kernel, dim3((H + 255) / 256), dim3(256), 0, s, emb, st, hidden, H);
}
void kr_launch_fused_add_rmsnorm(const KrNormSrc& src, float* hidden, const float* res_in, float* residual, const float* w, int n, float eps, int first, int bias_one, hipStream_t s,
                                 void* img_out) {
    if (n % 128) img_out = nullptr;
    hipLaunchKernelGGL(kr_fused_add_rmsnorm_kernel, dim3(1), dim3(KR_NORM_THREADS), (size_t)(n + 4 + 8 * (n / 8 + 4)) * 4, s, src, hidden, res_in, residual, w, n, eps, first, bias_one, img_out);
}
void kr_launch_la_conv(const KrLaArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(kr_la_conv_kernel, dim3(a.nk), dim3(256), (size_t)(2 * a.dk + 4) * 4, s, a);
}
int kr_launch_la_recurrent_gnorm(float* state, const float* q, const float* k, const float* v, const float* g, const float* beta, const float* z,
                                 const float* w, float* out, int nv, int dk, int dv, float eps, hipStream_t s, void* img_out) {
    if (dv > 256 || dv % 8 != 0) return 1;
    if (dv != 128) img_out = nullptr;
    const int img_k = nv * dv;
    if (dk == 128) hipLaunchKernelGGL(kr_la_recurrent_gnorm_kernel<128>, dim3(nv), dim3(dv), 0, s, state, q, k, v, g, beta, z, w, out, dv, eps, img_out, img_k);
    else if (dk == 64) hipLaunchKernelGGL(kr_la_recurrent_gnorm_kernel<64>, dim3(nv), dim3(dv), 0, s, state, q, k, v, g, beta, z, w, out, dv, eps, img_out, img_k);
    else return 1;
    return 0;
}
// fused conv + recurrence + gated norm (one launch); returns 1 when the geometry needs the two-launch path
// the recurrence + gated norm of kr_la_step with one workgroup per VALUE head; conv outputs and gates come from the in-projection launch (kr_launch_multi_matvec_la):
// a.q = [nk][2 dk] conv + SiLU outputs of q | k, a.v / a.z [nv][dv], a.ba = the raw in_proj_ba rows (gates are formed here).  Non-zero: geometry not covered
int kr_launch_la_step_heads(const KrLaArgs& a, float* state, const float* w, float* out, float eps, hipStream_t s, void* img_out) {
    if (a.nv != a.nk * a.hr || (a.dk != 128 && a.dk != 64) || (a.dv != 128 && a.dv != 64)) return 1;
    if (a.dv != 128) img_out = nullptr;
    const size_t lds = (size_t)(2 * a.dk + 2 * a.dv + 2 + 3 + 4) * 4;
#define KR_LAH(DK_, DV_) hipLaunchKernelGGL((kr_la_step_kernel<DK_, DV_, false>), dim3(a.nv), dim3(a.dv), lds, s, a, state, w, out, eps, img_out, a.nv * a.dv)
    if (a.dk == 128) { if (a.dv == 128) KR_LAH(128, 128); else KR_LAH(128, 64); }
    else { if (a.dv == 128) KR_LAH(64, 128); else KR_LAH(64, 64); }
#undef KR_LAH
    return 0;
}
int kr_launch_la_step(const KrLaArgs& a, float* state, const float* w, float* out, float eps, hipStream_t s, void* img_out) {
    const int nt = a.hr * a.dv;
    if (nt > 256 || nt % 64 || a.nv != a.nk * a.hr || (a.dk != 128 && a.dk != 64) || (a.dv != 128 && a.dv != 64)) return 1;
    if (a.dv != 128) img_out = nullptr;
    const size_t lds = (size_t)(2 * a.dk + 2 * nt + 2 + 3 * a.hr + 4) * 4;
#define KR_LA(DK_, DV_) hipLaunchKernelGGL((kr_la_step_kernel<DK_, DV_>), dim3(a.nk), dim3(nt), lds, s, a, state, w, out, eps, img_out, a.nv * a.dv)
    if (a.dk == 128) { if (a.dv == 128) KR_LA(128, 128); else KR_LA(128, 64); }
    else { if (a.dv == 128) KR_LA(64, 128); else KR_LA(64, 64); }
#undef KR_LA
    return 0;
}
void kr_launch_gated_rmsnorm_silu(const float* recur, const float* z, const float* w, float* out, int nv, int dv, float eps, hipStream_t s) {
    hipLaunchKernelGGL(kr_gated_rmsnorm_silu_kernel, dim3(nv), dim3(256), 0, s, recur, z, w, out, dv, eps);
}
static size_t kr_gqa_attn_lds(int max_seq, int hd, int fp8) {
    const size_t row_major = (size_t)KR_GQA_ROWS * ((size_t)hd * (fp8 ? 1 : 2) + 16), col_major = (size_t)hd * (KR_GQA_ROWS * 2 + 16);   // K stage / FP16 V stage
    return ((((size_t)max_seq + 40) * 4 + 15) & ~(size_t)15) + (row_major > col_major || fp8 ? row_major : col_major);
}
// Raises the kernel's dynamic-LDS window (gfx950: 160 KiB per workgroup).  Called outside graph capture, before the first launch.
// LDS-resident score rows fit up to ~23 k positions; beyond that the softmax / p.v launch streams the row from HBM (PHASE 3)
static bool kr_gqa_resident(int max_seq, int hd, int fp8) { return kr_gqa_attn_lds(max_seq, hd, fp8) <= 160 * 1024; }
int kr_gqa_attn_prepare(int max_seq, int hd, int fp8) {
    const size_t lds = kr_gqa_resident(max_seq, hd, fp8) ? kr_gqa_attn_lds(max_seq, hd, fp8) : kr_gqa_attn_lds(4096, hd, fp8);
    {
#define KR_F(F_, N_) (const void*)kr_gqa_attn_kernel<F_, N_, 0>, (const void*)kr_gqa_attn_kernel<F_, N_, 1>, (const void*)kr_gqa_attn_kernel<F_, N_, 2>, (const void*)kr_gqa_attn_kernel<F_, N_, 3>
        const void* f16[16] = {KR_F(false, 8), KR_F(false, 16), KR_F(false, 32), KR_F(false, 0)};
        const void* f8[16] = {KR_F(true, 8), KR_F(true, 16), KR_F(true, 32), KR_F(true, 0)};
#undef KR_F
        for (int i = 0; i < 16; i++)
            if (kr_lds_optin(fp8 ? f8[i] : f16[i], lds)) return -2;
    }
    if (hd == 64 || hd == 128 || hd == 256) {    // the producer / consumer softmax + p.v kernel of long caches
        const bool res = kr_gqa_pv_lds(max_seq, hd, fp8) <= 160 * 1024;
        const size_t lp = kr_gqa_pv_lds(res ? max_seq : 4096, hd, fp8);
        {
#define KR_F(N_, F_) (const void*)kr_gqa_pv_kernel<N_, false, F_>, (const void*)kr_gqa_pv_kernel<N_, true, F_>
            const void* f16[6] = {KR_F(8, false), KR_F(16, false), KR_F(32, false)};
            const void* f8[6] = {KR_F(8, true), KR_F(16, true), KR_F(32, true)};
#undef KR_F
            for (int i = 0; i < 6; i++) if (kr_lds_optin(fp8 ? f8[i] : f16[i], lp)) return -2;
        }
    }
    return 0;
}
template <int PHASE>
static void kr_launch_gqa_phase(const KrGqaArgs& a, int max_seq, dim3 grid, int lds_seq, hipStream_t s) {
    const size_t lds = kr_gqa_attn_lds(lds_seq, a.hd, a.kv_fp8);
#define KR_GQA(F_, N_) hipLaunchKernelGGL((kr_gqa_attn_kernel<F_, N_, PHASE>), grid, dim3(256), lds, s, a, max_seq, lds_seq)
    if (a.kv_fp8) { if (a.hd == 256) KR_GQA(true, 32); else if (a.hd == 128) KR_GQA(true, 16); else if (a.hd == 64) KR_GQA(true, 8); else KR_GQA(true, 0); }
    else { if (a.hd == 256) KR_GQA(false, 32); else if (a.hd == 128) KR_GQA(false, 16); else if (a.hd == 64) KR_GQA(false, 8); else KR_GQA(false, 0); }
#undef KR_GQA
}
void kr_launch_gqa(const KrGqaArgs& a, int max_seq, hipStream_t s) {
    hipLaunchKernelGGL(kr_gqa_prep_kernel, dim3(a.nh + a.nkv), dim3(256), 0, s, a);
    if (a.sc_g && a.fd_o && a.fd_ml) {      // FAST mode, long cache: one split-KV flash-decode launch + merge (no score scratch)
        KrFdFlashArgs f{};
        f.step = a.step; f.q = a.q_out; f.k_cache = a.k_cache; f.v_cache = a.v_cache; f.fd_o = a.fd_o; f.fd_ml = a.fd_ml; f.nh = a.nh; f.nkv = a.nkv;
        f.sm_scale = a.sm_scale; f.gate = a.gate; f.gated = a.gated; f.out = a.attn_out; f.img_out = a.img_out;
        if (kr_launch_fd_flash(f, a.hd, a.kv_fp8, max_seq, s) == 0) return;
    }
    if (a.sc_g) {      // long cache: scores over nh x max_seq / 256 workgroups (those past the current length leave at once), then softmax + p.v
        kr_launch_gqa_phase<1>(a, max_seq, dim3(a.nh, (max_seq + 255) / 256), 0, s);
        const bool stream_hook = a.force_stream != 0;
        if (a.hd == 64 || a.hd == 128 || a.hd == 256) {
            const bool res = kr_gqa_pv_lds(max_seq, a.hd, a.kv_fp8) <= 160 * 1024 && !stream_hook;
            const int lds_seq = res ? max_seq : 4096;
            const size_t lds = kr_gqa_pv_lds(lds_seq, a.hd, a.kv_fp8);
#define KR_PVK(N_, S_, F_) hipLaunchKernelGGL((kr_gqa_pv_kernel<N_, S_, F_>), dim3(a.nh), dim3(512), lds, s, a, max_seq, lds_seq)
#define KR_PVS(N_, F_) do { if (res) KR_PVK(N_, false, F_); else KR_PVK(N_, true, F_); } while (0)
            if (a.hd == 256) { if (a.kv_fp8) KR_PVS(32, true); else KR_PVS(32, false); }
            else if (a.hd == 128) { if (a.kv_fp8) KR_PVS(16, true); else KR_PVS(16, false); }
            else { if (a.kv_fp8) KR_PVS(8, true); else KR_PVS(8, false); }
#undef KR_PVS
#undef KR_PVK
        } else if (kr_gqa_resident(max_seq, a.hd, a.kv_fp8) && !stream_hook) kr_launch_gqa_phase<2>(a, max_seq, dim3(a.nh), max_seq, s);
        else kr_launch_gqa_phase<3>(a, max_seq, dim3(a.nh), 4096, s);
    } else kr_launch_gqa_phase<0>(a, max_seq, dim3(a.nh), max_seq, s);
}
void kr_launch_moe_combine_decode(const float* eo, const int32_t* ids, const float* wts, int topk, int has_shared, const float* gate_val,
                                  float rsf, float* hidden, int H, hipStream_t s) {
    hipLaunchKernelGGL(kr_moe_combine_decode_kernel, dim3((H + 255) / 256), dim3(256), 0, s, eo, ids, wts, topk, has_shared, gate_val, rsf, hidden, H);
}
void kr_launch_argmax(const float* x, int n, int* out, float* scratch /* >= 129 words, word 128 = counter (zeroed once) */, hipStream_t s) {
    hipLaunchKernelGGL(kr_argmax_kernel, dim3(KR_ARGMAX_BLOCKS), dim3(1024), 0, s, x, n, out, scratch, (int*)(scratch + 64), (unsigned*)(scratch + 128));
}
