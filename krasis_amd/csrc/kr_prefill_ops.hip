// kr_prefill_ops.hip -- batched (M tokens) forms of the decode-graph operators, for the whole-model prompt pass kr_decode_prefill.
//
// Contract: processing tokens t = 0..C-1 of a chunk with these kernels leaves every buffer (logits, FP16 KV, conv / recurrent state)
// BIT-IDENTICAL to C successive decode steps (src/decode.rs:2690-3520), i.e. prefill == decode == oracle.  Every per-token reduction
// keeps the decode kernel's order (8 fma lanes + hsum for the norms, one sequential chain per state column for the gated delta rule,
// sequential softmax sums); tokens only add independent parallel work.  The GEMM-shaped work (projections, experts) runs on the
// int8-MFMA grouped GEMM of kr_prefill.hip with the exact INT16-digit arithmetic.
#include "kr_lds_optin.h"
#include "kr_device.h"
#include "kr_libm.h"
#include "kr_prefill_ops.h"
#include "kr_pfh_dev.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>

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
    __shared__ uint32_t s_max;
    const int t = blockIdx.x, H = a.H, tid = threadIdx.x;
    if (tid == 0) s_max = 0;
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
    float mx = 0.0f;
    for (int c = tid; c < H / 8; c += 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = c * 8 + u; v[u] = (r[i] * rms) * (a.bias_one ? (a.w[i] + 1.0f) : a.w[i]); mx = fmaxf(mx, fabsf(v[u])); }
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
    if (a.xf) {      // the f16 row image of the values just stored, exactly as kr_pfh_rows_kernel<0> forms it from `out` (uniform per launch: every thread meets the barrier inside)
        mx = pfh_block_max(mx, &s_max);
        float scl, inv; pfh_row_scale(mx, scl, inv);
        for (int c = tid; c < H / 8; c += 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = c * 8 + u; v[u] = (r[i] * rms) * (a.bias_one ? (a.w[i] + 1.0f) : a.w[i]); }      // the same expression: the same bits
            u32x4 o;
            o.x = pfh_pack_h2(v[0] * scl, v[4] * scl); o.y = pfh_pack_h2(v[1] * scl, v[5] * scl);      // image order (0,4,1,5,2,6,3,7)
            o.z = pfh_pack_h2(v[2] * scl, v[6] * scl); o.w = pfh_pack_h2(v[3] * scl, v[7] * scl);
            *reinterpret_cast<u32x4*>(a.xf + (size_t)t * H + c * 8) = o;
        }
        if (tid == 0) a.xfm[t] = inv * 0.0625f;
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
// grid (nk, ceil(C / 16)): a workgroup takes one key head and SIXTEEN consecutive tokens, as two tiles of eight; a thread takes FOUR consecutive channels of one tile
// and walks them down the 8 tokens with the 4-tap window in registers: 11 row reads of 16 bytes per lane (round 2-4: one channel per thread, 4-byte reads -- the
// launch moved 180 MB per 2731-token chunk at 1.1 TB/s, 5.9 % of the tolerance prompt pass).  Tap j of token t is X(t-3+j): a chunk row for >= 0, the carried conv
// state slot 4+i for i < 0.  Per value the arithmetic is unchanged: the tap products are added left to right, SiLU with the degree-5 sigmoid, the two L2 norms as
// 8-lane fma chains over the head's conv outputs.  Needs dk % 4 == 0, dv % 4 == 0 and ld_qkvz % 4 == 0 (kr_launch_pfm_la refuses other shapes); any channel count and
// any hr: the channel-quad loop, the z copy and the gate block all stride over the workgroup's threads.
#define PFC_TT 8
#define PFC_WT 16      // tokens per workgroup
__global__ void __launch_bounds__(256) kr_pfm_la_conv_kernel(const KrPfmLaArgs a, int C) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int kh = blockIdx.x, w0 = blockIdx.y * PFC_WT, dk = a.dk, dv = a.dv, hr = a.hr, nk = a.nk;
    const int nw = C - w0 < PFC_WT ? C - w0 : PFC_WT;                       // tokens of this workgroup
    const int group_dim = 2 * dk + 2 * dv * hr, key_dim = nk * dk, nvdk = a.nv * dk, nvdv = a.nv * dv;
    float* qc = sm; float* kc = sm + PFC_WT * dk; float* nrm = sm + 2 * PFC_WT * dk;       // [16][dk], [16][dk], [16][2]
    const int nch = 2 * dk + hr * dv, nq = nch / 4;
    const int half = threadIdx.x >> 7, t0 = w0 + half * PFC_TT;               // this thread's token tile
    const int nt = C - t0 < PFC_TT ? (C - t0 < 0 ? 0 : C - t0) : PFC_TT;
    for (int cq = threadIdx.x & 127; cq < nq; cq += 128) {
        const int c = cq * 4;
        int ch, off;
        if (c < dk) { ch = kh * dk + c; off = c; }
        else if (c < 2 * dk) { ch = key_dim + kh * dk + (c - dk); off = c; }
        else { const int r = (c - 2 * dk) / dv, i = (c - 2 * dk) % dv; ch = 2 * key_dim + (kh * hr + r) * dv + i; off = 2 * dk + r * dv + i; }
        const float* cs = a.conv_state + (size_t)ch * 4;
        float4 cw[4];
#pragma unroll
        for (int q = 0; q < 4; q++) cw[q] = *reinterpret_cast<const float4*>(a.conv_w + (size_t)(ch + q) * 4);
        const float* col = a.qkvz + (size_t)kh * group_dim + off;
        float4 x[PFC_TT + 3];                       // X(t0 - 3) .. X(t0 + 7) of the four channels, all requested before the first use
#pragma unroll
        for (int j = 0; j < PFC_TT + 3; j++) {
            const int i = t0 - 3 + j;
            if (i < 0) x[j] = float4{cs[4 + i], cs[8 + i], cs[12 + i], cs[16 + i]};        // slot 4 + i of channels ch .. ch + 3 (first tile of the chunk only)
            else x[j] = i < C ? *reinterpret_cast<const float4*>(col + (size_t)i * a.ld_qkvz) : float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int tt = 0; tt < PFC_TT; tt++)
            if (tt < nt) {
                float4 co;
                co.x = x[tt].x * cw[0].x + x[tt + 1].x * cw[0].y + x[tt + 2].x * cw[0].z + x[tt + 3].x * cw[0].w;
                co.y = x[tt].y * cw[1].x + x[tt + 1].y * cw[1].y + x[tt + 2].y * cw[1].z + x[tt + 3].y * cw[1].w;
                co.z = x[tt].z * cw[2].x + x[tt + 1].z * cw[2].y + x[tt + 2].z * cw[2].z + x[tt + 3].z * cw[2].w;
                co.w = x[tt].w * cw[3].x + x[tt + 1].w * cw[3].y + x[tt + 2].w * cw[3].z + x[tt + 3].w * cw[3].w;
                co.x = co.x * kr_sigmoid_poly5(co.x); co.y = co.y * kr_sigmoid_poly5(co.y); co.z = co.z * kr_sigmoid_poly5(co.z); co.w = co.w * kr_sigmoid_poly5(co.w);
                const int wt = half * PFC_TT + tt;
                if (c < dk) *reinterpret_cast<float4*>(qc + wt * dk + c) = co;
                else if (c < 2 * dk) *reinterpret_cast<float4*>(kc + wt * dk + (c - dk)) = co;
                else *reinterpret_cast<float4*>(a.v + (size_t)(t0 + tt) * nvdv + (ch - 2 * key_dim)) = co;
            }
    }
    for (int c = threadIdx.x; c < nw * hr * dv / 4; c += 256) {
        const int e = c * 4, tt = e / (hr * dv), rem = e % (hr * dv), r = rem / dv, i = rem % dv;
        *reinterpret_cast<float4*>(a.z + (size_t)(w0 + tt) * nvdv + (size_t)(kh * hr + r) * dv + i) =
            *reinterpret_cast<const float4*>(a.qkvz + (size_t)(w0 + tt) * a.ld_qkvz + (size_t)kh * group_dim + 2 * dk + hr * dv + r * dv + i);
    }
    for (int gi = threadIdx.x; gi < nw * hr; gi += 256) {   // decode.rs:3891-3901; strided: 16 tokens x hr value heads per key head can exceed the 256 threads (hr > 16)
        const int tt = gi / hr, r = gi % hr, vh = kh * hr + r, t = w0 + tt;
        const float* ba = a.ba + (size_t)t * a.ld_ba;
        const float b_raw = ba[kh * 2 * hr + r], a_p = ba[kh * 2 * hr + hr + r];
        a.beta[(size_t)t * a.nv + vh] = 1.0f / (1.0f + kr_expf(-b_raw));
        const float ap_dt = a_p + a.dt_bias[vh];
        const float softplus = ap_dt > 20.0f ? ap_dt : kr_logf(1.0f + kr_expf(ap_dt));
        const float g = -(kr_expf(a.a_log[vh])) * softplus;
        a.gexp[(size_t)t * a.nv + vh] = kr_expf(g);   // decode.rs:1293 decays the state by exp(g)
    }
    __syncthreads();
    if (threadIdx.x < 16 * PFC_WT) {            // 8 lanes per (token, q | k) chain
        const int tt = threadIdx.x >> 4, which = (threadIdx.x >> 3) & 1, l = threadIdx.x & 7;
        if (tt < nw) {
            const float ss = kr_pfm_sumsq8((which ? kc : qc) + tt * dk, dk, l);
            if (l == 0) nrm[tt * 2 + which] = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < nw * hr * dk / 4; c += 256) {
        const int e = c * 4, tt = e / (hr * dk), rem = e % (hr * dk), r = rem / dk, i = rem % dk, vh = kh * hr + r;
        const float inv_q = nrm[tt * 2] * a.scale, inv_k = nrm[tt * 2 + 1] * 1.0f;
        const float4 qv = *reinterpret_cast<const float4*>(qc + tt * dk + i), kv = *reinterpret_cast<const float4*>(kc + tt * dk + i);
        *reinterpret_cast<float4*>(a.q + (size_t)(w0 + tt) * nvdk + (size_t)vh * dk + i) = float4{qv.x * inv_q, qv.y * inv_q, qv.z * inv_q, qv.w * inv_q};
        *reinterpret_cast<float4*>(a.k + (size_t)(w0 + tt) * nvdk + (size_t)vh * dk + i) = float4{kv.x * inv_k, kv.y * inv_k, kv.z * inv_k, kv.w * inv_k};
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
// The two DK-long fma chains per token are inherent to the reference order, and a dependent v_fmac issues only every ~15 cycles on this part: run one after the other they
// cost 2 x DK x 15 cycles per token (1.8 us at DK = 128: the round-5 form).  Round 6: the OUTPUT chain of token t and the kv chain of token t + 1 are independent once
// S(t)[i] exists, so they advance together, element by element -- update S[i] with delta_t, step o_t, decay S[i] by e^g(t+1), step kv_(t+1): four instructions per element,
// two chains in each other's latency shadow; every value is produced by the same operation in the same order as before (bit-identical).  Behind the last token runs a
// phantom one with e^g = 1 (S * 1.0f is S) whose chain nobody reads.  The rows (k, q, v) and the two gate scalars of a token reach an LDS ring of RS slots by LDS-DMA
// (global_load_lds_dword: lane i's word lands at M0 + 4 i), requested three tokens ahead: no staging registers, one loop body; the state slice is addressed through a
// buffer descriptor (scalar row offsets) so that no per-row 64-bit addresses stay live across the loop.
// (Measured earlier and slower: LDS-staged token blocks; one packed fma advancing both chains on a {S*e^g, S} register pair -- 2 x DK registers; the same interleave with
// the rows prefetched into rotating register sets -- three copies of the loop, 328 registers.)
__device__ __forceinline__ void pfm_dma4(uint32_t lds_dst, const void* gptr) {      // this lane's 4 bytes at gptr -> lds_dst + 4 * lane; not counted by the compiler's waits
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(lds_dst), "v"(gptr) : "memory");
}
template <int DK>
__global__ void __launch_bounds__(256) kr_pfm_la_recur_kernel(float* __restrict__ state, const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, const float* __restrict__ gexp, const float* __restrict__ beta,
                                                             float* __restrict__ out, int nv, int dv, int C) {
    constexpr int RS = 5, ROW = 256;                 // ring slots; floats per row region (dv <= 256)
    constexpr int SLOT = 3 * ROW + 128;              // floats per slot: k row, q row, v row, e^g at [3 ROW], beta at [3 ROW + 1] (wave 0's other lanes land behind them), a 64-word dump at [3 ROW + 64]
    __shared__ __attribute__((aligned(16))) float ring[RS][SLOT];
    const int h = blockIdx.x, j = threadIdx.x, lane = j & 63, wv = __builtin_amdgcn_readfirstlane(j >> 6), nvdk = nv * DK, nvdv = nv * dv;
    float S[DK];
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(state + (size_t)h * DK * dv, 0, DK * dv * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < DK; i++) S[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srd, j * 4, i * dv * 4, 0));
    const uint32_t ring0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)&ring[0][0];
    const bool kq_wave = wv * 64 < DK;                                   // wave-uniform: this wave's columns have k / q words
    const int jk = kq_wave ? j : lane;                                    // other waves fetch (harmless) words into the slot's spare area: every wave issues 5 DMAs per token
    auto request = [&](int tt, int slot) {           // token tt (clamped: a phantom token behind the chunk re-reads the last one; its gates are forced below)
        const int tc = tt < C ? tt : C - 1;
        const uint32_t base = ring0 + (uint32_t)slot * (SLOT * 4), dump = base + (3 * ROW + 64) * 4;
        pfm_dma4(kq_wave ? base + wv * 256 : dump, k + (size_t)tc * nvdk + (size_t)h * DK + jk);
        pfm_dma4(kq_wave ? base + ROW * 4 + wv * 256 : dump, q + (size_t)tc * nvdk + (size_t)h * DK + jk);
        pfm_dma4(base + 2 * ROW * 4 + wv * 256, v + (size_t)tc * nvdv + (size_t)h * dv + j);
        pfm_dma4(wv == 0 ? base + 3 * ROW * 4 : dump, (lane == 0 ? gexp : beta) + (size_t)tc * nv + h);      // wave 0: lane 0 -> e^g, lane 1 -> beta (lanes 2 .. 63 land behind them, unread)
        pfm_dma4(dump, beta + (size_t)tc * nv + h);                                                          // filler: every wave issues exactly 5 requests per token (the wait below counts them)
    };
    request(0, 0); request(1, 1); request(2, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // prologue: decay + kv chain of token 0 alone
    float delta;
    {
        const float4* k4 = reinterpret_cast<const float4*>(&ring[0][0]);
        const float ge0 = ring[0][3 * ROW], bt0 = ring[0][3 * ROW + 1], vj0 = ring[0][2 * ROW + j];
        float kv = 0.0f;
#pragma unroll
        for (int u = 0; u < DK / 4; u++) {
            const float4 kk = k4[u];
            float* Sb = S + 4 * u;
            Sb[0] = Sb[0] * ge0; kv = __builtin_fmaf(Sb[0], kk.x, kv);
            Sb[1] = Sb[1] * ge0; kv = __builtin_fmaf(Sb[1], kk.y, kv);
            Sb[2] = Sb[2] * ge0; kv = __builtin_fmaf(Sb[2], kk.z, kv);
            Sb[3] = Sb[3] * ge0; kv = __builtin_fmaf(Sb[3], kk.w, kv);
        }
        delta = (vj0 - kv) * bt0;
    }
    int slot = 0;
    for (int t = 0; t < C; t++) {
        // token t: S update + output chain of token t, together with decay + kv chain of token t + 1; blocks of 16 elements, the next block's rows leave LDS while the
        // current block's steps run (the scheduling fences keep the reads from sinking next to their uses)
        const int s1 = slot + 1 >= RS ? slot + 1 - RS : slot + 1, s3 = slot + 3 >= RS ? slot + 3 - RS : slot + 3;
        request(t + 3, s3);          // slot s3 held token t - 2: last read two barriers ago
        const float4* k4 = reinterpret_cast<const float4*>(&ring[slot][0]); const float4* q4 = reinterpret_cast<const float4*>(&ring[slot][ROW]);
        const float4* n4 = reinterpret_cast<const float4*>(&ring[s1][0]);
        const bool has_next = t + 1 < C;
        const float ge1 = has_next ? ring[s1][3 * ROW] : 1.0f, bt1 = ring[s1][3 * ROW + 1], vj1 = ring[s1][2 * ROW + j];
        float ob = 0.0f, kv = 0.0f;
        float4 ka[4] = {k4[0], k4[1], k4[2], k4[3]}, qa[4] = {q4[0], q4[1], q4[2], q4[3]}, na[4] = {n4[0], n4[1], n4[2], n4[3]};
#pragma unroll
        for (int b = 0; b < DK / 16; b++) {
            float4 kb[4] = {ka[0], ka[1], ka[2], ka[3]}, qb[4] = {qa[0], qa[1], qa[2], qa[3]}, nb[4] = {na[0], na[1], na[2], na[3]};
            if (b + 1 < DK / 16) {
#pragma unroll
                for (int u = 0; u < 4; u++) { kb[u] = k4[4 * b + 4 + u]; qb[u] = q4[4 * b + 4 + u]; nb[u] = n4[4 * b + 4 + u]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float* Sb = S + 16 * b + 4 * u;
                const float kk[4] = {ka[u].x, ka[u].y, ka[u].z, ka[u].w}, qq[4] = {qa[u].x, qa[u].y, qa[u].z, qa[u].w}, nn[4] = {na[u].x, na[u].y, na[u].z, na[u].w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    Sb[e] = __builtin_fmaf(kk[e], delta, Sb[e]); ob = __builtin_fmaf(Sb[e], qq[e], ob);
                    Sb[e] = Sb[e] * ge1; kv = __builtin_fmaf(Sb[e], nn[e], kv);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; u++) { ka[u] = kb[u]; qa[u] = qb[u]; na[u] = nb[u]; }
        }
        out[(size_t)t * nvdv + (size_t)h * dv + j] = ob;
        delta = (vj1 - kv) * bt1;
        // the next token reads slots t + 1 and t + 2: this wave's requests for token t + 2 (issued one token ago) are older than its 5 requests for t + 3 and the two
        // output stores since -- at most 7 vector-memory operations may still be in flight
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        __syncthreads();
        slot = s1;
    }
#pragma unroll
    for (int i = 0; i < DK; i++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(S[i]), srd, j * 4, i * dv * 4, 0);
}

// ---- gated RMSNorm + SiLU gate per (token, head) (decode.rs:3979) ----------------------------------------------------------------
// grid (nv, ceil(C / 8)), 256 threads: a WAVE takes one (token, head) row at a time (two rows per wave: 8 tokens per workgroup), the row in a wave-private LDS slice,
// the sum of squares as the reference's 8-lane chain + hsum (kr_pfm_sumsq8) on lanes 0..7, no workgroup barrier.  (Rounds 1-5: one workgroup of dv threads per
// (token, head) -- 262 144 workgroups of two waves for an 8192-token chunk, 140 us against ~80 us of traffic.)  z may sit inside the in-projection's output
// (z_ld = its row stride, z_head = floats between the z blocks of consecutive value heads): the stand-alone copy of z is then never made.
#define PFG_TT 8
__global__ void __launch_bounds__(256) kr_pfm_gated_norm_kernel(const float* __restrict__ recur, const float* __restrict__ z, size_t z_ld, int z_hr, int z_kstride, const float* __restrict__ w,
                                                               float* __restrict__ out, int nv, int dv, int C, float eps) {
    __shared__ __attribute__((aligned(16))) float rs[4][256 + 8];
    const int h = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* r = rs[wave];
    const int nc = dv / 4;                                          // 16-byte chunks of a row (dv % 8 == 0, dv <= 256: one chunk per lane at most)
    // z of value head h: block (h / z_hr) of z_kstride floats, then head h % z_hr inside it (plain [C][nv * dv] rows: z_hr = nv, one block)
    const size_t zoff = (size_t)(h / z_hr) * z_kstride + (size_t)(h % z_hr) * dv;
    const bool act = lane < nc;
    const float4 wv = act ? *reinterpret_cast<const float4*>(w + (size_t)h * dv + 4 * lane) : float4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int tw = 0; tw < PFG_TT / 4; tw++) {
        const int t = blockIdx.y * PFG_TT + tw * 4 + wave;
        if (t >= C) break;                                        // wave-uniform
        const size_t o = (size_t)t * nv * dv + (size_t)h * dv;
        float4 v = float4{0.0f, 0.0f, 0.0f, 0.0f}, zz = v;
        if (act) { v = *reinterpret_cast<const float4*>(recur + o + 4 * lane); zz = *reinterpret_cast<const float4*>(z + (size_t)t * z_ld + zoff + 4 * lane); *reinterpret_cast<float4*>(r + 4 * lane) = v; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
        if (lane < 8) { const float ss = kr_pfm_sumsq8(r, dv, lane); if (lane == 0) r[256] = 1.0f / sqrtf(ss / (float)dv + eps); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
        const float rms = r[256];
        if (act) {
            float4 ov;
            ov.x = (zz.x * kr_sigmoid_poly5(zz.x)) * ((v.x * rms) * wv.x); ov.y = (zz.y * kr_sigmoid_poly5(zz.y)) * ((v.y * rms) * wv.y);
            ov.z = (zz.z * kr_sigmoid_poly5(zz.z)) * ((v.z * rms) * wv.z); ov.w = (zz.w * kr_sigmoid_poly5(zz.w)) * ((v.w * rms) * wv.w);
            *reinterpret_cast<float4*>(out + o + 4 * lane) = ov;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();      // the slice is rewritten by the wave's next row
    }
}

// ---- GQA: gated split, per-head RMS norm, RoPE, FP16 KV append for every token (decode.rs:2873-2966); grid (nh + nkv, C) ---------
// (round 6) grid (nh + nkv, ceil(C / 8)), 256 threads: a WAVE takes one (head, token) row at a time (two per wave: 8 tokens per workgroup) in a wave-private LDS slice;
// the per-head RMS norm's sum of squares stays the reference's sequential sum (lane 0, values read four at a time).  Rounds 1-5 ran one 256-thread workgroup per row --
// 147 k workgroups per 8192-token chunk with ONE lane busy in the sum: 341 us per launch.
#define PFQ_TT 8
__global__ void __launch_bounds__(256) kr_pfm_gqa_prep_kernel(const KrPfmGqaArgs a, int C) {
    __shared__ __attribute__((aligned(16))) float xs[4][256 + 8];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hd = a.hd;
    const bool is_q = b < a.nh;
    const int h = is_q ? b : b - a.nh;
    float* x = xs[wave];
    const int nc = hd / 4, njc = (nc + 63) / 64;                  // 16-byte chunks of a row, chunks per lane (hd % 32 == 0: the launcher checks)
    const float* nw = is_q ? a.q_norm : a.k_norm;
    const int per_head = is_q ? a.q_norm_per_head : a.k_norm_per_head;
    const int d2 = a.rope_half;
    for (int tw = 0; tw < PFQ_TT / 4; tw++) {
        const int t = blockIdx.y * PFQ_TT + tw * 4 + wave;
        if (t >= C) break;                                        // wave-uniform
        const int pos = a.pos0 + t;
        const float* q_in = a.q_in + (size_t)t * a.ld_q; const float* k_in = a.k_in + (size_t)t * a.ld_k; const float* v_in = a.v_in + (size_t)t * a.ld_v;
        float4 vv[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {                             // hd <= 256: at most one chunk per lane (two iterations keep hd up to 512 correct)
            const int ci = lane + 64 * j;
            vv[j] = float4{0.0f, 0.0f, 0.0f, 0.0f};
            if (j < njc && ci < nc) {
                float4 xv;
                if (is_q) {
                    if (a.gated) {
                        xv = *reinterpret_cast<const float4*>(q_in + (size_t)h * hd * 2 + 4 * ci);
                        *reinterpret_cast<float4*>(a.gate + (size_t)t * a.nh * hd + (size_t)h * hd + 4 * ci) = *reinterpret_cast<const float4*>(q_in + (size_t)h * hd * 2 + hd + 4 * ci);
                    } else xv = *reinterpret_cast<const float4*>(q_in + (size_t)h * hd + 4 * ci);
                } else { xv = *reinterpret_cast<const float4*>(k_in + (size_t)h * hd + 4 * ci); vv[j] = *reinterpret_cast<const float4*>(v_in + (size_t)h * hd + 4 * ci); }
                *reinterpret_cast<float4*>(x + 4 * ci) = xv;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
        if (nw) {
            if (lane == 0) {
                float ss = 0.0f;
                const float4* x4 = reinterpret_cast<const float4*>(x);
                for (int i = 0; i < nc; i++) { const float4 v = x4[i]; ss += v.x * v.x; ss += v.y * v.y; ss += v.z * v.z; ss += v.w * v.w; }      // index order (decode.rs:2891)
                x[hd] = 1.0f / sqrtf(ss / (float)hd + a.eps);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
            const float rms = x[hd];
#pragma unroll
            for (int j = 0; j < 2; j++) {                         // a lane rewrites only the chunk it owns: no hazard inside the wave
                const int ci = lane + 64 * j;
                if (j < njc && ci < nc) {
                    const float4 xv = *reinterpret_cast<const float4*>(x + 4 * ci);
                    const float4 wv = *reinterpret_cast<const float4*>(nw + (per_head ? h * hd : 0) + 4 * ci);
                    *reinterpret_cast<float4*>(x + 4 * ci) = float4{xv.x * (rms * wv.x), xv.y * (rms * wv.y), xv.z * (rms * wv.z), xv.w * (rms * wv.w)};
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int ci = lane + 64 * j;
            if (j < njc && ci < nc) {
                float val[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int d = 4 * ci + e;
                    val[e] = x[d];
                    if (d < 2 * d2) {
                        const float c = a.rope_cos[(size_t)pos * d2 + (d % d2)], sn = a.rope_sin[(size_t)pos * d2 + (d % d2)];
                        if (d < d2) val[e] = x[d] * c - x[d2 + d] * sn;
                        else val[e] = x[d] * c + x[d - d2] * sn;
                    }
                }
                if (is_q) *reinterpret_cast<float4*>(a.q_out + (size_t)t * a.nh * hd + (size_t)h * hd + 4 * ci) = float4{val[0], val[1], val[2], val[3]};
                else {
                    const size_t o = (size_t)pos * a.nkv * hd + (size_t)h * hd + 4 * ci;
                    const float ve[4] = {vv[j].x, vv[j].y, vv[j].z, vv[j].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) { kr_kv_store(a.k_cache, o + e, val[e], a.kv_fp8); kr_kv_store(a.v_cache, o + e, ve[e], a.kv_fp8); }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();      // the slice is rewritten by the wave's next row
    }
}

// ---- causal attention over the FP16 cache, exact decode order (decode.rs:4194-4281), three passes that share K / V across queries ----
// A query is (token t, head h); the R = group * TT queries of one kv head and TT consecutive tokens form a tile.  Scores live in an
// HBM scratch sc[row = t*nh + h][pos] (row stride sc_ld) so the softmax keeps the reference's max / exp / sequential-sum / scale
// order over the WHOLE row, and P.V accumulates position by position.
#define PFA_TT_MAX 32          // queries per tile (registers of pass C)
#define PFA_LDB 36             // floats per lane record (32 + pad) in the transposed LDS images

// pass A: scores.  grid (position tiles of 256, token tiles, nkv); 256 threads = 32 lane-groups of 8; a lane-group owns one position
// per pass, keeps that K row in registers (lane l: elements 8b + l) and evaluates every query of the tile against it:
//   s = hsum8( chain_b fma(q[8b+l], k[8b+l]) ) * sm_scale      (single accumulator, decode.rs:4229-4242)
__global__ void __launch_bounds__(256) kr_pfm_gqa_scores_kernel(const KrPfmGqaArgs a, float* __restrict__ sc, int sc_ld, int TT, int C) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int hd = a.hd, nb = hd / 8, group = a.nh / a.nkv, kvh = blockIdx.z, t0 = blockIdx.y * TT, kvs = a.nkv * hd;
    const int tn = min(TT, C - t0), R = group * tn;
    const int p_lo = blockIdx.x * 256, p_max = a.pos0 + t0 + tn - 1;     // last position any query of the tile may see
    if (p_lo > p_max) return;
    float* qT = lds;                              // [R][8 lanes][PFA_LDB]   q[r][8b + l] at (r*8 + l)*PFA_LDB + b
    float* kT = lds + (size_t)group * TT * 8 * PFA_LDB;   // [32 positions][8 lanes][PFA_LDB]
    const int tid = threadIdx.x, g = tid >> 3, l = tid & 7;
    for (int i = tid; i < R * hd; i += 256) {
        const int r = i / hd, d = i % hd, tt = r / group, hh = kvh * group + r % group;
        qT[(r * 8 + (d & 7)) * PFA_LDB + (d >> 3)] = a.q_out[(size_t)(t0 + tt) * a.nh * hd + (size_t)hh * hd + d];
    }
    for (int pass = 0; pass < 8; pass++) {
        const int pb = p_lo + pass * 32;
        if (pb > p_max) break;
        __syncthreads();                          // q staged / previous K tile consumed
        for (int i = tid; i < 32 * (hd / 8); i += 256) {   // 16-byte loads: 8 halves = one b-block of one position
            const int pp = i / (hd / 8), b = i % (hd / 8), pos = pb + pp;
            if (a.kv_fp8) {       // 8 bytes = 8 e4m3 values of one b-block
                u32x2 w = {0, 0};
                if (pos <= p_max) w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint8_t*>(a.k_cache) + (size_t)pos * kvs + (size_t)kvh * hd + b * 8);
                const uint32_t ww[2] = {w.x, w.y};
#pragma unroll
                for (int j = 0; j < 8; j++) kT[(pp * 8 + j) * PFA_LDB + b] = kr_e4m3_to_f32((uint8_t)(ww[j >> 2] >> (8 * (j & 3))));
            } else {
                u32x4 w = {0, 0, 0, 0};
                if (pos <= p_max) w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(a.k_cache) + (size_t)pos * kvs + (size_t)kvh * hd + b * 8);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    kT[(pp * 8 + 2 * j) * PFA_LDB + b] = __half2float(__ushort_as_half((uint16_t)(ww[j] & 0xFFFFu)));
                    kT[(pp * 8 + 2 * j + 1) * PFA_LDB + b] = __half2float(__ushort_as_half((uint16_t)(ww[j] >> 16)));
                }
            }
        }
        __syncthreads();
        const int pos = pb + g;
        float kr[32];
#pragma unroll
        for (int b4 = 0; b4 < 8; b4++) {
            if (b4 * 4 < nb) {
                const float4 v = *reinterpret_cast<const float4*>(kT + (g * 8 + l) * PFA_LDB + b4 * 4);
                kr[b4 * 4] = v.x; kr[b4 * 4 + 1] = v.y; kr[b4 * 4 + 2] = v.z; kr[b4 * 4 + 3] = v.w;
            }
        }
        for (int r0 = 0; r0 < R; r0 += 4) {       // 4 queries at a time: 4 independent fma chains per lane
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const float* qb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) qb[u] = qT + ((r0 + u < R ? r0 + u : R - 1) * 8 + l) * PFA_LDB;
#pragma unroll
            for (int b4 = 0; b4 < 8; b4++) {
                if (b4 * 4 < nb) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float4 qv = *reinterpret_cast<const float4*>(qb[u] + b4 * 4);
                        acc[u] = __builtin_fmaf(qv.x, kr[b4 * 4], acc[u]); acc[u] = __builtin_fmaf(qv.y, kr[b4 * 4 + 1], acc[u]);
                        acc[u] = __builtin_fmaf(qv.z, kr[b4 * 4 + 2], acc[u]); acc[u] = __builtin_fmaf(qv.w, kr[b4 * 4 + 3], acc[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + u, tt = r / group, hh = kvh * group + r % group, qpos = a.pos0 + t0 + tt;
                const float sv = kr_pfm_hsum8(acc[u]);
                if (r < R && l == 0 && pos <= qpos) sc[((size_t)(t0 + tt) * a.nh + hh) * sc_ld + pos] = sv * a.sm_scale;
            }
        }
    }
}

// pass A, specialised: head_dim / 8 and the KV dtype are template parameters, so the per-lane loops carry no uniform guards (each
// guarded LDS / global read would sit in its own basic block behind a wait), and the NEXT 32-position K tile is fetched into
// registers while the current one is evaluated.  Same arithmetic, same order as the generic kernel above.
template <bool FP8, int NB>
__global__ void __launch_bounds__(256) kr_pfm_gqa_scores_t_kernel(const KrPfmGqaArgs a, float* __restrict__ sc, int sc_ld, int TT, int C) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int hd = NB * 8, NCH = 32 * NB / 256;            // 16-byte (FP16) / 8-byte (E4M3) chunks of a K tile per thread
    const int group = a.nh / a.nkv, kvh = blockIdx.z, t0 = blockIdx.y * TT, kvs = a.nkv * hd;
    const int tn = min(TT, C - t0), R = group * tn;
    const int p_lo = blockIdx.x * 256, p_max = a.pos0 + t0 + tn - 1;     // last position any query of the tile may see
    if (p_lo > p_max) return;
    float* qT = lds;                              // [R][8 lanes][PFA_LDB]   q[r][8b + l] at (r*8 + l)*PFA_LDB + b
    float* kT = lds + (size_t)group * TT * 8 * PFA_LDB;   // [32 positions][8 lanes][PFA_LDB]
    const int tid = threadIdx.x, g = tid >> 3, l = tid & 7;
    u32x4 kw[NCH];
    auto issue_k = [&](int pb) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int i = tid + 256 * c, pp = i / NB, b = i % NB, pos = pb + pp;
            kw[c] = u32x4{0, 0, 0, 0};
            if (pos <= p_max) {
                if (FP8) { const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint8_t*>(a.k_cache) + (size_t)pos * kvs + (size_t)kvh * hd + b * 8); kw[c].x = w.x; kw[c].y = w.y; }
                else kw[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(a.k_cache) + (size_t)pos * kvs + (size_t)kvh * hd + b * 8);
            }
        }
    };
    auto commit_k = [&]() {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int i = tid + 256 * c, pp = i / NB, b = i % NB;
            const uint32_t ww[4] = {kw[c].x, kw[c].y, kw[c].z, kw[c].w};
            if (FP8) {
#pragma unroll
                for (int j = 0; j < 8; j++) kT[(pp * 8 + j) * PFA_LDB + b] = kr_e4m3_to_f32((uint8_t)(ww[j >> 2] >> (8 * (j & 3))));
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    kT[(pp * 8 + 2 * j) * PFA_LDB + b] = __half2float(__ushort_as_half((uint16_t)(ww[j] & 0xFFFFu)));
                    kT[(pp * 8 + 2 * j + 1) * PFA_LDB + b] = __half2float(__ushort_as_half((uint16_t)(ww[j] >> 16)));
                }
            }
        }
    };
    issue_k(p_lo);
    for (int i = tid; i < R * hd; i += 256) {
        const int r = i / hd, d = i % hd, tt = r / group, hh = kvh * group + r % group;
        qT[(r * 8 + (d & 7)) * PFA_LDB + (d >> 3)] = a.q_out[(size_t)(t0 + tt) * a.nh * hd + (size_t)hh * hd + d];
    }
    for (int pass = 0; pass < 8; pass++) {
        const int pb = p_lo + pass * 32;
        if (pb > p_max) break;
        __syncthreads();                          // q staged / previous K tile consumed
        commit_k();
        if (pass + 1 < 8 && pb + 32 <= p_max) issue_k(pb + 32);
        __syncthreads();
        const int pos = pb + g;
        float kr[NB];
#pragma unroll
        for (int b4 = 0; b4 < NB / 4; b4++) {
            const float4 v = *reinterpret_cast<const float4*>(kT + (g * 8 + l) * PFA_LDB + b4 * 4);
            kr[b4 * 4] = v.x; kr[b4 * 4 + 1] = v.y; kr[b4 * 4 + 2] = v.z; kr[b4 * 4 + 3] = v.w;
        }
        for (int r0 = 0; r0 < R; r0 += 4) {       // 4 queries at a time: 4 independent fma chains per lane
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const float* qb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) qb[u] = qT + ((r0 + u < R ? r0 + u : R - 1) * 8 + l) * PFA_LDB;
#pragma unroll
            for (int b4 = 0; b4 < NB / 4; b4++) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float4 qv = *reinterpret_cast<const float4*>(qb[u] + b4 * 4);
                    acc[u] = __builtin_fmaf(qv.x, kr[b4 * 4], acc[u]); acc[u] = __builtin_fmaf(qv.y, kr[b4 * 4 + 1], acc[u]);
                    acc[u] = __builtin_fmaf(qv.z, kr[b4 * 4 + 2], acc[u]); acc[u] = __builtin_fmaf(qv.w, kr[b4 * 4 + 3], acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + u, tt = r / group, hh = kvh * group + r % group, qpos = a.pos0 + t0 + tt;
                const float sv = kr_pfm_hsum8(acc[u]);
                if (r < R && l == 0 && pos <= qpos) sc[((size_t)(t0 + tt) * a.nh + hh) * sc_ld + pos] = sv * a.sm_scale;
            }
        }
    }
}

// pass B: per row  max -> e = exp(s - max) (libm) -> sequential sum in position order -> inv = 1 / sum  (decode.rs:4244-4262).
// one wave per row: 64 positions at a time are exponentiated in parallel, lane 0 adds them in order.
__global__ void __launch_bounds__(256) kr_pfm_gqa_softmax_kernel(float* __restrict__ sc, int sc_ld, float* __restrict__ inv, int nh, int pos0, int rows,
                                                                 const float* __restrict__ tmax) {
    __shared__ __attribute__((aligned(16))) float buf[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int seq = pos0 + row / nh + 1;
    float* s = sc + (size_t)row * sc_ld;
    float mx = -__builtin_inff();
    if (tmax) {            // pass A left the maximum of every 32 positions (the maximum is the same whatever the order it is taken in)
        const float* tm = tmax + (size_t)row * (sc_ld >> 5);
        for (int b = lane; b < (seq + 31) >> 5; b += 64) mx = fmaxf(mx, tm[b]);
    } else for (int p = lane; p < seq; p += 64) mx = fmaxf(mx, s[p]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float se = 0.0f;
    for (int p0 = 0; p0 < seq; p0 += 64) {
        const int p = p0 + lane;
        float e = 0.0f;
        if (p < seq) { e = kr_expf(s[p] - mx); s[p] = e; }
        buf[wave][lane] = e;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const int n = min(64, seq - p0);
            if (n == 64) {
                const float4* b4 = reinterpret_cast<const float4*>(buf[wave]);
#pragma unroll
                for (int u = 0; u < 16; u++) { const float4 v = b4[u]; se += v.x; se += v.y; se += v.z; se += v.w; }
            } else for (int u = 0; u < n; u++) se += buf[wave][u];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    }
    // sc[s] *= inv (decode.rs:4260): the row leaves this pass as probabilities, so the vector pass C reads them as they are; with tmax (matrix-core
    // passes) the row stays as exponentials and pass C forms e * inv itself -- one read and one write of the scratch less
    const float iv = __shfl(1.0f / se, 0);
    if (lane == 0) inv[row] = iv;
    if (!tmax) for (int p = lane; p < seq; p += 64) s[p] = s[p] * iv;
}

// pass C: out[t][h][d] = chain_pos fma(p[pos], V[pos][d]) (decode.rs:4264-4273), gated by sigmoid(gate) (:4277).
// grid (token tiles, nkv), ONE wave per workgroup; lane l owns CPL = head_dim / 64 adjacent value columns and keeps one accumulator
// per (query of the tile, column).  The probabilities are the same for every lane, so each one read from the LDS image (a broadcast
// ds_read_b128 carries 4 of them) feeds CPL fma per lane: with 4 columns per lane the LDS return path (1 KiB per wave64 b128 read,
// 8 LDS cycles) stays at half the VALU time instead of twice it (the 1-column form), and the scalar-load form (3.4 ms per chunk) was
// bound by the scalar cache streaming a 1 GB score scratch.  The next 64-position tile of probabilities is fetched into registers
// during the current tile (double-buffered LDS image), the next 8 V rows are in flight while the current 8 are consumed, and the KV
// dtype / GQA group / columns per lane are template parameters.  Every accumulator sees its positions in ascending order.
template <bool FP8, int CPL> __device__ __forceinline__ void kr_pfm_v_load(const void* base, size_t i, float (&v)[CPL]) {
    if (FP8) {
        const uint8_t* p = reinterpret_cast<const uint8_t*>(base) + i;
        uint32_t w;
        if (CPL == 4) w = *reinterpret_cast<const uint32_t*>(p); else if (CPL == 2) w = *reinterpret_cast<const uint16_t*>(p); else w = *p;
#pragma unroll
        for (int c = 0; c < CPL; c++) v[c] = kr_e4m3_to_f32((uint8_t)(w >> (8 * c)));
    } else {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(base) + i;
        uint32_t w0 = 0, w1 = 0;
        if (CPL == 4) { const u32x2 w = *reinterpret_cast<const u32x2*>(p); w0 = w.x; w1 = w.y; }
        else if (CPL == 2) w0 = *reinterpret_cast<const uint32_t*>(p);
        else w0 = *p;
#pragma unroll
        for (int c = 0; c < CPL; c++) { const uint32_t w = c < 2 ? w0 : w1; v[c] = __half2float(__ushort_as_half((uint16_t)(w >> (16 * (c & 1))))); }
    }
}
#define PFV_R 16               // queries per pass-C workgroup (one wave): 64 accumulators per lane at 4 columns, two waves per SIMD
template <bool FP8, int GROUP, int CPL>     // GROUP = query heads per KV head (0 = run-time value); CPL = head_dim / 64
__global__ void __launch_bounds__(64) kr_pfm_gqa_pv_kernel(const KrPfmGqaArgs a, const float* __restrict__ sc, int sc_ld, int TT, int C) {
    __shared__ __attribute__((aligned(16))) float P[2][PFV_R][64];
    const int hd = a.hd, group = GROUP ? GROUP : a.nh / a.nkv, kvh = blockIdx.y, t0 = blockIdx.x * TT, kvs = a.nkv * hd;
    const int tn = min(TT, C - t0), R = group * tn, lane = threadIdx.x, p_max = a.pos0 + t0 + tn - 1;
    float acc[PFV_R][CPL];
#pragma unroll
    for (int r = 0; r < PFV_R; r++)
#pragma unroll
        for (int c = 0; c < CPL; c++) acc[r][c] = 0.0f;
    const size_t vcb = (size_t)kvh * hd + (size_t)lane * CPL;
    // probability tile p0: row r = query r of the tile, column = position p0 + lane
    // buffer addressing: descriptor = this tile's rows, voffset = lane, soffset = row / tile offset (scalar) -- 32 row pointers would
    // otherwise stay live in 64 VGPRs
    const __amdgpu_buffer_rsrc_t psrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc) + ((size_t)t0 * a.nh + (size_t)kvh * group) * sc_ld, 0,
                                                                          (int)(((size_t)(tn - 1) * a.nh + group) * sc_ld * 4), 0x00020000);
    float pf[PFV_R];
    auto fetch_p = [&](int p0) {
        const int pos = p0 + lane;
#pragma unroll
        for (int r = 0; r < PFV_R; r++) {
            const int tt = r / group, g = r % group;
            const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(psrd, lane * 4, ((tt * a.nh + g) * sc_ld + p0) * 4, 0));   // rows past the tile: out of range -> 0
            pf[r] = (r < R && pos <= a.pos0 + t0 + tt) ? v : 0.0f;
        }
    };
    auto commit_p = [&](int buf) {
#pragma unroll
        for (int r = 0; r < PFV_R; r++) P[buf][r][lane] = pf[r];
    };
    fetch_p(0);
    commit_p(0);
    float va[8][CPL], vb[8][CPL];
    // V rows are loaded UNCONDITIONALLY (row index clamped to p_max; rows past it are never consumed): a guard around a load + its
    // conversion puts each load in its own basic block behind a wait -- one exposed memory latency per position
#pragma unroll
    for (int u = 0; u < 8; u++) kr_pfm_v_load<FP8, CPL>(a.v_cache, vcb + (size_t)min(u, p_max) * kvs, va[u]);
    int cur = 0;
    for (int p0 = 0; p0 <= p_max; p0 += 64, cur ^= 1) {
        __syncthreads();                           // one wave: P[cur] committed, P[cur ^ 1] free
        const bool more = p0 + 64 <= p_max;
        if (more) fetch_p(p0 + 64);
        const bool full = p0 + 63 <= a.pos0 + t0 && R == PFV_R;   // tile entirely below the diagonal, full query tile: no masks
        const int np = min(64, p_max + 1 - p0);
#pragma unroll 1
        for (int pp0 = 0; pp0 < 64; pp0 += 8) {
            if (pp0 >= np) break;
            const int nx = p0 + pp0 + 8;           // next 8 rows (may belong to the next tile); rows past p_max are never used
#pragma unroll
            for (int u = 0; u < 8; u++) kr_pfm_v_load<FP8, CPL>(a.v_cache, vcb + (size_t)min(nx + u, p_max) * kvs, vb[u]);
            if (full) {
                // rows in groups of 4, the next group's probabilities (8 broadcast ds_read_b128) in flight under the current group's
                // 4 * 8 * CPL fma; the scheduling barriers keep the compiler from hoisting all 64 reads (256 registers) up front
                float4 pa[8], pb[8];
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) { pa[2 * q4] = *reinterpret_cast<const float4*>(&P[cur][q4][pp0]); pa[2 * q4 + 1] = *reinterpret_cast<const float4*>(&P[cur][q4][pp0 + 4]); }
#pragma unroll
                for (int r0 = 0; r0 < PFV_R; r0 += 4) {
                    if (r0 + 4 < PFV_R) {
#pragma unroll
                        for (int q4 = 0; q4 < 4; q4++) { pb[2 * q4] = *reinterpret_cast<const float4*>(&P[cur][r0 + 4 + q4][pp0]); pb[2 * q4 + 1] = *reinterpret_cast<const float4*>(&P[cur][r0 + 4 + q4][pp0 + 4]); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // position-major issue order: the 4 * CPL accumulators of the group advance one position at a time, so consecutive
                    // instructions are independent (each accumulator still takes its 8 positions in ascending order)
                    float pr[4][8];
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        pr[q4][0] = pa[2 * q4].x; pr[q4][1] = pa[2 * q4].y; pr[q4][2] = pa[2 * q4].z; pr[q4][3] = pa[2 * q4].w;
                        pr[q4][4] = pa[2 * q4 + 1].x; pr[q4][5] = pa[2 * q4 + 1].y; pr[q4][6] = pa[2 * q4 + 1].z; pr[q4][7] = pa[2 * q4 + 1].w;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++)
#pragma unroll
                        for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
                            for (int c = 0; c < CPL; c++) acc[r0 + q4][c] = __builtin_fmaf(pr[q4][u], va[u][c], acc[r0 + q4][c]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q8 = 0; q8 < 8; q8++) pa[q8] = pb[q8];
                }
            } else {
#pragma unroll
                for (int r = 0; r < PFV_R; r++) {
                    if (r < R) {
                        const int lim = min(a.pos0 + t0 + r / group - (p0 + pp0), np - 1 - pp0);   // positions pp0 + u with u <= lim are visible to query r
                        const float4 p0v = *reinterpret_cast<const float4*>(&P[cur][r][pp0]), p1v = *reinterpret_cast<const float4*>(&P[cur][r][pp0 + 4]);
                        const float pr[8] = {p0v.x, p0v.y, p0v.z, p0v.w, p1v.x, p1v.y, p1v.z, p1v.w};
#pragma unroll
                        for (int u = 0; u < 8; u++)
                            if (u <= lim)
#pragma unroll
                                for (int c = 0; c < CPL; c++) acc[r][c] = __builtin_fmaf(pr[u], va[u][c], acc[r][c]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int c = 0; c < CPL; c++) va[u][c] = vb[u][c];
        }
        if (more) commit_p(cur ^ 1);
    }
#pragma unroll
    for (int r = 0; r < PFV_R; r++) {
        if (r < R) {
            const int tt = r / group, hh = kvh * group + r % group;
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                float o = acc[r][c];
                const size_t oi = (size_t)(t0 + tt) * a.nh * hd + (size_t)hh * hd + (size_t)lane * CPL + c;
                if (a.gated) { const float gt = a.gate[oi]; o *= 1.0f / (1.0f + kr_expf(-gt)); }
                a.attn_out[oi] = o;
            }
        }
    }
}

// ---- MoE block epilogue (decode.rs:3343-3402): hidden = moe (*rsf) + shared (*sigmoid(gate)) -------------------------------------
// (round 6: four columns per thread, 16-byte accesses, the gate's sigmoid once per thread instead of once per element; the same operations per value)
__global__ void __launch_bounds__(256) kr_pfm_moe_epilogue_kernel(const float* __restrict__ moe, const float* __restrict__ shared, const float* __restrict__ gate_val,
                                                                 int gate_ld, float rsf, float* __restrict__ hidden, int H) {
    const int t = blockIdx.y, j = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= H) return;
    const float sg = (shared && gate_val) ? 1.0f / (1.0f + kr_expf(-gate_val[(size_t)t * gate_ld])) : 1.0f;
    if (j + 4 <= H && (H & 3) == 0) {
        float4 acc = *reinterpret_cast<const float4*>(moe + (size_t)t * H + j);
        if (rsf != 1.0f) { acc.x *= rsf; acc.y *= rsf; acc.z *= rsf; acc.w *= rsf; }
        if (shared) {
            float4 sh = *reinterpret_cast<const float4*>(shared + (size_t)t * H + j);
            if (gate_val) { sh.x *= sg; sh.y *= sg; sh.z *= sg; sh.w *= sg; }
            acc.x = acc.x + sh.x; acc.y = acc.y + sh.y; acc.z = acc.z + sh.z; acc.w = acc.w + sh.w;
        }
        *reinterpret_cast<float4*>(hidden + (size_t)t * H + j) = acc;
        return;
    }
    for (int jj = j; jj < H && jj < j + 4; jj++) {
        float acc = moe[(size_t)t * H + jj];
        if (rsf != 1.0f) acc *= rsf;
        if (shared) {
            float sh = shared[(size_t)t * H + jj];
            if (gate_val) sh *= sg;
            acc = acc + sh;
        }
        hidden[(size_t)t * H + jj] = acc;
    }
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
static void pfm_la_conv_state(const KrPfmLaArgs& a, int C, hipStream_t st, const KrPfSync* sy) {      // advance the carried conv slots, then publish them to the next chunk
    const int conv_dim = 2 * a.nk * a.dk + a.nv * a.dv;
    hipLaunchKernelGGL(kr_pfm_la_conv_state_kernel, dim3((conv_dim + 255) / 256), dim3(256), 0, st, a, C);
    kr_pf_rec(st, sy->rec_a);
}
int kr_launch_pfm_la(const KrPfmLaArgs& a, float* recur_state, float* recur_out, const float* norm_w, float* gated_out, int C, float eps, hipStream_t st, const KrPfSync* sy) {
    if (a.dv > 256 || a.dv % 8 || a.dv < a.dk || a.dk % 8) return 1;
    const KrPfSync none{};
    if (!sy) sy = &none;
    kr_pf_wait(st, sy->wait_a);              // the previous chunk's carried conv slots
    if (a.dk % 4 || a.dv % 4 || a.ld_qkvz % 4) return 1;      // 16-byte row reads
    // tolerance pass, round 6: no conv launch -- the delta rule's prep launch forms q / k / v / gates from the in-projection's output (same arithmetic), the carried conv
    // slots are advanced behind it, the gated norm reads z in place
    if (a.fast && a.conv_fused && a.lac && kr_pfm_la_chunk_ok(a.dk, a.dv, C) && a.nv == a.nk * a.hr &&
        kr_launch_pfm_la_chunked(a, recur_state, recur_out, a.lac, C, st, sy, 1, pfm_la_conv_state) == 0) {
        kr_pf_rec(st, sy->rec_b);
        const int gd = 2 * a.dk + 2 * a.dv * a.hr;
        hipLaunchKernelGGL(kr_pfm_gated_norm_kernel, dim3(a.nv, (C + PFG_TT - 1) / PFG_TT), dim3(256), 0, st, recur_out, a.qkvz + 2 * a.dk + a.hr * a.dv, (size_t)a.ld_qkvz, a.hr, gd, norm_w,
                           gated_out, a.nv, a.dv, C, eps);
        return 0;
    }
    hipLaunchKernelGGL(kr_pfm_la_conv_kernel, dim3(a.nk, (C + PFC_WT - 1) / PFC_WT), dim3(256), (size_t)(2 * PFC_WT * a.dk + 2 * PFC_WT) * 4, st, a, C);
    pfm_la_conv_state(a, C, st, sy);
    // the recurrent state: the chunked form waits between its two launches (the first needs no state), the per-token kernel before its one launch
    if (a.fast && a.lac && kr_pfm_la_chunk_ok(a.dk, a.dv, C) && kr_launch_pfm_la_chunked(a, recur_state, recur_out, a.lac, C, st, sy) == 0) {}
    else {
        if (a.dk != 128 && a.dk != 64) return 1;
        kr_pf_wait(st, sy->wait_b);
        if (a.dk == 128) hipLaunchKernelGGL(kr_pfm_la_recur_kernel<128>, dim3(a.nv), dim3(a.dv), 0, st, recur_state, a.q, a.k, a.v, a.gexp, a.beta, recur_out, a.nv, a.dv, C);
        else hipLaunchKernelGGL(kr_pfm_la_recur_kernel<64>, dim3(a.nv), dim3(a.dv), 0, st, recur_state, a.q, a.k, a.v, a.gexp, a.beta, recur_out, a.nv, a.dv, C);
    }
    kr_pf_rec(st, sy->rec_b);
    // z: the stand-alone copy [C][nv * dv] the conv launch makes (z_hr = nv: one block)
    hipLaunchKernelGGL(kr_pfm_gated_norm_kernel, dim3(a.nv, (C + PFG_TT - 1) / PFG_TT), dim3(256), 0, st, recur_out, (const float*)a.z, (size_t)a.nv * a.dv, a.nv, 0, norm_w, gated_out, a.nv, a.dv, C, eps);
    return 0;
}
// the recurrence alone (stand-alone operator linear_attention_recurrent, decode.rs:609): gexp = e^g per (token, head); non-zero = unsupported geometry
int kr_launch_pfm_la_recur(float* state, const float* q, const float* k, const float* v, const float* gexp, const float* beta, float* out, int nv, int dk, int dv, int C, hipStream_t st) {
    if (dv > 256 || dv % 8 || dv < dk) return 1;
    if (dk == 128) hipLaunchKernelGGL(kr_pfm_la_recur_kernel<128>, dim3(nv), dim3(dv), 0, st, state, q, k, v, gexp, beta, out, nv, dv, C);
    else if (dk == 64) hipLaunchKernelGGL(kr_pfm_la_recur_kernel<64>, dim3(nv), dim3(dv), 0, st, state, q, k, v, gexp, beta, out, nv, dv, C);
    else return 1;
    return 0;
}
int kr_pfm_gqa_tile(int nh, int nkv) { const int group = nh / nkv; int tt = PFA_TT_MAX / group; return tt < 1 ? 0 : tt; }
void kr_launch_pfm_gqa_prep(const KrPfmGqaArgs& a, int C, hipStream_t st) {
    hipLaunchKernelGGL(kr_pfm_gqa_prep_kernel, dim3(a.nh + a.nkv, (C + PFQ_TT - 1) / PFQ_TT), dim3(256), 0, st, a, C);
}
int kr_launch_pfm_gqa(const KrPfmGqaArgs& a, int C, float* sc, int sc_ld, float* inv, hipStream_t st, const KrPfSync* sy) {
    const int group = a.nh / a.nkv, TT = kr_pfm_gqa_tile(a.nh, a.nkv);
    if (TT == 0 || a.hd > 256 || a.hd % 32 || a.nh % a.nkv) return 1;
    hipLaunchKernelGGL(kr_pfm_gqa_prep_kernel, dim3(a.nh + a.nkv, (C + PFQ_TT - 1) / PFQ_TT), dim3(256), 0, st, a, C);
    if (sy) { kr_pf_wait(st, sy->wait_b); kr_pf_rec(st, sy->rec_b); }      // this chunk's rows are appended; the earlier chunks' rows are what the passes below read
    static const bool no_mfma = getenv("KR_EXACT_ATTN_VALU") != nullptr;       // tuning / A-B hook: keep the vector-ALU passes
    if (!no_mfma && kr_pfm_gqa_exact_mfma_ok(a)) {                              // scores and P.V on the f32 matrix cores, same bits (kr_attn_exact_mfma.hip)
        const int rows = C * a.nh;
        float* tmax = inv + rows;                                                // [rows][sc_ld / 32]
        kr_launch_pfm_gqa_scores_mfma(a, C, sc, sc_ld, tmax, st);
        hipLaunchKernelGGL(kr_pfm_gqa_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, sc, sc_ld, inv, a.nh, a.pos0, rows, (const float*)tmax);
        kr_launch_pfm_gqa_pv_mfma(a, C, sc, sc_ld, inv, st);
        return 0;
    }
    const int ntt = (C + TT - 1) / TT, npt = (a.pos0 + C + 255) / 256;
    const size_t lds = ((size_t)group * TT * 8 + 32 * 8) * PFA_LDB * 4;
    {                                  // > 64 KiB of dynamic LDS needs the opt-in (160 KiB per CU on gfx950), per (kernel, device)
        const void* fns[7] = {(const void*)kr_pfm_gqa_scores_kernel, (const void*)kr_pfm_gqa_scores_t_kernel<false, 8>, (const void*)kr_pfm_gqa_scores_t_kernel<false, 16>,
                              (const void*)kr_pfm_gqa_scores_t_kernel<false, 32>, (const void*)kr_pfm_gqa_scores_t_kernel<true, 8>, (const void*)kr_pfm_gqa_scores_t_kernel<true, 16>,
                              (const void*)kr_pfm_gqa_scores_t_kernel<true, 32>};
        for (const void* f : fns) (void)kr_lds_optin(f, 96 * 1024);
    }
#define KR_SC(F_, N_) hipLaunchKernelGGL((kr_pfm_gqa_scores_t_kernel<F_, N_>), dim3(npt, ntt, a.nkv), dim3(256), lds, st, a, sc, sc_ld, TT, C)
    if (a.hd == 256) { if (a.kv_fp8) KR_SC(true, 32); else KR_SC(false, 32); }
    else if (a.hd == 128) { if (a.kv_fp8) KR_SC(true, 16); else KR_SC(false, 16); }
    else if (a.hd == 64) { if (a.kv_fp8) KR_SC(true, 8); else KR_SC(false, 8); }
    else hipLaunchKernelGGL(kr_pfm_gqa_scores_kernel, dim3(npt, ntt, a.nkv), dim3(256), lds, st, a, sc, sc_ld, TT, C);
#undef KR_SC
    const int rows = C * a.nh;
    hipLaunchKernelGGL(kr_pfm_gqa_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, sc, sc_ld, inv, a.nh, a.pos0, rows, (const float*)nullptr);
    const int TTV = PFV_R / group < 1 ? 0 : PFV_R / group, nttv = TTV ? (C + TTV - 1) / TTV : 0;   // pass C has its own (smaller) token tile
    if (!TTV) return 1;
#define KR_PV(F_, G_, C_) hipLaunchKernelGGL((kr_pfm_gqa_pv_kernel<F_, G_, C_>), dim3(nttv, a.nkv), dim3(64), 0, st, a, sc, sc_ld, TTV, C)
#define KR_PVG(F_, C_) do { switch (group) { case 1: KR_PV(F_, 1, C_); break; case 2: KR_PV(F_, 2, C_); break; case 4: KR_PV(F_, 4, C_); break; case 8: KR_PV(F_, 8, C_); break; \
                                           case 16: KR_PV(F_, 16, C_); break; default: KR_PV(F_, 0, C_); } } while (0)
    if (a.hd == 256) { if (a.kv_fp8) KR_PVG(true, 4); else KR_PVG(false, 4); }
    else if (a.hd == 128) { if (a.kv_fp8) KR_PVG(true, 2); else KR_PVG(false, 2); }
    else if (a.hd == 64) { if (a.kv_fp8) KR_PVG(true, 1); else KR_PVG(false, 1); }
    else return 1;
#undef KR_PVG
#undef KR_PV
    return 0;
}
void kr_launch_pfm_softmax_rows(float* sc, int sc_ld, float* inv, int nh, int pos0, int rows, const float* tmax, hipStream_t st) {
    hipLaunchKernelGGL(kr_pfm_gqa_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, sc, sc_ld, inv, nh, pos0, rows, tmax);
}
void kr_launch_pfm_moe_epilogue(const float* moe, const float* shared, const float* gate_val, int gate_ld, float rsf, float* hidden, int C, int H, hipStream_t st) {
    hipLaunchKernelGGL(kr_pfm_moe_epilogue_kernel, dim3((H + 1023) / 1024, C), dim3(256), 0, st, moe, shared, gate_val, gate_ld, rsf, hidden, H);
}

// ---- negative log-likelihood of the next token per prompt position (perplexity/measure_ppl.py:218-227: cross_entropy(logits[:-1], tokens[1:],
// reduction="none") in f32).  One workgroup per row: maximum, then sum of expf(x - max) accumulated in double (the order of a double sum
// moves the f32 result by far less than one ulp), nll = log(sum) + max - x[label] rounded once.  HBM-bound: one read of the row.
__global__ void __launch_bounds__(1024) kr_pfm_nll_kernel(const float* __restrict__ logits, size_t ld, const int* __restrict__ labels, float* __restrict__ nll, int V) {
    __shared__ float redf[16];
    __shared__ double redd[16];
    const float* x = logits + (size_t)blockIdx.x * ld;
    const int t = threadIdx.x;
    float mx = -__builtin_inff();
    for (int i = t; i < V; i += 1024) mx = fmaxf(mx, x[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((t & 63) == 0) redf[t >> 6] = mx;
    __syncthreads();
    mx = redf[0];
#pragma unroll
    for (int w = 1; w < 16; w++) mx = fmaxf(mx, redf[w]);
    double sm = 0.0;
    for (int i = t; i < V; i += 1024) sm += (double)kr_expf(x[i] - mx);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sm += __shfl_xor(sm, off);
    if ((t & 63) == 0) redd[t >> 6] = sm;
    __syncthreads();
    if (t == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; w++) tot += redd[w];
        nll[blockIdx.x] = (float)(log(tot) + (double)mx - (double)x[labels[blockIdx.x]]);
    }
}
void kr_launch_pfm_nll(const float* logits, size_t ld, const int* labels, float* nll, int rows, int V, hipStream_t st) {
    if (rows > 0) hipLaunchKernelGGL(kr_pfm_nll_kernel, dim3(rows), dim3(1024), 0, st, logits, ld, labels, nll, V);
}
