// kr_decode_fast.hip -- the decode step in TOLERANCE mode (mode bit KR_DECODE_FAST of kr_decode_set_attention_mode) for gfx950.
//
// Same operators as the exact decode graph (kr_decode_ops.hip, kr_moe_decode.hip, kr_router.hip; reference src/decode.rs:2690-3520,
// src/moe.rs:572-715, src/kernel/avx2.rs:1066-1206), same products: INT16 activation digits per 128-group (avx2.rs:234-304), exact integer
// group sums (v_dot4), bf16(w_scale) * a_scale per group, the reference's polynomial sigmoid / libm functions.  What changes is the ORDER OF
// THE f32 SUMS: a lane accumulates its own groups and lanes / waves are combined by trees, the norm and softmax sums are wave trees, the
// state chains of the gated delta rule are split over 16 row slices.  That removes the serial chains that set the duration of the exact
// launches and lets the launch structure follow the data dependences instead of the reference's call structure:
// 6 launches per linear-attention MoE layer (7 before), none of them a single-workgroup kernel.
// The exact kernels stay the default and the yardstick: tests/test_decode_fast_gpu.py states and checks the tolerances.
#include "kr_decode_fast.h"
#include "kr_gguf_dev.h"

#include "kr_device.h"
#include "kr_libm.h"
#include "kr_matvec_dev.h"
#include "kr_topk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) u32x4 kr_fsm[];

#ifdef KR_FTIMING   // libkrasis_hip_timing.so (make timing): wall-clock stamps (10 ns units) of wave 0 of the middle workgroup, read back by tools/probes/decode_fast_stamps.py
__device__ unsigned long long kr_fstamps[8][16];
#define KR_FSTAMP(k, i) do { if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x == 0) kr_fstamps[k][i] = wall_clock64(); } while (0)
extern "C" int kr_debug_fstamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kr_fstamps), sizeof(kr_fstamps)); }
// entry / exit time and hardware id of EVERY workgroup of a launch (wave 0), for the launches of the last layer of a step: where a launch's tail comes from
__device__ unsigned long long kr_fwg[6][1024][3];
#define KR_FWG(k, i) do { if (threadIdx.x == 0) { const unsigned b_ = blockIdx.x + gridDim.x * blockIdx.y; if (b_ < 1024) { kr_fwg[k][b_][i] = wall_clock64(); \
        if ((i) == 0) kr_fwg[k][b_][2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32); } } } while (0)
extern "C" int kr_debug_fwg(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kr_fwg), sizeof(kr_fwg)); }
#else
#define KR_FSTAMP(k, i) do { } while (0)
#define KR_FWG(k, i) do { } while (0)
#endif

#define KR_FW2_LDS_MAX (64 * 1024)   // dynamic LDS of kr_fw2_kernel without a per-device opt-in (kr_lds_optin.h)
#define KR_FU 16   // 16-byte weight records a lane keeps in flight per tile in the generic (guarded) form

// ---------------------------------------------------------------------------------------------------------------------------------
// reductions (all 64 lanes active at every call site that spans rows)
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float kr_f_red8(float v) {
    v += __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_XOR1));
    v += __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_XOR2));
    v += __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_HALF_MIRROR));
    return v;
}
__device__ __forceinline__ float kr_f_red16(float v) {
    v = kr_f_red8(v);
    v += __int_as_float(KR_DPP(__float_as_int(v), KR_DPP_MIRROR));
    return v;
}
__device__ __forceinline__ float kr_f_rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float kr_f_wave_sum(float v) {
    v = kr_f_red16(v);
    return (kr_f_rl(v, 0) + kr_f_rl(v, 16)) + (kr_f_rl(v, 32) + kr_f_rl(v, 48));
}
__device__ __forceinline__ float kr_f_wave_max(float v) {
    v = kr_red16_max_f32(v);
    return fmaxf(fmaxf(kr_f_rl(v, 0), kr_f_rl(v, 16)), fmaxf(kr_f_rl(v, 32), kr_f_rl(v, 48)));
}
__device__ __forceinline__ void kr_f_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// ---------------------------------------------------------------------------------------------------------------------------------
// weight records of one 8-column tile: NU > 0 = exactly NU units per wave (no guards: one basic block), NU == 0 = guarded, up to KR_FU per pass
// unit = group pair (INT4) / group (INT8), as in kr_moe_decode.hip
// ---------------------------------------------------------------------------------------------------------------------------------
template <int BITS, int NU, int FU = KR_FU> struct KrFw { u32x4 w[NU > 0 ? NU : FU]; uint32_t sc[NU > 0 ? NU : FU]; };

template <int BITS, int NU, int FU = KR_FU>
__device__ __forceinline__ void kr_f_fetch(KrFw<BITS, NU, FU>& p, const KrMatDev& m, const void* qbase, const uint32_t* sbase, int tile, int lane, int ub, int u1) {
    const int units = BITS == 4 ? m.ngp : m.ng;
    const u32x4* q = reinterpret_cast<const u32x4*>(qbase) + (size_t)tile * units * 64 + lane;
    const uint32_t* s = sbase + (size_t)tile * m.ngp * 8 + (lane >> 3);
    if constexpr (NU > 0) {
#pragma unroll
        for (int u = 0; u < NU; u++) { p.w[u] = kr_ldg_nt(q + (size_t)(ub + u) * 64); p.sc[u] = kr_ldg_nt(s + (BITS == 4 ? (ub + u) : ((ub + u) >> 1)) * 8); }
    } else {
#pragma unroll
        for (int u = 0; u < FU; u++)
            if (ub + u < u1) { p.w[u] = kr_ldg_nt(q + (size_t)(ub + u) * 64); p.sc[u] = kr_ldg_nt(s + (BITS == 4 ? (ub + u) : ((ub + u) >> 1)) * 8); }
    }
}

// the lane's f32 partial over units [ub, u1): exact integer sum of each group times bf16(w_scale) * a_scale (avx2.rs:1171), accumulated per lane
template <int BITS, int NU, int FU = KR_FU>
__device__ __forceinline__ float kr_f_dot(const KrFw<BITS, NU, FU>& p, const KrMatDev& m, int ub, int u1, int l8, const KrActLds& L, float acc) {
    constexpr int N = NU > 0 ? NU : FU;
#pragma unroll
    for (int u = 0; u < N; u++) {
        const int un = ub + u;
        if (NU > 0 || un < u1) {
            if constexpr (BITS == 4) {
                const int g0 = 2 * un, g1 = g0 + 1;
                const int i0 = kr_group_i4(p.w[u].x, p.w[u].y, g0, l8, L);
                acc = __builtin_fmaf((float)i0, __uint_as_float(p.sc[u] << 16) * L.ascale[g0], acc);
                if (NU > 0 || g1 < m.ng) {   // NU > 0 is only launched for an even group count
                    const int i1 = kr_group_i4(p.w[u].z, p.w[u].w, g1, l8, L);
                    acc = __builtin_fmaf((float)i1, __uint_as_float(p.sc[u] & 0xFFFF0000u) * L.ascale[g1], acc);
                }
            } else {
                const int i0 = kr_group_i8(p.w[u], un, l8, L);
                const uint32_t sb = (un & 1) ? (p.sc[u] & 0xFFFF0000u) : (p.sc[u] << 16);
                acc = __builtin_fmaf((float)i0, __uint_as_float(sb) * L.ascale[un], acc);
            }
        }
    }
    return acc;
}

// whole tile for one wave's unit range [u0, u1); `p` holds the first pass (already requested)
template <int BITS, int NU, int FU = KR_FU>
__device__ __forceinline__ float kr_f_tile(KrFw<BITS, NU, FU>& p, const KrMatDev& m, const void* qbase, const uint32_t* sbase, int tile, int lane, int u0, int u1, const KrActLds& L) {
    float acc = kr_f_dot<BITS, NU, FU>(p, m, u0, u1, lane & 7, L, 0.0f);
    if constexpr (NU == 0) {
        for (int ub = u0 + FU; ub < u1; ub += FU) {
            kr_f_fetch<BITS, NU, FU>(p, m, qbase, sbase, tile, lane, ub, u1);
            acc = kr_f_dot<BITS, NU, FU>(p, m, ub, u1, lane & 7, L, acc);
        }
    }
    return kr_f_red8(acc);
}

// pre-built activation image (global memory, byte layout == the LDS image for INT4 weights) -> LDS by threads [t0, t0 + nthr) of the workgroup.
// Two phases: the first KR_FIMG records of a thread are requested by kr_f_image_load (before the caller's weight stream, see kr_f_norm_load) and stored by
// kr_f_image_store, which also walks whatever lies beyond them (K > 4096 * nthr / 256).
#define KR_FIMG 3
struct KrFImg { u32x4 r[KR_FIMG]; };
__device__ __forceinline__ void kr_f_image_load(const void* img, int K, KrFImg& R, int tt, int nthr) {
    const int n16 = (int)(kr_lds_bytes(K, false) / 16);
    const u32x4* src = reinterpret_cast<const u32x4*>(img);
#pragma unroll
    for (int j = 0; j < KR_FIMG; j++)
        if (j == 0 || n16 > j * nthr) { const int i = tt + j * nthr; R.r[j] = src[i < n16 ? i : n16 - 1]; }      // (uniform guard, every lane loads: see kr_f_norm_load)
}
template <int BITS>
__device__ __forceinline__ void kr_f_image_put(const u32x4 r, int i, int K, u32x4* smem, const KrActLds& L) {
    smem[i] = r;
    if constexpr (BITS == 8) {
        if (i < K / 8) {
            uint32_t* p = reinterpret_cast<uint32_t*>(L.planes8) + (i >> 1) * 8 + (i & 1) * 2;
            p[0] = __builtin_amdgcn_perm(r.y, r.x, 0x05010400u);
            p[1] = __builtin_amdgcn_perm(r.y, r.x, 0x07030602u);
            p[4] = __builtin_amdgcn_perm(r.w, r.z, 0x05010400u) ^ 0x80808080u;
            p[5] = __builtin_amdgcn_perm(r.w, r.z, 0x07030602u) ^ 0x80808080u;
        }
    }
}
template <int BITS>
__device__ __forceinline__ void kr_f_image_store(const void* img, int K, const KrFImg& R, u32x4* smem, const KrActLds& L, int tt, int nthr) {
    const int n16 = (int)(kr_lds_bytes(K, false) / 16);
    const u32x4* src = reinterpret_cast<const u32x4*>(img);
#pragma unroll
    for (int j = 0; j < KR_FIMG; j++) { const int i = tt + j * nthr; if (i < n16) kr_f_image_put<BITS>(R.r[j], i, K, smem, L); }
    for (int i = tt + KR_FIMG * nthr; i < n16; i += nthr) kr_f_image_put<BITS>(src[i], i, K, smem, L);
}
template <int BITS>
__device__ __forceinline__ void kr_f_image_copy(const void* img, int K, u32x4* smem, const KrActLds& L, int tt, int nthr) {
    KrFImg R;
    kr_f_image_load(img, K, R, tt, nthr);
    kr_f_image_store<BITS>(img, K, R, smem, L, tt, nthr);
}

struct KrFNormIn { const float *hid, *res, *w; float* res_out; int first; float eps; int bias_one; int n; };
// The inputs of the norm are REQUESTED by kr_f_norm_load and consumed by kr_f_norm_finish: a kernel issues these (and every other small load whose address it
// knows) BEFORE its weight stream.  The memory counter of a wave is in-order: a small load issued behind 16 weight records is not back before all of them are,
// and (round 4's kernels) a prologue that loads, waits, computes, loads again walks one memory round trip after the other behind the whole weight stream
// (profiles/r05_decode_fast_stamps.txt: "norm + image" 2.1 us of the in-projection launch).
struct KrFNormRegs { float h[2][8], r[2][8], w[2][8]; };
// (Every lane loads: a lane without a chunk reads the last one and never uses it.  A load under a lane mask whose destination has a default value costs a
// wait for the WHOLE memory counter at the merge -- the default's v_mov is a write-after-write on a register with a load in flight.)
__device__ __forceinline__ void kr_f_norm_load(const KrFNormIn& in, KrFNormRegs& R) {
    const int t = threadIdx.x, nch = in.n / 8;
    {
        const int c = t < nch ? t : nch - 1;
        kr_load8(in.hid, c, R.h[0]);
        if (!in.first) kr_load8(in.res, c, R.r[0]);
        kr_load8(in.w, c, R.w[0]);
    }
    if (nch > 256) {      // workgroup-uniform
        const int c = t + 256 < nch ? t + 256 : nch - 1;
        kr_load8(in.hid, c, R.h[1]);
        if (!in.first) kr_load8(in.res, c, R.r[1]);
        kr_load8(in.w, c, R.w[1]);
    }
}
__device__ __forceinline__ void kr_f_norm_finish(const KrFNormIn& in, const KrFNormRegs& R, float (&x)[2][8], float* s_red, bool write_res) {
    const int t = threadIdx.x, nch = in.n / 8;
    float ss = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int c = t + 256 * u;
        if (c < nch) {
#pragma unroll
            for (int i = 0; i < 8; i++) { x[u][i] = in.first ? R.h[u][i] : (R.h[u][i] + R.r[u][i]); ss = __builtin_fmaf(x[u][i], x[u][i], ss); }
            if (write_res) {
                float4* ro = reinterpret_cast<float4*>(in.res_out + (size_t)c * 8);
                ro[0] = float4{x[u][0], x[u][1], x[u][2], x[u][3]}; ro[1] = float4{x[u][4], x[u][5], x[u][6], x[u][7]};
            }
        }
    }
    ss = kr_f_wave_sum(ss);
    if ((t & 63) == 0) s_red[t >> 6] = ss;
    __syncthreads();
    const float tot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    const float rms = 1.0f / sqrtf(tot / (float)in.n + in.eps);
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int i = 0; i < 8; i++) x[u][i] = (x[u][i] * rms) * (in.bias_one ? (R.w[u][i] + 1.0f) : R.w[u][i]);
}
// fused add + RMSNorm (decode.rs:1199) of a vector of n <= 4096 values by a 256-thread workgroup: thread t owns chunks t and t + 256 (8 values
// each).  Returns the normalised values in x[u][*]; the caller quantises / stores them.  The sum of squares is a tree (lane, wave, workgroup).
__device__ __forceinline__ void kr_f_norm(const KrFNormIn& in, float (&x)[2][8], float* s_red, bool write_res) {
    KrFNormRegs R;
    kr_f_norm_load(in, R);
    kr_f_norm_finish(in, R, x, s_red, write_res);
}

// quantize_activation_int16_f32 (avx2.rs:274) of the thread's chunk straight from registers (its 128-group = its 16-lane row)
template <bool I8>
__device__ __forceinline__ void kr_f_quant_chunk(const float (&v)[8], int c, const KrActLds& L, bool round_bf16) {
    float y[8];
    float mx = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) { y[i] = round_bf16 ? kr_bf16_to_f32(kr_f32_to_bf16(v[i])) : v[i]; mx = fmaxf(mx, fabsf(y[i])); }
    float scale, inv;
    kr_group_scale(mx, scale, inv);
    int q[8];
    kr_quant8<false>(y, inv, q);
    kr_store_chunk<I8>(L, c, q);
    if ((c & 15) == 0) L.ascale[c >> 4] = scale;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K1 / K3: multi-matrix dequant-matvec.  KS waves split the K range of a tile, 4 / KS tiles per workgroup.
// ---------------------------------------------------------------------------------------------------------------------------------
// Kernel arguments: the pointers a wave needs for its FIRST requests are leading scalar arguments, which the Makefile asks the compiler to have preloaded into
// SGPRs (-amdgpu-kernarg-preload-count): those requests leave without the scalar round trip to the argument block that the by-value structs cost.
//   p0: MODE 0 the INT16 image, MODE 1 / 2 the input vector (null: first layer, the embedding row of the step's token); p1: residual (null: none); p2: norm weights
// (Measured and removed, round 5: a BALANCED form of the in-projection launch -- exactly 256 workgroups of 6 or 7 one-tile waves, one per CU, the norm built once per
//  CU -- against the 386 four-tile workgroups that put two workgroups on 130 CUs.  Per-workgroup times say why it is worth nothing: alone on a CU a workgroup takes
//  4.3 us, sharing one 5.2, and a 6 - 7-wave workgroup 5.1: the launch follows the bytes a CU streams (profiles/r05_decode_fast_wg_times.txt), first entry -> last
//  exit 6.34 -> 6.13 us, 620.9 vs 620.9 tok/s in an interleaved A/B.)
template <int BITS, int KS, int NU, int MODE>
__global__ void __launch_bounds__(256) kr_fdm_kernel(const void* p0, const float* p1, const float* p2, int Kp, const KrFdmArgs a) {
    constexpr int TW = 4 / KS;
    __shared__ float s_red[4];
    __shared__ float s_x[TW][KS][8];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, l8 = lane & 7, cl = lane >> 3;
    const int tw = wave / KS, ks = wave - tw * KS;
    [[maybe_unused]] constexpr int sk = MODE == 1 ? 0 : 2;
    KR_FSTAMP(sk, 0); KR_FWG(sk, 0);
    // ---- requests, in the order they are needed back (a wave's memory counter is in-order): the input vector / image, then the weight stream, then the
    //      epilogue's operands.  MODE is a template parameter so that no register of one mode's loads is ever seen as pending by another mode's code.
    const int K = Kp;
    const KrActLds L = kr_carve_lds(kr_fsm, K, BITS == 8);
    [[maybe_unused]] KrFNormRegs NR; [[maybe_unused]] KrFImg IR; [[maybe_unused]] float x8[2][8];
    [[maybe_unused]] KrFNormIn in{reinterpret_cast<const float*>(p0), p1, p2, a.res_out, p1 == nullptr, a.eps, a.bias_one, K};
    if constexpr (MODE == 0) kr_f_image_load(p0, K, IR, t, 256);
    else if constexpr (MODE == 2) {
        const int nch = K / 8;
        kr_load8(reinterpret_cast<const float*>(p0), t < nch ? t : nch - 1, x8[0]);
        if (nch > 256) kr_load8(reinterpret_cast<const float*>(p0), t + 256 < nch ? t + 256 : nch - 1, x8[1]);
    } else {
        if (p0 == nullptr) in.hid = a.emb + (size_t)a.step->token * K;
        kr_f_norm_load(in, NR);
    }
    // (Measured and removed, round 5: requesting the weight records of matrix-0 tiles from preloaded arguments right behind the input requests -- 1 us earlier than the
    //  descriptor fields of the argument struct allow.  The stream then queues ahead of the norm's inputs in the memory system: "norm + image" 1.4 -> 2.9 us, the launch
    //  6.1 -> 6.5 us, 626 -> 613 tok/s; a workgroup barrier between the two request groups did not help (615).  The scalar work below is a useful delay.)
    const int bx = blockIdx.x;
    const int gt0 = bx * TW + tw;
    // the matrix of this tile: every field sits at a CONSTANT kernarg offset (one scalar round trip for all of them) and is selected
    // afterwards -- indexing the argument struct with a computed matrix index costs a second, dependent scalar round trip before the
    // first weight request can be issued.  The host sets tile_end[i] = total for every i >= n - 1.
    const int te0 = a.mm.tile_end[0], te1 = a.mm.tile_end[1], te2 = a.mm.tile_end[2], total = a.mm.tile_end[3];
    const bool active = gt0 < total;
    const int gt = active ? gt0 : total - 1;      // a wave past the last tile walks the last tile again and stores nothing: every wave issues the same requests (see kr_f_norm_load)
    const int mi = (gt >= te0) + (gt >= te1) + (gt >= te2);
#define KR_MSEL(f) (mi == 0 ? a.mm.m[0].f : (mi == 1 ? a.mm.m[1].f : (mi == 2 ? a.mm.m[2].f : a.mm.m[3].f)))
    KrMatDev m{};
    m.q = KR_MSEL(q); m.s = KR_MSEL(s); m.ng = KR_MSEL(ng); m.ngp = KR_MSEL(ngp); m.N = KR_MSEL(N);
    float* const my = mi == 0 ? a.mm.y[0] : (mi == 1 ? a.mm.y[1] : (mi == 2 ? a.mm.y[2] : a.mm.y[3]));
#undef KR_MSEL
    const int tile = gt - (mi == 0 ? 0 : (mi == 1 ? te0 : (mi == 2 ? te1 : te2)));
    const int units = BITS == 4 ? m.ngp : m.ng;
    int u0, u1;
    if (NU > 0) { u0 = ks * NU; u1 = u0 + NU; }
    else { const int uw = (units + KS - 1) / KS; u0 = ks * uw; u1 = u0 + uw < units ? u0 + uw : units; }
    KrFw<BITS, NU> W;
    kr_f_fetch<BITS, NU>(W, m, m.q, m.s, tile, lane, u0, u1);
    KR_FSTAMP(sk, 1);
    if constexpr (MODE == 0) kr_f_image_store<BITS>(p0, K, IR, kr_fsm, L, t, 256);
    else if constexpr (MODE == 2) {      // plain f32 input vector (MLA: the w_vc output feeding o_proj): every workgroup quantises it (quantize_activation_int16_f32, avx2.rs:274)
#pragma unroll
        for (int u = 0; u < 2; u++) { const int c = t + 256 * u; if (c < K / 8) kr_f_quant_chunk<BITS == 8>(x8[u], c, L, false); }
    } else {
        float x[2][8];
        kr_f_norm_finish(in, NR, x, s_red, blockIdx.x == 0);
#pragma unroll
        for (int u = 0; u < 2; u++) { const int c = t + 256 * u; if (c < K / 8) kr_f_quant_chunk<BITS == 8>(x[u], c, L, false); }
    }
    // the lane that will hold column `col` asks for what its epilogue needs -- conv state / taps of the linear-attention channels, gate constants -- BEHIND the
    // weight stream and after the norm / image work (its index arithmetic is ~100 instructions of a lone wave: placed here it runs while the weights are still in
    // flight instead of delaying the norm; nothing of it is touched before the dot is done).  Every lane loads (index 0 where it has nothing to ask for) and no destination has a default.
    const int col = tile * 8 + cl;
    const bool out_lane = active && ks == 0 && l8 == 0 && col < m.N;
    int kind = -1, dst = 0, ch = 0;       // 0 q, 1 k, 2 v (conv channels), 3 z, 4 beta, 5 decay gate
    float4 cs = float4{0.0f, 0.0f, 0.0f, 0.0f}, cw = cs;      // (MODE 1 assigns all four unconditionally below: the initial values are dead there and cost nothing)
    float g_al = 0.0f, g_dt = 0.0f;
    if constexpr (MODE == 1) {
        const bool la = a.conv_state != nullptr;      // workgroup-uniform; without the epilogue the four requests read the norm weights (any valid address) and are dropped
        if (la && out_lane && mi == a.conv_mi) {
            // (24-bit multiplies and selects instead of branches: a 64-bit v_mad here pairs its 32-bit addend with whatever register follows it -- one that a weight
            //  request is still writing, as it happened -- and the wave waits for the whole stream before it may start on the norm)
            const int nt = a.hr * a.dv, gd = 2 * a.dk + 2 * nt, key_dim = a.nk * a.dk;
            const int kh = col / gd, cc = col - __mul24(kh, gd);
            const int khk = __mul24(kh, a.dk), kht = __mul24(kh, nt);
            kind = cc < a.dk ? 0 : (cc < 2 * a.dk ? 1 : (cc < 2 * a.dk + nt ? 2 : 3));
            ch = kind == 0 ? khk + cc : (kind == 1 ? key_dim + khk + (cc - a.dk) : 2 * key_dim + kht + (cc - 2 * a.dk));
            dst = kind < 2 ? 2 * khk + cc : (kind == 2 ? kht + (cc - 2 * a.dk) : kht + (cc - 2 * a.dk - nt));
        } else if (la && out_lane && mi == a.gate_mi) {   // ba row: [kh][beta raw (hr) | a raw (hr)] (decode.rs:3891-3901)
            const int kh = col / (2 * a.hr), idx = col - __mul24(kh, 2 * a.hr);
            kind = idx < a.hr ? 4 : 5;
            dst = __mul24(kh, a.hr) + (idx < a.hr ? idx : idx - a.hr);
        }
        const int chl = (kind >= 0 && kind < 3) ? ch : 0, gl = kind == 5 ? dst : 0;
        cs = reinterpret_cast<const float4*>(la ? a.conv_state : a.norm_w)[chl]; cw = reinterpret_cast<const float4*>(la ? a.conv_w : a.norm_w)[chl];
        g_al = (la ? a.a_log : a.norm_w)[gl]; g_dt = (la ? a.dt_bias : a.norm_w)[gl];
    }
    KR_FSTAMP(sk, 2);
    __syncthreads();
    KR_FSTAMP(sk, 3);
    float acc = kr_f_tile<BITS, NU>(W, m, m.q, m.s, tile, lane, u0, u1, L);
    KR_FSTAMP(sk, 4);
    if constexpr (KS > 1) {
        if (l8 == 0) s_x[tw][ks][cl] = acc;
        __syncthreads();
        if (ks == 0) {
            acc = s_x[tw][0][cl];
#pragma unroll
            for (int k = 1; k < KS; k++) acc += s_x[tw][k][cl];
        }
    }
    if (out_lane) {
        if (kind < 0) my[col] = acc;
        else if (kind == 3) a.z_out[dst] = acc;
        else if (kind == 4) a.beta_out[dst] = 1.0f / (1.0f + kr_expf(-acc));
        else if (kind == 5) {
            const float ap_dt = acc + g_dt;
            const float softplus = ap_dt > 20.0f ? ap_dt : kr_logf(1.0f + kr_expf(ap_dt));
            a.ge_out[dst] = kr_expf(-(kr_expf(g_al)) * softplus);
        } else {   // decode.rs:3815-3890: depthwise conv1d (kernel 4) over the shifted state, SiLU; the state shift is this lane's alone
            reinterpret_cast<float4*>(a.conv_state)[ch] = float4{cs.y, cs.z, cs.w, acc};
            float co = cs.y * cw.x + cs.z * cw.y + cs.w * cw.z + acc * cw.w;
            co = co * kr_sigmoid_poly5(co);
            if (kind == 2) a.v_out[dst] = co; else a.qk_out[dst] = co;
        }
    }
    KR_FSTAMP(sk, 5); KR_FWG(sk, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K2: gated delta-rule step of one VALUE head (decode.rs:3891-3945, 1293, 3979).  512 threads: NS slices of RPS state rows x DV / 4 column quads.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int DK, int DV>
__global__ void __launch_bounds__(512) kr_fla_kernel(const float* p_qk, const float* p_z, const float* p_v, const float* p_ge, const float* p_beta, float* p_state, int p_hr, const KrFlaArgs a) {
    constexpr int JQ = DV / 4, NS = 512 / JQ, RPS = DK / NS, WQ = DK / 64;
    static_assert(RPS >= 1 && NS * RPS == DK, "geometry");
    __shared__ float s_qk[2 * DK];
    __shared__ float s_red[8];
    __shared__ __attribute__((aligned(16))) float s_part[NS][DV];
    __shared__ __attribute__((aligned(16))) float s_vec[DV];
    const int vh = blockIdx.x, kh = vh / p_hr;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int slice = t / JQ, jq = t - slice * JQ;
    f32x4* S4 = reinterpret_cast<f32x4*>(p_state + (size_t)vh * DK * DV);
    KR_FSTAMP(1, 0); KR_FWG(1, 0);
    // the head's vectors first (every thread loads, clamped: kr_f_norm_load), the 64 KB of state behind them -- the q / k sums run while the state streams in
    const float g_exp = p_ge[vh], beta = p_beta[vh];   // e^g and beta of this head: formed by the ba lanes of the projection launch
    float qv = p_qk[(size_t)kh * 2 * DK + (t < 2 * DK ? t : 0)];
    const size_t ov_ = (size_t)vh * DV + (t < DV ? t : 0);
    float zz = p_z[ov_], vv = p_v[ov_], wn = a.norm_w[ov_];
    asm volatile("" ::: "memory");      // (keeps the compiler from sinking these requests below the state's)
    f32x4 c[RPS];
#pragma unroll
    for (int u = 0; u < RPS; u++) c[u] = __builtin_nontemporal_load(S4 + (size_t)(slice * RPS + u) * JQ + jq);
    if (t < 2 * DK) s_qk[t] = qv; else qv = 0.0f;
    const float sq = kr_f_wave_sum(qv * qv);
    if (lane == 0) s_red[wave] = sq;
    KR_FSTAMP(1, 1);
    __syncthreads();
    KR_FSTAMP(1, 2);
    float ssq = s_red[0], ssk = s_red[WQ];
#pragma unroll
    for (int w = 1; w < WQ; w++) { ssq += s_red[w]; ssk += s_red[WQ + w]; }
    const float inv_q = (ssq > 0.0f ? 1.0f / sqrtf(ssq) : 0.0f) * a.scale, inv_k = ssk > 0.0f ? 1.0f / sqrtf(ssk) : 0.0f;
    float kk[RPS], qq[RPS];
#pragma unroll
    for (int u = 0; u < RPS; u++) { kk[u] = s_qk[DK + slice * RPS + u] * inv_k; qq[u] = s_qk[slice * RPS + u] * inv_q; }
    f32x4 kvp = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < RPS; u++) {
        c[u] = c[u] * g_exp;
        kvp.x = __builtin_fmaf(c[u].x, kk[u], kvp.x); kvp.y = __builtin_fmaf(c[u].y, kk[u], kvp.y);
        kvp.z = __builtin_fmaf(c[u].z, kk[u], kvp.z); kvp.w = __builtin_fmaf(c[u].w, kk[u], kvp.w);
    }
    reinterpret_cast<f32x4*>(s_part[slice])[jq] = kvp;
    KR_FSTAMP(1, 3);
    __syncthreads();
    if (t < DV) {
        float kv = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; s++) kv += s_part[s][t];
        s_vec[t] = (vv - kv) * beta;
    }
    __syncthreads();
    KR_FSTAMP(1, 4);
    const f32x4 d4 = reinterpret_cast<const f32x4*>(s_vec)[jq];
    f32x4 op = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < RPS; u++) {
        f32x4 sn;
        sn.x = __builtin_fmaf(kk[u], d4.x, c[u].x); sn.y = __builtin_fmaf(kk[u], d4.y, c[u].y);
        sn.z = __builtin_fmaf(kk[u], d4.z, c[u].z); sn.w = __builtin_fmaf(kk[u], d4.w, c[u].w);
        __builtin_nontemporal_store(sn, S4 + (size_t)(slice * RPS + u) * JQ + jq);
        op.x = __builtin_fmaf(sn.x, qq[u], op.x); op.y = __builtin_fmaf(sn.y, qq[u], op.y);
        op.z = __builtin_fmaf(sn.z, qq[u], op.z); op.w = __builtin_fmaf(sn.w, qq[u], op.w);
    }
    reinterpret_cast<f32x4*>(s_part[slice])[jq] = op;    // every read of the first partials happened before the previous barrier
    KR_FSTAMP(1, 5);
    __syncthreads();
    float ob = 0.0f;
    if (t < DV) {
#pragma unroll
        for (int s = 0; s < NS; s++) ob += s_part[s][t];
        const float so = kr_f_wave_sum(ob * ob);
        if (lane == 0) s_red[wave] = so;
    }
    __syncthreads();
    KR_FSTAMP(1, 6);
    float ov = 0.0f;
    if (t < DV) {
        float ss = s_red[0];
#pragma unroll
        for (int w = 1; w < DV / 64; w++) ss += s_red[w];
        const float rms = 1.0f / sqrtf(ss / (float)DV + a.eps);
        const float normed = (ob * rms) * wn;
        ov = (zz * kr_sigmoid_poly5(zz)) * normed;
        a.out[(size_t)vh * DV + t] = ov;
        s_vec[t] = ov;
    }
    if (a.img_out && DV == 128) {   // the head is one quantization group of the out-projection's input
        __syncthreads();
        if (t < 16) {
            const KrActLds Lg = kr_carve_lds(reinterpret_cast<u32x4*>(a.img_out), a.img_k, false);
            float v8[8];
            kr_load8(s_vec, t, v8);
            float mx = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v8[i]));
            float scale, inv;
            kr_group_scale(mx, scale, inv);
            int q8[8];
            kr_quant8<false>(v8, inv, q8);
            kr_store_chunk<false>(Lg, vh * 16 + t, q8);
            if (t == 0) Lg.ascale[vh] = scale;
        }
    }
    KR_FSTAMP(1, 7); KR_FWG(1, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K4: post-attention add + RMSNorm folded into the router gate GEMV (decode.rs:1385-1429).  One workgroup per block of 4 experts, its 4 waves
// split the K range of the chain-major gate rows.  Workgroups 0 / 1 also publish the residual and the two INT16 images.
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool GATE_BF16, int CPW>      // CPW: 16-byte gate chunks per lane and wave kept in flight, the power of two >= ceil(chunks / 4) (H <= 4096: bf16 <= 8, f32 <= 16)
__global__ void __launch_bounds__(256) kr_frt_kernel(const float* p_hid, const float* p_res, const float* p_nw, const void* p_gate, int p_H, const KrFrtArgs a) {
    float* xs = reinterpret_cast<float*>(kr_fsm);   // [16][ld] chain-major copy of the normalised hidden
    __shared__ float s_red[4];
    __shared__ float s_part[4][4];
    const int H = p_H, ld = H / 16 + 4;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int eb = blockIdx.x;
    KR_FSTAMP(3, 0); KR_FWG(3, 0);
    const int ncg = GATE_BF16 ? H / 128 : H / 64;
    const int cpw = (ncg + 3) / 4, c0 = wave * cpw, c1 = c0 + cpw < ncg ? c0 + cpw : ncg;
    const u32x4* gp = reinterpret_cast<const u32x4*>(p_gate) + (size_t)eb * ncg * 64 + lane;
    KrFNormIn in{p_hid, p_res, p_nw, a.res_out, 0, a.eps, a.bias_one, H};
    KrFNormRegs NR;
    kr_f_norm_load(in, NR);      // the norm's inputs come back first, the gate rows stream behind them (kr_f_norm_load)
    u32x4 gw[CPW];
#pragma unroll
    for (int u = 0; u < CPW; u++) { const int c = c0 + u < ncg ? c0 + u : ncg - 1; gw[u] = kr_ldg_nt(gp + (size_t)c * 64); }      // (every wave issues CPW requests: past the row's end it re-reads the last chunk)
    float x[2][8];
    kr_f_norm_finish(in, NR, x, s_red, blockIdx.x == 0);
    KR_FSTAMP(3, 1);
    const bool img_f = a.img_f32 && blockIdx.x == 0, img_b = a.img_bf16 && blockIdx.x == (gridDim.x > 1 ? 1 : 0);
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int c = t + 256 * u;
        if (c < H / 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) { const int e = c * 8 + i; xs[(e & 15) * ld + (e >> 4)] = x[u][i]; }
            if (blockIdx.x == 0 && a.hid_out) {
                float4* ho = reinterpret_cast<float4*>(a.hid_out + (size_t)c * 8);
                ho[0] = float4{x[u][0], x[u][1], x[u][2], x[u][3]}; ho[1] = float4{x[u][4], x[u][5], x[u][6], x[u][7]};
            }
            if (img_f) kr_f_quant_chunk<false>(x[u], c, kr_carve_lds(reinterpret_cast<u32x4*>(a.img_f32), H, false), false);
            if (img_b) kr_f_quant_chunk<false>(x[u], c, kr_carve_lds(reinterpret_cast<u32x4*>(a.img_bf16), H, false), true);
        }
    }
    __syncthreads();
    KR_FSTAMP(3, 2);
    const int j = lane & 15;
    const float* xj = xs + j * ld;
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < CPW; u++) if (c0 + u < c1) {
        if (GATE_BF16) {
            const float* xx = xj + (c0 + u) * 8;
            const uint32_t ww[4] = {gw[u].x, gw[u].y, gw[u].z, gw[u].w};
#pragma unroll
            for (int p = 0; p < 4; p++) {
                acc = __builtin_fmaf(__uint_as_float(ww[p] << 16), xx[2 * p], acc);
                acc = __builtin_fmaf(__uint_as_float(ww[p] & 0xFFFF0000u), xx[2 * p + 1], acc);
            }
        } else {
            const float* xx = xj + (c0 + u) * 4;
            acc = __builtin_fmaf(__uint_as_float(gw[u].x), xx[0], acc);
            acc = __builtin_fmaf(__uint_as_float(gw[u].y), xx[1], acc);
            acc = __builtin_fmaf(__uint_as_float(gw[u].z), xx[2], acc);
            acc = __builtin_fmaf(__uint_as_float(gw[u].w), xx[3], acc);
        }
    }
    acc = kr_f_red16(acc);
    KR_FSTAMP(3, 3);
    if (j == 0) s_part[wave][lane >> 4] = acc;
    __syncthreads();
    if (t < 4) {
        const int e = eb * 4 + t;
        if (e < a.E) {
            float v = (s_part[0][t] + s_part[1][t]) + (s_part[2][t] + s_part[3][t]);
            if (a.bias) v += a.bias[e];   // decode.rs:3292-3294
            a.logits[e] = v;
        }
    }
    KR_FSTAMP(3, 4); KR_FWG(3, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// wave-wide top-np in (value desc, index asc) order WITHOUT a tournament over all elements (kr_topk_wave_reg walks np rounds of: scan the
// lane's NV keys, wave maximum, retire the winner -- ~65 instructions each on a lone wave, and a lone wave issues one instruction per ~9 cycles):
//   (1) T = a lower bound of the np-th largest element from ceil(np / 4) rounds PER 16-LANE ROW: the row maximum of the lane maxima (four DPP steps, no cross-row
//       step), the lanes that hold it retire.  Every row-round retires at least one lane whose maximum is >= that row's last maximum, so at least 4 ceil(np / 4) >= np
//       elements are >= T = the smallest of the four rows' last maxima.  (Round 4 ran np rounds of a whole-wave maximum here: 0.9 us of the 3.3 us select.)
//   (2) the elements >= T (np of them plus, typically, ten more) are compacted into an LDS list of 64-bit records (key << 32 | ~index: one unsigned compare orders two
//       candidates by value desc, index asc),
//   (3) every candidate lane counts the records that precede its own -- its rank -- reading the list from LDS four records at a time (uniform addresses: broadcasts);
//       ranks < np are written to pv / pi.  (Round 4: one lane-broadcast pair + three compares per candidate, 0.7 us.)
// Returns false (nothing written) when more than 64 elements reach T (a row ran out of lanes: tiny or heavily tied inputs): the caller runs the tournament.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ bool kr_f_topk(const float (&val)[NV], int n, int np, float* pv, int* pi, uint64_t* cand /* LDS [64 + 4], 8-byte aligned */) {
    const int lane = threadIdx.x & 63;
    uint32_t key[NV], h = 0u;
#pragma unroll
    for (int i = 0; i < NV; i++) { const int e = lane * NV + i; key[i] = e < n ? kr_make_key(val[i]) : 0u; h = kr_umax(h, key[i]); }
    KR_FSTAMP(6, 2);
    uint32_t trow = 0u;
    const int rounds = (np + 3) >> 2;
    for (int r = 0; r < rounds; r++) {
        uint32_t w = h;
        w = kr_umax(w, (uint32_t)KR_DPP((int)w, KR_DPP_XOR1));
        w = kr_umax(w, (uint32_t)KR_DPP((int)w, KR_DPP_XOR2));
        w = kr_umax(w, (uint32_t)KR_DPP((int)w, KR_DPP_HALF_MIRROR));
        w = kr_umax(w, (uint32_t)KR_DPP((int)w, KR_DPP_MIRROR));      // every lane of the row holds the row maximum
        trow = w;
        if (h == w) h = 0u;
    }
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)trow, 0), t1 = (uint32_t)__builtin_amdgcn_readlane((int)trow, 16);
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)trow, 32), t3 = (uint32_t)__builtin_amdgcn_readlane((int)trow, 48);
    const uint32_t T01 = t0 < t1 ? t0 : t1, T23 = t2 < t3 ? t2 : t3, T = T01 < T23 ? T01 : T23;
    KR_FSTAMP(6, 3);
    int base = 0;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const bool c = key[i] != 0u && key[i] >= T;
        const uint64_t mask = __ballot(c);
        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (c && pos < 64) cand[pos] = ((uint64_t)key[i] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(lane * NV + i));
        base += __popcll(mask);
    }
    const int M = base;
    if (M > 64) return false;
    if (lane < 4) cand[M + lane] = 0ull;      // the rank loop reads four records at a time: an empty record precedes nothing
    kr_f_wave_sync();
    KR_FSTAMP(6, 4);
    const uint64_t cj = cand[lane < M ? lane : 0];
    int rank = 0;
    for (int i = 0; i < M; i += 4) {
        const uint64_t c0 = cand[i], c1 = cand[i + 1], c2 = cand[i + 2], c3 = cand[i + 3];
        rank += (c0 > cj ? 1 : 0) + (c1 > cj ? 1 : 0) + (c2 > cj ? 1 : 0) + (c3 > cj ? 1 : 0);
    }
    if (lane < M && rank < np) { pv[rank] = kr_key_value((uint32_t)(cj >> 32)); pi[rank] = (int)(0xFFFFFFFFu - (uint32_t)cj); }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// scoring + top-k by ONE wave (moe_route_score_topk + topk_indices, decode.rs:4088-4186, 1495-1535): the scores and the renormalisation are
// wave trees; the selection is the (value desc, index asc) wave top-k of kr_topk.h; softmax without a correction bias selects on the LOGITS
// (softmax is monotone), so for identical logits the ids are those of the exact kernel unless two of the leading k + 1 are EQUAL -- then the
// reference's heap order decides and is emulated serially, as in the exact kernel.  (Stated exception of the `lean` path below, krasis_hip.h KR_DECODE_FAST: it compares
// LOGITS; two distinct logits whose f32 softmax scores coincide are a tie for the exact kernel only.)
// sm: [E] scores, [E] selection values, [33] pv, [33] pi, [32] hv, [32] hi, [2] pad, [136] candidate list (68 records of 8 bytes)
// ---------------------------------------------------------------------------------------------------------------------------------
// The `lean` rule on its own straight path: softmax scoring without a correction bias, weights renormalised over the k leaders (QCN, Qwen3-235B).  The selection runs on
// the logits (softmax is monotone) and the renormalised weight of leader i is e^{l_i - m} / sum over the LEADERS of e^{l_j - m} -- the full-softmax denominator cancels,
// so the E exponentials, their wave sum and the score / selection arrays in LDS are never formed.  Against the reference's (e_i / S) / sum_j (e_j / S): two roundings fewer
// per weight, within 3e-7 relative (test bound).  The leaders' logits are read back from the sorted list (pv), not from global memory.
template <int NV>
__device__ __forceinline__ void kr_f_select_lean(const float* logits, int E, int k, float* sm, int* s_ids, float* s_w) {
    float* sel = sm + E; float* pv = sel + E; int* pi = reinterpret_cast<int*>(pv + 33);
    float* hv = reinterpret_cast<float*>(pi + 33); int* hi = reinterpret_cast<int*>(hv + 32);
    uint64_t* cand = reinterpret_cast<uint64_t*>(hi + 32 + 2);
    const int lane = threadIdx.x & 63;
    KR_FSTAMP(6, 0);
    float lg[NV];
    if (NV % 4 == 0 && E % 4 == 0) {
#pragma unroll
        for (int i = 0; i + 3 < NV; i += 4) {
            const int e = lane * NV + i;
            const float4 v = *reinterpret_cast<const float4*>(logits + (e < E ? e : 0));
            lg[i] = v.x; lg[i + 1] = v.y; lg[i + 2] = v.z; lg[i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = lane * NV + i; lg[i] = logits[e < E ? e : 0]; }
    }
#ifdef KR_FTIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    KR_FSTAMP(6, 1);
    const int np = k + 1 <= E ? k + 1 : k;
    if (!kr_f_topk<NV>(lg, E, np, pv, pi, cand)) kr_topk_wave_reg<NV>(lg, E, np, pv, pi);      // (elements e >= E carry key 0 in both: never selected)
    kr_f_wave_sync();
    KR_FSTAMP(6, 5);
    const float pa = lane < np ? pv[lane] : 0.0f, pb = lane + 1 < np ? pv[lane + 1] : 0.0f;
    if (__ballot(lane + 1 < np && pa == pb) != 0ull) {   // two of the leading k + 1 are equal: the heap order governs (decode.rs:1531); rare
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = lane * NV + i; if (e < E) sel[e] = lg[i]; }
        kr_f_wave_sync();
        if (lane == 0) kr_topk_heap_serial(sel, E, k, hv, hi, pi);
        kr_f_wave_sync();
    }
    // the leaders come out in descending order, pv[0] is the largest logit; under a tie the heap order may permute EQUAL values only, so pv[lane] is the logit of leader `lane`
    float wv = lane < k ? __builtin_amdgcn_exp2f((pa - pv[0]) * 1.4426950408889634f) : 0.0f;
    const float se = kr_f_wave_sum(wv);
    wv = wv / se;
    if (lane < k) { s_ids[lane] = pi[lane]; s_w[lane] = wv; }
    KR_FSTAMP(6, 6);
}

template <int NV>
__device__ __forceinline__ void kr_f_select(const float* logits, const float* esc, int E, int k, int scoring, int norm, float* sm, int* s_ids, float* s_w) {
    if (scoring == 1 && esc == nullptr && norm != 0) { kr_f_select_lean<NV>(logits, E, k, sm, s_ids, s_w); return; }
    float* scores = sm; float* sel = sm + E; float* pv = sel + E; int* pi = reinterpret_cast<int*>(pv + 33);
    float* hv = reinterpret_cast<float*>(pi + 33); int* hi = reinterpret_cast<int*>(hv + 32);
    uint64_t* cand = reinterpret_cast<uint64_t*>(hi + 32 + 2);      // [64 + 4] records of 8 bytes (the offset from `sm` is 8 E + 528 bytes)
    const int lane = threadIdx.x & 63;
    const bool raw = scoring == 2;
    float lg[NV], sc[NV], sl[NV];
    bool wide = false;
    if constexpr (NV % 4 == 0) {
        if (E % 4 == 0) {
            wide = true;
#pragma unroll
            for (int i = 0; i < NV; i += 4) {
                const int e = lane * NV + i;
                const float4 v = e < E ? *reinterpret_cast<const float4*>(logits + e) : float4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
                lg[i] = v.x; lg[i + 1] = v.y; lg[i + 2] = v.z; lg[i + 3] = v.w;
            }
        }
    }
    if (!wide) {
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = lane * NV + i; lg[i] = e < E ? logits[e] : -__builtin_inff(); }
    }
    if (raw) {
#pragma unroll
        for (int i = 0; i < NV; i++) sc[i] = lg[i];
    } else if (scoring == 0) {
        const int e8 = (E / 8) * 8;
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = lane * NV + i; sc[i] = e < E ? (e < e8 ? kr_sigmoid_poly4(lg[i]) : 1.0f / (1.0f + kr_expf(-lg[i]))) : 0.0f; }
    } else {
        float mx = lg[0];
#pragma unroll
        for (int i = 1; i < NV; i++) mx = fmaxf(mx, lg[i]);
        mx = kr_f_wave_max(mx);
        float se = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; i++) { const int e = lane * NV + i; sc[i] = e < E ? __builtin_amdgcn_exp2f((lg[i] - mx) * 1.4426950408889634f) : 0.0f; se += sc[i]; }
        se = kr_f_wave_sum(se);
        const float inv = 1.0f / se;
#pragma unroll
        for (int i = 0; i < NV; i++) sc[i] *= inv;
    }
    const bool on_logits = scoring == 1 && esc == nullptr;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int e = lane * NV + i;
        sl[i] = on_logits ? lg[i] : ((!raw && esc && e < E) ? sc[i] + esc[e] : sc[i]);
        if (e < E) { scores[e] = sc[i]; sel[e] = sl[i]; }
    }
    const int np = k + 1 <= E ? k + 1 : k;
    if (!kr_f_topk<NV>(sl, E, np, pv, pi, cand)) kr_topk_wave_reg<NV>(sl, E, np, pv, pi);
    kr_f_wave_sync();
    {
        const float pa = lane < np ? pv[lane] : 0.0f, pb = lane + 1 < np ? pv[lane + 1] : 0.0f;
        const bool tie = __ballot(lane + 1 < np && pa == pb) != 0ull;
        if (tie) {   // heap order governs ties (decode.rs:1531)
            if (lane == 0) kr_topk_heap_serial(sel, E, k, hv, hi, pi);
            kr_f_wave_sync();
        }
    }
    const int my = lane < k ? pi[lane] : 0;
    float wv = lane < k ? scores[my] : 0.0f;
    if (raw) {
        const float mx = kr_f_wave_max(lane < k ? wv : -__builtin_inff());
        wv = lane < k ? kr_expf(wv - mx) : 0.0f;
    }
    if (raw || norm) {
        const float se = kr_f_wave_sum(wv);
        if (raw) wv = wv * (1.0f / se);
        else if (se > 0.0f) wv = wv / se;
    }
    if (lane < k) { s_ids[lane] = my; s_w[lane] = wv; }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K5: gate | up of the k routed experts + the shared expert (+ its sigmoid-gate row).  grid (units, n_slots); 2 tile PAIRS per workgroup
// (gate tile t and up tile t + I / 8 of the same hidden column), 2 waves split K; epilogue h = silu(g) * u (avx2.rs:2331-2333, decode.rs:1731-1733).
// ---------------------------------------------------------------------------------------------------------------------------------
// (leading scalars = what the select wave and the image copy need for their first requests, preloaded into SGPRs: see kr_fdm_kernel)
template <int BITS, int NU>
__global__ void __launch_bounds__(256) kr_fw13_kernel(const float* p_logits, const float* p_esc, const void* p_img_bf16, int p_E, int p_topk, int p_scoring, int p_norm, int p_H, int p_I, int p_gguf,
                                                      const KrFmoeArgs fa) {
    const KrMoeArgs& a = fa.m;
    constexpr int KS = 2, TW = 2;
    __shared__ int s_ids[32];
    __shared__ float s_w[32];
    __shared__ float s_x[TW][KS][2][8];
    const int slot = blockIdx.y;
    const bool shared = slot >= p_topk;
    // the grid's x extent follows the widest slot (a shared expert may be wider than the routed ones): a routed workgroup past its expert's last tile pair leaves
    // before the select and the image copy, not after them (DeepSeek-V2-Lite: half of the launch's workgroups)
    if (!shared && (int)blockIdx.x * 2 /* TW */ >= p_I / 8) return;      // (workgroup (0, 0), which publishes the routing, always has a tile pair)
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, l8 = lane & 7, cl = lane >> 3;
    const int tw = wave >> 1, ks = wave & 1;
    const KrActLds L = kr_carve_lds(kr_fsm, p_H, BITS == 8);
    const bool gg = p_gguf && !shared;                  // routed slot on native GGUF blocks: per-32 activation image instead of the per-128 one
    const GgAct GA = gg_carve(reinterpret_cast<char*>(kr_fsm), p_H);
    const size_t img_bytes = p_gguf ? (kr_lds_bytes(p_H, BITS == 8) > gg_lds_bytes(p_H, false) ? kr_lds_bytes(p_H, BITS == 8) : gg_lds_bytes(p_H, false)) : kr_lds_bytes(p_H, BITS == 8);
    KR_FSTAMP(4, 0); KR_FWG(4, 0);
    if (shared) kr_f_image_copy<BITS>(a.act_img, p_H, kr_fsm, L, t, 256);
    else if (wave > 0) {
        if (gg) {      // quantize_bf16_to_int16 of bf16(hidden) (gguf_kernels.rs:110, decode.rs:3307-3309), 8 values per thread, 4 consecutive lanes per sub-block
            for (int c = t - 64; c < a.H / 8; c += 192) {
                float v[8];
                kr_load8(fa.act_f32, c, v);
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = kr_bf16_to_f32(kr_f32_to_bf16(v[i]));
                gg_quant_store(v, c, GA);
            }
        } else kr_f_image_copy<BITS>(p_img_bf16, p_H, kr_fsm, L, t - 64, 192);
    } else {
        float* selsm = reinterpret_cast<float*>(reinterpret_cast<char*>(kr_fsm) + img_bytes);
        const int nv = (p_E + 63) / 64;
        if (nv <= 1) kr_f_select<1>(p_logits, p_esc, p_E, p_topk, p_scoring, p_norm, selsm, s_ids, s_w);
        else if (nv <= 2) kr_f_select<2>(p_logits, p_esc, p_E, p_topk, p_scoring, p_norm, selsm, s_ids, s_w);
        else if (nv <= 4) kr_f_select<4>(p_logits, p_esc, p_E, p_topk, p_scoring, p_norm, selsm, s_ids, s_w);
        else kr_f_select<8>(p_logits, p_esc, p_E, p_topk, p_scoring, p_norm, selsm, s_ids, s_w);
        if (blockIdx.x == 0 && slot == 0 && lane < p_topk) {   // the routing of this token, for the w2 launch (and anyone who asks)
            const_cast<int32_t*>(a.ids)[lane] = s_ids[lane]; const_cast<float*>(a.wts)[lane] = s_w[lane];
        }
    }
    KR_FSTAMP(4, 1);
    __syncthreads();
    KR_FSTAMP(4, 2);
    if (gg) {
        // ---- native GGUF blocks: tile pair (gate tile `unit`, up tile `unit`) of expert e, the row's blocks split over the two k-waves
        int e = s_ids[slot];
        if (a.e_hi > 0) { if (e < a.e_lo || e >= a.e_hi) return; e -= a.e_sub; }
        const GgMat gm = gg_expert_mat(fa.ggate, e), um = gg_expert_mat(fa.gup, e);
        const int ntp = gm.N / 8, unit = blockIdx.x * TW + tw;
        float ag = 0.0f, au = 0.0f;
        if (unit < ntp) {
            if (gm.type == GG_Q4_K) ag = gg_tile_q4k(gm, unit, GA, lane, ks, KS); else if (gm.type == GG_Q8_0) ag = gg_tile_q8_0(gm, unit, GA, lane, ks, KS); else ag = gg_tile_q4_0(gm, unit, GA, lane, ks, KS);
            if (um.type == GG_Q4_K) au = gg_tile_q4k(um, unit, GA, lane, ks, KS); else if (um.type == GG_Q8_0) au = gg_tile_q8_0(um, unit, GA, lane, ks, KS); else au = gg_tile_q4_0(um, unit, GA, lane, ks, KS);
        }
        if (l8 == 0) { s_x[tw][ks][0][cl] = ag; s_x[tw][ks][1][cl] = au; }
        __syncthreads();
        if (ks == 0 && l8 == 0 && unit < ntp) {
            const float g = s_x[tw][0][0][cl] + s_x[tw][1][0][cl], u = s_x[tw][0][1][cl] + s_x[tw][1][1][cl];
            a.gu[(size_t)slot * a.gu_ld + unit * 8 + cl] = (g / (1.0f + kr_expf(-g))) * u;      // gguf_kernels.rs:733-737
        }
        return;
    }
    // (Measured and removed, round 5: reading the struct fields this part needs BEFORE the select and pinning them in SGPRs -- the pin is a wait: the select wave
    //  then starts behind those scalar loads, "weight fetch issued" 1.34 -> 1.80 us.)
    const KrMatDev& m = shared ? a.sw13 : a.w13;
    const void* qb = m.q; const uint32_t* sb = m.s;
    const int inter = shared ? a.I_shared : a.I;
    if (!shared) {
        int e = s_ids[slot];
        if (a.e_hi > 0) {      // expert-parallel decode: this rank evaluates the slots whose expert lies in its slice only (workgroup-uniform exit, no barrier is left behind it)
            if (e < a.e_lo || e >= a.e_hi) return;
            e -= a.e_sub;
        }
        qb = reinterpret_cast<const char*>(m.q) + (size_t)e * m.q_stride;
        sb = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(m.s) + (size_t)e * m.s_stride);
    } else if (fa.shared_skip) return;      // expert-parallel decode: another rank evaluates this layer's (replicated) shared expert
    const int ntp = inter / 8;
    const int unit = blockIdx.x * TW + tw;
    const bool pair = unit < ntp;
    const bool gate_row = shared && a.sgate.q != nullptr && unit == ntp;
    const int units = BITS == 4 ? m.ngp : m.ng;
    int u0, u1;
    if (NU > 0) { u0 = ks * NU; u1 = u0 + NU; }
    else { const int uw = (units + KS - 1) / KS; u0 = ks * uw; u1 = u0 + uw < units ? u0 + uw : units; }
    KrFw<BITS, NU, 8> Wg, Wu;      // guarded form: 8 records per tile in flight (two K-waves: enough for K <= 4096 in one pass; 16 made the launch a one-workgroup-per-CU kernel)
    float ag = 0.0f, au = 0.0f;
    if (pair) {
        kr_f_fetch<BITS, NU, 8>(Wg, m, qb, sb, unit, lane, u0, u1);
        kr_f_fetch<BITS, NU, 8>(Wu, m, qb, sb, unit + ntp, lane, u0, u1);
        KR_FSTAMP(4, 3);
        ag = kr_f_tile<BITS, NU, 8>(Wg, m, qb, sb, unit, lane, u0, u1, L);
        au = kr_f_tile<BITS, NU, 8>(Wu, m, qb, sb, unit + ntp, lane, u0, u1, L);
    } else if (gate_row) {
        kr_f_fetch<BITS, NU, 8>(Wg, a.sgate, a.sgate.q, a.sgate.s, 0, lane, u0, u1);
        ag = kr_f_tile<BITS, NU, 8>(Wg, a.sgate, a.sgate.q, a.sgate.s, 0, lane, u0, u1, L);
    }
    KR_FSTAMP(4, 4);
    if (l8 == 0) { s_x[tw][ks][0][cl] = ag; s_x[tw][ks][1][cl] = au; }
    __syncthreads();
    if (ks == 0 && l8 == 0) {
        const float g = s_x[tw][0][0][cl] + s_x[tw][1][0][cl], u = s_x[tw][0][1][cl] + s_x[tw][1][1][cl];
        if (pair) a.gu[(size_t)slot * a.gu_ld + unit * 8 + cl] = (g * kr_sigmoid_poly5(g)) * u;
        else if (gate_row && cl == 0) a.gate_out[0] = 1.0f / (1.0f + kr_expf(-g));   // sigmoid of the shared expert's gate row (decode.rs:3379-3393): the w2 launch multiplies by it
    }
    KR_FSTAMP(4, 5); KR_FWG(4, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K6: down projection of ALL slots for one 8-column tile + the weighted combine.  One wave per slot: it quantises its expert's hidden
// (silu_quantize_int16_avx2's cvtps rounding for routed experts, avx2.rs:2357; f32::round for the decode-store shared expert,
// decode.rs:3364-3374) into its own LDS region, walks the I / 128 groups and leaves 8 column values; 8 lanes then form
// rsf * sum_i w_i y_i (routing order, moe.rs:661-667) + shared * sigmoid(gate) (decode.rs:3379-3402).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int BITS, int NU, bool MULTI>
__global__ void __launch_bounds__(1024) kr_fw2_kernel(const float* p_gu, const int32_t* p_ids, const float* p_wts, int p_gu_ld, int p_topk, int slot_lds, int pr_, int ps_, const KrFmoeArgs fa) {
    const KrMoeArgs& a = fa.m;
    const int pr = MULTI ? pr_ : 1, ps = MULTI ? ps_ : 1;      // MULTI = false: one wave per slot, the part arithmetic below folds away (the headline shapes: every cycle of this launch is on the token's path)
    __shared__ float s_y[16][8];
    __shared__ __attribute__((aligned(16))) float s_wt[16];
    // A slot may be walked by SEVERAL waves ("parts": pr per routed slot, ps for the shared one; blockDim = 64 (topk pr + ps)): each takes a slice of the expert's
    // units and quantises the matching chunks of its hidden, the column sums meet in the combine.  With one wave per slot a wide expert (I = 1408: 176 chunks to
    // quantise on 64 lanes, then 6 units) made the launch twice as long as at I = 512, and a shared expert twice as wide again was the workgroup's critical path.
    const int t = threadIdx.x, vslot = t >> 6, lane = t & 63, l8 = lane & 7, cl = lane >> 3;
    const int nrw = p_topk * pr;                             // waves of the routed slots
    const int slot = vslot < nrw ? vslot / pr : p_topk, part = vslot < nrw ? vslot % pr : vslot - nrw, parts = vslot < nrw ? pr : ps;
    const int tile = blockIdx.x;
    KR_FSTAMP(5, 0); KR_FWG(5, 0);
    // sigmoid(gate row) of the shared expert, formed by the gate|up launch.  A rank that skips this layer's shared expert (expert-parallel decode) never wrote it:
    // it neither reads the value nor adds the term (0 * stale bits could be NaN and the all-reduce would spread it -- ADVICE r4 #2)
    const float sig = (a.gate_out && !fa.shared_skip) ? a.gate_out[0] : 1.0f;
    const bool shared = slot >= p_topk;
    const bool gg = fa.gguf && !shared;
    const KrMatDev& m = (shared || fa.gguf) ? a.sw2 : a.w2;      // (a GGUF slot never reads m: the shared expert's matrix stands in so that the fields are defined)
    const int inter = shared ? a.I_shared : a.I;
    // one wave per slot over an exact unit count: the lane's chunks of the expert hidden are requested before anything else (ahead of the routing record and the
    // weight stream, which waits for the record: see kr_f_norm_load)
    constexpr int HCN = (NU > 0 && !MULTI) ? (BITS == 4 ? NU / 2 : NU / 4) : 0;      // exact-unit forms: chunks per lane
    constexpr int HC = HCN > 0 ? 1 : 0;
    // this wave's units of the slot's expert and the FIRST chunk of the hidden it will quantise: requested here, ahead of the routing record and the weight records
    // (16 waves of a workgroup leave 128 registers a lane: one chunk ahead, the others in the loops below)
    const int units = BITS == 4 ? m.ngp : m.ng;
    const int u0 = units * part / parts, u1 = units * (part + 1) / parts;
    const int cpu = (BITS == 4 ? 256 : 128) / 8;            // 8-value chunks per unit
    const int c0 = u0 * cpu + lane, cend = u1 * cpu < inter / 8 ? u1 * cpu : inter / 8;
    constexpr bool PRE = !MULTI;      // (the several-waves-per-slot form is at its 128 registers: one chunk ahead spills there)
    [[maybe_unused]] float hpre[1][8];
    if constexpr (PRE) kr_load8(p_gu + (size_t)slot * p_gu_ld, c0 < inter / 8 ? c0 : inter / 8 - 1, hpre[0]);
    const void* qb = m.q; const uint32_t* sb = m.s;
    bool valid = true; float wt = 1.0f;
    bool skip = false;        // expert-parallel decode: slot evaluated by another rank -- this wave contributes 0 and reads no weights
    size_t ee = 0;
    if (!shared) {
        int e = p_ids[slot];
        valid = e >= 0 && e < a.E; wt = p_wts[slot];
        if (a.e_hi > 0) { skip = !valid || e < a.e_lo || e >= a.e_hi; e -= a.e_sub; }
        ee = valid && !skip ? (size_t)e : 0;
        if (!gg) {
            qb = reinterpret_cast<const char*>(m.q) + ee * m.q_stride;
            sb = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(m.s) + ee * m.s_stride);
        }
    } else skip = fa.shared_skip != 0;
    if (gg) {
        // ---- native GGUF down projection: per-32 INT16 image of the hidden (f32::round, gguf_kernels.rs:143), the block kernel's row tile, result into the combine below
        const GgAct GA = gg_carve(reinterpret_cast<char*>(kr_fsm) + (size_t)slot * slot_lds, inter);
        const float* hg = a.gu + (size_t)slot * a.gu_ld;
        float accg = 0.0f;
        if (!skip && valid) {
            for (int c = lane; c < inter / 8; c += 64) { float v[8]; kr_load8(hg, c, v); gg_quant_store(v, c, GA); }
            kr_f_wave_sync();
            const GgMat dm = gg_expert_mat(fa.gdown, (int)ee);
            accg = dm.type == GG_Q4_K ? gg_tile_q4k(dm, tile, GA, lane) : dm.type == GG_Q8_0 ? gg_tile_q8_0(dm, tile, GA, lane) : gg_tile_q4_0(dm, tile, GA, lane);
        }
        if (l8 == 0) s_y[vslot][cl] = accg;                 // (GGUF routed slots: pr = 1, the launcher sees to it)
        if (lane == 0) s_wt[slot] = valid ? wt : 0.0f;
    }
    constexpr int FU2 = 8;      // 16 waves per workgroup leave 128 registers per lane: the guarded form keeps 8 records in flight
    KrFw<BITS, NU, FU2> W;
    if (gg) skip = true;     // handled above: the rest of the slot's work is the barrier and the combine
    if (!skip) kr_f_fetch<BITS, NU, FU2>(W, m, qb, sb, tile, lane, u0, u1);
    const KrActLds L = kr_carve_lds(reinterpret_cast<u32x4*>(reinterpret_cast<char*>(kr_fsm) + (size_t)slot * slot_lds), inter, BITS == 8);
    const float* h = a.gu + (size_t)slot * a.gu_ld;
    KR_FSTAMP(5, 1);
    const bool half_away = shared && a.shared_decode;
    if constexpr (HC > 0) {
        if (!skip)
#pragma unroll
        for (int j = 0; j < HCN; j++) {
            const int c = lane + 64 * j;
            float v[8];
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = hpre[0][i];
            } else kr_load8(h, c, v);
            float mx = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v[i]));
            float scale, inv;
            kr_group_scale(mx, scale, inv);
            int q[8];
            if (half_away) kr_quant8<false>(v, inv, q); else kr_quant8<true>(v, inv, q);
            kr_store_chunk<BITS == 8>(L, c, q);
            if ((c & 15) == 0) L.ascale[c >> 4] = scale;
        }
    } else if (!skip)
    for (int c = c0; c < cend; c += 64) {
        float v[8];
        if (PRE && c == c0) {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = hpre[0][i];
        } else kr_load8(h, c, v);
        float mx = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) mx = fmaxf(mx, fabsf(v[i]));
        float scale, inv;
        kr_group_scale(mx, scale, inv);
        int q[8];
        if (half_away) kr_quant8<false>(v, inv, q); else kr_quant8<true>(v, inv, q);
        kr_store_chunk<BITS == 8>(L, c, q);
        if ((c & 15) == 0) L.ascale[c >> 4] = scale;
    }
    kr_f_wave_sync();
    KR_FSTAMP(5, 2);
    const float acc = skip ? 0.0f : kr_f_tile<BITS, NU, FU2>(W, m, qb, sb, tile, lane, u0, u1, L);
    KR_FSTAMP(5, 3);
    if (!gg) {
        if (l8 == 0) s_y[vslot][cl] = acc;
        if (lane == 0 && !part) s_wt[slot] = valid ? wt : 0.0f;
    }
    if (t >= a.n_slots && t < 16) s_wt[t] = 0.0f;
    __syncthreads();
    KR_FSTAMP(5, 4);
    if (t < 8) {
        float y[16], w[16];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) { const float4 v = reinterpret_cast<const float4*>(s_wt)[s4]; w[4 * s4] = v.x; w[4 * s4 + 1] = v.y; w[4 * s4 + 2] = v.z; w[4 * s4 + 3] = v.w; }
#pragma unroll
        for (int sl = 0; sl < 16; sl++) y[sl] = s_y[sl][t];
        float o = 0.0f;
        if (pr == 1) {
#pragma unroll
            for (int sl = 0; sl < 15; sl++) { const float pd = w[sl] * y[sl]; o += sl < a.topk ? pd : 0.0f; }    // routing order (moe.rs:661-667); an invalid id carries weight 0
        } else {
#pragma unroll
            for (int sl = 0; sl < 8; sl++) { const float pd = w[sl] * (y[2 * sl] + y[2 * sl + 1]); o += sl < a.topk ? pd : 0.0f; }      // pr == 2: the two parts of a slot sit side by side
        }
        if (a.rsf != 1.0f) o *= a.rsf;
        if (a.n_slots > a.topk && !fa.shared_skip) {
            float sh = 0.0f;
#pragma unroll
            for (int p = 0; p < 4; p++) sh += p < ps ? s_y[(nrw + p) & 15][t] : 0.0f;
            if (a.gate_out) sh *= sig;
            o = o + sh;
        }
        const int col = tile * 8 + t;
        if (col < a.H) fa.hid_out[col] = o;
    }
    KR_FSTAMP(5, 5); KR_FWG(5, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// GQA layers over a SHORT cache (kv_max_seq <= KR_FGQA_MAX, the reference's decode benchmark runs 256): what the exact path does in two launches -- kr_gqa_prep_kernel
// (gated split, per-head QK RMSNorm, half-split RoPE, KV append; decode.rs:2873-2966) and kr_gqa_attn_kernel (decode.rs:4194-4281: q.k, softmax, p.v, output gate)
// -- as ONE launch of one workgroup per query head.  Same operations and the same libm functions; what changes is the order of the f32 sums (lane / wave trees) and
// who walks what: 16 lanes per cache position in the score pass (a wave takes 4 positions per step, the workgroup 16), one wave per cache row in the p.v pass (the
// four waves take every fourth position and meet in LDS).  The new K / V row of this step is used as the cache will hold it (rounded to FP16 / E4M3), every query head
// of a KV group forms it for itself and the group's first head stores it.  Requests run a batch of positions ahead of the arithmetic.
// (Round 3 tried a single launch with one thread per output dimension in the p.v pass: faster below 100 positions, slower at 200 -- not kept.  The two passes here
// spread every position over lanes, so a 256-position cache costs ~16 steps of each.)
// ---------------------------------------------------------------------------------------------------------------------------------
#define KR_FGQA_MAX 1024
template <int N, bool FP8> struct KrKvRaw { static constexpr int W = FP8 ? (N + 3) / 4 : (N + 1) / 2; uint32_t w[W]; };
template <int N, bool FP8>
__device__ __forceinline__ void kr_kv_raw_load(KrKvRaw<N, FP8>& r, const void* base, size_t elem) {
    constexpr int bytes = FP8 ? N : 2 * N;
    const char* p = reinterpret_cast<const char*>(base) + elem * (FP8 ? 1 : 2);
    if constexpr (bytes >= 16) {
#pragma unroll
        for (int i = 0; i < bytes / 16; i++) { const u32x4 v = *reinterpret_cast<const u32x4*>(p + 16 * i); r.w[4 * i] = v.x; r.w[4 * i + 1] = v.y; r.w[4 * i + 2] = v.z; r.w[4 * i + 3] = v.w; }
    } else if constexpr (bytes == 8) { const u32x2 v = *reinterpret_cast<const u32x2*>(p); r.w[0] = v.x; r.w[1] = v.y; }
    else if constexpr (bytes == 4) r.w[0] = *reinterpret_cast<const uint32_t*>(p);
    else if constexpr (bytes == 2) r.w[0] = *reinterpret_cast<const uint16_t*>(p);
    else r.w[0] = *reinterpret_cast<const uint8_t*>(p);
}
template <int N, bool FP8>
__device__ __forceinline__ void kr_kv_raw_f32(const KrKvRaw<N, FP8>& r, float (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        if constexpr (FP8) {
            const int w = (int)r.w[i >> 2];
            x[i] = (i & 3) == 0 ? __builtin_amdgcn_cvt_f32_fp8(w, 0) : ((i & 3) == 1 ? __builtin_amdgcn_cvt_f32_fp8(w, 1) : ((i & 3) == 2 ? __builtin_amdgcn_cvt_f32_fp8(w, 2) : __builtin_amdgcn_cvt_f32_fp8(w, 3)));
        } else {
            const uint16_t hb = (uint16_t)((i & 1) ? (r.w[i >> 1] >> 16) : (r.w[i >> 1] & 0xFFFFu));
            x[i] = (float)__builtin_bit_cast(_Float16, hb);
        }
    }
}
template <bool FP8> __device__ __forceinline__ float kr_kv_round(float v) {      // the value a cache element holds after kr_kv_store (kr_device.h)
    if constexpr (FP8) return kr_e4m3_to_f32(kr_f32_to_e4m3(v));
    else return (float)(_Float16)v;
}

template <int HD, bool FP8>
__global__ void __launch_bounds__(256) kr_fgqa_kernel(const KrStep* p_step, const float* p_q, const float* p_k, const float* p_v, const void* p_kc, const void* p_vc,
                                                      int p_nh, int p_nkv, const KrGqaArgs a) {
    constexpr int D16 = HD / 16;        // dimensions per lane in the score pass (16 lanes per position)
    constexpr int DPL = HD / 64;        // dimensions per lane in the p.v pass (a wave spans one cache row)
    constexpr int SB = 2;               // score pass: 16-position steps requested per batch
    constexpr int PB = 8;               // p.v pass: positions per wave requested per batch
    __shared__ __attribute__((aligned(16))) float s_q[HD], s_k[HD], s_v[HD], s_sc[KR_FGQA_MAX + 64], s_part[4][HD];
    __shared__ float s_red[2][4];
    const int h = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int grp = p_nh / p_nkv, kvh = h / grp;
    const int pos = p_step->pos, seq = pos + 1;
    const size_t rowe = (size_t)p_nkv * HD, headoff = (size_t)kvh * HD;
    const int d = t < HD ? t : HD - 1;     // every thread loads (clamped)
    const bool own = t < HD;
    // ---- requests: this step's q / gate / k / v of the head, norm weights, the rope row; then the first batch of K rows
    const float qraw = a.gated ? p_q[(size_t)h * HD * 2 + d] : p_q[(size_t)h * HD + d];
    const float graw = a.gated ? p_q[(size_t)h * HD * 2 + HD + d] : 0.0f;
    const float kraw = p_k[headoff + d], vraw = p_v[headoff + d];
    const float qw = a.q_norm ? a.q_norm[(a.q_norm_per_head ? h * HD : 0) + d] : 1.0f;
    const float kw = a.k_norm ? a.k_norm[(a.k_norm_per_head ? kvh * HD : 0) + d] : 1.0f;
    const int d2 = a.rope_half;
    const bool rot = t < 2 * d2 && own;
    const int ri = rot ? (t < d2 ? t : t - d2) : 0;
    const float rc = a.rope_cos[(size_t)pos * d2 + ri], rs = a.rope_sin[(size_t)pos * d2 + ri];
    const int l16 = lane & 15, pw = lane >> 4;
    KrKvRaw<D16, FP8> kqA[SB], kqB[SB];      // two request batches, named (a run-time buffer index would send the arrays to scratch)
    auto issue_k = [&](KrKvRaw<D16, FP8> (&kq)[SB], int s0) {
#pragma unroll
        for (int u = 0; u < SB; u++) { const int sp = s0 + 16 * u + wave * 4 + pw; kr_kv_raw_load<D16, FP8>(kq[u], p_kc, (size_t)(sp < pos ? sp : 0) * rowe + headoff + l16 * D16); }
    };
    issue_k(kqA, 0);
    // ---- per-head QK RMSNorm (trees), RoPE
    float sq = own ? qraw * qraw : 0.0f, sk = own ? kraw * kraw : 0.0f;
    sq = kr_f_wave_sum(sq); sk = kr_f_wave_sum(sk);
    if (lane == 0) { s_red[0][wave] = sq; s_red[1][wave] = sk; }
    __syncthreads();
    float qx = qraw, kx = kraw;
    if (a.q_norm) { const float ss = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]); qx = qraw * ((1.0f / sqrtf(ss / (float)HD + a.eps)) * qw); }
    if (a.k_norm) { const float ss = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]); kx = kraw * ((1.0f / sqrtf(ss / (float)HD + a.eps)) * kw); }
    if (own) { s_q[t] = qx; s_k[t] = kx; }
    __syncthreads();
    float qv = qx, kv = kx;
    if (rot) {
        if (t < d2) { qv = qx * rc - s_q[d2 + t] * rs; kv = kx * rc - s_k[d2 + t] * rs; }      // x1 cos - x2 sin
        else { qv = qx * rc + s_q[t - d2] * rs; kv = kx * rc + s_k[t - d2] * rs; }              // x2 cos + x1 sin
    }
    __syncthreads();
    if (own) {
        s_q[t] = qv; s_k[t] = kr_kv_round<FP8>(kv); s_v[t] = kr_kv_round<FP8>(vraw);
        if (h == kvh * grp) {      // the group's first head appends the row
            const size_t o = (size_t)pos * rowe + headoff + t;
            kr_kv_store(const_cast<void*>(p_kc), o, kv, FP8 ? 1 : 0);
            kr_kv_store(const_cast<void*>(p_vc), o, vraw, FP8 ? 1 : 0);
        }
    }
    __syncthreads();
    // ---- scores: 16 lanes per position
    float qreg[D16], knew[D16];
#pragma unroll
    for (int i = 0; i < D16; i++) { qreg[i] = s_q[l16 * D16 + i]; knew[i] = s_k[l16 * D16 + i]; }
    auto score_batch = [&](const KrKvRaw<D16, FP8> (&kq)[SB], int s0) {
#pragma unroll
        for (int u = 0; u < SB; u++) {
            const int sp = s0 + 16 * u + wave * 4 + pw;
            float kx_[D16];
            kr_kv_raw_f32<D16, FP8>(kq[u], kx_);
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < D16; i++) acc = __builtin_fmaf(qreg[i], sp == pos ? knew[i] : kx_[i], acc);
            acc = kr_f_red16(acc);
            if (l16 == 0 && sp < seq) s_sc[sp] = acc * a.sm_scale;
        }
    };
    constexpr int SSTEP = 16 * SB;
    for (int s0 = 0; s0 < seq; s0 += 2 * SSTEP) {
        if (s0 + SSTEP < seq) issue_k(kqB, s0 + SSTEP);
        score_batch(kqA, s0);
        if (s0 + 2 * SSTEP < seq) issue_k(kqA, s0 + 2 * SSTEP);
        if (s0 + SSTEP < seq) score_batch(kqB, s0 + SSTEP);
    }
    // ---- the first batch of V rows leaves before the softmax
    KrKvRaw<DPL, FP8> vqA[PB], vqB[PB];
    auto issue_v = [&](KrKvRaw<DPL, FP8> (&vq)[PB], int s0) {
#pragma unroll
        for (int u = 0; u < PB; u++) { const int sp = s0 + 4 * u + wave; kr_kv_raw_load<DPL, FP8>(vq[u], p_vc, (size_t)(sp < pos ? sp : 0) * rowe + headoff + lane * DPL); }
    };
    issue_v(vqA, 0);
    __syncthreads();
    float mx = -__builtin_inff();
    for (int sp = t; sp < seq; sp += 256) mx = fmaxf(mx, s_sc[sp]);
    mx = kr_f_wave_max(mx);
    if (lane == 0) s_red[0][wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3]));
    float se = 0.0f;
    for (int sp = t; sp < seq; sp += 256) { const float e = kr_expf(s_sc[sp] - mx); s_sc[sp] = e; se += e; }
    se = kr_f_wave_sum(se);
    if (lane == 0) s_red[1][wave] = se;
    __syncthreads();
    const float inv = 1.0f / ((s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]));
    for (int sp = t; sp < seq; sp += 256) s_sc[sp] *= inv;      // sc[s] *= inv (decode.rs:4260)
    float vnew[DPL];
#pragma unroll
    for (int i = 0; i < DPL; i++) vnew[i] = s_v[lane * DPL + i];
    __syncthreads();
    // ---- p.v: wave w takes positions w, w + 4, ...; lane l the dimensions l DPL .. l DPL + DPL
    float o[DPL];
#pragma unroll
    for (int i = 0; i < DPL; i++) o[i] = 0.0f;
    auto pv_batch = [&](const KrKvRaw<DPL, FP8> (&vq)[PB], int s0) {
#pragma unroll
        for (int u = 0; u < PB; u++) {
            const int sp = s0 + 4 * u + wave;
            float vx[DPL];
            kr_kv_raw_f32<DPL, FP8>(vq[u], vx);
            const float pr = sp < seq ? s_sc[sp] : 0.0f;
#pragma unroll
            for (int i = 0; i < DPL; i++) o[i] = __builtin_fmaf(pr, sp == pos ? vnew[i] : vx[i], o[i]);
        }
    };
    constexpr int PSTEP = 4 * PB;
    for (int s0 = 0; s0 < seq; s0 += 2 * PSTEP) {
        if (s0 + PSTEP < seq) issue_v(vqB, s0 + PSTEP);
        pv_batch(vqA, s0);
        if (s0 + 2 * PSTEP < seq) issue_v(vqA, s0 + 2 * PSTEP);
        if (s0 + PSTEP < seq) pv_batch(vqB, s0 + PSTEP);
    }
#pragma unroll
    for (int i = 0; i < DPL; i++) s_part[wave][lane * DPL + i] = o[i];
    __syncthreads();
    float ov = 0.0f;
    if (own) {
        ov = (s_part[0][t] + s_part[1][t]) + (s_part[2][t] + s_part[3][t]);
        if (a.gated) ov *= 1.0f / (1.0f + kr_expf(-graw));
        a.attn_out[(size_t)h * HD + t] = ov;
        s_q[t] = ov;
    }
    if (a.img_out) {   // HD % 128 == 0: the head's output is HD / 128 whole quantization groups of the o-projection's input
        __syncthreads();
        const KrActLds Lg = kr_carve_lds(reinterpret_cast<u32x4*>(a.img_out), p_nh * HD, false);
        constexpr int nch = HD / 8;
        if (t < nch) {
            float v8[8];
            kr_load8(s_q, t, v8);
            float mxq = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) mxq = fmaxf(mxq, fabsf(v8[i]));
            float scale, invq;
            kr_group_scale(mxq, scale, invq);
            int q8[8];
            kr_quant8<false>(v8, invq, q8);
            const int gc = h * nch + t;
            kr_store_chunk<false>(Lg, gc, q8);
            if ((gc & 15) == 0) Lg.ascale[gc >> 4] = scale;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// MLA layers over a SHORT cache (kv_max_seq <= KR_FGQA_MAX) in the mode: scores over the latent + rope caches, softmax and the weighted sum of the latent rows of one
// head per workgroup (decode.rs:3150-3225; the exact path: kr_mla_attn_staged_kernel with the reference's 16-lane dot order, position-ordered softmax sum and one
// fma chain per latent element).  Same products, tree sums; the passes spread positions over lanes as kr_fgqa_kernel does: 16 lanes per position in the score
// pass (KLR / 16 latent + RD / 16 rope elements per lane), one wave per cache row in the weighted sum.  The row of the current position was appended by the prep launch.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int KLR, int RD, bool FP8>
__global__ void __launch_bounds__(256) kr_fmla_kernel(const KrStep* p_step, const float* p_qabs, const float* p_qpe, const void* p_ckv, const void* p_kpe, float* p_out, float p_scale) {
    constexpr int DC = KLR / 16, DR = RD / 16, DPL = KLR / 64, SB = 2, PB = 8;
    __shared__ __attribute__((aligned(16))) float s_sc[KR_FGQA_MAX + 64], s_part[4][KLR];
    __shared__ float s_red[2][4];
    const int h = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63, l16 = lane & 15, pw = lane >> 4;
    const int pos = p_step->pos, seq = pos + 1;
    KrKvRaw<DC, FP8> kcA[SB], kcB[SB]; KrKvRaw<DR, FP8> krA[SB], krB[SB];
    auto issue_k = [&](KrKvRaw<DC, FP8> (&kc)[SB], KrKvRaw<DR, FP8> (&kr)[SB], int s0) {
#pragma unroll
        for (int u = 0; u < SB; u++) {
            const int sp = s0 + 16 * u + wave * 4 + pw, sr = sp < seq ? sp : 0;
            kr_kv_raw_load<DC, FP8>(kc[u], p_ckv, (size_t)sr * KLR + l16 * DC);
            kr_kv_raw_load<DR, FP8>(kr[u], p_kpe, (size_t)sr * RD + l16 * DR);
        }
    };
    float qc[DC], qr[DR];
#pragma unroll
    for (int i = 0; i < DC; i += 4) { const float4 v = *reinterpret_cast<const float4*>(p_qabs + (size_t)h * KLR + l16 * DC + i); qc[i] = v.x; qc[i + 1] = v.y; qc[i + 2] = v.z; qc[i + 3] = v.w; }
#pragma unroll
    for (int i = 0; i < DR; i += 4) { const float4 v = *reinterpret_cast<const float4*>(p_qpe + (size_t)h * RD + l16 * DR + i); qr[i] = v.x; qr[i + 1] = v.y; qr[i + 2] = v.z; qr[i + 3] = v.w; }
    issue_k(kcA, krA, 0);
    auto score_batch = [&](const KrKvRaw<DC, FP8> (&kc)[SB], const KrKvRaw<DR, FP8> (&kr)[SB], int s0) {
#pragma unroll
        for (int u = 0; u < SB; u++) {
            const int sp = s0 + 16 * u + wave * 4 + pw;
            float xc[DC], xr[DR];
            kr_kv_raw_f32<DC, FP8>(kc[u], xc); kr_kv_raw_f32<DR, FP8>(kr[u], xr);
            float acc = 0.0f, acr = 0.0f;
#pragma unroll
            for (int i = 0; i < DC; i++) acc = __builtin_fmaf(qc[i], xc[i], acc);
#pragma unroll
            for (int i = 0; i < DR; i++) acr = __builtin_fmaf(qr[i], xr[i], acr);
            acc = kr_f_red16(acc); acr = kr_f_red16(acr);
            if (l16 == 0 && sp < seq) s_sc[sp] = (acc + acr) * p_scale;      // latent dot + rope dot, then the scale (decode.rs:3176-3184)
        }
    };
    constexpr int SSTEP = 16 * SB;
    for (int s0 = 0; s0 < seq; s0 += 2 * SSTEP) {
        if (s0 + SSTEP < seq) issue_k(kcB, krB, s0 + SSTEP);
        score_batch(kcA, krA, s0);
        if (s0 + 2 * SSTEP < seq) issue_k(kcA, krA, s0 + 2 * SSTEP);
        if (s0 + SSTEP < seq) score_batch(kcB, krB, s0 + SSTEP);
    }
    KrKvRaw<DPL, FP8> vqA[PB], vqB[PB];
    auto issue_v = [&](KrKvRaw<DPL, FP8> (&vq)[PB], int s0) {
#pragma unroll
        for (int u = 0; u < PB; u++) { const int sp = s0 + 4 * u + wave; kr_kv_raw_load<DPL, FP8>(vq[u], p_ckv, (size_t)(sp < seq ? sp : 0) * KLR + lane * DPL); }
    };
    issue_v(vqA, 0);
    __syncthreads();
    float mx = -__builtin_inff();
    for (int sp = t; sp < seq; sp += 256) mx = fmaxf(mx, s_sc[sp]);
    mx = kr_f_wave_max(mx);
    if (lane == 0) s_red[0][wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3]));
    float se = 0.0f;
    for (int sp = t; sp < seq; sp += 256) { const float e = kr_expf(s_sc[sp] - mx); s_sc[sp] = e; se += e; }
    se = kr_f_wave_sum(se);
    if (lane == 0) s_red[1][wave] = se;
    __syncthreads();
    const float inv = 1.0f / ((s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]));
    for (int sp = t; sp < seq; sp += 256) s_sc[sp] *= inv;
    __syncthreads();
    float o[DPL];
#pragma unroll
    for (int i = 0; i < DPL; i++) o[i] = 0.0f;
    auto pv_batch = [&](const KrKvRaw<DPL, FP8> (&vq)[PB], int s0) {
#pragma unroll
        for (int u = 0; u < PB; u++) {
            const int sp = s0 + 4 * u + wave;
            float vx[DPL];
            kr_kv_raw_f32<DPL, FP8>(vq[u], vx);
            const float pr = sp < seq ? s_sc[sp] : 0.0f;
#pragma unroll
            for (int i = 0; i < DPL; i++) o[i] = __builtin_fmaf(pr, vx[i], o[i]);
        }
    };
    constexpr int PSTEP = 4 * PB;
    for (int s0 = 0; s0 < seq; s0 += 2 * PSTEP) {
        if (s0 + PSTEP < seq) issue_v(vqB, s0 + PSTEP);
        pv_batch(vqA, s0);
        if (s0 + 2 * PSTEP < seq) issue_v(vqA, s0 + 2 * PSTEP);
        if (s0 + PSTEP < seq) pv_batch(vqB, s0 + PSTEP);
    }
#pragma unroll
    for (int i = 0; i < DPL; i++) s_part[wave][lane * DPL + i] = o[i];
    __syncthreads();
    for (int j = t; j < KLR; j += 256) p_out[(size_t)h * KLR + j] = (s_part[0][j] + s_part[1][j]) + (s_part[2][j] + s_part[3][j]);
}

int kr_launch_fmla(const KrMlaArgs& a, int max_seq, hipStream_t st) {
    if (!a.step || max_seq > KR_FGQA_MAX || a.sc_g || a.rd != 64 || (a.klr != 512 && a.klr != 256)) return 1;
#define KR_FMLA(K_, F_) hipLaunchKernelGGL((kr_fmla_kernel<K_, 64, F_>), dim3(a.nh), dim3(256), 0, st, a.step, (const float*)a.q_abs, (const float*)a.q_pe, (const void*)a.ckv_cache, (const void*)a.kpe_cache, a.attn_lat, a.sm_scale)
    if (a.klr == 512) { if (a.kv_fp8) KR_FMLA(512, true); else KR_FMLA(512, false); }
    else { if (a.kv_fp8) KR_FMLA(256, true); else KR_FMLA(256, false); }
#undef KR_FMLA
    return 0;
}

int kr_launch_fgqa(const KrGqaArgs& a, int max_seq, hipStream_t st) {
    if (max_seq > KR_FGQA_MAX || a.sc_g || a.nh % a.nkv || 2 * a.rope_half > a.hd || (a.img_out && a.hd % 128)) return 1;
#define KR_FGQA(H_, F_) hipLaunchKernelGGL((kr_fgqa_kernel<H_, F_>), dim3(a.nh), dim3(256), 0, st, a.step, a.q_in, a.k_in, a.v_in, (const void*)a.k_cache, (const void*)a.v_cache, a.nh, a.nkv, a)
    if (a.hd == 256) { if (a.kv_fp8) KR_FGQA(256, true); else KR_FGQA(256, false); }
    else if (a.hd == 128) { if (a.kv_fp8) KR_FGQA(128, true); else KR_FGQA(128, false); }
    else if (a.hd == 64) { if (a.kv_fp8) KR_FGQA(64, true); else KR_FGQA(64, false); }
    else return 1;
#undef KR_FGQA
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------------------
template <int BITS, int KS, int MODE>
static int kr_fdm_launch_nu2(const KrFdmArgs& a, int nu, dim3 grid, size_t lds, hipStream_t st) {
    const void* p0 = MODE == 0 ? a.img : ((MODE == 1 && a.emb) ? nullptr : (const void*)a.hid_in);
    const float* p1 = (MODE == 1 && !a.first) ? a.res_in : nullptr;
    const int Kp = a.mm.m[0].ng * 128;
#define KR_FDM(N_) hipLaunchKernelGGL((kr_fdm_kernel<BITS, KS, N_, MODE>), grid, dim3(256), lds, st, p0, p1, a.norm_w, Kp, a)
    switch (nu) {
        case 2: KR_FDM(2); break;
        case 4: KR_FDM(4); break;
        case 8: KR_FDM(8); break;
        case 16: KR_FDM(16); break;
        default: KR_FDM(0); break;
    }
#undef KR_FDM
    return 0;
}
template <int BITS, int KS>
static int kr_fdm_launch_nu(const KrFdmArgs& a, int nu, dim3 grid, size_t lds, hipStream_t st) {
    if (a.mode == 0) return kr_fdm_launch_nu2<BITS, KS, 0>(a, nu, grid, lds, st);
    if (a.mode == 2) return kr_fdm_launch_nu2<BITS, KS, 2>(a, nu, grid, lds, st);
    return kr_fdm_launch_nu2<BITS, KS, 1>(a, nu, grid, lds, st);
}

int kr_launch_fdm(const KrFdmArgs& a, hipStream_t st) {
    const KrMatDev& m0 = a.mm.m[0];
    const int bits = m0.bits, K = m0.ng * 128;
    for (int i = 1; i < a.mm.n; i++) if (a.mm.m[i].bits != bits || a.mm.m[i].ng != m0.ng) return 1;
    if ((a.mode == 1 || a.mode == 2) && (K > 4096 || K != m0.K)) return 1;
    if (a.conv_state && (a.conv_mi < 0 || a.conv_mi >= a.mm.n || a.mm.m[a.conv_mi].N != a.nk * (2 * a.dk + 2 * a.hr * a.dv) || a.gate_mi < 0 || a.gate_mi >= a.mm.n ||
                         a.mm.m[a.gate_mi].N != a.nk * 2 * a.hr || !a.ge_out || !a.beta_out)) return 1;
    const int total = a.mm.tile_end[a.mm.n - 1];
    const int units = bits == 4 ? m0.ngp : m0.ng;
    // K split: near one or two workgroups per CU, and a wave's chain no longer than 8 units
    int ks = total >= 1024 ? 1 : (total >= 512 ? 2 : 4);
    while (ks < 4 && units / ks > 8) ks *= 2;
    const int tw = 4 / ks;
    const bool even = (bits == 8 || (m0.ng % 2) == 0) && units % ks == 0;
    const int nu_exact = even ? units / ks : 0;
    const int nu = (nu_exact == 2 || nu_exact == 4 || nu_exact == 8 || nu_exact == 16) ? nu_exact : 0;
    dim3 grid((total + tw - 1) / tw);
    const size_t lds = kr_lds_bytes(K, bits == 8);
    KrFdmArgs b = a;
    for (int i = a.mm.n - 1; i < KR_MAX_MULTI; i++) b.mm.tile_end[i] = total;   // the kernel reads all four at constant offsets
    if (bits == 4) {
        if (ks == 1) return kr_fdm_launch_nu<4, 1>(b, nu, grid, lds, st);
        if (ks == 2) return kr_fdm_launch_nu<4, 2>(b, nu, grid, lds, st);
        return kr_fdm_launch_nu<4, 4>(b, nu, grid, lds, st);
    }
    if (ks == 1) return kr_fdm_launch_nu<8, 1>(b, nu, grid, lds, st);
    if (ks == 2) return kr_fdm_launch_nu<8, 2>(b, nu, grid, lds, st);
    return kr_fdm_launch_nu<8, 4>(b, nu, grid, lds, st);
}

int kr_launch_fla(const KrFlaArgs& a, hipStream_t st) {
    if (a.nv != a.nk * a.hr) return 1;
#define KR_FLA(DK_, DV_) hipLaunchKernelGGL((kr_fla_kernel<DK_, DV_>), dim3(a.nv), dim3(512), 0, st, a.qk, a.z, a.v, a.ge, a.beta, a.state, a.hr, a)
    if (a.dk == 128 && a.dv == 128) KR_FLA(128, 128);
    else if (a.dk == 64 && a.dv == 128) KR_FLA(64, 128);
    else return 1;
#undef KR_FLA
    return 0;
}

int kr_launch_frt(const KrFrtArgs& a, hipStream_t st) {
    if (a.H % 128 || a.H > 4096 || a.H < 256) return 1;
    const size_t lds = (size_t)16 * (a.H / 16 + 4) * 4;
    dim3 grid((a.E + 3) / 4);
    const int ncg = a.gate_bf16 ? a.H / 128 : a.H / 64, cpw = (ncg + 3) / 4;
#define KR_FRT(B_, C_) hipLaunchKernelGGL((kr_frt_kernel<B_, C_>), grid, dim3(256), lds, st, a.hid_in, a.res_in, a.norm_w, a.gate_cm, a.H, a)
    if (a.gate_bf16) { if (cpw <= 1) KR_FRT(true, 1); else if (cpw <= 2) KR_FRT(true, 2); else if (cpw <= 4) KR_FRT(true, 4); else KR_FRT(true, 8); }
    else { if (cpw <= 2) KR_FRT(false, 2); else if (cpw <= 4) KR_FRT(false, 4); else if (cpw <= 8) KR_FRT(false, 8); else KR_FRT(false, 16); }
#undef KR_FRT
    return 0;
}

static bool kr_fmoe_ok(const KrFmoeArgs& fa) {
    const KrMoeArgs& a = fa.m;
    const bool has_shared = a.n_slots > a.topk;
    if (a.B != 1 || a.n_slots > 16 || a.topk > 15 || a.E > 512 || a.act_mode != KR_ACT_SILU_FUSED) return false;
    if (fa.gguf) {      // routed experts on native GGUF blocks: the integer block kernels only, a transposed shared expert (if any) of one bit width
        auto ok_mat = [&](const GgMat& g, int K, int N) { return (g.type == GG_Q4_K ? K % 256 == 0 : ((g.type == GG_Q8_0 || g.type == GG_Q4_0) && K % 32 == 0)) && g.K == K && g.N == N && N % 8 == 0; };
        if (!fa.act_f32 || !ok_mat(fa.ggate, a.H, a.I) || !ok_mat(fa.gup, a.H, a.I) || !ok_mat(fa.gdown, a.I, a.H) || a.H % 128) return false;
        if (has_shared) {
            if (!a.act_img || a.sw13.bits != a.sw2.bits || a.I_shared % 128 || a.sw13.ng * 128 != a.H || a.sw2.ng * 128 != a.I_shared) return false;
            if (a.sgate.q && (a.sgate.bits != a.sw13.bits || a.sgate.ng != a.sw13.ng || !a.gate_out)) return false;
        }
        return true;
    }
    if (!a.act_img || !a.act_img_bf16 || a.H % 128 || a.I % 128) return false;
    if (a.w13.ng * 128 != a.H || a.w2.ng * 128 != a.I) return false;
    if (has_shared) {
        if (a.sw13.bits != a.w13.bits || a.sw2.bits != a.w2.bits || a.I_shared % 128 || a.sw13.ng != a.w13.ng || a.sw2.ng * 128 != a.I_shared) return false;
        if (a.sgate.q && (a.sgate.bits != a.w13.bits || a.sgate.ng != a.w13.ng || !a.gate_out)) return false;
    }
    return true;
}

// dynamic LDS of one slot of the down + combine launch: the larger of the two activation images a slot may build (per-128 transposed form / per-32 GGUF form)
static size_t kr_fw2_slot_lds(const KrFmoeArgs& fa) {
    const KrMoeArgs& a = fa.m;
    const bool has_shared = a.n_slots > a.topk;
    const int imax = has_shared && a.I_shared > a.I ? a.I_shared : a.I;
    size_t n = fa.gguf ? (has_shared ? kr_lds_bytes(a.I_shared, a.sw2.bits == 8) : 0) : kr_lds_bytes(imax, a.w2.bits == 8);
    if (fa.gguf) { const size_t g = (gg_lds_bytes(a.I, false) + 15) & ~(size_t)15; if (g > n) n = g; }
    return n;
}
int kr_fmoe_check(const KrFmoeArgs& fa) {
    if (!kr_fmoe_ok(fa)) return 1;
    const KrMoeArgs& a = fa.m;
    // kr_fw2_kernel takes slot_lds * n_slots of dynamic LDS and no opt-in is made for it: past the 64 KiB every kernel may use the layer falls back to the exact MoE kernels
    return kr_fw2_slot_lds(fa) * a.n_slots > KR_FW2_LDS_MAX ? 1 : 0;
}

int kr_launch_fw13(const KrFmoeArgs& fa, hipStream_t st) {
    if (!kr_fmoe_ok(fa)) return 1;
    const KrMoeArgs& a = fa.m;
    const bool has_shared = a.n_slots > a.topk;
    int ntp = a.I / 8;
    if (has_shared && a.I_shared / 8 + (a.sgate.q ? 1 : 0) > ntp) ntp = a.I_shared / 8 + (a.sgate.q ? 1 : 0);
    dim3 grid((ntp + 1) / 2, a.n_slots);
    const KrMatDev& wm = fa.gguf ? a.sw13 : a.w13;      // GGUF routed experts: the template follows the (transposed) shared expert; INT4 form when there is none
    const int wbits = fa.gguf && !has_shared ? 4 : wm.bits;
    size_t img = kr_lds_bytes(a.H, wbits == 8);
    if (fa.gguf && gg_lds_bytes(a.H, false) > img) img = gg_lds_bytes(a.H, false);
    const size_t lds = img + (size_t)(2 * a.E + 33 + 33 + 32 + 32 + 4 + 136) * 4;
    const int units = fa.gguf && !has_shared ? 0 : (wbits == 4 ? wm.ngp : wm.ng);
    const bool even = !fa.gguf && (a.w13.bits == 8 || (a.w13.ng % 2) == 0) && units % 2 == 0;
    const int nu = even ? units / 2 : 0;
#define KR_FW13(B_, N_) hipLaunchKernelGGL((kr_fw13_kernel<B_, N_>), grid, dim3(256), lds, st, fa.logits, fa.esc, a.act_img_bf16, a.E, a.topk, fa.scoring, fa.norm_topk, a.H, a.I, fa.gguf, fa)
    if (wbits == 4) { if (nu == 4) KR_FW13(4, 4); else if (nu == 8) KR_FW13(4, 8); else KR_FW13(4, 0); }
    else { if (nu == 8) KR_FW13(8, 8); else KR_FW13(8, 0); }
#undef KR_FW13
    return 0;
}

int kr_launch_fw2(const KrFmoeArgs& fa, hipStream_t st) {
    if (!kr_fmoe_ok(fa)) return 1;
    const KrMoeArgs& a = fa.m;
    const bool has_shared = a.n_slots > a.topk;
    const size_t slot_lds = kr_fw2_slot_lds(fa);
    if (slot_lds * a.n_slots > KR_FW2_LDS_MAX) return 1;
    dim3 grid((a.H + 7) / 8);
    // exact unit count (no guards) when every slot has the same, even, group count
    const int wbits = fa.gguf ? (has_shared ? a.sw2.bits : 4) : a.w2.bits;
    const int units = fa.gguf ? 0 : (a.w2.bits == 4 ? a.w2.ngp : a.w2.ng);
    const bool uniform = !fa.gguf && (!has_shared || a.I_shared == a.I) && (a.w2.bits == 8 || a.w2.ng % 2 == 0);
    const int nu = uniform && (units == 2 || units == 4 || units == 6 || units == 8) ? units : 0;      // (6: I = 1536, Qwen3-235B)
    // waves per slot: a wide routed expert (guarded form only: the exact-unit forms walk the whole expert) on two waves, the shared expert on as many as keep its
    // slice no longer than a routed one's -- within the 16 waves of a workgroup
    const int sunits = has_shared ? (a.sw2.bits == 4 ? a.sw2.ngp : a.sw2.ng) : 0;
    int pr = 1, ps = has_shared ? 1 : 0;
    if (nu == 0 && !fa.gguf && units >= 4 && a.I >= 1024 && a.topk <= 7) pr = 2;
    if (has_shared && nu == 0 && !fa.shared_skip) {
        ps = (a.I_shared * pr + a.I - 1) / a.I;
        if (ps > 4) ps = 4;
        while (ps > 1 && (a.topk * pr + ps > 16 || ps > sunits)) ps--;
    }
    const bool multi = pr > 1 || ps > 1;
#define KR_FW2(B_, N_, M_) hipLaunchKernelGGL((kr_fw2_kernel<B_, N_, M_>), grid, dim3(64 * (a.topk * pr + ps)), slot_lds * a.n_slots, st, a.gu, a.ids, a.wts, a.gu_ld, a.topk, (int)slot_lds, pr, ps, fa)
    if (multi) { if (wbits == 4) KR_FW2(4, 0, true); else KR_FW2(8, 0, true); }      // (several waves per slot only with the guarded form)
    else if (wbits == 4) { if (nu == 2) KR_FW2(4, 2, false); else if (nu == 4) KR_FW2(4, 4, false); else if (nu == 6) KR_FW2(4, 6, false); else if (nu == 8) KR_FW2(4, 8, false); else KR_FW2(4, 0, false); }
    else { if (nu == 4) KR_FW2(8, 4, false); else if (nu == 8) KR_FW2(8, 8, false); else KR_FW2(8, 0, false); }
#undef KR_FW2
    return 0;
}
