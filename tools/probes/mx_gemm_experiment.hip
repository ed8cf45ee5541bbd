// kr_prefill_mx.hip -- the MX form of the tolerance GEMM of the prompt pass: block-scaled FP6 operands on v_mfma_scale_f32_32x32x64_f8f6f4.
//
// Why.  The f16 form (kr_prefill_h.hip) is bound by the vector port: 12 instructions rebuild 8 INT4 weights as f16 for every MFMA that consumes them
// (docs/design/06-prompt-pass.md "Round 6": 1130 vector cycles per 1024 matrix cycles and SIMD).  gfx950's block-scaled matrix instruction takes FP6 (E2M3)
// operands at FOUR times the f16 rate (tools/probes/mx_mfma_probe.hip: 16 ns per 32x32x64 instruction and SIMD, the time of one 32x32x16 f16 instruction),
// and an INT4 weight is an E2M3 value: (n - 8) / 2 is one of -4, -3.5, .., 3.5, every one of them representable.  So the weights are re-coded ONCE (6 bits
// instead of 4, a second image in HBM like the reference's Marlin copy, gpu_prefill.py:64-239) and reach the matrix cores without touching the vector unit.
//
// Numerics (tolerance mode; tests/test_gemm_mx_gpu.py states the bounds).
//   B: exact.  code(n) = E2M3((n - 8) / 2); the bf16 group scale s enters after the matrix instruction: the products of a 128-wide quantization group are summed
//      into a group accumulator (f32, inside the MFMA), then acc += (32 s) * group_acc -- one fma per output and group, as the exact kernel does with its i32 sums.
//   A: a row is scaled by a power of two (largest |value| in [1, 2), like the f16 form: same row multipliers) and every 32-wide block of it is written as the sum of
//      PFX_NT = 3 E2M3 terms with their own power-of-two block scales (E8M0 bytes, the MX block scale of the instruction): x = q0 2^E0 + q1 2^(E0-4) + q2 2^(E0-9),
//      E0 from the block's largest value (|x| / 2^E0 <= 7.5), each q the round-to-nearest E2M3 code of the residual.  |error| <= 2^-12 of the block's 2^E0 .. 2^(E0+3)
//      range: 12 - 14 significant bits at the block's largest values -- the f16 form has 11 per value, the exact path's INT16 digits 15 per 128-group.
//      Each term is one matrix instruction: 3 at 4x rate = 0.75 of the f16 form's matrix time, and no conversion work.
//   Products are exact in the MFMA's f32 accumulation (4 x 4 significant bits).
//
// Operand layout (established by the probe): lane l of the wave supplies row / column l & 31, k = 32 (l >> 5) + p for the 6-bit field p of its 192-bit operand;
// the E8M0 scale is per lane (its row / column and k half), byte `opsel` of the scale register.  C / D as every 32 x 32 MFMA.
//   A image   : row-major, per 128-wide group NT * 128 bytes = 4 NT fragment slots of 32 B: slot ((b >> 1) NT + t) 2 + (b & 1) holds term t of block b = 0..3 (24-byte
//               code string); bytes 24..27 of a block's t = 0 slot hold its scale word (byte t = E_t + 127)
//   B image   : [expert][N / 32 column blocks][K / 64 k-steps] x 1536 B: lane l's 24-byte string as 16 B at l * 16 and 8 B at 1024 + l * 8 -- a wave streams its
//               column block's k-steps as consecutive 1.5-KiB records straight into registers (the fragment of a column block is used by ONE wave of the workgroup:
//               no LDS for B at all)
// Tile: 64 rows x 256 columns per workgroup of 4 waves (each 64 x 64), two workgroups per CU.  The A rows of one group are an LDS stage of NT segments
// [64 rows][128 B] (three stages), filled by LDS-DMA: one global_load_lds_dwordx4 copies a 128-byte segment of 8 rows (whole cache lines; lane i's 16 bytes land
// at M0 + 16 i), the 16-byte chunk index XOR-ed with (row >> 1) & 7 on the SOURCE side so that the fragment reads of 32 consecutive rows spread over the banks
// (the A path of kr_prefill_ring.hip): no staging registers, one barrier per group.  History of the staging, each measured on the 4096 x 2048 -> 12288 problem:
// through registers 410 TFLOP/s-equivalent (the accumulators spilled, and a spill reload waits for EVERY outstanding load of the wave: vmcnt(0)); LDS-DMA in
// 12-byte pieces (dwordx3 -- which lands lane i at M0 + 16 i, not 12 i) took ~700 cycles to issue per instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <type_traits>

#include "../../krasis_amd/csrc/kr_lds_optin.h"
#include "../../krasis_amd/csrc/kr_device.h"
#include "../../krasis_amd/csrc/kr_libm.h"
#include "../../krasis_amd/csrc/kr_kernels.h"
#include "../../krasis_amd/csrc/kr_prefill.h"
#include "../../krasis_amd/csrc/kr_pfh_dev.h"

// launch arguments: the tolerance GEMM's, plus the FP6 weight image (an experiment: the product structs are not touched)
struct PfxArgs { KrPfGemmHArgs g; const void* mxq; size_t mxq_stride; };

#ifndef PFX_NT
#define PFX_NT 3
#endif
#ifndef PFX_ABL
#define PFX_ABL 0        // tools/probes/gemm_mx_probe.hip: timing-only ablations (results wrong): 1 no A DMA, 2 no B loads in the loop, 4 no MFMA, 8 no fold, 16 no fragment reads
#endif
#define PFX_GB (PFX_NT * 128)                   // bytes of a row's group record: 4 * NT fragment slots of 32 B = NT segments of 128 B
#define PFX_STAGE (PFX_NT * 64 * 128)          // bytes of an LDS stage: [segment][row] x 128 B
#define PFX_NDMA (PFX_NT * 8 / 4)              // LDS-DMA instructions per wave and stage (NT segments x 8 row groups over 4 waves)
#define PFX_BM 64
#ifdef KR_TIMING   // tools/probes/gemm_mx_probe.hip: shader-clock stamps of wave 0 of one mid-grid workgroup in group 6; no-op in the product build
__device__ unsigned long long kr_xstamps[16];
#define PFX_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && g == 6) kr_xstamps[i] = clock64(); } while (0)
#else
#define PFX_STAMP(i) do { } while (0)
#endif
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
// the instruction reads 6 registers of an FP6 operand; the builtin's type has 8: the upper two stay undefined (clang narrows the operand again)
__device__ __forceinline__ v8i pfx_wide(v6i x) { return __builtin_shufflevector(x, x, 0, 1, 2, 3, 4, 5, -1, -1); }
typedef float pfx_f2 __attribute__((ext_vector_type(2)));

size_t kr_pfx_row_bytes(int K) { return (size_t)(K / 128) * PFX_GB; }
size_t kr_pfx_wimage_bytes(int K, int N) { return (size_t)(N / 32) * (K / 64) * 1536; }
bool kr_pfx_shape_ok(int K, int N, int bits) { return bits == 4 && K % 256 == 0 && N % 256 == 0; }

// 16 six-bit codes -> 96 bits
__device__ __forceinline__ void pfx_pack16(const uint32_t* c, uint32_t* d) {
    d[0] = c[0] | (c[1] << 6) | (c[2] << 12) | (c[3] << 18) | (c[4] << 24) | (c[5] << 30);
    d[1] = (c[5] >> 2) | (c[6] << 4) | (c[7] << 10) | (c[8] << 16) | (c[9] << 22) | (c[10] << 28);
    d[2] = (c[10] >> 4) | (c[11] << 2) | (c[12] << 8) | (c[13] << 14) | (c[14] << 20) | (c[15] << 26);
}

// ------------------------------------------------------------------------------------------
// B image: INT4 lane tiles (kr_kernels.h) -> E2M3 fragment records.  thread = one lane of one (expert, column block, k-step) record
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kr_pfx_wimage_kernel(KrMatDev m, int count, char* __restrict__ out, size_t out_stride) {
    const int nks = m.K / 64, ncb = m.N / 32;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t frag = idx >> 6; const int lane = (int)(idx & 63);
    if (frag >= (size_t)count * ncb * nks) return;
    const int e = (int)(frag / ((size_t)ncb * nks)); const int rem = (int)(frag - (size_t)e * ncb * nks), cb = rem / nks, ks = rem - cb * nks;
    const int col = cb * 32 + (lane & 31), k0 = ks * 64 + 32 * (lane >> 5);
    const int g = k0 >> 7, i0 = (k0 & 127) >> 3, st = g >> 1, tile = col >> 3, c = col & 7;
    const char* q = reinterpret_cast<const char*>(m.q) + (size_t)e * m.q_stride;
    uint32_t code[32];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int i = i0 + w, l = i >> 1, wi = (g & 1) * 2 + (i & 1);
        const uint32_t pk = *reinterpret_cast<const uint32_t*>(q + (((size_t)tile * m.ngp + st) * 64 + c * 8 + l) * 16 + wi * 4);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int v = (int)((pk >> (4 * j)) & 15u) - 8, mag = v < 0 ? -v : v;          // value (n - 8) / 2: magnitude mag / 2
            code[8 * w + j] = (uint32_t)(mag <= 4 ? 4 * mag : 8 + 2 * mag) | (v < 0 ? 0x20u : 0u);
        }
    }
    uint32_t d[6];
    pfx_pack16(code, d); pfx_pack16(code + 16, d + 3);
    char* rec = out + (size_t)e * out_stride + ((size_t)cb * nks + ks) * 1536;
    *reinterpret_cast<u32x4*>(rec + lane * 16) = u32x4{d[0], d[1], d[2], d[3]};
    *reinterpret_cast<u32x2*>(rec + 1024 + lane * 8) = u32x2{d[4], d[5]};
}
void kr_launch_pfx_wimage(const KrMatDev& m, int count, void* out, size_t out_stride, hipStream_t st) {
    const size_t n = (size_t)count * (m.N / 32) * (m.K / 64) * 64;
    hipLaunchKernelGGL(kr_pfx_wimage_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m, count, (char*)out, out_stride);
}

// ------------------------------------------------------------------------------------------
// A image: rows -> PFX_NT E2M3 terms per 32-wide block.  thread = one block of one row
// ------------------------------------------------------------------------------------------
// magnitude a >= 0 (already divided by the block scale, <= 7.5 + rounding) -> round-to-nearest E2M3: v8 = the value in eighths, returns the 5-bit magnitude code
__device__ __forceinline__ uint32_t pfx_e2m3(float a, float& deq) {
    const float t = a * 8.0f;
    float v;
    if (t < 16.0f) v = rintf(t);                         // 0 .. 2: steps of 1/8 (subnormals and the first binade share one spacing)
    else if (t < 32.0f) v = 2.0f * rintf(t * 0.5f);      // 2 .. 4: steps of 1/4
    else v = fminf(4.0f * rintf(t * 0.25f), 60.0f);      // 4 .. 7.5: steps of 1/2
    deq = v * 0.125f;
    const int iv = (int)v;
    return (uint32_t)(iv < 16 ? iv : (iv < 32 ? 8 + (iv >> 1) : 16 + (iv >> 2)));
}
__device__ __forceinline__ float pfx_pow2(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }      // -126 <= e <= 127

// SRC 0: f32 rows, 1: bf16 rows (ld elements apart).  grid (ceil(rows / RPW)), 256 threads: TPR = K / 32 threads per row, RPW = 256 / TPR rows per workgroup
// (K <= 8192); pass 1 the row maximum (LDS), pass 2 the codes.  The row multiplier is the f16 form's: mul = 2^e / 16 with 2^-e the scaling that puts the row's
// largest magnitude in [1, 2) -- the GEMM applies the group scales as 32 s, so acc * mul is the product (see the kernel).
template <int SRC>
__global__ void __launch_bounds__(256) kr_pfx_rows_kernel(const void* __restrict__ x, int rows, int ld, int K, char* __restrict__ out, float* __restrict__ mul) {
    __shared__ uint32_t smax[32];
    const int tpr = K / 32, rpw = 256 / tpr, lr = threadIdx.x / tpr, b = threadIdx.x - lr * tpr;
    const int row = blockIdx.x * rpw + lr;
    const bool live = lr < rpw && row < rows;
    if (threadIdx.x < 32) smax[threadIdx.x] = 0;
    __syncthreads();
    float v[32];
    float bm = 0.0f;
    if (live) {
        if (SRC == 0) {
            const float* p = reinterpret_cast<const float*>(x) + (size_t)row * ld + (size_t)b * 32;
#pragma unroll
            for (int i = 0; i < 8; i++) { const float4 f = *reinterpret_cast<const float4*>(p + 4 * i); v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
        } else {
            const uint16_t* p = reinterpret_cast<const uint16_t*>(x) + (size_t)row * ld + (size_t)b * 32;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u32x4 r = *reinterpret_cast<const u32x4*>(p + 8 * i);
                const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int j = 0; j < 4; j++) { v[8 * i + 2 * j] = __uint_as_float(w[j] << 16); v[8 * i + 2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
            }
        }
#pragma unroll
        for (int i = 0; i < 32; i++) bm = fmaxf(bm, fabsf(v[i]));
        atomicMax(&smax[lr], __float_as_uint(bm));        // non-negative floats order like their bit patterns
    }
    __syncthreads();
    if (!live) return;
    float scl, inv; pfh_row_scale(__uint_as_float(smax[lr]), scl, inv);
    bm *= scl;
    // E0: |y| / 2^E0 <= 7.5 for every y of the block, as large a mantissa as that allows
    int E0 = -100;
    if (bm > 0.0f) {
        E0 = (int)(__float_as_uint(bm) >> 23) - 127 - 2;            // bm / 2^E0 in [4, 8)
        if (E0 < -100) E0 = -100;
        if (bm * pfx_pow2(-E0) > 7.5f) E0 += 1;
    }
    const int E[3] = {E0, E0 - 4, E0 - 9};
    uint32_t code[PFX_NT][32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        float r = v[i] * scl;
#pragma unroll
        for (int t = 0; t < PFX_NT; t++) {
            float dq;
            const uint32_t c = pfx_e2m3(fabsf(r) * pfx_pow2(-E[t]), dq);
            code[t][i] = c | (r < 0.0f ? 0x20u : 0u);
            r -= copysignf(dq * pfx_pow2(E[t]), r);              // exact: both sides are multiples of 2^(E_t - 3) below 2^(E_t + 3)
        }
    }
    char* rec = out + (size_t)row * ((size_t)(K / 128) * PFX_GB) + (size_t)(b >> 2) * PFX_GB;
    uint32_t sw = 0;
#pragma unroll
    for (int t = 0; t < PFX_NT; t++) sw |= (uint32_t)(E[t] + 127) << (8 * t);
    const int bb = b & 3;
#pragma unroll
    for (int t = 0; t < PFX_NT; t++) {
        uint32_t d[6];
        pfx_pack16(code[t], d); pfx_pack16(code[t] + 16, d + 3);
        u32x4* o = reinterpret_cast<u32x4*>(rec + (((bb >> 1) * PFX_NT + t) * 2 + (bb & 1)) * 32);
        o[0] = u32x4{d[0], d[1], d[2], d[3]}; o[1] = u32x4{d[4], d[5], t == 0 ? sw : 0u, 0u};
    }
    if (b == 0) mul[row] = inv * 0.0625f;
}
static int pfx_rows_launch(int src, const void* x, int rows, int ld, int K, void* out, float* mul, hipStream_t st) {
    if (rows <= 0) return 0;
    if (K % 128 || K > 8192 || 256 % (K / 32)) return 1;
    const int rpw = 256 / (K / 32);
    if (src == 0) hipLaunchKernelGGL(kr_pfx_rows_kernel<0>, dim3((rows + rpw - 1) / rpw), dim3(256), 0, st, x, rows, ld, K, (char*)out, mul);
    else hipLaunchKernelGGL(kr_pfx_rows_kernel<1>, dim3((rows + rpw - 1) / rpw), dim3(256), 0, st, x, rows, ld, K, (char*)out, mul);
    return 0;
}
int kr_launch_pfx_rows_f32(const float* x, int rows, int ld, int K, void* out, float* mul, hipStream_t st) { return pfx_rows_launch(0, x, rows, ld, K, out, mul, st); }
int kr_launch_pfx_rows_bf16(const uint16_t* x, int rows, int ld, int K, void* out, float* mul, hipStream_t st) { return pfx_rows_launch(1, x, rows, ld, K, out, mul, st); }

// one LDS-DMA wave-instruction: 64 lanes x 16 B, lane i's bytes land at lds_dst + 16 i; source = sbase (wave-uniform) + voff (per lane).  M0 carries the LDS
// address and is the compiler's register: saved and restored inside the statement.  The compiler does not count this load: the waits on it are by hand.
__device__ __forceinline__ void pfx_dma16(uint32_t lds_dst, uint32_t voff, const void* sbase) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    const uint64_t v = (uint64_t)(uintptr_t)sbase;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    const void* sb = (const void*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sb) : "memory");
}
// wait until at most N of this wave's vector-memory instructions are outstanding (they retire in order) and its LDS reads have returned, then the workgroup barrier
template <int N> __device__ __forceinline__ void pfx_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------
// the GEMM
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) kr_pfx_gemm_kernel(const PfxArgs xa) {
    const KrPfGemmHArgs& a = xa.g;
    constexpr int NC = 2, NS = 2, NT = PFX_NT, BN = 256, GB = PFX_GB, STG = PFX_STAGE, NDMA = PFX_NDMA;
    constexpr int NVB = NC * 2 * 2 + NC;          // vector-memory instructions of load_B x 2 + load_S: what a wave issues per group besides its DMAs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As0 = smem;                                              // 3 stages of [NT segments][64 rows][128 B]
    float* rmul = reinterpret_cast<float*>(As0 + 3 * STG);         // [64]
    int* row_src = reinterpret_cast<int*>(rmul + PFX_BM);          // [64] byte offset of the row in the A image
    int* row_dst = row_src + PFX_BM;                               // [64]

    const bool actf = a.act_fused != 0;       // N = 2 I: a tile = 128 gate + the 128 matching up columns
    const int ncb = a.m.N / BN, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int mt, cb;
    if (a.single_expert) {      // dense: super-tiles of sr x sc (row tile, column block) pairs per XCD (kr_prefill_h.hip)
        const int nrt = (a.total_rows + PFX_BM - 1) / PFX_BM, nsc = (ncb + a.sc - 1) / a.sc, ssz = a.sr * a.sc;
        const int sup = (slot / ssz) * 8 + xcd, w = slot % ssz;
        mt = (sup / nsc) * a.sr + w % a.sr; cb = (sup % nsc) * a.sc + w / a.sr;
        if (mt >= nrt || cb >= ncb) return;
    } else {
        const int per = ncb * a.run, grp = slot / per, local = slot - grp * per;
        mt = (grp * 8 + xcd) * a.run + local / ncb; cb = local % ncb;
    }
    int expert, row0, rows;
    if (a.single_expert) { expert = 0; row0 = mt * PFX_BM; rows = a.total_rows - row0 < PFX_BM ? a.total_rows - row0 : PFX_BM; if (rows <= 0) return; }
    else { if (mt >= a.n_tiles[0]) return; expert = a.tile_expert[mt]; row0 = a.tile_row0[mt]; rows = a.tile_rows[mt]; }
    const KrMatDev& m = a.m;
    const int n0 = cb * BN, ng = m.ng, nks = 2 * ng;
    const bool two = rows > 32;
    const uint32_t* wsc = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(m.s) + (size_t)expert * m.s_stride);
    const char* bimg = reinterpret_cast<const char*>(xa.mxq) + (size_t)expert * xa.mxq_stride;
    const char* abase = reinterpret_cast<const char*>(a.a);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n31 = lane & 31, khalf = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(wave);

    const int half_n = m.N >> 1, n0h = cb * 128;
    int col[NC];
    const char* bbase[NC];          // wave-uniform: the column block's first record; lanes add 16 l (+ 1024 - 8 l for the second part) as a 32-bit offset
    uint32_t soff[NC];              // byte offset of the column's scale word inside a group pair's 8-column record row
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int cblk = actf ? (c * half_n + n0h) / 32 + wv : n0 / 32 + wv * 2 + c;           // 32-column block of this wave (uniform)
        col[c] = cblk * 32 + n31;
        bbase[c] = bimg + (size_t)cblk * nks * 1536;
        soff[c] = (uint32_t)(((col[c] >> 3) * m.ngp * 8 + (col[c] & 7)) * 4);
    }
    const uint32_t l16 = (uint32_t)lane * 16, l8 = 1024u + (uint32_t)lane * 8;
    // B fragments: a ring of 4 k-steps in registers (two groups ahead of the MFMAs)
    v6i bq[4][NC];
    uint32_t sraw[2][NC];          // group scale words (bf16 pair of the group pair) of the two groups in flight
    auto load_B = [&](int kk, int slot_) {
        const int kc = __builtin_amdgcn_readfirstlane(kk < nks ? kk : nks - 1);            // past the end: re-read the last record (never multiplied)
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const char* p = bbase[c] + (size_t)kc * 1536;          // uniform
            const u32x4 lo = kr_ldg_nt(reinterpret_cast<const u32x4*>(p + l16));
            const u32x2 hi = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p + l8));
            bq[slot_][c] = v6i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y};
        }
    };
    auto load_S = [&](int g, int slot_) {
        const int gc = __builtin_amdgcn_readfirstlane(g < ng ? g : ng - 1);
        const char* sp = reinterpret_cast<const char*>(wsc) + (size_t)(gc >> 1) * 32;     // uniform
#pragma unroll
        for (int c = 0; c < NC; c++) sraw[slot_][c] = *reinterpret_cast<const uint32_t*>(sp + soff[c]);
    };
    if (wave != 0) { load_B(0, 0); load_B(1, 1); load_S(0, 0); }
    float mulv = 0.0f;
    const uint32_t rowbytes = (uint32_t)ng * GB;
    if (tid < PFX_BM) {
        int src = -1;
        if (tid < rows) {
            if (a.single_expert) src = row0 + tid;
            else { const int pair = a.row_pair[row0 + tid]; src = a.gather_tokens ? pair / a.topk : row0 + tid; }
        }
        row_src[tid] = (int)((uint32_t)(src < 0 ? 0 : src) * rowbytes);
        row_dst[tid] = (a.scatter_rows && !a.single_expert && tid < rows) ? a.row_pair[row0 + tid] : row0 + tid;
        if (src >= 0) mulv = a.a_mul[src];
    }
    __syncthreads();
    if (wave == 0) { load_B(0, 0); load_B(1, 1); load_S(0, 0); }
    // A staging by LDS-DMA: instruction d = wave + 4 j (j < NDMA) of a stage copies segment d >> 3 of rows 8 (d & 7) .. + 8 to stage_base + 1024 d; lane (row
    // lane >> 3, slot lane & 7) fetches chunk slot ^ ((row >> 1) & 7) of the row's 128-byte segment.  Every wave issues exactly NDMA per stage: the waits are constants.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    const uint32_t* rofs = reinterpret_cast<const uint32_t*>(row_src);
    uint32_t aoff[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; j++) {
        const int d = wv + 4 * j, r = 8 * (d & 7) + (lane >> 3);
        aoff[j] = rofs[r] + (uint32_t)(d >> 3) * 128 + (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
    }
    auto dma_A = [&](int g, int stage) {           // group g -> LDS stage `stage` (= g % 3)
        const int gc = g < ng ? g : ng - 1;
        const char* sb = abase + (size_t)gc * GB;
        const uint32_t st0 = lds0 + (uint32_t)stage * STG;
#pragma unroll
        for (int j = 0; j < NDMA; j++) pfx_dma16(st0 + (uint32_t)(wv + 4 * j) * 1024, aoff[j], sb);
    };

    v16f acc[NS][NC];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[s][c][r] = 0.0f;

    // one group (128 k = 2 k-steps of 64): row block by row block -- 2 * NT fragments of the row block, each against the NC column blocks, into the row block's
    // group accumulators; then acc += (32 s) * group accumulator.  Fragment f + 1 is read from LDS before the instructions of fragment f (two fragment buffers); a scheduling fence per
    // fragment keeps the compiler from hoisting all twelve reads to the top (72 registers).
    auto group_mfma = [&](int g, auto par, const char* As) {
        constexpr int PAR = decltype(par)::value, NF = 2 * NT;       // PAR: parity of the group (which ring slots hold its k-steps)
        float sc[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) sc[c] = __uint_as_float(((g & 1) ? (sraw[PAR][c] >> 16) : (sraw[PAR][c] & 0xFFFFu)) << 16) * 32.0f;
        // fragment (ks, t) of this lane's block 2 ks + khalf: slot (ks NT + t) 2 + khalf of the row record = segment (ks NT + t) >> 1, chunks 4 ((ks NT + t) & 1) +
        // 2 khalf (16 B) and + 1 (8 B of codes; for t = 0 also the block's scale word): chunk c of row r sits at slot c ^ ((r >> 1) & 7) of the row's 128 bytes
        v6i fr[2]; int sa[2];
        auto rd = [&](int i, int buf) {          // fragment i = rb * NF + ks * NT + t
            const int rb = i / NF, ks = (i % NF) / NT, t = i % NT, u = ks * NT + t;
            const int r = rb * 32 + n31, swz = (r >> 1) & 7;
            const char* seg = As + ((u >> 1) * 64 + r) * 128;
            const int c0 = 4 * (u & 1) + 2 * khalf;
            if (PFX_ABL & 16) { fr[buf] = v6i{i, lane, i, lane, i, lane}; if (t == 0) sa[ks] = 0x7F7F7F7F; return; }
            const u32x4 lo = *reinterpret_cast<const u32x4*>(seg + ((c0 ^ swz) << 4));
            if (t == 0) {
                const u32x4 hi = *reinterpret_cast<const u32x4*>(seg + (((c0 + 1) ^ swz) << 4));
                fr[buf] = v6i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y}; sa[ks] = (int)hi.z;
            } else {
                const u32x2 hi = *reinterpret_cast<const u32x2*>(seg + (((c0 + 1) ^ swz) << 4));
                fr[buf] = v6i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y};
            }
        };
        v16f ga[NC];
        auto fold = [&](int rb) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const pfx_f2 s2 = {sc[c], sc[c]};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const pfx_f2 o = __builtin_elementwise_fma(pfx_f2{ga[c][r], ga[c][r + 1]}, s2, pfx_f2{acc[rb][c][r], acc[rb][c][r + 1]});
                    acc[rb][c][r] = o.x; acc[rb][c][r + 1] = o.y;
                }
            }
        };
        auto block = [&](int rb) {
            rd(rb * NF, 0);
#pragma unroll
            for (int j = 0; j < NF; j++) {
                const int i = rb * NF + j, ks = j / NT, t = j % NT, buf = j & 1;
                if (j + 1 < NF) rd(i + 1, buf ^ 1);
                const v8i af = pfx_wide(fr[buf]);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const v8i bf = pfx_wide(bq[2 * PAR + ks][c]);
                    v16f cin_;
                    if (j == 0) {
#pragma unroll
                        for (int r = 0; r < 16; r++) cin_[r] = 0.0f;
                    } else cin_ = ga[c];
                    if (PFX_ABL & 4) { ga[c] = cin_; ga[c][0] += (float)(af[0] + bf[0] + sa[ks]); continue; }
                    if (t == 0) ga[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af, bf, cin_, 2, 2, 0, sa[ks], 0, 0x7F7F7F7F);
                    else if (t == 1) ga[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af, bf, cin_, 2, 2, 1, sa[ks], 0, 0x7F7F7F7F);
                    else ga[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af, bf, cin_, 2, 2, 2, sa[ks], 0, 0x7F7F7F7F);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(PFX_ABL & 8)) fold(rb); else { acc[rb][0][0] += ga[0][0]; acc[rb][1][0] += ga[1][0]; }         // the row block's sums (the other wave of the SIMD has the matrix pipe meanwhile)
            __builtin_amdgcn_sched_barrier(0);
        };
        PFX_STAMP(2);
        block(0);
        PFX_STAMP(3);
        block(1);               // (a uniform branch around the second row block of short tiles costs 100 registers of spills: both blocks always run)
    };
    // Vector-memory order of a wave (it retires in order): DMA(0) DMA(1) B(1) | per group g: [wait, barrier] MFMA(g) DMA(g + 2) B(g + 2).  B(g) = the two k-step
    // records of both column blocks + the scale words of group g (NVB instructions, the compiler's loads); DMA(g) = NDMA instructions.  Behind DMA(g) the wave has
    // issued at least B(g) and DMA(g + 1) when it reaches the top of group g: vmcnt(NVB + NDMA) there means DMA(g) -- and B(g), needed next -- have landed.
    // Stage (g + 2) % 3 was last read by MFMA(g - 1), which every wave has left when it passes the barrier of group g.
    dma_A(0, 0); dma_A(1, 1);
    load_B(2, 2); load_B(3, 3); load_S(1, 1);
    int s0 = 0;          // stage of group g: g % 3
    for (int g = 0; g < ng; g += 2) {
        const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        PFX_STAMP(0);
        pfx_wait_barrier<(PFX_ABL & 3) ? 0 : NVB + NDMA>();
        PFX_STAMP(1);
        group_mfma(g, std::integral_constant<int, 0>{}, As0 + s0 * STG);
        PFX_STAMP(4);
        if (!(PFX_ABL & 1)) dma_A(g + 2, s2);
        if (!(PFX_ABL & 2)) { load_B(2 * g + 4, 0); load_B(2 * g + 5, 1); load_S(g + 2, 0); }
        PFX_STAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        pfx_wait_barrier<(PFX_ABL & 3) ? 0 : NVB + NDMA>();
        PFX_STAMP(6);
        group_mfma(g + 1, std::integral_constant<int, 1>{}, As0 + s1 * STG);
        if (!(PFX_ABL & 1)) dma_A(g + 3, s0);
        if (!(PFX_ABL & 2)) { load_B(2 * g + 6, 2); load_B(2 * g + 7, 3); load_S(g + 3, 1); }
        __builtin_amdgcn_sched_barrier(0);
        s0 = s2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA of this wave may land after the workgroup has given its LDS back

    if (tid < PFX_BM) rmul[tid] = a.out_bf16 == 2 ? 1.0f : mulv;
    __syncthreads();
    {
        const bool full = rows == (two ? 64 : 32) && !(a.scatter_rows && !a.single_expert);
        const int nsb = two ? 2 : 1;
        float* out_p = a.out; const int out_ld = a.out_ld;
#define PFX_ST(F_, OT_, A_) pfh_store_tile<NS, NC, F_, OT_, A_>(acc, nsb, rows, row0, rmul, row_dst, out_p, out_ld, col, m.N, lane, a.act_fused, a.act_limit, a.act_alpha)
        if (actf) { if (full) PFX_ST(true, 0, true); else PFX_ST(false, 0, true); }
        else if (a.out_bf16 == 1) { if (full) PFX_ST(true, 1, false); else PFX_ST(false, 1, false); }
        else if (a.out_bf16 == 2) { if (full) PFX_ST(true, 2, false); else PFX_ST(false, 2, false); }
        else { if (full) PFX_ST(true, 0, false); else PFX_ST(false, 0, false); }
#undef PFX_ST
    }
}

static int pfx_launch(const KrPfGemmHArgs& a, const void* mxq, size_t mxq_stride, int mt, hipStream_t st) {
    const KrMatDev& m = a.m;
    if (!mxq || !kr_pfx_shape_ok(m.K, m.N, m.bits) || m.qs || a.n_extra) return 1;
    if (a.act_fused && (m.N / 2) % 128) return 1;
    const size_t lds = (size_t)3 * PFX_STAGE + 3 * PFX_BM * 4;
    if (kr_lds_optin((const void*)kr_pfx_gemm_kernel, 80 * 1024)) return 1;
    const int ncb = m.N / 256;
    PfxArgs x{a, mxq, mxq_stride};
    KrPfGemmHArgs& b = x.g;
    dim3 grid;
    if (a.single_expert) { int n_super; kr_pf_super_tile(mt, ncb, &b.sr, &b.sc, &n_super); grid = dim3(((n_super + 7) / 8) * 8 * b.sr * b.sc); }
    else { const int span = 8 * a.run; grid = dim3(((mt + span - 1) / span) * span * ncb); }
    hipLaunchKernelGGL(kr_pfx_gemm_kernel, grid, dim3(256), lds, st, x);
    return 0;
}
int kr_launch_pfx_gemm(const KrMatDev& m, const void* mxq, size_t mxq_stride, const void* a_mx, const float* a_mul, const KrPfSort* sort, int topk, int gather_tokens, int max_tiles,
                       int single_expert_rows, float* out, int out_ld, hipStream_t st, int scatter_rows = 0, int out_bf16 = 0, int run = 1) {
    KrPfGemmHArgs a{};
    a.m = m; a.a = reinterpret_cast<const uint16_t*>(a_mx); a.a_mul = a_mul; a.topk = topk; a.gather_tokens = gather_tokens; a.scatter_rows = scatter_rows; a.out_bf16 = out_bf16;
    if (sort) { a.row_pair = sort->row_pair; a.tile_expert = sort->tile_expert; a.tile_row0 = sort->tile_row0; a.tile_rows = sort->tile_rows; a.n_tiles = sort->n_tiles; }
    a.out = out; a.out_ld = out_ld; a.single_expert = single_expert_rows > 0; a.total_rows = single_expert_rows;
    a.run = (single_expert_rows > 0 || run < 1) ? 1 : run;
    const int mt = single_expert_rows > 0 ? (single_expert_rows + PFX_BM - 1) / PFX_BM : max_tiles;
    return pfx_launch(a, mxq, mxq_stride, mt, st);
}
// gate | up GEMM with the activation in its epilogue (gu = [rows][I] f32 hidden values), then the MX image of the hidden rows
int kr_launch_pfx_w13_act(const KrMatDev& m, const void* mxq, size_t mxq_stride, const void* a_mx, const float* a_mul, const KrPfSort* sort, int topk, int gather_tokens, int max_tiles,
                          int single_expert_rows, float* gu, int rows, int act_mode, float swiglu_limit, float alpha, void* h_mx, float* h_mul, hipStream_t st, int run) {
    const int I = m.N / 2;
    if (I % 128 || I > 8192 || 256 % (I / 32)) return 1;
    KrPfGemmHArgs a{};
    a.m = m; a.a = reinterpret_cast<const uint16_t*>(a_mx); a.a_mul = a_mul; a.topk = topk; a.gather_tokens = gather_tokens;
    if (sort) { a.row_pair = sort->row_pair; a.tile_expert = sort->tile_expert; a.tile_row0 = sort->tile_row0; a.tile_rows = sort->tile_rows; a.n_tiles = sort->n_tiles; }
    a.out = gu; a.out_ld = I; a.single_expert = single_expert_rows > 0; a.total_rows = single_expert_rows;
    a.run = (single_expert_rows > 0 || run < 1) ? 1 : run;
    a.act_fused = act_mode == KR_ACT_GPTOSS ? 2 : (act_mode == 3 /* libm SiLU, kr_prefill_h.hip */ ? 3 : 1); a.act_limit = swiglu_limit; a.act_alpha = alpha;
    const int mt = single_expert_rows > 0 ? (single_expert_rows + PFX_BM - 1) / PFX_BM : max_tiles;
    if (pfx_launch(a, mxq, mxq_stride, mt, st)) return 1;
    return kr_launch_pfx_rows_f32(gu, rows, I, I, h_mx, h_mul, st);
}
