// kr_prefill_ring.hip -- the tolerance GEMM of the prompt pass (KR_GEMM_FAST) as an LDS RING fed by LDS-DMA: f16 activation rows x INT4-g128 weights
// de-quantized in registers on v_mfma_f32_32x32x16_f16, f32 accumulation over the whole k range.  Same products, same accumulation order per output as
// kr_pfh_gemm_kernel (kr_prefill_h.hip): results are BIT-IDENTICAL to it (tests/test_gemm_ring_gpu.py); reference dataflow gpu_prefill.py:64-239.
//
// Why a second form.  The register-staged kernel requests stage s + 1 after the commit of stage s and needs it landed before the commit of stage s + 1: one stage of
// prefetch distance, two barriers per stage, and two workgroups per CU that fall into the same phase (profiles/r04_gemm_experiments.txt: 0.46 ms of operand
// skeleton + 0.31 ms of matrix pipe = 0.84 ms per expert layer, they do not overlap).  Here ONE workgroup of 8 waves owns the CU and nothing is staged through
// registers: every operand byte travels HBM / L2 -> LDS by `global_load_lds_dwordx4` (1 KiB per wave-instruction, no VGPR destination) into rings whose depth is
// bounded by the 160 KiB of LDS:
//   A ring  : DA = 3 | 4 slots of one 64-k UNIT (TM rows x 128 B), requested DA - 1 units ahead; DA = 4 publishes a unit one barrier early, so the A fragments of
//             its first k-step are read during the previous unit;
//   B ring  : DB slots of one 256-k STAGE (TN / 8 tile records of 1 KiB -- the lane-tiled HBM layout keeps a group PAIR in every 16-byte lane record, so a
//             stage is the smallest piece LDS-DMA can fetch), requested DB - 1 stages ahead and published one barrier early, with the group scales beside it;
//   ONE barrier per unit: before it every wave waits (counted vmcnt) for its own DMA of the unit about to be published, after it the slot read last is refilled.
// A wave's memory counter retires in order, so a wave that requested both operands would have to wait for the far-ahead weight records whenever it needs the
// near-ahead A rows: waves 0-3 request only A, waves 4-7 only B and scales (one of each per SIMD); all 8 waves run the same MFMA work.  The k loop exists once per
// role and has no branch inside a unit: requests past the end of K are clamped and land in a dead slot, so every counted wait is a constant.
// LDS-DMA writes lane-linear (base + lane * 16), so bank conflicts are avoided on the SOURCE side: the lane -> source-chunk map is an XOR swizzle and the
// fragment reads apply the same swizzle (A: chunk ^ (row >> 1 & 7); B: k-slice ^ (2 (col >> 1 & 3) + (record & 1)) inside the record's column block);
// tools/probes/lds_read_probe.hip: both patterns read at the rate of the linear pattern, the un-swizzled ones 2.5 - 3.7 x slower.
// Inside a unit the issue order is pinned with scheduling fences: MFMA, 8 mask / shift operations, MFMA, 4 packed fma + one request or fragment read -- the
// conversion of a B fragment runs one MFMA pair behind the pair that released its registers.
//
// What it reaches (one MI355X, profiles/r06_gemm_ring_experiments.txt): dense 4096 x 2048 -> 12288: 783 - 813 TFLOP/s against 794 - 796 of the register-staged
// kernel; experts only, 8192 tokens: 1.07 against 1.03 ms per layer.  Timing-only ablations of the same loop: without the conversion 1050 - 1080 TFLOP/s,
// without requests, reads and conversion 1290, the 16 MFMAs of a unit alone 1024 matrix-pipe cycles per SIMD against 1890 measured -- the two waves of a SIMD
// issue 2 x (16 MFMA + ~125 VALU) x 4 cycles = 1130 cycles through one vector port per unit: the conversion (7.8 vector instructions per MFMA at a 64 x 64
// wave tile), not the operand staging, is what bounds BOTH forms.  So the dispatch keeps the register-staged kernel (two workgroups per CU also hide each
// other's prologue and store phase, ~10 k cycles per tile here); this form stays selectable (kr_moe_set_gemm_mode 5, option "gemm_ring" 2) and tested.
//
// Tiles: 8 waves as WM x WN, wave tile 64 x 64 (2 x 2 accumulators of 32 x 32, as the register-staged kernel): 64 x 512 for the experts (the 64-row tile table
// of kr_launch_pf_sort; B ring of 2 stages = 128 KiB, A ring of 3) and 128 x 256 for the dense projections (B ring of 2 = 64 KiB, A ring of 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kr_lds_optin.h"
#include "kr_pfh_dev.h"
#include "kr_prefill.h"

#ifndef PFR_ABL    // probe builds only (tools/probes/gemm_ring_probe.hip -DPFR_ABL=bits): 1 no de-quantization VALU, 2 no requests inside the loop, 4 no fragment reads inside the loop, 8 no barriers
#define PFR_ABL 0  // inside the loop -- results wrong by construction, timing only
#endif
#ifdef KR_TIMING   // tools/probes/gemm_ring_probe.hip: shader-clock stamps of wave 0 / wave 4 of one mid-grid workgroup; no-op in the product build
__device__ unsigned long long kr_rstamps[64];
#define PFR_STAMP(i) do { if ((threadIdx.x & 255) == 0 && blockIdx.x == gridDim.x / 2) kr_rstamps[(i) + 32 * (threadIdx.x >> 8)] = clock64(); } while (0)
#else
#define PFR_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ uint32_t pfr_lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
// one LDS-DMA wave-instruction: 64 lanes x 16 B, lane i's bytes land at lds_dst + 16 i; source = sbase (wave-uniform) + voff (per lane).  M0 carries the LDS
// address and is the compiler's register: saved and restored inside the statement (guide §5.7).  The compiler does not count this load: every wait is by hand.
__device__ __forceinline__ const void* pfr_uniform_ptr(const void* p) {      // the operands below must sit in SGPRs: no-ops when the compiler already knows
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const void*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void pfr_dma16(uint32_t lds_dst, uint32_t voff, const void* sbase) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst); sbase = pfr_uniform_ptr(sbase);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
__device__ __forceinline__ void pfr_dma4(uint32_t lds_dst, uint32_t voff, const void* sbase) {      // 64 lanes x 4 B
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst); sbase = pfr_uniform_ptr(sbase);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
// wait until at most N of this wave's DMAs are outstanding and its LDS reads have returned, then the workgroup barrier
template <int N> __device__ __forceinline__ void pfr_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}

// WM x WN waves (= 8), DB = depth of the B ring in stages (2 | 3), DA = depth of the A ring in units: 3 = a unit's rows are published by the barrier that opens it,
// 4 = one barrier early (the A fragments of a unit's first k-step are then read during the previous unit, like the B words)
template <int WM, int WN, int DB, int DA>
__global__ void __launch_bounds__(512) kr_pfr_gemm_kernel(const KrPfGemmHArgs a) {
    static_assert(WM * WN == 8 && (DA == 3 || DA == 4) && (DB == 2 || DB == 3), "8 waves; ring depths");
    constexpr bool AE = DA == 4;
    constexpr int TM = 64 * WM, TN = 64 * WN, NR = TN / 8, NS = 2, NC = 2;
    constexpr int B_SLOT = NR * 1024, S_SLOT = TN * 4, A_SLOT = TM * 128;
    constexpr int OFF_B = 0, OFF_A = DB * B_SLOT, OFF_S = OFF_A + DA * A_SLOT, OFF_RM = OFF_S + DB * S_SLOT, OFF_RD = OFF_RM + TM * 4;
    constexpr int NAW = TM / 32;            // A requests per A wave and unit
    constexpr int NRW = NR / 4;             // tile records per B wave and stage
    constexpr int NSW = WN / 4;             // scale requests (64 columns x 4 B) per B wave and stage
    constexpr int PB = (NRW + 2) / 3;       // records per B wave and unit: a stage's records are requested in units 0..2 of an earlier stage (unit 3 opens with the wait for them)
    static_assert(NAW <= 8 && NSW + PB <= 8, "one request per gap between the 8 MFMA pairs of a unit");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float* rmul = reinterpret_cast<float*>(smem + OFF_RM);         // [TM]
    int* row_dst = reinterpret_cast<int*>(smem + OFF_RD);          // [TM]
    const uint32_t lds0 = pfr_lds_addr(smem);

    PFR_STAMP(0);
    const bool actf = a.act_fused != 0;      // uniform; N = 2 I with I % (TN / 2) == 0 (the launcher checks): a tile = TN / 2 gate + the matching up columns
    const int ncb0 = (a.m.N + TN - 1) / TN, ncb1 = a.n_extra > 0 ? (a.mx[0].N + TN - 1) / TN : 0, ncb2 = a.n_extra > 1 ? (a.mx[1].N + TN - 1) / TN : 0;
    const int ncb = ncb0 + ncb1 + ncb2, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int mt, cb;
    if (a.single_expert) {       // dense: super-tiles of sr x sc (row tile, column block) pairs per XCD, as kr_pfh_gemm_kernel
        const int nrt = (a.total_rows + TM - 1) / TM, nsc = (ncb + a.sc - 1) / a.sc, ssz = a.sr * a.sc;
        const int sup = (slot / ssz) * 8 + xcd, w = slot % ssz;
        mt = (sup / nsc) * a.sr + w % a.sr; cb = (sup % nsc) * a.sc + w / a.sr;
        if (mt >= nrt || cb >= ncb) return;
    } else {
        const int per = ncb * a.run, grp = slot / per, local = slot - grp * per;
        mt = (grp * 8 + xcd) * a.run + local / ncb; cb = local % ncb;
    }
    int expert, row0, rows;
    if (a.single_expert) { expert = 0; row0 = mt * TM; rows = a.total_rows - row0 < TM ? a.total_rows - row0 : TM; if (rows <= 0) return; }
    else { if (mt >= a.n_tiles[0]) return; expert = a.tile_expert[mt]; row0 = a.tile_row0[mt]; rows = a.tile_rows[mt]; }
    KrMatDev m = a.m; float* out_p = a.out; int out_ld = a.out_ld;
    if (cb >= ncb0 + ncb1) { cb -= ncb0 + ncb1; m = a.mx[1]; out_p = a.outx[1]; out_ld = a.out_ldx[1]; }
    else if (cb >= ncb0) { cb -= ncb0; m = a.mx[0]; out_p = a.outx[0]; out_ld = a.out_ldx[0]; }
    const int n0 = cb * TN, half_n = m.N >> 1, n0h = cb * (TN / 2);
    const int K = m.ng * 128, U = 2 * m.ng, nst = m.ngp;      // the launcher takes ng >= 2: U >= 4
    const char* wq = reinterpret_cast<const char*>(m.q) + (size_t)expert * m.q_stride;
    const char* wsc = reinterpret_cast<const char*>(m.s) + (size_t)expert * m.s_stride;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int n31 = lane & 31, khalf = lane >> 5;
    const int last_tile = (m.N - 1) >> 3;

    // ---------------- request side ----------------
    // local record ri of the tile (8 columns each): wave column ri >> 3, accumulator column block c = ri >> 2 & 1, q = ri & 3 -> global column tile
    auto rec_tile = [&](int ri) {
        int t = actf ? (((ri >> 2) & 1) ? (half_n >> 3) : 0) + (n0h >> 3) + (ri >> 3) * 4 + (ri & 3) : (n0 >> 3) + ri;
        return t < last_tile ? t : last_tile;       // a column tile past the last one re-reads the last tile (its columns are never stored)
    };
    // A waves (0..3): this lane's source rows.  Request j of wave `wave` fills LDS rows 8 (wave * NAW + j) .. + 8 of the unit slot: lane i -> row + (i >> 3), position i & 7,
    // which holds chunk (i & 7) ^ (row >> 1 & 7) of the row's 128 bytes.
    // B waves (4..7): lane i of a record request lands at position i = cc * 8 + x of the record and fetches the record's chunk cc * 8 + (x ^ (2 (cc >> 1) + (ri & 1)))
    // (one register set for both roles: a wave has one role)
    constexpr int NRO = NAW > NSW + 1 ? NAW : NSW + 1;
    uint32_t roff[NRO];
#pragma unroll
    for (int j = 0; j < NRO; j++) roff[j] = 0;
    if (wave < 4) {
#pragma unroll
        for (int j = 0; j < NAW; j++) {
            const int r = 8 * (wave * NAW + j) + (lane >> 3);
            int src = 0;
            if (r < rows) {
                if (a.single_expert) src = row0 + r;
                else src = a.gather_tokens ? a.row_pair[row0 + r] / a.topk : row0 + r;
            }
            roff[j] = (uint32_t)src * (uint32_t)(K * 2) + (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
        }
    } else {
        const int cc = lane >> 3, x = lane & 7;
        roff[NSW] = (uint32_t)((cc * 8 + (x ^ (2 * (cc >> 1)))) * 16);          // parity-0 records; parity 1: ^ 16
#pragma unroll
        for (int j = 0; j < NSW; j++) {
            const int lc = ((wave - 4) * NSW + j) * 64 + lane;              // local column
            int col = actf ? ((lc >> 5) & 1) * half_n + n0h + (lc >> 6) * 32 + (lc & 31) : n0 + lc;
            col = col < m.N ? col : m.N - 1;
            roff[j] = (uint32_t)(((col >> 3) * m.ngp * 8 + (col & 7)) * 4);
        }
    }
#define aoff roff
#define soff roff
#define boff roff[NSW]
    auto dma_A = [&](int u, int j) {        // request j of this A wave for unit u -> slot u % DA
        pfr_dma16(lds0 + OFF_A + (uint32_t)(u % DA) * A_SLOT + (uint32_t)(wave * NAW + j) * 1024, aoff[j], reinterpret_cast<const char*>(a.a) + (size_t)u * 128);
    };
    auto dma_S = [&](int st, int j) {       // scale request j of this B wave for stage st
        pfr_dma4(lds0 + OFF_S + (uint32_t)(st % DB) * S_SLOT + (uint32_t)(((wave - 4) * NSW + j) * 256), soff[j], wsc + (size_t)st * 32);
    };
    auto dma_B = [&](int st, int i) {       // record i (0 .. NRW - 1) of this B wave for stage st -> slot st % DB
        const int ri = (wave - 4) * NRW + i;
        pfr_dma16(lds0 + OFF_B + (uint32_t)(st % DB) * B_SLOT + (uint32_t)ri * 1024, boff ^ ((ri & 1) ? 16u : 0u), wq + (size_t)rec_tile(ri) * m.ngp * 1024 + (size_t)st * 1024);
    };
    auto issue_B_stage = [&](int st) {      // prologue: a whole stage
#pragma unroll
        for (int j = 0; j < NSW; j++) dma_S(st, j);
#pragma unroll
        for (int i = 0; i < NRW; i++) dma_B(st, i);
    };

    // ---------------- prologue ----------------
    if (wave >= 4) {
        issue_B_stage(0);
        if (DB >= 3 && nst > 1) issue_B_stage(1);
    } else {
#pragma unroll
        for (int u0 = 0; u0 < DA - 1; u0++)
#pragma unroll
            for (int j = 0; j < NAW; j++) dma_A(u0, j);
    }
    if (tid < TM) {
        int src = -1;
        if (tid < rows) {
            if (a.single_expert) src = row0 + tid;
            else src = a.gather_tokens ? a.row_pair[row0 + tid] / a.topk : row0 + tid;
        }
        row_dst[tid] = (a.scatter_rows && !a.single_expert && tid < rows) ? a.row_pair[row0 + tid] : row0 + tid;
        rmul[tid] = (a.out_bf16 == 2 || src < 0) ? (a.out_bf16 == 2 ? 1.0f : 0.0f) : a.a_mul[src];
    }

    v16f acc[NS][NC];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[s][c][r] = 0.0f;
    uint32_t M0 = 0x000F000Fu, M1 = 0x00F000F0u, MH = 0x03C003C0u, Kc = 0x64006400u;      // de-quantization masks, kept in registers (see pfh_dq4)
    asm volatile("" : "+v"(M0), "+v"(M1), "+v"(MH), "+v"(Kc));

    // ---------------- fragment side ----------------
    // A: row wr * 64 + s2 * 32 + n31 of the unit slot; MFMA (tt, h) of a unit takes chunk j = 4 tt + 2 khalf + h of the row, stored at j ^ (row >> 1 & 7):
    //    address = (rowbase + 16 g) ^ (64 tt + 16 h) with g = (2 khalf) ^ (n31 >> 1 & 7)
    uint32_t a_lane_v = (uint32_t)((wr * 64 + n31) * 128 + (((2 * khalf) ^ ((n31 >> 1) & 7)) * 16));
    // B: column block c of this wave = local records wc * 8 + c * 4 + (n31 >> 3), column cc = n31 & 7 inside the record; k-slice l = 2 t' + khalf of the stage
    //    (t' = k-step inside a group, 0..3) sits at position cc * 8 + (l ^ sw), sw = 2 (cc >> 1) + (n31 >> 3 & 1): address = (recbase + 128 cc + 16 e) ^ (32 t'),
    //    e = khalf ^ sw; the 16 bytes are {g0: 2 words, g1: 2 words}: + 8 hh
    uint32_t b_lane[NC];
    uint32_t s_lane[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int ri = wc * 8 + c * 4 + (n31 >> 3), cc = n31 & 7, sw = 2 * (cc >> 1) + ((n31 >> 3) & 1);
        b_lane[c] = (uint32_t)(ri * 1024 + cc * 128 + ((khalf ^ sw) * 16));
        s_lane[c] = (uint32_t)((wc * 64 + c * 32 + n31) * 4);
    }
    const char* Asm = smem + OFF_A;
    const char* Bsm = smem + OFF_B;
    const char* Ssm = smem + OFF_S;

    v2h sqc[NC], cqc[NC];          // scale / constant of the group being de-quantized
    uint32_t spv[NC] = {0, 0};     // bf16 scale pair of the stage
    auto load_scales = [&](int st) {
#pragma unroll
        for (int c = 0; c < NC; c++) spv[c] = *reinterpret_cast<const uint32_t*>(Ssm + (st % DB) * S_SLOT + s_lane[c]);
    };
    auto set_group = [&](int hh) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float sc = __uint_as_float((hh ? (spv[c] >> 16) : (spv[c] & 0xFFFFu)) << 16);
            const _Float16 s1 = (_Float16)(sc * 0.25f);
            sqc[c] = v2h{s1, s1};
            const _Float16 c1 = (_Float16)(-1536.0f * (float)s1);
            cqc[c] = v2h{c1, c1};
        }
    };
    v8h af[2][NS][2];              // [tt][row block][h]
    v8h bf[2 * NC];                // de-quantized B fragments, f = 2 h + c: ONE k-step deep -- fragment f of the next k-step is rebuilt one MFMA pair behind its last use
    u32x2 br[2][NC];               // raw words of two k-steps, by k-step parity
    auto rd_B = [&](int st, int tq, int buf) {          // tq = k-step of the stage, 0..7
        const int hh = tq >> 2, tp = tq & 3;
#pragma unroll
        for (int c = 0; c < NC; c++) br[buf][c] = *reinterpret_cast<const u32x2*>(Bsm + (st % DB) * B_SLOT + ((b_lane[c] ^ (uint32_t)(tp * 32)) + hh * 8));
    };
    auto rd_A = [&](int u, int tt) {
        const char* base = Asm + (u % DA) * A_SLOT;
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++)
#pragma unroll
            for (int h = 0; h < 2; h++)
                af[tt][s2][h] = *reinterpret_cast<const v8h*>(base + ((a_lane_v + s2 * 32 * 128) ^ (uint32_t)(tt * 64 + h * 16)));
    };
    // pfh_dq4 in two halves (same operations, same results): the 8 mask / shift operations, then the 4 packed fma
    uint32_t dt[4];
    auto dq_a = [&](int buf, int f) {
        if (PFR_ABL & 1) return;
        const uint32_t w = (f >> 1) ? br[buf][f & 1].y : br[buf][f & 1].x;
        dt[0] = ((w & M0) << 6) | Kc; dt[1] = ((w & M1) << 2) | Kc; dt[2] = ((w >> 2) & MH) | Kc; dt[3] = ((w >> 6) & MH) | Kc;
    };
    auto dq_b = [&](int f) {
        if (PFR_ABL & 1) { asm volatile("" : "+v"(bf[f])); return; }
        const int c = f & 1;
        const v2h r0 = __builtin_elementwise_fma(__builtin_bit_cast(v2h, dt[0]), sqc[c], cqc[c]), r1 = __builtin_elementwise_fma(__builtin_bit_cast(v2h, dt[1]), sqc[c], cqc[c]);
        const v2h r2 = __builtin_elementwise_fma(__builtin_bit_cast(v2h, dt[2]), sqc[c], cqc[c]), r3 = __builtin_elementwise_fma(__builtin_bit_cast(v2h, dt[3]), sqc[c], cqc[c]);
        bf[f] = v8h{r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y};
    };

    // ---------------- main loop ----------------
    // unit u = 4 s + q (q = 0..3: group q >> 1 of stage s), k-steps ks = 2 q + tt of the stage; per k-step four MFMA pairs f = 2 h + c.  Software pipeline, one pair
    // deep: in the slot of pair (ks, f) the wave rebuilds the fragment the PREVIOUS pair released -- (ks + 1, f - 1), or (ks, 3) when f = 0 -- so every MFMA is followed by
    // half a conversion (8 logic operations behind the first MFMA of a pair, 4 packed fma behind the second: the wave issues in order, and a pair issued back to back
    // leaves it parked on the busy matrix pipe while its vector work waits).  The raw words of k-step ks are read during k-step ks - 2 (B is published a barrier early);
    // the scales switch between the pairs 0 and 1 of the k-step before a group's first.
    PFR_STAMP(1);
    if (wave >= 4) { if (DB >= 3 && nst > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NRW + NSW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else if (AE) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DA - 2) * NAW) : "memory");          // unit 0 landed, units 1 .. DA - 2 in flight
    __syncthreads();               // B(0) + its scales are in LDS (DA = 4: and A(0)), and the row tables
    load_scales(0); set_group(0);
    rd_B(0, 0, 0); rd_B(0, 1, 1);
    if (AE) rd_A(0, 0);
#pragma unroll
    for (int f = 0; f < 2 * NC - 1; f++) { dq_a(0, f); dq_b(f); }      // k-step 0, fragments 0..2; fragment 3 is built in the slot of pair (0, 0) like every k-step's
    PFR_STAMP(2);

    // The loop exists twice, once per request role (A waves / B waves): inside a unit there is no branch at all -- requests past the end of K are clamped to the last
    // unit / stage and land in a slot nobody reads any more, so the request count per unit, and with it every counted wait, is a constant.
    auto unit = [&](int s, auto qq, auto role) {
        constexpr int q = decltype(qq)::value;
        constexpr bool RA = decltype(role)::value == 0;
        const int u = 4 * s + q;
#ifdef KR_TIMING
        const bool stamp = q == 0 && s == (nst >> 1);
        if (stamp) PFR_STAMP(8);
#endif
        // (1) counted wait for this wave's requests, then the barrier.  A waves: the unit published here is u (DA = 3) or u + 1 (DA = 4); the unit requested behind it stays in
        // flight.  B waves: before the barrier that opens the stage's last unit the next stage must have landed (requested in units 0..2 of this stage or, DB = 3, of the previous one)
        if (PFR_ABL & 8) { }
        else if (RA) pfr_wait_barrier<NAW>();
        else if (q < 3) pfr_wait_barrier<63>();
        else if (DB >= 3) pfr_wait_barrier<NRW + NSW>();
        else pfr_wait_barrier<0>();
#ifdef KR_TIMING
        if (stamp) PFR_STAMP(9);
#endif
        // (2) the A fragments still missing (DA = 3: both k-steps; DA = 4: the second -- the first was read during the previous unit).  The lane parts of the fragment
        // addresses are made opaque per unit: left alone the optimizer keeps all 12 XOR-ed variants in registers across both loop copies and the kernel spills
        asm volatile("" : "+v"(a_lane_v), "+v"(b_lane[0]), "+v"(b_lane[1]));
        if (!(PFR_ABL & 4)) { if (!AE) rd_A(u, 0); rd_A(u, 1); }
        __builtin_amdgcn_sched_barrier(0);
        const int sn = (q == 3) ? (s + 1 < nst ? s + 1 : s) : s;       // stage of the next unit (past the last stage: valid bytes that are never used)
        const int ua = u + DA - 1 < U ? u + DA - 1 : U - 1;            // unit / stage this unit's requests fetch (clamped past the end)
        const int sb = s + DB - 1 < nst ? s + DB - 1 : nst - 1;
        const uint32_t a_dst = lds0 + OFF_A + (uint32_t)((u + DA - 1) % DA) * A_SLOT + (uint32_t)(wave * NAW) * 1024;
        const char* a_src = reinterpret_cast<const char*>(a.a) + (size_t)ua * 128;
        const uint32_t b_slot = (uint32_t)((s + DB - 1) % DB);
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int f = 0; f < 2 * NC; f++) {
                const int h = f >> 1, c = f & 1, ks = 2 * q + tt, g = tt * 4 + f;       // ks = k-step of the stage, g = pair of the unit 0..7
                acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tt][0][h], bf[f], acc[0][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // fragment released by the previous pair: (ks, 3) in pair 0, else (ks + 1, f - 1); raw words by k-step parity
                if (f == 0) dq_a(ks & 1, 3);
                else dq_a((ks + 1) & 1, f - 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tt][1][h], bf[f], acc[1][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (f == 0) {
                    dq_b(3);
                    // a group's first k-step is next (ks + 1 = 0 or 4 mod 8): from here on every conversion belongs to it (the next stage's scales were read in pair 2 of k-step 6)
                    if (((ks + 1) & 3) == 0) set_group(((ks + 1) & 7) ? 1 : 0);
                } else dq_b(f - 1);
                // reads: the raw words of k-step ks + 2 (two k-steps ahead, into the buffer k-step ks released in pair 0); in pair 2 of a stage's k-step 6 also the next stage's scales;
                // DA = 4: behind the first k-step's last pair the next unit's first A fragments
                if (f == 1 && !(PFR_ABL & 4)) rd_B(ks + 2 < 8 ? s : sn, (ks + 2) & 7, ks & 1);
                if (f == 2 && ks == 6 && !(PFR_ABL & 4)) load_scales(sn);
                if (AE && tt == 0 && f == 3 && !(PFR_ABL & 4)) rd_A(u + 1, 0);
                // this wave's request of pair g
                if (PFR_ABL & 2) { }
                else if (RA) { if (g < NAW) pfr_dma16(a_dst + g * 1024, aoff[g < NAW ? g : 0], a_src); }
                else if (q < 3) {
                    const int gi = g - (q == 0 ? NSW : 0), i = q * PB + gi;
                    if (q == 0 && g < NSW) pfr_dma4(lds0 + OFF_S + b_slot * S_SLOT + (uint32_t)(((wave - 4) * NSW + g) * 256), soff[g < NSW ? g : 0], wsc + (size_t)sb * 32);
                    else if (gi >= 0 && gi < PB && i < NRW) {
                        const int ri = (wave - 4) * NRW + i;
                        pfr_dma16(lds0 + OFF_B + b_slot * B_SLOT + (uint32_t)ri * 1024, boff ^ ((ri & 1) ? 16u : 0u), wq + (size_t)rec_tile(ri) * m.ngp * 1024 + (size_t)sb * 1024);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef KR_TIMING
        if (stamp) PFR_STAMP(10);
#endif
    };
    auto run = [&](auto role) {
        for (int s = 0; s < nst; s++) {
            unit(s, std::integral_constant<int, 0>{}, role);
            unit(s, std::integral_constant<int, 1>{}, role);
            if (4 * s + 2 >= U) break;         // odd group count: the last stage holds one group
            unit(s, std::integral_constant<int, 2>{}, role);
            unit(s, std::integral_constant<int, 3>{}, role);
        }
    };
    if (wave < 4) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    PFR_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int rows_w = rows - wr * 64;                 // rows of this wave's 64-row block
        if (rows_w > 0) {
            // the lane id is formed again here (a different expression: nothing derived from threadIdx stays live across the k loop, which runs at the register limit)
            const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), n31_e = lane_e & 31;
            const int nsb = rows_w > 32 ? 2 : 1, rw = rows_w < 64 ? rows_w : 64;
            int col[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) col[c] = actf ? c * half_n + n0h + wc * 32 + n31_e : n0 + wc * 64 + c * 32 + n31_e;
            const bool full = rw == (nsb == 2 ? 64 : 32) && (actf || n0 + TN <= m.N) && !(a.scatter_rows && !a.single_expert);     // uniform per wave
#define PFR_ST(F_, OT_, A_) pfh_store_tile<NS, NC, F_, OT_, A_>(acc, nsb, rw, row0 + wr * 64, rmul + wr * 64, row_dst + wr * 64, out_p, out_ld, col, m.N, lane_e, a.act_fused, a.act_limit, a.act_alpha)
            if (a.act_fused) { if (full) PFR_ST(true, 0, true); else PFR_ST(false, 0, true); }
            else if (a.out_bf16 == 1) { if (full) PFR_ST(true, 1, false); else PFR_ST(false, 1, false); }
            else if (a.out_bf16 == 2) { if (full) PFR_ST(true, 2, false); else PFR_ST(false, 2, false); }
            else { if (full) PFR_ST(true, 0, false); else PFR_ST(false, 0, false); }
#undef PFR_ST
        }
    }
    PFR_STAMP(7);
#undef aoff
#undef soff
#undef boff
}

template <int WM, int WN, int DB, int DA>
static int pfr_launch(const KrPfGemmHArgs& a, int mt, hipStream_t st) {
    constexpr int TM = 64 * WM, TN = 64 * WN, NR = TN / 8;
    const size_t lds = (size_t)DB * NR * 1024 + (size_t)DA * TM * 128 + (size_t)DB * TN * 4 + (size_t)TM * 8;
    if (kr_lds_optin((const void*)kr_pfr_gemm_kernel<WM, WN, DB, DA>, lds)) return 1;
    int ncb = (a.m.N + TN - 1) / TN;
    for (int i = 0; i < a.n_extra; i++) ncb += (a.mx[i].N + TN - 1) / TN;
    KrPfGemmHArgs b = a;
    dim3 grid;
    if (a.single_expert) { int n_super; kr_pf_super_tile(mt, ncb, &b.sr, &b.sc, &n_super); grid = dim3(((n_super + 7) / 8) * 8 * b.sr * b.sc); }
    else { const int span = 8 * a.run; grid = dim3(((mt + span - 1) / span) * span * ncb); }
    hipLaunchKernelGGL((kr_pfr_gemm_kernel<WM, WN, DB, DA>), grid, dim3(512), lds, st, b);
    return 0;
}

// 0 = launched.  mt64 = row tiles of 64 rows (the experts' tile table); dense problems carry their row count in a.total_rows.  kr_pfr_set_enabled(0) keeps the
// register-staged kernel (process-wide A/B and test hook: kr_moe_set_gemm_mode(e, 3), kr_decode_set_option(s, "gemm_ring", 0)); the caller falls back to it for
// every shape this form does not take.
// 0 off, 1 = the measured policy (today: never -- on one MI355X the ring form ties the register-staged kernel on dense problems, 780 - 810 TFLOP/s, and is 4 - 12 % behind it
// on the experts: both are bound by the 7.8 vector instructions per MFMA of the INT4 -> f16 conversion, profiles/r06_gemm_ring_experiments.txt), 2 = every shape the kernel takes
static std::atomic<int> g_pfr_on{1};
void kr_pfr_set_enabled(int on) { g_pfr_on.store(on < 0 ? 0 : (on > 2 ? 2 : on)); }
int kr_pfr_try_launch(const KrPfGemmHArgs& a, int mt64, hipStream_t st) {
    const int on = g_pfr_on.load(std::memory_order_relaxed);
    const long min_wg = 1;
    if (on != 2 || a.m.bits != 4 || a.m.qs) return 1;
    if (a.m.ng < 2) return 1;
    for (int i = 0; i < a.n_extra; i++) if (a.mx[i].bits != 4 || a.mx[i].qs || a.mx[i].ng != a.m.ng) return 1;
    if (a.single_expert) {
        // dense: 128 x 256 tiles when they still give every CU work, else keep the 64-row kernel
        long n256 = (a.m.N + 255) / 256;
        for (int i = 0; i < a.n_extra; i++) n256 += (a.mx[i].N + 255) / 256;
        const int mt128 = (a.total_rows + 127) / 128;
        if ((long)mt128 * n256 < min_wg || a.act_fused) return 1;
        return pfr_launch<2, 4, 2, 4>(a, mt128, st);
    }
    // experts: 64 x 512 over the 64-row tile table
    if (a.act_fused && ((a.m.N / 2) % 256 != 0)) return 1;
    if (!a.act_fused && a.m.N % 8 != 0) return 1;
    long n512 = (a.m.N + 511) / 512;
    if ((long)mt64 * n512 < min_wg) return 1;
    return pfr_launch<1, 8, 2, 3>(a, mt64, st);
}
