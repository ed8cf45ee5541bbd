// Probe: the LDS-ring tolerance GEMM (kr_prefill_ring.hip) against the register-staged kernel (kr_prefill_h.hip) on one dense problem M x K -> N, INT4 weights:
// bit comparison with a mismatch map, timing of both, shader-clock stamps of one workgroup of the ring kernel.
// build on the box: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/gemm_ring_probe tools/probes/gemm_ring_probe.hip
// argv: [M=4096] [K=2048] [N=12288] [pattern: 0 random, 1 A = 1.0 everywhere (tests the B path), 2 every weight = +1 (tests the A path)]
#define KR_TIMING 1
#include "../../krasis_amd/csrc/kr_prefill_h.hip"
#include "../../krasis_amd/csrc/kr_prefill_ring.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 2048, N = argc > 3 ? atoi(argv[3]) : 12288, pat = argc > 4 ? atoi(argv[4]) : 0;
    KrMatDev m{}; m.K = K; m.N = N; m.ng = K / 128; m.ngp = (m.ng + 1) / 2; m.bits = 4; m.n_fma = N;
    const size_t qb = kr_mat_q_bytes(K, N, 4), sb = kr_mat_s_bytes(K, N);
    void *q, *s; uint16_t* a; float *mul, *out0, *out1;
    CK(hipMalloc(&q, qb)); CK(hipMalloc(&s, sb)); CK(hipMalloc(&a, (size_t)M * K * 2)); CK(hipMalloc(&mul, M * 4));
    CK(hipMalloc(&out0, (size_t)M * N * 4)); CK(hipMalloc(&out1, (size_t)M * N * 4));
    std::vector<uint32_t> hq(qb / 4); for (auto& x : hq) x = (uint32_t)rand() * 2654435761u;
    std::vector<uint32_t> hs(sb / 4); for (auto& x : hs) x = 0x3C003C00u + (rand() & 0x7F) * 0x10001u;     // bf16 scales ~0.008
    std::vector<uint16_t> ha((size_t)M * K); for (auto& x : ha) x = 0x3800 + (rand() & 0x3FF) + ((rand() & 1) << 15);   // f16 in +-[0.5, 1)
    if (pat == 1) for (auto& x : ha) x = 0x3C00;
    if (pat == 2) for (auto& x : hq) x = 0x99999999u;
    std::vector<float> hm(M, 0.0625f);
    CK(hipMemcpy(q, hq.data(), qb, hipMemcpyHostToDevice)); CK(hipMemcpy(s, hs.data(), sb, hipMemcpyHostToDevice));
    CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(mul, hm.data(), M * 4, hipMemcpyHostToDevice));
    m.q = q; m.s = (const uint32_t*)s;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flop = 2.0 * M * (double)K * N;
    if (pat == 3) {       // which k does the ring kernel pair with image position p of the A rows?  one-hot rows, staged outputs as the dictionary
        std::vector<std::vector<float>> S(K, std::vector<float>(N)), R(K, std::vector<float>(N));
        for (int p = 0; p < K; p++) {
            std::fill(ha.begin(), ha.end(), 0); for (int r = 0; r < M; r++) ha[(size_t)r * K + p] = 0x3C00;
            CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
            for (int form = 0; form < 2; form++) {
                kr_pfr_set_enabled(form ? 2 : 0);
                kr_launch_pfh_gemm(m, a, mul, nullptr, 1, 0, 0, M, out0, N, st); CK(hipStreamSynchronize(st));
                CK(hipMemcpy((form ? R : S)[p].data(), out0 + (size_t)5 * N, N * 4, hipMemcpyDeviceToHost));      // row 5
            }
        }
        for (int p = 0; p < K; p++) {
            int found = -1;
            for (int p2 = 0; p2 < K; p2++) if (!memcmp(R[p].data(), S[p2].data(), N * 4)) { found = p2; break; }
            if (found != p) printf("image position %4d (unit %d chunk %d elem %d) -> ring pairs it with the weights of position %4d (unit %d chunk %d elem %d)\n", p, p / 64, (p / 8) % 8, p % 8, found,
                                   found / 64, (found / 8) % 8, found % 8);
        }
        printf("pattern 3 done\n");
        return 0;
    }
    for (int form = 0; form < 2; form++) {
        kr_pfr_set_enabled(form ? 2 : 0);
        float* out = form ? out1 : out0;
        CK(hipMemset(out, 0xFF, (size_t)M * N * 4));
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0, st));
            kr_launch_pfh_gemm(m, a, mul, nullptr, 1, 0, 0, M, out, N, st);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s rep %d: %.1f us, %.0f TFLOP/s\n", form ? "ring  " : "staged", rep, ms * 1e3, flop / (ms * 1e-3) / 1e12);
        }
    }
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    CK(hipMemcpy(h0.data(), out0, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), out1, h1.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; int shown = 0;
    std::vector<int> by_col(256, 0), by_row(128, 0);
    for (size_t i = 0; i < h0.size(); i++) if (memcmp(&h0[i], &h1[i], 4)) {
        bad++; by_col[(i % N) % 256]++; by_row[(i / N) % 128]++;
        if (shown++ < 12) printf("  mismatch row %zu col %zu: staged %g ring %g\n", i / N, i % N, h0[i], h1[i]);
    }
    printf("mismatching outputs: %zu of %zu\n", bad, h0.size());
    if (bad) {
        printf("by col %% 256 (blocks of 32): "); for (int b = 0; b < 8; b++) { long t = 0; for (int i = 0; i < 32; i++) t += by_col[b * 32 + i]; printf("%ld ", t); } printf("\n");
        printf("by row %% 128 (blocks of 32): "); for (int b = 0; b < 4; b++) { long t = 0; for (int i = 0; i < 32; i++) t += by_row[b * 32 + i]; printf("%ld ", t); } printf("\n");
    }
    unsigned long long t[64]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(kr_rstamps), sizeof(t)));
    for (int w = 0; w < 2; w++) {
        const unsigned long long* p = t + 32 * w;
        printf("ring stamps wave %d (shader clocks): prologue to first barrier %lld | to loop start %lld | mid-loop unit (q = 0): wait + barrier %lld, unit body %lld | whole loop %lld (%d units: %.0f per unit) | stores %lld\n",
               4 * w, (long long)(p[1] - p[0]), (long long)(p[2] - p[0]), (long long)(p[9] - p[8]), (long long)(p[10] - p[9]), (long long)(p[6] - p[2]), 2 * m.ng,
               (double)(p[6] - p[2]) / (2 * m.ng), (long long)(p[7] - p[6]));
    }
    return 0;
}
