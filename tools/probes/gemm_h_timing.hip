// Probe: phases of one k-stage of the tolerance GEMM (kr_prefill_h.hip), dense QCN-shaped problem M x 2048 -> 12288, INT4 weights.
// argv: [M=4096] ; env KR_PFH_FORM=1|2 picks the form
#define KR_TIMING 1
#include "../../krasis_amd/csrc/kr_prefill_h.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, K = 2048, N = 12288;
    KrMatDev m{}; m.K = K; m.N = N; m.ng = K / 128; m.ngp = m.ng / 2; m.bits = 4; m.n_fma = N;
    const size_t qb = kr_mat_q_bytes(K, N, 4), sb = kr_mat_s_bytes(K, N);
    void *q, *s; uint16_t* a; float *mul, *out;
    CK(hipMalloc(&q, qb)); CK(hipMalloc(&s, sb)); CK(hipMalloc(&a, (size_t)M * K * 2)); CK(hipMalloc(&mul, M * 4)); CK(hipMalloc(&out, (size_t)M * N * 4));
    std::vector<uint32_t> hq(qb / 4); for (auto& x : hq) x = (uint32_t)rand() * 2654435761u;
    std::vector<uint32_t> hs(sb / 4); for (auto& x : hs) x = 0x3C003C00u + (rand() & 0x7F) * 0x10001u;     // bf16 scales ~0.008
    std::vector<uint16_t> ha((size_t)M * K); for (auto& x : ha) x = 0x3800 + (rand() & 0x3FF) + ((rand() & 1) << 15);   // f16 in +-[0.5, 1)
    std::vector<float> hm(M, 0.0625f);
    CK(hipMemcpy(q, hq.data(), qb, hipMemcpyHostToDevice)); CK(hipMemcpy(s, hs.data(), sb, hipMemcpyHostToDevice));
    CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(mul, hm.data(), M * 4, hipMemcpyHostToDevice));
    m.q = q; m.s = (const uint32_t*)s;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, st));
        kr_launch_pfh_gemm(m, a, mul, nullptr, 1, 0, 0, M, out, N, st);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t[16]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(kr_hstamps), sizeof(t)));
        auto d = [&](int x, int y) { return (long long)(t[y] - t[x]); };
        const double flop = 2.0 * M * (double)K * N;
        printf("rep %d: %.1f us, %.0f TFLOP/s | stage 3 of a mid-grid workgroup, shader clocks: ", rep, ms * 1e3, flop / (ms * 1e-3) / 1e12);
        if (getenv("KR_PFH_FORM") && atoi(getenv("KR_PFH_FORM")) == 2) printf("stage (MFMA + commit) %lld | load issue %lld | barrier %lld | total %lld\n", d(0, 1), d(1, 2), d(2, 3), d(0, 3));
        else printf("commit %lld | load issue %lld | barrier %lld | MFMA block %lld | barrier %lld | total %lld || workgroup: prologue %lld | main loop %lld | epilogue %lld\n", d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(0, 5), d(6, 7), d(7, 8), d(8, 9));
    }
    return 0;
}
