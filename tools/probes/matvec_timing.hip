// Probe: what does one streaming matvec launch of the decode graph cost, and which part?  40 cold weight matrices (680 MB > Infinity Cache)
// chained in a captured graph, like the projection launches of consecutive layers.
//   build variants:  (none) | -DKR_ABL_NOPROLOGUE (skip activation quantisation) | -DKR_ABL_TINY (N = 64 rows: launch + prologue only)
#include "../../krasis_amd/csrc/kr_moe_decode.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int K = 2048, NM = 40;
    int N = argc > 1 ? atoi(argv[1]) : 12352;
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<KrMatDev> mats(NM);
    float *x, *y; CK(hipMalloc(&x, K * 4)); CK(hipMalloc(&y, (size_t)N * 4 + 1024));
    kr_launch_fill_uniform_f32(x, K, 0.5f, 3, st);
    for (int i = 0; i < NM; i++) {
        void* q; uint32_t* s; const size_t qb = kr_mat_q_bytes(K, N, 4), sb = kr_mat_s_bytes(K, N);
        CK(hipMalloc(&q, qb)); CK(hipMalloc((void**)&s, sb));
        kr_launch_fill_synth(q, qb, s, sb / 4, 100 + i, st);
        KrMatDev m{}; m.q = q; m.s = s; m.K = K; m.N = N; m.ng = K / 128; m.ngp = (m.ng + 1) / 2; m.bits = 4; m.n_fma = (N / 8) * 8;
        mats[i] = m;
    }
    CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NM; i++) kr_launch_matvec(mats[i], x, 1, y, st);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; r++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (5 * NM), mb = (kr_mat_q_bytes(K, N, 4) + kr_mat_s_bytes(K, N)) / 1e6;
    printf("N=%d K=%d  %.2f us per launch  %.1f MB  -> %.2f TB/s\n", N, K, us, mb, mb / us / 1e3 * 1e0);
    return 0;
}
