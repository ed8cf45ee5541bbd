// Probe: the MX form of the tolerance GEMM (kr_prefill_mx.hip) against the f16 form (kr_prefill_h.hip) and a double-precision product on one dense problem
// M x K -> N with INT4-g128 weights: errors of both forms against the exact product of the de-quantized weights, timing of both.
// build on the box: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/gemm_mx_probe tools/probes/gemm_mx_probe.hip
// argv: [M=4096] [K=2048] [N=12288]
#define KR_TIMING 1
#include "../../krasis_amd/csrc/kr_prefill_h.hip"
#include "../../krasis_amd/csrc/kr_prefill_ring.hip"
#include "mx_gemm_experiment.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static float bf16f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 2048, N = argc > 3 ? atoi(argv[3]) : 12288;
    KrMatDev m{}; m.K = K; m.N = N; m.ng = K / 128; m.ngp = (m.ng + 1) / 2; m.bits = 4; m.n_fma = N;
    const size_t qb = kr_mat_q_bytes(K, N, 4), sb = kr_mat_s_bytes(K, N), ib = kr_pfx_wimage_bytes(K, N);
    void *q, *s, *img, *amx; uint16_t* ah; float *x, *mulh, *mulx, *out0, *out1;
    CK(hipMalloc(&q, qb)); CK(hipMalloc(&s, sb)); CK(hipMalloc(&img, ib)); CK(hipMalloc(&x, (size_t)M * K * 4)); CK(hipMalloc(&ah, (size_t)M * K * 2)); CK(hipMalloc(&amx, (size_t)M * kr_pfx_row_bytes(K)));
    CK(hipMalloc(&mulh, M * 4)); CK(hipMalloc(&mulx, M * 4)); CK(hipMalloc(&out0, (size_t)M * N * 4)); CK(hipMalloc(&out1, (size_t)M * N * 4));
    srand(7);
    std::vector<uint32_t> hq(qb / 4); for (auto& v : hq) v = ((uint32_t)rand() << 16) ^ (uint32_t)rand() ^ ((uint32_t)rand() << 30);
    std::vector<uint32_t> hs(sb / 4); for (auto& v : hs) { const uint32_t a = 0x3C00 + (rand() & 0x7F), b = 0x3C00 + (rand() & 0x7F); v = a | (b << 16); }     // bf16 scales ~0.008 .. 0.016
    std::vector<float> hx((size_t)M * K);
    for (size_t i = 0; i < hx.size(); i++) { const float u = (rand() % 20001 - 10000) / 10000.0f; hx[i] = u * u * u * (1.0f + (i / K) % 7) * ((i % 97) == 0 ? 8.0f : 1.0f); }     // heavy-tailed rows, a few outliers
    CK(hipMemcpy(q, hq.data(), qb, hipMemcpyHostToDevice)); CK(hipMemcpy(s, hs.data(), sb, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    m.q = q; m.s = (const uint32_t*)s;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kr_launch_pfx_wimage(m, 1, img, ib, st); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
    const double flop = 2.0 * M * (double)K * N;
    kr_pfr_set_enabled(0);
    for (int form = 0; form < 2; form++)
        for (int rep = 0; rep < 4; rep++) {
            float ms_rows, ms;
            CK(hipEventRecord(e0, st));
            if (form == 0) kr_launch_pfh_rows_f32(x, M, K, K, ah, mulh, st); else if (kr_launch_pfx_rows_f32(x, M, K, K, amx, mulx, st)) { printf("rows: shape not covered\n"); return 1; }
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipGetLastError()); CK(hipEventElapsedTime(&ms_rows, e0, e1));
            CK(hipEventRecord(e0, st));
            if (form == 0) kr_launch_pfh_gemm(m, ah, mulh, nullptr, 1, 0, 0, M, out0, N, st);
            else if (kr_launch_pfx_gemm(m, img, ib, amx, mulx, nullptr, 1, 0, 0, M, out1, N, st)) { printf("gemm: shape not covered\n"); return 1; }
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipGetLastError()); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s rep %d: rows %.1f us, GEMM %.1f us = %.0f TFLOP/s\n", form ? "MX " : "f16", rep, ms_rows * 1e3, ms * 1e3, flop / (ms * 1e-3) / 1e12);
        }
    // reference: rows 0, 1, 77, M-1 and every column, in double over the de-quantized weights
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    CK(hipMemcpy(h0.data(), out0, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), out1, h1.size() * 4, hipMemcpyDeviceToHost));
    const int rows[6] = {0, 1, 77 % M, M / 2, M - 2, M - 1};
    double e_h = 0, e_x = 0, ref2 = 0, mx_h = 0, mx_x = 0, mxr = 0;
    for (int ri = 0; ri < 6; ri++) {
        const int r = rows[ri];
        for (int n = 0; n < N; n++) {
            double acc = 0;
            const int tile = n >> 3, c = n & 7;
            for (int k = 0; k < K; k++) {
                const int g = k >> 7, i = (k & 127) >> 3, j = k & 7, l = i >> 1, wi = (g & 1) * 2 + (i & 1);
                const uint32_t pk = hq[(((size_t)tile * m.ngp + (g >> 1)) * 64 + c * 8 + l) * 4 + wi];
                const int nib = (pk >> (4 * j)) & 15;
                const uint32_t sw = hs[((size_t)tile * m.ngp + (g >> 1)) * 8 + c];
                const float sc = bf16f((g & 1) ? (uint16_t)(sw >> 16) : (uint16_t)(sw & 0xFFFF));
                acc += (double)hx[(size_t)r * K + k] * (double)((nib - 8) * sc);
            }
            const double d0 = h0[(size_t)r * N + n] - acc, d1 = h1[(size_t)r * N + n] - acc;
            e_h += d0 * d0; e_x += d1 * d1; ref2 += acc * acc; mx_h = fmax(mx_h, fabs(d0)); mx_x = fmax(mx_x, fabs(d1)); mxr = fmax(mxr, fabs(acc));
        }
    }
    printf("against the double product (6 rows x %d columns): f16 form rms rel %.3e max|err| %.3e | MX form rms rel %.3e max|err| %.3e | max|ref| %.3e\n", N, sqrt(e_h / ref2), mx_h, sqrt(e_x / ref2), mx_x, mxr);
    unsigned long long t[16]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(kr_xstamps), sizeof(t)));
    printf("MX stamps, wave 0 of a mid-grid workgroup, group 6 (shader clocks): wait + barrier %lld | MFMA phase %lld (row block 0: %lld, row block 1: %lld) | DMA + B requests %lld | next wait + barrier %lld\n",
           (long long)(t[1] - t[0]), (long long)(t[4] - t[1]), (long long)(t[3] - t[2]), (long long)(t[4] - t[3]), (long long)(t[5] - t[4]), (long long)(t[6] - t[5]));
    size_t nan = 0; for (float v : h1) if (!(v == v)) nan++;
    printf("MX output NaNs: %zu\n", nan);
    return 0;
}
