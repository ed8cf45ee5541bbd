// Probe: where do the two launches of the chunked gated delta rule (kr_la_chunk.hip) spend their time?  QCN geometry (32 heads, 1024 tokens).
#define KR_TIMING 1
#include "../../krasis_amd/csrc/kr_la_chunk.hip"
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int nv = 32, C = argc > 1 ? atoi(argv[1]) : 1024;
    const size_t nq = (size_t)C * nv * 128;
    float *q, *k, *v, *ge, *be, *out, *state, *scr;
    CK(hipMalloc(&q, nq * 4)); CK(hipMalloc(&k, nq * 4)); CK(hipMalloc(&v, nq * 4)); CK(hipMalloc(&ge, (size_t)C * nv * 4)); CK(hipMalloc(&be, (size_t)C * nv * 4));
    CK(hipMalloc(&out, nq * 4)); CK(hipMalloc(&state, (size_t)nv * 128 * 128 * 4)); CK(hipMalloc(&scr, kr_pfm_la_chunk_scratch_floats(C, nv) * 4));
    std::vector<float> h(nq);
    for (auto& x : h) x = ((rand() % 2000) / 1000.f - 1.0f) * 0.088f;     // |row| ~ 1 at 128 elements
    CK(hipMemcpy(q, h.data(), nq * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(k, h.data(), nq * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v, h.data(), nq * 4, hipMemcpyHostToDevice));
    std::vector<float> g((size_t)C * nv); for (auto& x : g) x = 0.9f + 0.1f * (rand() % 1000) / 1000.f;
    CK(hipMemcpy(ge, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    for (auto& x : g) x = (rand() % 1000) / 1000.f;
    CK(hipMemcpy(be, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(state, 0, (size_t)nv * 128 * 128 * 4));
    KrPfmLaArgs p{}; p.q = q; p.k = k; p.v = v; p.gexp = ge; p.beta = be; p.nv = nv; p.nk = 16; p.dk = 128; p.dv = 128; p.hr = 2;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    if (lac_prepare()) { printf("prepare failed\n"); return 1; }
    KrLacArgs a{};
    a.q = q; a.k = k; a.v = v; a.gexp = ge; a.beta = be; a.nv = nv; a.C = C; a.n_sub = (C + 63) / 64;
    const size_t tiles = (size_t)a.n_sub * nv * 64;
    a.Y = scr; a.G = a.Y + tiles * 128; a.out = out; a.state = state;
    uint16_t* pl = reinterpret_cast<uint16_t*>(a.G + ((tiles + 3) / 4) * 4);
    a.Wh = pl; a.Wl = a.Wh + tiles * 128; a.Qh = a.Wl + tiles * 128; a.Ql = a.Qh + tiles * 128; a.Kh = a.Ql + tiles * 128; a.Kl = a.Kh + tiles * 128;
    // correctness first: one run from a zero state against the per-token recurrence (decode.rs:1293) in double on the host, a few heads
    {
        hipLaunchKernelGGL(kr_lac_prep_kernel, dim3(a.n_sub, nv), dim3(LC_PTH), LC_PREP_LDS, st, a);
        hipLaunchKernelGGL(kr_lac_scan_kernel, dim3(nv * 4), dim3(256), LC_SCAN_LDS, st, a);
        CK(hipStreamSynchronize(st));
        std::vector<float> ho(nq), hs((size_t)nv * 128 * 128), hg((size_t)C * nv), hb((size_t)C * nv);
        CK(hipMemcpy(ho.data(), out, nq * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), state, hs.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hg.data(), ge, hg.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), be, hb.size() * 4, hipMemcpyDeviceToHost));
        double eo = 0, mo = 0, es = 0, ms = 0;
        const int heads[3] = {0, 13, nv - 1};
        for (int hh : heads) {
            std::vector<double> S(128 * 128, 0.0), kv(128), dl(128);
            for (int t = 0; t < C; t++) {
                const float* row = h.data() + (size_t)t * nv * 128 + (size_t)hh * 128;      // q = k = v in this probe
                const double ga = hg[(size_t)t * nv + hh], b = hb[(size_t)t * nv + hh];
                for (auto& x : S) x *= ga;
                for (int j = 0; j < 128; j++) { double s2 = 0; for (int i = 0; i < 128; i++) s2 += S[i * 128 + j] * row[i]; kv[j] = s2; }
                for (int j = 0; j < 128; j++) dl[j] = (row[j] - kv[j]) * b;
                for (int i = 0; i < 128; i++) for (int j = 0; j < 128; j++) S[i * 128 + j] += (double)row[i] * dl[j];
                for (int j = 0; j < 128; j++) {
                    double o = 0; for (int i = 0; i < 128; i++) o += S[i * 128 + j] * row[i];
                    const double g2 = ho[(size_t)t * nv * 128 + (size_t)hh * 128 + j];
                    eo = fmax(eo, fabs(g2 - o)); mo = fmax(mo, fabs(o));
                }
            }
            for (int i = 0; i < 128 * 128; i++) { es = fmax(es, fabs(hs[(size_t)hh * 128 * 128 + i] - S[i])); ms = fmax(ms, fabs(S[i])); }
        }
        printf("check vs per-token recurrence (double, heads 0 / 13 / %d): out max|err| %.3e of max %.3e (rel %.2e), state max|err| %.3e of max %.3e (rel %.2e)\n",
               nv - 1, eo, mo, eo / mo, es, ms, es / ms);
        CK(hipMemset(state, 0, (size_t)nv * 128 * 128 * 4));
    }
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(kr_lac_prep_kernel, dim3(a.n_sub, nv), dim3(LC_PTH), LC_PREP_LDS, st, a);
        CK(hipEventRecord(e1, st));
        hipLaunchKernelGGL(kr_lac_scan_kernel, dim3(nv * 4), dim3(256), LC_SCAN_LDS, st, a);
        CK(hipEventRecord(e2, st)); CK(hipStreamSynchronize(st));
        float m0, m1; CK(hipEventElapsedTime(&m0, e0, e1)); CK(hipEventElapsedTime(&m1, e1, e2));
        unsigned long long s[64]; CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(kr_lstamps), sizeof(s)));
        auto d = [&](int x, int y) { return (double)(long long)(s[y] - s[x]) * 0.01; };
        printf("rep %d: prep %.1f us, scan %.1f us (%d sub-chunks)\n", rep, m0 * 1e3, m1 * 1e3, a.n_sub);
        printf("   prep wg(0,0): load+scan %.2f | KK^T,QK^T %.2f | block inverses + row scaling %.2f | block solve (3 products) %.2f | Q',O0 mfma %.2f | epilogue %.2f | total %.2f us\n",
               d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), d(0, 6));
        printf("   scan wg 0 step 1: lds write+barrier %.2f | fetch issue + y loads %.2f | mfma a %.2f | epilogue a %.2f | barrier %.2f | mfma b %.2f | barrier %.2f | S write %.2f | whole step 2 %.2f us\n",
               d(10, 11), d(11, 12), d(12, 13), d(13, 14), d(14, 15), d(15, 16), d(16, 17), d(17, 18), d(18, 19));
    }
    return 0;
}
