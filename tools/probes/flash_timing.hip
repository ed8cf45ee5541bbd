// Probe: phases of one K/V tile of the prompt-pass flash attention kernel (QCN geometry: 16 heads, 2 KV heads, head_dim 256, E4M3 cache),
// 1024 query tokens late in a 32768-position cache.
#define KR_TIMING 1
#include "../../krasis_amd/csrc/kr_attn_flash.hip"
#include <cstdio>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int nh = 16, nkv = 2, hd = 256, C = 1024, seq = argc > 1 ? atoi(argv[1]) : 32768, pos0 = seq - C;
    const size_t cache = (size_t)seq * nkv * hd;
    unsigned char *kc, *vc; float *q, *out, *gate;
    CK(hipMalloc(&kc, cache)); CK(hipMalloc(&vc, cache)); CK(hipMalloc(&q, (size_t)C * nh * hd * 4)); CK(hipMalloc(&out, (size_t)C * nh * hd * 4)); CK(hipMalloc(&gate, (size_t)C * nh * hd * 4));
    std::vector<unsigned char> hb(cache); for (auto& x : hb) x = (unsigned char)(0x20 + rand() % 0x30) | ((rand() & 1) << 7);      // finite E4M3 codes
    CK(hipMemcpy(kc, hb.data(), cache, hipMemcpyHostToDevice)); CK(hipMemcpy(vc, hb.data(), cache, hipMemcpyHostToDevice));
    std::vector<float> hq((size_t)C * nh * hd); for (auto& x : hq) x = (rand() % 2000) / 1000.f - 1.0f;
    CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(gate, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    KrPfmGqaArgs a{};
    a.k_cache = kc; a.v_cache = vc; a.kv_fp8 = 1; a.q_out = q; a.gate = gate; a.attn_out = out; a.gated = 1; a.nh = nh; a.nkv = nkv; a.hd = hd; a.pos0 = pos0; a.sm_scale = 0.0625f;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto e4m3 = [](unsigned char b) -> double {      // OCP E4M3 (finite codes only here)
        const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
        const double v = e == 0 ? m / 8.0 * 0.015625 : (1.0 + m / 8.0) * std::pow(2.0, e - 7);
        return s ? -v : v;
    };
    {   // correctness: a few (token, head) rows against a double softmax over the decoded cache
        if (kr_launch_pfm_gqa_flash(a, C, st)) { printf("launch refused\n"); return 1; }
        CK(hipStreamSynchronize(st));
        std::vector<float> ho((size_t)C * nh * hd); CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
        double emax = 0, omax = 0;
        const int rows[4][2] = {{0, 0}, {C / 2 + 3, 5}, {C - 1, nh - 1}, {17, 9}};
        for (auto& rw : rows) {
            const int t = rw[0], h = rw[1], kvh2 = h / (nh / nkv), np = pos0 + t + 1;
            std::vector<double> sc(np); double mx = -1e300;
            for (int p = 0; p < np; p++) { double d2 = 0; for (int i = 0; i < hd; i++) d2 += (double)hq[((size_t)t * nh + h) * hd + i] * e4m3(hb[((size_t)p * nkv + kvh2) * hd + i]); sc[p] = d2 * 0.0625; mx = std::max(mx, sc[p]); }
            double l = 0; for (auto& x : sc) { x = std::exp(x - mx); l += x; }
            for (int i = 0; i < hd; i++) {
                double o = 0; for (int p = 0; p < np; p++) o += sc[p] * e4m3(hb[((size_t)p * nkv + kvh2) * hd + i]);
                o /= l; const double g = hq[((size_t)t * nh + h) * hd + i]; o *= 1.0 / (1.0 + std::exp(-g));
                emax = std::max(emax, std::fabs(o - (double)ho[((size_t)t * nh + h) * hd + i])); omax = std::max(omax, std::fabs(o));
            }
        }
        printf("check (4 rows, double softmax over the decoded E4M3 cache): max|err| %.3e of max|out| %.3e\n", emax, omax);
    }
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, st));
        if (kr_launch_pfm_gqa_flash(a, C, st)) { printf("launch refused\n"); return 1; }
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long s[32]; CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(kr_fstamps), sizeof(s)));
        auto d = [&](int x, int y) { return (double)(long long)(s[y] - s[x]) * 0.01; };
        const double flop = 4.0 * C * nh * (double)(pos0 + C / 2) * hd;
        printf("rep %d: %.1f us, %.0f TFLOP/s | S wave, tile 40: barrier X %.2f | S^T %.2f | softmax + P store %.2f | barrier Y %.2f | K commit + loads %.2f | tile %.2f us\n",
               rep, ms * 1e3, flop / (ms * 1e-3) / 1e12, d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(0, 5));
        printf("        PV wave, tile 40: barrier X %.2f | fragment requests + V commit + loads %.2f | rescale + P.V %.2f | barrier Y %.2f | tile %.2f us\n", d(8, 9), d(9, 10), d(10, 11), d(11, 12), d(8, 12));
    }
    return 0;
}
