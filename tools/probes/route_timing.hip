// Probe: where does the fused norm+router launch spend its time?  Includes the product source with KR_TIMING stamps.
#define KR_TIMING 1
#include "../../krasis_amd/csrc/kr_router.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    const int E = 512, H = 2048, k = 10;
    std::vector<uint16_t> gate((size_t)E * H); for (auto& g : gate) g = (uint16_t)(0x3C00 + (rand() & 0x3FF) - ((rand() & 1) << 15));
    std::vector<float> hid(H), res(H), nw(H); for (int i = 0; i < H; i++) { hid[i] = (rand() % 1000) / 1000.f - 0.5f; res[i] = (rand() % 1000) / 1000.f - 0.5f; nw[i] = 0.1f; }
    void *dg; float *dh, *dr, *dn, *dh2, *dr2, *dl, *dw; int32_t* di; unsigned* dc;
    CK(hipMalloc(&dg, gate.size() * 2)); CK(hipMemcpy(dg, gate.data(), gate.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dh, H * 4)); CK(hipMalloc(&dr, H * 4)); CK(hipMalloc(&dn, H * 4)); CK(hipMalloc(&dh2, H * 4)); CK(hipMalloc(&dr2, H * 4));
    CK(hipMalloc(&dl, E * 4)); CK(hipMalloc(&dw, 256)); CK(hipMalloc(&di, 256)); CK(hipMalloc(&dc, 64)); CK(hipMemset(dc, 0, 64));
    CK(hipMemcpy(dh, hid.data(), H * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, res.data(), H * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, nw.data(), H * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int scoring = 0; scoring < 2; scoring++) for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, st));
        kr_launch_route_fused_decode(dg, 1, nullptr, dl, dc, nullptr, di, dw, E, H, k, scoring, 1, nullptr, dh, dr, dn, dh2, dr2, 1e-6f, 1, st);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long s[32]; CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(kr_stamps), sizeof(s)));
        auto d = [&](int a, int b) { return (double)(long long)(s[b] - s[a]) * 0.01; };
        printf("scoring %d rep %d: event %.1f us | last-block: norm %.2f logits %.2f sync %.2f | select: fence %.2f score %.2f (seqsum %.2f) topk %.2f tail %.2f | total-in-block %.2f us\n",
               scoring, rep, ms * 1e3, d(0, 1), d(1, 2), d(2, 3), d(3, 8), d(8, 9), scoring ? d(12, 13) : 0.0, d(9, 10), d(10, 11), d(0, 11));
    }
    // standalone select for comparison
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, st));
        kr_launch_route_select(dl, nullptr, di, dw, 1, E, k, 1, 1, 1, 0, st);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long s[32]; CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(kr_stamps), sizeof(s)));
        auto d = [&](int a, int b) { return (double)(long long)(s[b] - s[a]) * 0.01; };
        printf("standalone select rep %d: event %.1f us | score %.2f (seqsum %.2f) topk %.2f\n", rep, ms * 1e3, d(8, 9), d(12, 13), d(9, 10));
    }
    return 0;
}
