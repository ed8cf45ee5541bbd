// Probe: where does the fused add + RMSNorm launch (MoE-epilogue gather form) spend its time?  Includes the product source with stamps.
#define KR_TIMING 1
#include "../../krasis_amd/csrc/kr_decode_ops.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    const int H = 2048, k = 10;
    float *eo, *hid, *res, *res2, *w, *wts, *gv; int* ids; void* img;
    CK(hipMalloc(&eo, (size_t)(k + 1) * H * 4)); CK(hipMalloc(&hid, H * 4)); CK(hipMalloc(&res, H * 4)); CK(hipMalloc(&res2, H * 4)); CK(hipMalloc(&w, H * 4));
    CK(hipMalloc(&wts, 64 * 4)); CK(hipMalloc(&gv, 64)); CK(hipMalloc(&ids, 64 * 4)); CK(hipMalloc(&img, 16384));
    std::vector<float> h((size_t)(k + 1) * H); for (auto& x : h) x = (rand() % 1000) / 1000.f - 0.5f;
    CK(hipMemcpy(eo, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(res, h.data(), H * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, h.data(), H * 4, hipMemcpyHostToDevice));
    std::vector<float> wv(64, 0.1f); std::vector<int> iv(64, 3);
    CK(hipMemcpy(wts, wv.data(), 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ids, iv.data(), 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(gv, wv.data(), 64, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    KrNormSrc src{}; src.mode = 2; src.eo = eo; src.ids = ids; src.wts = wts; src.topk = k; src.has_shared = 1; src.gate_val = gv; src.rsf = 1.0f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0, st));
        kr_launch_fused_add_rmsnorm(src, hid, res, res2, w, H, 1e-6f, 0, 1, st, img);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long s[32]; CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(kr_dstamps), sizeof(s)));
        auto d = [&](int x, int y) { return (double)(long long)(s[y] - s[x]) * 0.01; };
        printf("rep %d: event %.1f us | gather+store %.2f  barrier %.2f  chain %.2f  barrier+scale+write %.2f  image %.2f | in-kernel %.2f us\n", rep, ms * 1e3,
               d(10, 11), d(11, 12), d(12, 13), d(13, 14), d(14, 15), d(10, 15));
    }
    return 0;
}
