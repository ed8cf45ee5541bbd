// Probe: where does the decode GQA attention launch spend its time, as a function of the sequence length?
// Includes the product source with KR_TIMING stamps (thread 0 of workgroup 0).
#define KR_TIMING 1
#include "../../krasis_amd/csrc/kr_decode_ops.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    const int nh = 16, nkv = 2, hd = 256, max_seq = 512;
    const size_t nkvb = (size_t)max_seq * nkv * hd;
    std::vector<uint16_t> kc(nkvb), vc(nkvb);
    for (size_t i = 0; i < nkvb; i++) { kc[i] = 0x3000 + (rand() & 0x7FF) + ((rand() & 1) << 15); vc[i] = 0x3000 + (rand() & 0x7FF) + ((rand() & 1) << 15); }
    std::vector<float> q(nh * hd), gate(nh * hd);
    for (auto& x : q) x = (rand() % 1000) / 1000.f - 0.5f;
    for (auto& x : gate) x = (rand() % 1000) / 1000.f - 0.5f;
    void *dk, *dv; float *dq, *dg, *dout; KrStep* dstep;
    CK(hipMalloc(&dk, nkvb * 2)); CK(hipMalloc(&dv, nkvb * 2)); CK(hipMalloc(&dq, q.size() * 4)); CK(hipMalloc(&dg, q.size() * 4)); CK(hipMalloc(&dout, q.size() * 4));
    CK(hipMalloc(&dstep, sizeof(KrStep)));
    CK(hipMemcpy(dk, kc.data(), nkvb * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, vc.data(), nkvb * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, gate.data(), q.size() * 4, hipMemcpyHostToDevice));
    if (kr_gqa_attn_prepare(max_seq, hd, 0)) { printf("prepare failed\n"); return 1; }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    KrGqaArgs a{};
    a.step = dstep; a.k_cache = dk; a.v_cache = dv; a.kv_fp8 = 0; a.q_out = dq; a.gate = dg; a.attn_out = dout; a.gated = 1; a.nh = nh; a.nkv = nkv; a.hd = hd;
    a.sm_scale = 0.0625f; a.img_out = nullptr;
    const size_t lds = kr_gqa_attn_lds(max_seq, hd, 0);
    for (int pos : {9, 39, 79, 119, 127, 128, 199, 249, 400, 511}) for (int rep = 0; rep < 2; rep++) {
        KrStep hs{}; hs.token = 0; hs.pos = pos;
        CK(hipMemcpy(dstep, &hs, sizeof(hs), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((kr_gqa_attn_kernel<false, 32, 0>), dim3(nh), dim3(256), lds, st, a, max_seq, max_seq);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long s[32]; CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(kr_dstamps), sizeof(s)));
        auto d = [&](int x, int y) { return (double)(long long)(s[y] - s[x]) * 0.01; };
        if (rep) printf("seq %3d: event %5.1f us | K latency %.2f q->regs+commit %.2f scores(last stage) %.2f max+exp %.2f seqsum %.2f scale+V commit %.2f pv(last stage) %.2f | in-kernel %.2f us\n",
                        pos + 1, ms * 1e3, d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), d(6, 7), d(0, 7));
    }
    // ---- long cache: scores launch (PHASE 1) + softmax / p.v per head (PHASE 2)
    {
        const int ms2 = 8192; const size_t n2 = (size_t)ms2 * nkv * hd;
        std::vector<uint16_t> k2(n2), v2(n2);
        for (size_t i = 0; i < n2; i++) { k2[i] = 0x3000 + (rand() & 0x7FF) + ((rand() & 1) << 15); v2[i] = 0x3000 + (rand() & 0x7FF) + ((rand() & 1) << 15); }
        void *dk2, *dv2; float* dsc;
        CK(hipMalloc(&dk2, n2 * 2)); CK(hipMalloc(&dv2, n2 * 2)); CK(hipMalloc(&dsc, (size_t)nh * ms2 * 4));
        CK(hipMemcpy(dk2, k2.data(), n2 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dv2, v2.data(), n2 * 2, hipMemcpyHostToDevice));
        if (kr_gqa_attn_prepare(ms2, hd, 0)) { printf("prepare failed\n"); return 1; }
        a.k_cache = dk2; a.v_cache = dv2; a.sc_g = dsc;
        const size_t lds2 = kr_gqa_attn_lds(ms2, hd, 0), lds1 = kr_gqa_attn_lds(0, hd, 0);
        for (int pos : {1023, 4095, 8190}) for (int rep = 0; rep < 2; rep++) {
            KrStep hs{}; hs.token = 0; hs.pos = pos;
            CK(hipMemcpy(dstep, &hs, sizeof(hs), hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL((kr_gqa_attn_kernel<false, 32, 1>), dim3(nh, ms2 / 256), dim3(256), lds1, st, a, ms2, 0);
            CK(hipEventRecord(e1, st));
            hipEvent_t e2; CK(hipEventCreate(&e2));
            hipLaunchKernelGGL((kr_gqa_pv_kernel<32, false, false>), dim3(nh), dim3(512), kr_gqa_pv_lds(ms2, hd, 0), st, a, ms2, ms2);
            CK(hipEventRecord(e2, st)); CK(hipStreamSynchronize(st));
            float m1, m2; CK(hipEventElapsedTime(&m1, e0, e1)); CK(hipEventElapsedTime(&m2, e1, e2));
            unsigned long long s[32]; CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(kr_dstamps), sizeof(s)));
            auto d = [&](int x, int y) { return (double)(long long)(s[y] - s[x]) * 0.01; };
            if (rep) printf("seq %4d split: scores launch %6.1f us | softmax+pv launch %6.1f us: load scores %.1f max+exp %.1f seqsum %.1f scale + p.v (all stages) %.1f\n",
                            pos + 1, m1 * 1e3, m2 * 1e3, d(0, 3), d(3, 4), d(4, 5), d(5, 7));
        }
    }
    // ---- the same long cache as FP8-E4M3 bytes: scores launch + producer / consumer softmax + p.v
    {
        const int ms2 = 8192; const size_t n2 = (size_t)ms2 * nkv * hd;
        std::vector<uint8_t> k8(n2), v8(n2);
        for (size_t i = 0; i < n2; i++) { k8[i] = (uint8_t)(0x28 + (rand() & 0x1F) + ((rand() & 1) << 7)); v8[i] = (uint8_t)(0x28 + (rand() & 0x1F) + ((rand() & 1) << 7)); }
        void *dk8, *dv8; float* dsc;
        CK(hipMalloc(&dk8, n2)); CK(hipMalloc(&dv8, n2)); CK(hipMalloc(&dsc, (size_t)nh * ms2 * 4));
        CK(hipMemcpy(dk8, k8.data(), n2, hipMemcpyHostToDevice)); CK(hipMemcpy(dv8, v8.data(), n2, hipMemcpyHostToDevice));
        if (kr_gqa_attn_prepare(ms2, hd, 1)) { printf("prepare failed\n"); return 1; }
        a.k_cache = dk8; a.v_cache = dv8; a.sc_g = dsc; a.kv_fp8 = 1;
        const size_t lds2 = kr_gqa_attn_lds(ms2, hd, 1), lds1 = kr_gqa_attn_lds(0, hd, 1);
        hipEvent_t e2; CK(hipEventCreate(&e2));
        for (int pos : {1023, 8190}) for (int rep = 0; rep < 2; rep++) {
            KrStep hs{}; hs.token = 0; hs.pos = pos;
            CK(hipMemcpy(dstep, &hs, sizeof(hs), hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL((kr_gqa_attn_kernel<true, 32, 1>), dim3(nh, ms2 / 256), dim3(256), lds1, st, a, ms2, 0);
            CK(hipEventRecord(e1, st));
            hipLaunchKernelGGL((kr_gqa_pv_kernel<32, false, true>), dim3(nh), dim3(512), kr_gqa_pv_lds(ms2, hd, 1), st, a, ms2, ms2);
            CK(hipEventRecord(e2, st)); CK(hipStreamSynchronize(st));
            float m1, m2; CK(hipEventElapsedTime(&m1, e0, e1)); CK(hipEventElapsedTime(&m2, e1, e2));
            if (rep) printf("seq %4d FP8 split: scores launch %6.1f us | softmax+pv launch %6.1f us\n", pos + 1, m1 * 1e3, m2 * 1e3);
        }
    }
    return 0;
}
