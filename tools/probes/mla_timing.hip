// Probe: phases of the staged MLA decode attention launch vs cache length (stamps: thread 0 of workgroup 0).
#define KR_TIMING 1
#include "../../krasis_amd/csrc/kr_mla.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    const int nh = 16, klr = 512, rd = 64, max_seq = 2048;
    std::vector<uint16_t> ck((size_t)max_seq * klr), kp((size_t)max_seq * rd);
    for (auto& x : ck) x = 0x3000 + (rand() & 0x7FF) + ((rand() & 1) << 15);
    for (auto& x : kp) x = 0x3000 + (rand() & 0x7FF) + ((rand() & 1) << 15);
    std::vector<float> qa(nh * klr), qp(nh * rd); for (auto& x : qa) x = (rand() % 1000) / 1000.f - 0.5f; for (auto& x : qp) x = (rand() % 1000) / 1000.f - 0.5f;
    void *dck, *dkp; float *dqa, *dqp, *dlat; KrStep* dstep;
    CK(hipMalloc(&dck, ck.size() * 2)); CK(hipMalloc(&dkp, kp.size() * 2)); CK(hipMalloc(&dqa, qa.size() * 4)); CK(hipMalloc(&dqp, qp.size() * 4)); CK(hipMalloc(&dlat, qa.size() * 4));
    CK(hipMalloc(&dstep, sizeof(KrStep)));
    CK(hipMemcpy(dck, ck.data(), ck.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dkp, kp.data(), kp.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dqa, qa.data(), qa.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dqp, qp.data(), qp.size() * 4, hipMemcpyHostToDevice));
    KrMlaArgs a{}; a.step = dstep; a.ckv_cache = dck; a.kpe_cache = dkp; a.kv_fp8 = 0; a.q_abs = dqa; a.q_pe = dqp; a.attn_lat = dlat; a.nh = nh; a.klr = klr; a.rd = rd; a.sm_scale = 0.07f;
    kr_mla_attn_prepare(a, max_seq);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = kr_mla_staged_lds(a, max_seq);
    for (int pos : {9, 63, 127, 511, 1023, 2047}) for (int rep = 0; rep < 2; rep++) {
        KrStep hs{}; hs.token = 0; hs.pos = pos; CK(hipMemcpy(dstep, &hs, sizeof(hs), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((kr_mla_attn_staged_kernel<false, 64, 8>), dim3(nh, 1), dim3(512), lds, st, a, max_seq);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long s[32]; CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(kr_mstamps), sizeof(s)));
        auto d = [&](int x, int y) { return (double)(long long)(s[y] - s[x]) * 0.01; };
        if (rep) printf("seq %4d: event %6.1f us | first load %.2f  scores (all stages, start of last) %.2f  last scores %.2f  barrier %.2f  max+exp %.2f  seqsum %.2f  wsum to last stage %.2f  last wsum %.2f | in-kernel %.2f\n",
                        pos + 1, ms * 1e3, d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), d(6, 7), d(7, 8), d(0, 8));
    }
    // ---- long cache: head-shared scores launch + softmax / weighted sum per head (old PHASE 2 vs the producer / consumer kernel)
    {
        const int ms2 = 8192;
        std::vector<uint16_t> c2((size_t)ms2 * klr), k2((size_t)ms2 * rd);
        for (auto& x : c2) x = 0x3000 + (rand() & 0x7FF) + ((rand() & 1) << 15);
        for (auto& x : k2) x = 0x3000 + (rand() & 0x7FF) + ((rand() & 1) << 15);
        void *dc2, *dk2; float* dsc;
        CK(hipMalloc(&dc2, c2.size() * 2)); CK(hipMalloc(&dk2, k2.size() * 2)); CK(hipMalloc(&dsc, (size_t)nh * ms2 * 4));
        CK(hipMemcpy(dc2, c2.data(), c2.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dk2, k2.data(), k2.size() * 2, hipMemcpyHostToDevice));
        a.ckv_cache = dc2; a.kpe_cache = dk2; a.sc_g = dsc;
        kr_mla_attn_prepare(a, ms2);
        const size_t lds_sc = (size_t)KR_MLA_HG * (klr + rd) * 4 + (size_t)KR_MLA_ROWS * ((size_t)(klr + rd) * 2 + 16);
        hipEvent_t e2; CK(hipEventCreate(&e2));
        const size_t lds_pv = kr_mla_pv_lds<64, false>();
        for (int pos : {1023, 4095, 8190}) for (int variant = 1; variant < 2; variant++) for (int rep = 0; rep < 2; rep++) {
            KrStep hs{}; hs.token = 0; hs.pos = pos; CK(hipMemcpy(dstep, &hs, sizeof(hs), hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL((kr_mla_scores_kernel<false, 64, 8>), dim3((ms2 + KR_MLA_ROWS - 1) / KR_MLA_ROWS, (nh + KR_MLA_HG - 1) / KR_MLA_HG), dim3(512), lds_sc, st, a, ms2);
            CK(hipEventRecord(e1, st));
            hipLaunchKernelGGL((kr_mla_pv_kernel<64, false>), dim3(nh), dim3(512 / KR_MPV_EPT + 256), lds_pv, st, a, ms2);
            CK(hipEventRecord(e2, st)); CK(hipStreamSynchronize(st));
            float m1, m2; CK(hipEventElapsedTime(&m1, e0, e1)); CK(hipEventElapsedTime(&m2, e1, e2));
            if (rep) printf("seq %4d split: scores launch %6.1f us | softmax + weighted sum (%s) %6.1f us\n", pos + 1, m1 * 1e3, variant ? "producer/consumer" : "staged PHASE 2", m2 * 1e3);
        }
    }
    return 0;
}
