// kr_marlin.cpp -- import / export of the reference's Marlin GPU weight layout.
//
// The reference keeps a second copy of every expert in the CUDA tensor-core layout of sglang's fused_marlin_moe: written by
// marlin_repack / marlin_repack_int8 (src/weights/marlin.rs:256-491, :587-758), cached on disk in that form (src/weights/mod.rs:856-934),
// exported through KrasisEngine.get_expert_w13_packed / get_expert_w13_scales / ... (src/moe.rs:1972-2709) and undone in Python by
// inverse_marlin_repack / inverse_scale_permute (python/krasis/triton_moe.py:71-170).  A CDNA kernel has no use for that layout: this
// file only converts it -- so a caller that holds Marlin-packed experts (the reference's disk cache, its GPU store) can hand them over,
// and a caller that expects them (get_expert_*) can have them back.  The permutation is built ONCE as an index table
// (source element of every destination element inside one [16 x N] k-tile row) and applied in both directions.
//
//   row-major   packed [N, K/8] u32 (nibble j of a word = k 8c+j, value q+8)  |  data [N, K] i8      scales bf16 [N, K/gs]
//   Marlin      packed [K/16, 2N] u32                                          |  [K/16, 4N] u32      scales bf16 [K/gs, N] (64-permuted)
#include <cstdint>
#include <cstring>
#include <vector>

#include "kr_engine_internal.h"

namespace {
constexpr int TILE = 16;

// destination -> source inside one 1024-element chunk (marlin.rs:256-298 INT4, :587-631 INT8)
void weight_perm(int bits, int (&out)[1024]) {
    int perm[1024]; int idx = 0;
    for (int i = 0; i < 32; i++) {
        const int col = i / 4; int perm1[8]; int p = 0;
        for (int block = 0; block < 2; block++)
            for (int row : {2 * (i % 4), 2 * (i % 4) + 1, 2 * (i % 4 + 4), 2 * (i % 4 + 4) + 1}) perm1[p++] = 16 * row + col + 8 * block;
        for (int j = 0; j < 4; j++) for (int q = 0; q < 8; q++) perm[idx++] = perm1[q] + 256 * j;
    }
    if (bits == 4) { const int il[8] = {0, 2, 4, 6, 1, 3, 5, 7}; for (int g = 0; g < 128; g++) for (int d = 0; d < 8; d++) out[g * 8 + d] = perm[g * 8 + il[d]]; }
    else { const int il[4] = {0, 2, 1, 3}; for (int g = 0; g < 256; g++) for (int d = 0; d < 4; d++) out[g * 4 + d] = perm[g * 4 + il[d]]; }
}

// scale permutation (marlin.rs:302-323): 64-element chunks when grouped, 32 when channelwise
void scale_perm(bool grouped, std::vector<int>& sp) {
    if (grouped) { sp.resize(64); for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) sp[i * 8 + j] = i + 8 * j; }
    else { sp.resize(32); const int off[8] = {0, 1, 8, 9, 16, 17, 24, 25}; for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) sp[i * 8 + j] = 2 * i + off[j]; }
}

int check_dims(int N, int K, int gs, int bits) {
    if (bits != 4 && bits != 8) return kr_fail(KR_ERR_VALUE, "Unsupported num_bits: %d", bits);
    if (K <= 0 || K % TILE) return kr_fail(KR_ERR_VALUE, "K (%d) must be divisible by 16", K);           // marlin.rs:336
    if (N <= 0 || N % 64) return kr_fail(KR_ERR_VALUE, "N (%d) must be divisible by 64 (Marlin tile constraint)", N);   // marlin.rs:337
    if (gs <= 0 || K % gs) return kr_fail(KR_ERR_VALUE, "K (%d) must be divisible by group_size (%d)", K, gs);
    return KR_OK;
}

// unsigned element (row n, k) of a row-major matrix: nibble (INT4) or q + 128 (INT8)
inline uint8_t rm_get(const void* rm, int bits, int K, int n, int k) {
    if (bits == 4) return (uint8_t)((((const uint32_t*)rm)[(size_t)n * (K / 8) + k / 8] >> ((k & 7) * 4)) & 0xF);
    return (uint8_t)((int)((const int8_t*)rm)[(size_t)n * K + k] + 128);
}
}  // namespace

extern "C" int kr_marlin_repack(const void* rowmajor, const uint16_t* scales, int N, int K, int group_size, int bits, uint32_t* out_packed,
                                uint16_t* out_scales) {
    if (!rowmajor || !scales || !out_packed || !out_scales) return kr_fail(KR_ERR_VALUE, "null pointer argument");
    if (int rc = check_dims(N, K, group_size, bits)) return rc;
    int perm[1024]; weight_perm(bits, perm);
    const int kt_n = K / TILE, row_len = N * TILE, pf = 32 / bits, out_cols = row_len / pf;
    std::vector<uint8_t> row(row_len);
    for (int kt = 0; kt < kt_n; kt++) {
        // destination element d of the k-tile row <- tiled position perm[d]: (nt, tk, tn) = original (n = nt*16+tn, k = kt*16+tk)
        for (int chunk = 0; chunk < row_len / 1024; chunk++)
            for (int i = 0; i < 1024; i++) {
                const int src = chunk * 1024 + perm[i], nt = src / 256, tk = (src % 256) / 16, tn = src % 16;
                row[chunk * 1024 + i] = rm_get(rowmajor, bits, K, nt * 16 + tn, kt * TILE + tk);
            }
        for (int c = 0; c < out_cols; c++) {
            uint32_t w = 0;
            for (int i = 0; i < pf; i++) w |= (uint32_t)row[c * pf + i] << (i * bits);
            out_packed[(size_t)kt * out_cols + c] = w;
        }
    }
    const int ng = K / group_size; std::vector<int> sp; scale_perm(group_size < K, sp);
    const size_t total = (size_t)ng * N; const int pl = (int)sp.size();
    std::vector<uint16_t> tr(total);
    for (int n = 0; n < N; n++) for (int g = 0; g < ng; g++) tr[(size_t)g * N + n] = scales[(size_t)n * ng + g];
    for (size_t base = 0; base + pl <= total; base += pl) for (int i = 0; i < pl; i++) out_scales[base + i] = tr[base + sp[i]];
    return KR_OK;
}

extern "C" int kr_marlin_unpack(const uint32_t* marlin_packed, const uint16_t* marlin_scales, int N, int K, int group_size, int bits,
                                void* out_rowmajor, uint16_t* out_scales) {
    if (!marlin_packed || !marlin_scales || !out_rowmajor || !out_scales) return kr_fail(KR_ERR_VALUE, "null pointer argument");
    if (int rc = check_dims(N, K, group_size, bits)) return rc;
    int perm[1024]; weight_perm(bits, perm);
    const int kt_n = K / TILE, row_len = N * TILE, pf = 32 / bits, out_cols = row_len / pf;
    if (bits == 4) memset(out_rowmajor, 0, (size_t)N * (K / 8) * 4);
    const uint32_t mask = bits == 4 ? 0xFu : 0xFFu;
    for (int kt = 0; kt < kt_n; kt++)
        for (int chunk = 0; chunk < row_len / 1024; chunk++)
            for (int i = 0; i < 1024; i++) {
                const int d = chunk * 1024 + i;
                const uint32_t v = (marlin_packed[(size_t)kt * out_cols + d / pf] >> ((d % pf) * bits)) & mask;
                const int src = chunk * 1024 + perm[i], nt = src / 256, tk = (src % 256) / 16, tn = src % 16;
                const int n = nt * 16 + tn, k = kt * TILE + tk;
                if (bits == 4) ((uint32_t*)out_rowmajor)[(size_t)n * (K / 8) + k / 8] |= v << ((k & 7) * 4);
                else ((int8_t*)out_rowmajor)[(size_t)n * K + k] = (int8_t)((int)v - 128);
            }
    const int ng = K / group_size; std::vector<int> sp; scale_perm(group_size < K, sp);
    const size_t total = (size_t)ng * N; const int pl = (int)sp.size();
    std::vector<uint16_t> tr(total);
    for (size_t base = 0; base + pl <= total; base += pl) for (int i = 0; i < pl; i++) tr[base + sp[i]] = marlin_scales[base + i];
    for (int n = 0; n < N; n++) for (int g = 0; g < ng; g++) out_scales[(size_t)n * ng + g] = tr[(size_t)g * N + n];
    return KR_OK;
}

// weights/mod.rs:942-949: w2's N is padded by 64 when hidden == intermediate and hidden % 256 != 0
static int marlin_w2_padded_n(int hidden, int inter) { return (hidden == inter && hidden % 256 != 0) ? hidden + 64 : hidden; }

// row-major [N, K/8] u32 | [N, K] i8 + scales [N, K/gs]  ->  the CPU transposed layout kr_upload_expert_unified takes ([K/8, N] | [K, N], [K/gs, N])
static void to_transposed(const void* rm, const uint16_t* sc, int N, int K, int gs, int bits, int n_keep, std::vector<uint8_t>& wt, std::vector<uint16_t>& st) {
    const int ng = K / gs;
    st.resize((size_t)ng * n_keep);
    for (int n = 0; n < n_keep; n++) for (int g = 0; g < ng; g++) st[(size_t)g * n_keep + n] = sc[(size_t)n * ng + g];
    if (bits == 4) {
        wt.resize((size_t)(K / 8) * n_keep * 4); uint32_t* o = (uint32_t*)wt.data(); const uint32_t* s = (const uint32_t*)rm;
        for (int n = 0; n < n_keep; n++) for (int c = 0; c < K / 8; c++) o[(size_t)c * n_keep + n] = s[(size_t)n * (K / 8) + c];
    } else {
        wt.resize((size_t)K * n_keep); int8_t* o = (int8_t*)wt.data(); const int8_t* s = (const int8_t*)rm;
        for (int n = 0; n < n_keep; n++) for (int k = 0; k < K; k++) o[(size_t)k * n_keep + n] = s[(size_t)n * K + k];
    }
    (void)N;
}
static void from_transposed(const void* wt, const uint16_t* st, int N, int K, int gs, int bits, int n_have, std::vector<uint8_t>& rm, std::vector<uint16_t>& sc) {
    const int ng = K / gs;
    sc.assign((size_t)N * ng, 0);
    for (int n = 0; n < n_have; n++) for (int g = 0; g < ng; g++) sc[(size_t)n * ng + g] = st[(size_t)g * n_have + n];
    if (bits == 4) {
        rm.assign((size_t)N * (K / 8) * 4, 0); uint32_t* o = (uint32_t*)rm.data(); const uint32_t* s = (const uint32_t*)wt;
        for (int n = 0; n < n_have; n++) for (int c = 0; c < K / 8; c++) o[(size_t)n * (K / 8) + c] = s[(size_t)c * n_have + n];
    } else {
        rm.assign((size_t)N * K, 0); int8_t* o = (int8_t*)rm.data(); const int8_t* s = (const int8_t*)wt;
        for (int n = 0; n < n_have; n++) for (int k = 0; k < K; k++) o[(size_t)n * K + k] = s[(size_t)k * n_have + n];
    }
}

// One expert in the reference's Marlin GPU format (UnifiedExpertWeights::from_expert_weights_marlin_int4/_int8, weights/mod.rs:506-640):
// w13 = marlin_repack(gate rows || up rows) [H/16, (2|4) * 2I], w2 = marlin_repack(down, N padded per marlin_w2_padded_n).
extern "C" int kr_upload_expert_marlin(kr_engine* e, int layer, int expert, int inter, const uint32_t* w13_packed, const uint16_t* w13_scales,
                                       const uint32_t* w2_packed, const uint16_t* w2_scales, int bits) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (!w13_packed || !w13_scales || !w2_packed || !w2_scales) return kr_fail(KR_ERR_VALUE, "null weight pointer");
    const int H = e->cfg.hidden_size, gs = e->cfg.group_size, N2 = marlin_w2_padded_n(H, inter);
    const size_t eb = bits == 4 ? 4 : 1;
    std::vector<uint8_t> rm13((size_t)2 * inter * (bits == 4 ? H / 8 : H) * eb), rm2((size_t)N2 * (bits == 4 ? inter / 8 : inter) * eb), t13, t2;
    std::vector<uint16_t> s13((size_t)2 * inter * (H / gs)), s2((size_t)N2 * (inter / gs)), ts13, ts2;
    if (int rc = kr_marlin_unpack(w13_packed, w13_scales, 2 * inter, H, gs, bits, rm13.data(), s13.data())) return rc;
    if (int rc = kr_marlin_unpack(w2_packed, w2_scales, N2, inter, gs, bits, rm2.data(), s2.data())) return rc;
    to_transposed(rm13.data(), s13.data(), 2 * inter, H, gs, bits, 2 * inter, t13, ts13);
    to_transposed(rm2.data(), s2.data(), N2, inter, gs, bits, H, t2, ts2);              // the padding rows are dropped (gpu_prefill.py:228-230)
    return kr_upload_expert_unified(e, layer, expert, inter, t13.data(), ts13.data(), bits, t2.data(), ts2.data(), bits);
}

// KrasisEngine.get_expert_w13_packed / _scales / get_expert_w2_packed / _scales (moe.rs:1972-2090) for one expert
extern "C" int kr_download_expert_marlin(kr_engine* e, int layer, int expert, uint32_t* w13_packed, uint16_t* w13_scales, uint32_t* w2_packed,
                                         uint16_t* w2_scales) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (layer < 0 || layer >= (int)e->layers.size()) return kr_fail(KR_ERR_VALUE, "moe_layer_idx %d out of range", layer);
    Layer& L = e->layers[layer];
    MatSet& a = expert == -1 ? L.sw13 : L.w13; MatSet& b = expert == -1 ? L.sw2 : L.w2;
    if (!a.allocated()) return kr_fail(KR_ERR_STATE, "expert %d of layer %d not loaded", expert, layer);
    if (a.bits != b.bits) return kr_fail(KR_ERR_VALUE, "Marlin export needs one weight width per expert (w13 %d-bit, w2 %d-bit)", a.bits, b.bits);
    const int H = e->cfg.hidden_size, gs = e->cfg.group_size, bits = a.bits, inter = b.K, N2 = marlin_w2_padded_n(H, inter);
    std::vector<uint8_t> t13((size_t)(bits == 4 ? H / 8 * 4 : H) * 2 * inter), t2((size_t)(bits == 4 ? inter / 8 * 4 : inter) * H), rm13, rm2;
    std::vector<uint16_t> ts13((size_t)(H / gs) * 2 * inter), ts2((size_t)(inter / gs) * H), s13, s2;
    if (int rc = kr_download_expert_unified(e, layer, expert, t13.data(), ts13.data(), t2.data(), ts2.data())) return rc;
    from_transposed(t13.data(), ts13.data(), 2 * inter, H, gs, bits, 2 * inter, rm13, s13);
    from_transposed(t2.data(), ts2.data(), N2, inter, gs, bits, H, rm2, s2);            // padding rows: zero words / zero scales (weights/mod.rs:541-547)
    if (int rc = kr_marlin_repack(rm13.data(), s13.data(), 2 * inter, H, gs, bits, w13_packed, w13_scales)) return rc;
    return kr_marlin_repack(rm2.data(), s2.data(), N2, inter, gs, bits, w2_packed, w2_scales);
}
