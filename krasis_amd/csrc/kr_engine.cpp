// kr_engine.cpp -- host side of libkrasis_hip.so: engine state, weight store in HBM, re-tiling of the
// reference's weight layouts, and the extern "C" entry points declared in include/krasis_hip.h.
//
// Mirrors the role of the reference's Rust host for this path: KrasisEngine (src/moe.rs:1377-3296) and
// WeightStore (src/weights/mod.rs:814-850).  With 288 GB of HBM every expert of every layer is resident;
// the reference's CPU/GPU split, expert DMA, LRU/HCS caches (python/krasis/gpu_prefill.py) do not exist here.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "kr_engine_internal.h"
#include "kr_router.h"
#include "kr_prefill.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
int kr_fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char* kr_last_error(void) { return g_err.c_str(); }
extern "C" int kr_version(void) { return 1; }
extern "C" long kr_alloc_count_total(void) { return kr_alloc_count().load(std::memory_order_relaxed); }

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// ------------------------------------------------------------------------------------------------
// re-tiling between the reference CPU layouts and the lane-tiled HBM layout
// ------------------------------------------------------------------------------------------------
// INT4: src packed [K/8, N] u32, scales [K/128, N] bf16 (weights/mod.rs:329-397)
static void retile_int4(const uint32_t* src, const uint16_t* sc, int K, int N, uint32_t* dq, uint32_t* ds) {
    const int ng = K / 128, ngp = (ng + 1) / 2, nt = (N + 7) / 8;
    for (int t = 0; t < nt; t++)
        for (int gp = 0; gp < ngp; gp++) {
            uint32_t* rec = dq + ((size_t)t * ngp + gp) * 64 * 4;
            for (int c = 0; c < 8; c++) {
                const int col = t * 8 + c;
                const int g0 = 2 * gp, g1 = g0 + 1;
                for (int l = 0; l < 8; l++) {
                    uint32_t* o = rec + (c * 8 + l) * 4;
                    for (int h = 0; h < 2; h++) {
                        const int g = h == 0 ? g0 : g1;
                        for (int i = 0; i < 2; i++)
                            o[h * 2 + i] = (g < ng && col < N) ? src[(size_t)(g * 16 + 2 * l + i) * N + col] : 0x88888888u;
                    }
                }
                const uint32_t s0 = (col < N) ? sc[(size_t)g0 * N + col] : 0;
                const uint32_t s1 = (g1 < ng && col < N) ? sc[(size_t)g1 * N + col] : 0;
                ds[((size_t)t * ngp + gp) * 8 + c] = s0 | (s1 << 16);
            }
        }
}
static void untile_int4(const uint32_t* dq, const uint32_t* ds, int K, int N, uint32_t* dst, uint16_t* sc) {
    const int ng = K / 128, ngp = (ng + 1) / 2, nt = (N + 7) / 8;
    for (int t = 0; t < nt; t++)
        for (int gp = 0; gp < ngp; gp++)
            for (int c = 0; c < 8; c++) {
                const int col = t * 8 + c;
                if (col >= N) continue;
                for (int l = 0; l < 8; l++)
                    for (int h = 0; h < 2; h++) {
                        const int g = 2 * gp + h;
                        if (g >= ng) continue;
                        for (int i = 0; i < 2; i++)
                            dst[(size_t)(g * 16 + 2 * l + i) * N + col] = dq[(((size_t)t * ngp + gp) * 64 + c * 8 + l) * 4 + h * 2 + i];
                    }
                const uint32_t sp = ds[((size_t)t * ngp + gp) * 8 + c];
                sc[(size_t)(2 * gp) * N + col] = (uint16_t)(sp & 0xFFFF);
                if (2 * gp + 1 < ng) sc[(size_t)(2 * gp + 1) * N + col] = (uint16_t)(sp >> 16);
            }
}
// INT8: src [K, N] i8, scales [K/128, N] bf16 (weights/mod.rs:403-470)
static void retile_int8(const int8_t* src, const uint16_t* sc, int K, int N, uint32_t* dq, uint32_t* ds) {
    const int ng = K / 128, ngp = (ng + 1) / 2, nt = (N + 7) / 8;
    for (int t = 0; t < nt; t++) {
        for (int g = 0; g < ng; g++) {
            uint8_t* rec = reinterpret_cast<uint8_t*>(dq) + ((size_t)t * ng + g) * 64 * 16;
            for (int c = 0; c < 8; c++) {
                const int col = t * 8 + c;
                for (int l = 0; l < 8; l++)
                    for (int j = 0; j < 16; j++)
                        rec[(c * 8 + l) * 16 + j] = (col < N) ? (uint8_t)src[(size_t)(g * 128 + 16 * l + j) * N + col] : 0;
            }
        }
        for (int gp = 0; gp < ngp; gp++)
            for (int c = 0; c < 8; c++) {
                const int col = t * 8 + c;
                const uint32_t s0 = (col < N) ? sc[(size_t)(2 * gp) * N + col] : 0;
                const uint32_t s1 = (2 * gp + 1 < ng && col < N) ? sc[(size_t)(2 * gp + 1) * N + col] : 0;
                ds[((size_t)t * ngp + gp) * 8 + c] = s0 | (s1 << 16);
            }
    }
}
static void untile_int8(const uint32_t* dq, const uint32_t* ds, int K, int N, int8_t* dst, uint16_t* sc) {
    const int ng = K / 128, ngp = (ng + 1) / 2, nt = (N + 7) / 8;
    for (int t = 0; t < nt; t++) {
        for (int g = 0; g < ng; g++) {
            const uint8_t* rec = reinterpret_cast<const uint8_t*>(dq) + ((size_t)t * ng + g) * 64 * 16;
            for (int c = 0; c < 8; c++) {
                const int col = t * 8 + c;
                if (col >= N) continue;
                for (int l = 0; l < 8; l++)
                    for (int j = 0; j < 16; j++) dst[(size_t)(g * 128 + 16 * l + j) * N + col] = (int8_t)rec[(c * 8 + l) * 16 + j];
            }
        }
        for (int gp = 0; gp < ngp; gp++)
            for (int c = 0; c < 8; c++) {
                const int col = t * 8 + c;
                if (col >= N) continue;
                const uint32_t sp = ds[((size_t)t * ngp + gp) * 8 + c];
                sc[(size_t)(2 * gp) * N + col] = (uint16_t)(sp & 0xFFFF);
                if (2 * gp + 1 < ng) sc[(size_t)(2 * gp + 1) * N + col] = (uint16_t)(sp >> 16);
            }
    }
}

int matset_alloc(kr_engine* e, MatSet& ms, int K, int N, int bits, int count) {
    if (ms.allocated()) {
        if (ms.K != K || ms.N != N || ms.bits != bits)
            return kr_fail(KR_ERR_VALUE, "expert shape/bits mismatch within layer: have K=%d N=%d bits=%d, got K=%d N=%d bits=%d",
                           ms.K, ms.N, ms.bits, K, N, bits);
        return KR_OK;
    }
    if (K % 128 != 0) return kr_fail(KR_ERR_VALUE, "reduction dim %d must be divisible by group_size 128", K);
    ms.K = K; ms.N = N; ms.bits = bits; ms.count = count;
    ms.q_stride = kr_mat_q_bytes(K, N, bits); ms.s_stride = kr_mat_s_bytes(K, N);
    if (ms.q.ensure(ms.q_stride * count) || ms.s.ensure(ms.s_stride * count))
        return kr_fail(KR_ERR_HIP, "hipMalloc of %zu bytes failed", ms.q_stride * count);
    e->weight_bytes += (ms.q_stride + ms.s_stride) * count;
    return KR_OK;
}

// the prefill nibble sums (MatSet.wsum) are built lazily for ALL matrices of a set: any upload / fill that changes a matrix drops them so
// the next prefill call rebuilds them (their bytes are counted once, when first built)
static void drop_wsum(MatSet& ms) { if (ms.wsum.p) ms.wsum.release(); }

int upload_mat(kr_engine* e, MatSet& ms, int idx, const void* w, const uint16_t* sc) {
    drop_wsum(ms);
    std::vector<uint32_t> dq(ms.q_stride / 4), ds(ms.s_stride / 4);
    if (ms.bits == 4) retile_int4((const uint32_t*)w, sc, ms.K, ms.N, dq.data(), ds.data());
    else retile_int8((const int8_t*)w, sc, ms.K, ms.N, dq.data(), ds.data());
    KR_HIP(hipMemcpy((char*)ms.q.p + (size_t)idx * ms.q_stride, dq.data(), ms.q_stride, hipMemcpyHostToDevice));
    KR_HIP(hipMemcpy((char*)ms.s.p + (size_t)idx * ms.s_stride, ds.data(), ms.s_stride, hipMemcpyHostToDevice));
    (void)e;
    return KR_OK;
}

int download_mat(kr_engine* e, MatSet& ms, int idx, void* dst, uint16_t* sc) {
    (void)e;
    std::vector<uint32_t> dq(ms.q_stride / 4), ds(ms.s_stride / 4);
    KR_HIP(hipMemcpy(dq.data(), (char*)ms.q.p + (size_t)idx * ms.q_stride, ms.q_stride, hipMemcpyDeviceToHost));
    KR_HIP(hipMemcpy(ds.data(), (char*)ms.s.p + (size_t)idx * ms.s_stride, ms.s_stride, hipMemcpyDeviceToHost));
    if (ms.bits == 4) untile_int4(dq.data(), ds.data(), ms.K, ms.N, (uint32_t*)dst, sc);
    else untile_int8(dq.data(), ds.data(), ms.K, ms.N, (int8_t*)dst, sc);
    return KR_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int kr_engine_create(int device, const kr_model_config* cfg, kr_engine** out) {
    if (!cfg || !out) return kr_fail(KR_ERR_VALUE, "null argument");
    if (cfg->hidden_size <= 0 || cfg->hidden_size % 128 != 0)
        return kr_fail(KR_ERR_VALUE, "hidden_size (%d) must be divisible by group_size (128)", cfg->hidden_size);
    if (cfg->group_size != 128) return kr_fail(KR_ERR_VALUE, "group_size %d unsupported (reference default 128, marlin.rs:12)", cfg->group_size);
    if (cfg->num_experts_per_tok > KR_MAX_TOPK) return kr_fail(KR_ERR_VALUE, "topk %d > MAX_TOPK 32", cfg->num_experts_per_tok);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return kr_fail(KR_ERR_HIP, "no HIP device available: libkrasis_hip.so has no CPU fallback");
    }
    if (device < 0 || device >= ndev) return kr_fail(KR_ERR_VALUE, "device ordinal %d out of range (%d devices)", device, ndev);
    KR_HIP(hipSetDevice(device));
    std::unique_ptr<kr_engine> e(new kr_engine);
    e->device = device; e->cfg = *cfg;
    e->layers.resize(cfg->num_moe_layers);
    for (auto& l : e->layers) l.present.assign(cfg->n_routed_experts, 0);
    KR_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    *out = e.release();
    return KR_OK;
}

// A decode store created before its MoE engine exists (the reference builds CpuDecodeStore first and calls set_moe_store last,
// decode_setup.py:824-1018) runs on a bare engine: device + stream, no MoE layers.  kr_decode_set_moe_store swaps the real one in.
kr_engine* kr_engine_new_bare(int device) {
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::unique_ptr<kr_engine> e(new kr_engine);
    e->device = device; e->cfg = kr_model_config{}; e->cfg.group_size = 128; e->cfg.routed_scaling_factor = 1.0f;
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e.release();
}

extern "C" void kr_engine_destroy(kr_engine* e) {
    if (!e) return;
    (void)kr_ep_destroy(e);
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (auto& l : e->layers) {
        for (MatSet* ms : {&l.w13, &l.w2, &l.sw13, &l.sw2}) ms->wsum.release();
        l.w13.q.release(); l.w13.s.release(); l.w2.q.release(); l.w2.s.release();
        l.sw13.q.release(); l.sw13.s.release(); l.sw2.q.release(); l.sw2.s.release();
        l.gate_cm.release(); l.gate_rm.release(); l.gate_row.release(); l.bias.release(); l.esc.release();
        for (GgufSet* g : {&l.g_gate, &l.g_up, &l.g_down, &l.gs_gate, &l.gs_up, &l.gs_down}) { g->q.release(); g->h.release(); g->ws.release(); }
    }
    for (DevBuf* b : {&e->gu, &e->eo, &e->st_act, &e->st_ids, &e->st_w, &e->st_out, &e->ptr_table, &e->r_logits, &e->r_ids, &e->r_w, &e->r_x}) b->release();
    for (auto& P : e->pf) for (DevBuf* b : {&P.i32, &P.xh, &P.xl, &P.xs, &P.xm, &P.gu, &P.hh, &P.hl, &P.hs, &P.hm, &P.eo, &P.sgu, &P.shh, &P.shl, &P.shs, &P.shm, &P.seo}) b->release();
    (void)hipStreamDestroy(e->stream);
    delete e;
    (void)hipGetLastError();
}

extern "C" int kr_engine_get_config(const kr_engine* e, kr_model_config* out) {
    if (!e || !out) return kr_fail(KR_ERR_VALUE, "null argument");
    *out = e->cfg; return KR_OK;
}
extern "C" size_t kr_engine_device_bytes(const kr_engine* e) { return e ? e->weight_bytes : 0; }
extern "C" int kr_synchronize(kr_engine* e) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    KR_HIP(hipSetDevice(e->device));
    KR_HIP(hipStreamSynchronize(e->stream));
    return KR_OK;
}

static int check_layer(kr_engine* e, int layer) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (layer < 0 || layer >= (int)e->layers.size())
        return kr_fail(KR_ERR_VALUE, "moe_layer_idx %d out of range (have %zu MoE layers)", layer, e->layers.size());
    return KR_OK;
}

extern "C" int kr_upload_expert_unified(kr_engine* e, int layer, int expert, int inter, const void* w13,
                                        const uint16_t* w13_scales, int w13_bits, const void* w2,
                                        const uint16_t* w2_scales, int w2_bits) {
    if (int rc = check_layer(e, layer)) return rc;
    if (!w13 || !w13_scales || !w2 || !w2_scales) return kr_fail(KR_ERR_VALUE, "null weight pointer");
    if ((w13_bits != 4 && w13_bits != 8) || (w2_bits != 4 && w2_bits != 8))
        return kr_fail(KR_ERR_VALUE, "Unsupported num_bits: %d/%d", w13_bits, w2_bits);
    if (inter <= 0 || inter % 128 != 0) return kr_fail(KR_ERR_VALUE, "intermediate size (%d) must be divisible by group_size (128)", inter);
    KR_HIP(hipSetDevice(e->device));
    Layer& L = e->layers[layer];
    const int H = e->cfg.hidden_size;
    if (expert == -1) {
        if (int rc = matset_alloc(e, L.sw13, H, 2 * inter, w13_bits, 1)) return rc;
        if (int rc = matset_alloc(e, L.sw2, inter, H, w2_bits, 1)) return rc;
        if (int rc = upload_mat(e, L.sw13, 0, w13, w13_scales)) return rc;
        if (int rc = upload_mat(e, L.sw2, 0, w2, w2_scales)) return rc;
        L.shared_present = true; L.shared_inter = inter;
        return KR_OK;
    }
    if (expert < 0 || expert >= e->cfg.n_routed_experts)
        return kr_fail(KR_ERR_VALUE, "expert index %d out of range (%d experts)", expert, e->cfg.n_routed_experts);
    if (int rc = matset_alloc(e, L.w13, H, 2 * inter, w13_bits, e->cfg.n_routed_experts)) return rc;
    if (int rc = matset_alloc(e, L.w2, inter, H, w2_bits, e->cfg.n_routed_experts)) return rc;
    if (int rc = upload_mat(e, L.w13, expert, w13, w13_scales)) return rc;
    if (int rc = upload_mat(e, L.w2, expert, w2, w2_scales)) return rc;
    L.present[expert] = 1; L.inter = inter;
    return KR_OK;
}

// load_from_hf path (weights/mod.rs:1181 -> load_and_quantize_expert -> marlin.rs:65,145): BF16 checkpoint tensors in the HF layout
// (gate/up [inter, hidden], down [hidden, inter], row-major; host or device) are quantized ON THE GPU with the reference's rule and
// written straight into the resident layout -- no host-side packing, one HtoD copy of the BF16 data.
extern "C" int kr_upload_expert_bf16(kr_engine* e, int layer, int expert, int inter, const uint16_t* gate, const uint16_t* up, const uint16_t* down,
                                     int w13_bits, int w2_bits) {
    if (int rc = check_layer(e, layer)) return rc;
    if (!gate || !up || !down) return kr_fail(KR_ERR_VALUE, "null weight pointer");
    if ((w13_bits != 4 && w13_bits != 8) || (w2_bits != 4 && w2_bits != 8)) return kr_fail(KR_ERR_VALUE, "Unsupported num_bits: %d/%d", w13_bits, w2_bits);
    if (inter <= 0 || inter % 128 != 0) return kr_fail(KR_ERR_VALUE, "intermediate size (%d) must be divisible by group_size (128)", inter);
    KR_HIP(hipSetDevice(e->device));
    Layer& L = e->layers[layer];
    const int H = e->cfg.hidden_size;
    const bool shared = expert == -1;
    if (!shared && (expert < 0 || expert >= e->cfg.n_routed_experts)) return kr_fail(KR_ERR_VALUE, "expert index %d out of range (%d experts)", expert, e->cfg.n_routed_experts);
    MatSet& a = shared ? L.sw13 : L.w13; MatSet& b = shared ? L.sw2 : L.w2;
    const int count = shared ? 1 : e->cfg.n_routed_experts, idx = shared ? 0 : expert;
    if (int rc = matset_alloc(e, a, H, 2 * inter, w13_bits, count)) return rc;
    if (int rc = matset_alloc(e, b, inter, H, w2_bits, count)) return rc;
    const size_t n_each = (size_t)inter * H;
    const uint16_t *dg = gate, *du = up, *dd = down;
    if (!is_device_ptr(gate) || !is_device_ptr(up) || !is_device_ptr(down)) {
        if (e->st_act.ensure(3 * n_each * 2)) return kr_fail(KR_ERR_HIP, "hipMalloc of the BF16 staging buffer failed");
        uint16_t* st = (uint16_t*)e->st_act.p;
        KR_HIP(hipMemcpyAsync(st, gate, n_each * 2, hipMemcpyDefault, e->stream));
        KR_HIP(hipMemcpyAsync(st + n_each, up, n_each * 2, hipMemcpyDefault, e->stream));
        KR_HIP(hipMemcpyAsync(st + 2 * n_each, down, n_each * 2, hipMemcpyDefault, e->stream));
        dg = st; du = st + n_each; dd = st + 2 * n_each;
    }
    char* q13 = (char*)a.q.p + (size_t)idx * a.q_stride; uint32_t* s13 = (uint32_t*)((char*)a.s.p + (size_t)idx * a.s_stride);
    char* q2 = (char*)b.q.p + (size_t)idx * b.q_stride; uint32_t* s2 = (uint32_t*)((char*)b.s.p + (size_t)idx * b.s_stride);
    kr_launch_quant_bf16(dg, inter, H, w13_bits, q13, s13, 0, e->stream);            // gate -> columns [0, inter)
    kr_launch_quant_bf16(du, inter, H, w13_bits, q13, s13, inter / 8, e->stream);    // up   -> columns [inter, 2 inter)
    kr_launch_quant_bf16(dd, H, inter, w2_bits, q2, s2, 0, e->stream);
    KR_HIP(hipStreamSynchronize(e->stream));   // the staging buffer is reused by the next call
    KR_HIP(hipGetLastError());
    drop_wsum(a); drop_wsum(b);                 // nibble sums are rebuilt on the next prefill call
    if (shared) { L.shared_present = true; L.shared_inter = inter; }
    else { L.present[expert] = 1; L.inter = inter; }
    return KR_OK;
}

extern "C" int kr_fill_layer_synthetic(kr_engine* e, int layer, int bits, uint64_t seed) {
    if (int rc = check_layer(e, layer)) return rc;
    if (bits != 4 && bits != 8) return kr_fail(KR_ERR_VALUE, "Unsupported num_bits: %d", bits);
    KR_HIP(hipSetDevice(e->device));
    Layer& L = e->layers[layer];
    const int H = e->cfg.hidden_size, I = e->cfg.moe_intermediate_size, E = e->cfg.n_routed_experts;
    if (int rc = matset_alloc(e, L.w13, H, 2 * I, bits, E)) return rc;
    if (int rc = matset_alloc(e, L.w2, I, H, bits, E)) return rc;
    for (MatSet* ms : {&L.w13, &L.w2, &L.sw13, &L.sw2}) drop_wsum(*ms);
    kr_launch_fill_synth(L.w13.q.p, L.w13.q_stride * E, (uint32_t*)L.w13.s.p, L.w13.s_stride * E / 4, seed * 4 + 0, e->stream);
    kr_launch_fill_synth(L.w2.q.p, L.w2.q_stride * E, (uint32_t*)L.w2.s.p, L.w2.s_stride * E / 4, seed * 4 + 1, e->stream);
    std::fill(L.present.begin(), L.present.end(), 1); L.inter = I;
    if (e->cfg.n_shared_experts > 0) {
        const int SI = I * e->cfg.n_shared_experts;
        if (int rc = matset_alloc(e, L.sw13, H, 2 * SI, bits, 1)) return rc;
        if (int rc = matset_alloc(e, L.sw2, SI, H, bits, 1)) return rc;
        kr_launch_fill_synth(L.sw13.q.p, L.sw13.q_stride, (uint32_t*)L.sw13.s.p, L.sw13.s_stride / 4, seed * 4 + 2, e->stream);
        kr_launch_fill_synth(L.sw2.q.p, L.sw2.q_stride, (uint32_t*)L.sw2.s.p, L.sw2.s_stride / 4, seed * 4 + 3, e->stream);
        L.shared_present = true; L.shared_inter = SI;
    }
    KR_HIP(hipStreamSynchronize(e->stream));
    return KR_OK;
}

extern "C" int kr_download_expert_unified(kr_engine* e, int layer, int expert, void* w13, uint16_t* w13_scales,
                                          void* w2, uint16_t* w2_scales) {
    if (int rc = check_layer(e, layer)) return rc;
    KR_HIP(hipSetDevice(e->device));
    Layer& L = e->layers[layer];
    MatSet& a = expert == -1 ? L.sw13 : L.w13;
    MatSet& b = expert == -1 ? L.sw2 : L.w2;
    const int idx = expert == -1 ? 0 : expert;
    if (!a.allocated() || (expert >= 0 && (expert >= a.count || !L.present[expert])) || (expert == -1 && !L.shared_present))
        return kr_fail(KR_ERR_STATE, "expert %d of layer %d not loaded", expert, layer);
    if (int rc = download_mat(e, a, idx, w13, w13_scales)) return rc;
    if (int rc = download_mat(e, b, idx, w2, w2_scales)) return rc;
    return KR_OK;
}

// ---- native GGUF blocks: raw row-major [rows][K/blk] -> lane-tiled records (kr_gguf.hip header comment) ----
static size_t ggml_block_bytes(int t) { return t == GG_Q4_K ? 144 : t == GG_Q8_0 ? 34 : t == GG_Q4_0 ? 18 : t == GG_Q5_0 ? 22 : t == GG_Q6_K ? 210 : 0; }
static int ggml_block_elems(int t) { return (t == GG_Q4_K || t == GG_Q6_K) ? 256 : 32; }

static void retile_gguf(int type, const uint8_t* src, int K, int N, uint8_t* dq, uint8_t* dh) {
    const int nt = (N + 7) / 8; const size_t bb = ggml_block_bytes(type); const int nb = K / ggml_block_elems(type);
    const size_t row_bytes = (size_t)nb * bb;
    if (type == GG_Q5_0 || type == GG_Q6_K) { memcpy(dq, src, (size_t)N * row_bytes); return; }
    memset(dq, 0, gg_q_bytes(type, K, N)); memset(dh, 0, gg_h_bytes(type, K, N));
    for (int t = 0; t < nt; t++) for (int r = 0; r < 8; r++) {
        const int row = t * 8 + r; if (row >= N) continue;
        const uint8_t* rp = src + (size_t)row * row_bytes;
        if (type == GG_Q4_K) {
            for (int b = 0; b < nb; b++) {
                const uint8_t* blk = rp + (size_t)b * 144; const uint8_t* qs = blk + 16;
                memcpy(dh + (((size_t)t * nb + b) * 8 + r) * 16, blk, 16);
                for (int l = 0; l < 8; l++) {
                    uint8_t* o = dq + (((size_t)t * nb + b) * 64 + r * 8 + l) * 16;
                    for (int j = 0; j < 4; j++) { o[4 * j] = qs[32 * j + 2 * l]; o[4 * j + 1] = qs[32 * j + 2 * l + 1]; o[4 * j + 2] = qs[32 * j + 16 + 2 * l]; o[4 * j + 3] = qs[32 * j + 17 + 2 * l]; }
                }
            }
        } else if (type == GG_Q8_0) {
            const int nbg = (nb + 3) / 4;
            for (int b = 0; b < nb; b++) {
                const uint8_t* blk = rp + (size_t)b * 34; const uint8_t* qs = blk + 2; const int bg = b / 4, u = b % 4;
                memcpy(dh + (((size_t)t * nbg + bg) * 8 + r) * 8 + u * 2, blk, 2);
                for (int l = 0; l < 8; l++) {
                    uint8_t* o = dq + (((size_t)t * nbg + bg) * 64 + r * 8 + l) * 16 + u * 4;
                    o[0] = qs[2 * l]; o[1] = qs[2 * l + 1]; o[2] = qs[16 + 2 * l]; o[3] = qs[17 + 2 * l];
                }
            }
        } else {  // Q4_0
            const int nbg = (nb + 7) / 8;
            for (int b = 0; b < nb; b++) {
                const uint8_t* blk = rp + (size_t)b * 18; const uint8_t* qs = blk + 2; const int bg = b / 8, u = b % 8;
                memcpy(dh + (((size_t)t * nbg + bg) * 8 + r) * 16 + u * 2, blk, 2);
                for (int l = 0; l < 8; l++) {
                    uint8_t* o = dq + (((size_t)t * nbg + bg) * 64 + r * 8 + l) * 16 + u * 2;
                    o[0] = qs[2 * l]; o[1] = qs[2 * l + 1];
                }
            }
        }
    }
}

static int ggset_alloc(kr_engine* e, GgufSet& gs, int type, int K, int N, int count) {
    if (gs.allocated()) {
        if (gs.type != type || gs.K != K || gs.N != N) return kr_fail(KR_ERR_VALUE, "GGUF expert type/shape mismatch within layer");
        return KR_OK;
    }
    if (!ggml_block_bytes(type)) return kr_fail(KR_ERR_VALUE, "GGUF matvec not implemented for type %d", type);
    if (K % ggml_block_elems(type)) return kr_fail(KR_ERR_VALUE, "K=%d not a multiple of the block size of type %d", K, type);
    gs.type = type; gs.K = K; gs.N = N; gs.count = count; gs.q_stride = gg_q_bytes(type, K, N); gs.h_stride = gg_h_bytes(type, K, N);
    if (gs.q.ensure(gs.q_stride * count) || gs.h.ensure(gs.h_stride * count)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    e->weight_bytes += (gs.q_stride + gs.h_stride) * count;
    return KR_OK;
}
// the KR_GEMM_FAST copy of a matrix set (gg_ensure_fast) is derived data: any change of the blocks drops it, and it is rebuilt on the next
// tolerance-mode call.  `up` has no copy of its own -- its columns live behind the gate set's -- so an upload to `up` passes the gate set as `owner`.
static void ggset_drop_fast(kr_engine* e, GgufSet& gs) {
    if (!gs.fq.p) return;
    const size_t bytes = (gs.fq_stride + (gs.fqo.p ? 2 : 1) * gs.fqs_stride) * (size_t)gs.count;
    e->weight_bytes = e->weight_bytes > bytes ? e->weight_bytes - bytes : 0;
    gs.fq.release(); gs.fqs.release(); gs.fqo.release(); gs.fq_stride = gs.fqs_stride = 0; gs.fN = 0;
}
static int ggset_upload(kr_engine* e, GgufSet& gs, int idx, const uint8_t* src, GgufSet* owner = nullptr) {
    gs.ws.release();                                  // prompt-pass quant sums are rebuilt on the next kr_moe_prefill
    ggset_drop_fast(e, gs);
    if (owner) ggset_drop_fast(e, *owner);
    std::vector<uint8_t> dq(gs.q_stride), dh(gs.h_stride);
    retile_gguf(gs.type, src, gs.K, gs.N, dq.data(), dh.data());
    KR_HIP(hipMemcpy((char*)gs.q.p + (size_t)idx * gs.q_stride, dq.data(), gs.q_stride, hipMemcpyHostToDevice));
    KR_HIP(hipMemcpy((char*)gs.h.p + (size_t)idx * gs.h_stride, dh.data(), gs.h_stride, hipMemcpyHostToDevice));
    return KR_OK;
}

extern "C" int kr_upload_expert_gguf(kr_engine* e, int layer, int expert, int inter, const uint8_t* gate, const uint8_t* up, int gate_up_type,
                                     const uint8_t* down, int down_type) {
    if (int rc = check_layer(e, layer)) return rc;
    if (!gate || !up || !down) return kr_fail(KR_ERR_VALUE, "null weight pointer");
    KR_HIP(hipSetDevice(e->device));
    Layer& L = e->layers[layer];
    const int H = e->cfg.hidden_size;
    const bool sh = expert == -1;
    if (!sh && (expert < 0 || expert >= e->cfg.n_routed_experts)) return kr_fail(KR_ERR_VALUE, "expert index %d out of range", expert);
    const int cnt = sh ? 1 : e->cfg.n_routed_experts, idx = sh ? 0 : expert;
    GgufSet& G = sh ? L.gs_gate : L.g_gate; GgufSet& U = sh ? L.gs_up : L.g_up; GgufSet& D = sh ? L.gs_down : L.g_down;
    if (int rc = ggset_alloc(e, G, gate_up_type, H, inter, cnt)) return rc;
    if (int rc = ggset_alloc(e, U, gate_up_type, H, inter, cnt)) return rc;
    if (int rc = ggset_alloc(e, D, down_type, inter, H, cnt)) return rc;
    if (int rc = ggset_upload(e, G, idx, gate)) return rc;
    if (int rc = ggset_upload(e, U, idx, up, &G)) return rc;
    if (int rc = ggset_upload(e, D, idx, down)) return rc;
    if (sh) { L.gguf_shared = true; L.shared_inter = inter; } else { L.gguf = true; L.inter = inter; L.present[expert] = 1; }
    return KR_OK;
}

// synthetic native-GGUF experts for a whole layer (Q4_K or Q8_0 blocks, both projections), generated on the GPU -- the GGUF twin of
// kr_fill_layer_synthetic with the block distribution SURVEY 8d defines
extern "C" int kr_fill_layer_synthetic_gguf(kr_engine* e, int layer, int gate_up_type, int down_type, uint64_t seed) {
    if (int rc = check_layer(e, layer)) return rc;
    for (int t : {gate_up_type, down_type}) if (t != GG_Q4_K && t != GG_Q8_0) return kr_fail(KR_ERR_VALUE, "synthetic GGUF fill supports Q4_K (12) and Q8_0 (8), got %d", t);
    KR_HIP(hipSetDevice(e->device));
    Layer& L = e->layers[layer];
    const int H = e->cfg.hidden_size, I = e->cfg.moe_intermediate_size, E = e->cfg.n_routed_experts;
    if (int rc = ggset_alloc(e, L.g_gate, gate_up_type, H, I, E)) return rc;
    if (int rc = ggset_alloc(e, L.g_up, gate_up_type, H, I, E)) return rc;
    if (int rc = ggset_alloc(e, L.g_down, down_type, I, H, E)) return rc;
    int k = 0;
    for (GgufSet* g : {&L.g_gate, &L.g_up, &L.g_down}) {
        g->ws.release();
        ggset_drop_fast(e, *g);
        kr_launch_gpf_fill_synth(g->q.p, g->q_stride * E, g->h.p, g->h_stride * E, g->type, seed * 8 + (uint64_t)(k++), e->stream);
    }
    KR_HIP(hipStreamSynchronize(e->stream));
    L.gguf = true; L.inter = I; std::fill(L.present.begin(), L.present.end(), 1);
    return KR_OK;
}

// stage an argument that may live on the host
static int stage_in(kr_engine* e, DevBuf& buf, const void* p, size_t bytes, const void** dev) {
    if (is_device_ptr(p)) { *dev = p; return KR_OK; }
    if (buf.ensure(bytes)) return kr_fail(KR_ERR_HIP, "hipMalloc of staging buffer failed");
    KR_HIP(hipMemcpyAsync(buf.p, p, bytes, hipMemcpyHostToDevice, e->stream));
    *dev = buf.p; return KR_OK;
}

// expert ids handed over in HOST memory are checked here the way the reference's indexing would fail (a panic on an out-of-range or
// unloaded expert, moe.rs:2904-2932): ValueError instead of an out-of-bounds device read.  Device-resident ids are clamped to "skip"
// by the kernels (kr_resolve_slot, the prefill sort).
static int check_host_ids(kr_engine* e, const Layer& L, const int32_t* ids, size_t n) {
    const int E = e->cfg.n_routed_experts;
    for (size_t i = 0; i < n; i++) {
        const int id = ids[i];
        if (id < 0) continue;                                  // -1 = skip (moe.rs:2904)
        if (id >= E) return kr_fail(KR_ERR_VALUE, "expert id %d out of range (%d experts)", id, E);
        if (!L.present.empty() && !L.present[id]) return kr_fail(KR_ERR_STATE, "expert %d is not loaded", id);
    }
    return KR_OK;
}

// body of kr_moe_forward on a resolved stream; the caller holds e->mu
static int moe_forward_locked(kr_engine* e, int layer, const void* act, const int32_t* ids, const float* wts,
                              void* out, int batch, int topk, int out_dtype, int routed_only, hipStream_t st) {
    if (int rc = check_layer(e, layer)) return rc;
    if (!act || !ids || !wts || !out) return kr_fail(KR_ERR_VALUE, "null pointer argument");
    if (batch <= 0) return kr_fail(KR_ERR_VALUE, "batch_size must be > 0");
    if (topk <= 0 || topk > KR_MAX_TOPK) return kr_fail(KR_ERR_VALUE, "topk %d exceeds MAX_TOPK %d", topk, KR_MAX_TOPK);
    if (batch > 65535) return kr_fail(KR_ERR_VALUE, "batch %d too large for the decode path (use the prefill entry point)", batch);
    Layer& L = e->layers[layer];
    if (!L.w13.allocated() && !L.gguf) return kr_fail(KR_ERR_STATE, "Model not loaded -- call load() first (layer %d has no experts)", layer);
    KR_HIP(hipSetDevice(e->device));
    if (!is_device_ptr(ids)) if (int rc = check_host_ids(e, L, ids, (size_t)batch * topk)) return rc;
    const int H = e->cfg.hidden_size;
    if (L.gguf) {
        // moe_forward_gguf (moe.rs:990): native GGUF blocks, per-32 INT16 activations
        const bool ush = L.gguf_shared && !routed_only;
        GgMoeArgs g{};
        g.B = batch; g.topk = topk; g.n_slots = topk + (ush ? 1 : 0); g.H = H; g.E = e->cfg.n_routed_experts;
        g.gate = L.g_gate.view(); g.up = L.g_up.view(); g.down = L.g_down.view();
        if (ush) { g.sgate = L.gs_gate.view(); g.sup = L.gs_up.view(); g.sdown = L.gs_down.view(); }
        g.I_max = ush && L.shared_inter > L.inter ? L.shared_inter : L.inter; g.gu_ld = 2 * g.I_max;
        if (e->gu.ensure((size_t)batch * g.n_slots * g.gu_ld * 4) || e->eo.ensure((size_t)batch * g.n_slots * H * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc of MoE scratch failed");
        g.gu = (float*)e->gu.p; g.eo = (float*)e->eo.p;
        hipStream_t saved = e->stream; e->stream = st;
        const void *d_act, *d_ids, *d_w;
        int rc = stage_in(e, e->st_act, act, (size_t)batch * H * 2, &d_act);
        if (!rc) rc = stage_in(e, e->st_ids, ids, (size_t)batch * topk * 4, &d_ids);
        if (!rc) rc = stage_in(e, e->st_w, wts, (size_t)batch * topk * 4, &d_w);
        e->stream = saved;
        if (rc) return rc;
        g.act = (const uint16_t*)d_act; g.ids = (const int32_t*)d_ids;
        const size_t out_bytes = (size_t)batch * H * (out_dtype == KR_OUT_BF16 ? 2 : 4);
        const bool out_dev = is_device_ptr(out);
        if (!out_dev && e->st_out.ensure(out_bytes)) return kr_fail(KR_ERR_HIP, "hipMalloc of staging buffer failed");
        kr_launch_gguf_moe(g, st);
        KrMoeArgs c{};
        c.B = batch; c.topk = topk; c.n_slots = g.n_slots; c.H = H; c.E = e->cfg.n_routed_experts; c.ids = g.ids; c.wts = (const float*)d_w; c.eo = g.eo;
        c.out = out_dev ? out : e->st_out.p; c.out_bf16 = out_dtype == KR_OUT_BF16; c.rsf = e->cfg.routed_scaling_factor;
        kr_launch_moe_combine(c, st);
        KR_HIP(hipGetLastError());
        if (!out_dev) { KR_HIP(hipMemcpyAsync(out, e->st_out.p, out_bytes, hipMemcpyDeviceToHost, st)); KR_HIP(hipStreamSynchronize(st)); }
        return KR_OK;
    }
    const bool use_shared = L.shared_present && !routed_only;
    KrMoeArgs a{};
    a.B = batch; a.topk = topk; a.n_slots = topk + (use_shared ? 1 : 0); a.E = e->cfg.n_routed_experts;
    a.H = H; a.I = L.inter; a.I_shared = L.shared_inter;
    a.w13 = L.w13.view(); a.w2 = L.w2.view();
    if (use_shared) { a.sw13 = L.sw13.view(); a.sw2 = L.sw2.view(); }
    const int imax = use_shared && L.shared_inter > L.inter ? L.shared_inter : L.inter;
    a.gu_ld = 2 * imax;
    if (e->gu.ensure((size_t)batch * a.n_slots * a.gu_ld * 4) || e->eo.ensure((size_t)batch * a.n_slots * H * 4))
        return kr_fail(KR_ERR_HIP, "hipMalloc of MoE scratch failed");
    a.gu = (float*)e->gu.p; a.eo = (float*)e->eo.p;
    hipStream_t saved = e->stream; e->stream = st;
    const void *d_act, *d_ids, *d_w;
    int rc = stage_in(e, e->st_act, act, (size_t)batch * H * 2, &d_act);
    if (!rc) rc = stage_in(e, e->st_ids, ids, (size_t)batch * topk * 4, &d_ids);
    if (!rc) rc = stage_in(e, e->st_w, wts, (size_t)batch * topk * 4, &d_w);
    e->stream = saved;
    if (rc) return rc;
    a.act = (const uint16_t*)d_act; a.ids = (const int32_t*)d_ids; a.wts = (const float*)d_w;
    const size_t out_bytes = (size_t)batch * H * (out_dtype == KR_OUT_BF16 ? 2 : 4);
    const bool out_dev = is_device_ptr(out);
    if (!out_dev && e->st_out.ensure(out_bytes)) return kr_fail(KR_ERR_HIP, "hipMalloc of staging buffer failed");
    a.out = out_dev ? out : e->st_out.p; a.out_bf16 = out_dtype == KR_OUT_BF16;
    // moe.rs:703-706 applies rsf only together with a shared expert; callers that pass routed_only
    // (GpuPrefillManager.forward(routed_only=True), gpu_prefill.py:4467) get the bare weighted sum.
    a.rsf = e->cfg.routed_scaling_factor; a.swiglu_limit = e->cfg.swiglu_limit; a.alpha = e->cfg.activation_alpha;
    a.act_mode = e->cfg.swiglu_limit > 0.0f ? KR_ACT_GPTOSS : KR_ACT_SILU_FUSED;
    if (!e->prof) {
        kr_launch_moe_decode(a, st);
    } else {
        for (int i = 0; i < 4; i++) if (!e->pev[i]) KR_HIP(hipEventCreate(&e->pev[i]));
        KR_HIP(hipEventRecord(e->pev[0], st)); kr_launch_moe_w13(a, st);
        KR_HIP(hipEventRecord(e->pev[1], st)); kr_launch_moe_w2(a, st);
        KR_HIP(hipEventRecord(e->pev[2], st)); kr_launch_moe_combine(a, st);
        KR_HIP(hipEventRecord(e->pev[3], st));
        KR_HIP(hipEventSynchronize(e->pev[3]));
        for (int i = 0; i < 3; i++) { float ms = 0; KR_HIP(hipEventElapsedTime(&ms, e->pev[i], e->pev[i + 1])); e->prof_ms[i] += ms; e->prof_n[i]++; }
    }
    KR_HIP(hipGetLastError());
    if (!out_dev) {
        KR_HIP(hipMemcpyAsync(out, e->st_out.p, out_bytes, hipMemcpyDeviceToHost, st));
        KR_HIP(hipStreamSynchronize(st));
    }
    return KR_OK;
}

extern "C" int kr_moe_forward(kr_engine* e, int layer, const void* act, const int32_t* ids, const float* wts,
                              void* out, int batch, int topk, int out_dtype, int routed_only, void* stream) {
    if (int rc = check_layer(e, layer)) return rc;
    std::lock_guard<std::mutex> lk(e->mu);
    return moe_forward_locked(e, layer, act, ids, wts, out, batch, topk, out_dtype, routed_only, kr_pick_stream(e, stream));
}

extern "C" int kr_reduce_sum_bf16(kr_engine* e, const void* const* inputs, int n_inputs, void* out, size_t n, void* stream) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (n_inputs <= 0) return KR_OK;  // moe.rs:2511
    KR_HIP(hipSetDevice(e->device));
    hipStream_t st = kr_pick_stream(e, stream);
    for (int i = 0; i < n_inputs; i++)
        if (!is_device_ptr(inputs[i])) return kr_fail(KR_ERR_VALUE, "kr_reduce_sum_bf16 expects device pointers");
    if (!is_device_ptr(out)) return kr_fail(KR_ERR_VALUE, "kr_reduce_sum_bf16 expects device pointers");
    if (e->ptr_table.ensure(sizeof(void*) * 64)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    if (n_inputs > 64) return kr_fail(KR_ERR_VALUE, "too many inputs (%d > 64)", n_inputs);
    KR_HIP(hipMemcpyAsync(e->ptr_table.p, inputs, sizeof(void*) * n_inputs, hipMemcpyHostToDevice, st));
    kr_launch_reduce_sum_bf16((const uint16_t* const*)e->ptr_table.p, n_inputs, (uint16_t*)out, n, st);
    KR_HIP(hipGetLastError());
    return KR_OK;
}

// routing entry points live in kr_router.cpp-equivalent section below (implemented with kr_router.hip)
extern "C" int kr_set_routing_config(kr_engine* e, int scoring, int norm_topk_prob, int topk, int n_experts, int hidden) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (scoring < 0 || scoring > 2) return kr_fail(KR_ERR_VALUE, "unknown scoring_func %d", scoring);
    if (topk <= 0 || topk > KR_MAX_TOPK || topk > n_experts) return kr_fail(KR_ERR_VALUE, "bad topk %d for %d experts", topk, n_experts);
    e->routing_set = true; e->r_scoring = scoring; e->r_norm = norm_topk_prob; e->r_topk = topk; e->r_ne = n_experts; e->r_hidden = hidden;
    return KR_OK;
}

static inline uint16_t f32_to_bf16_bits(float f) {
    uint32_t b; memcpy(&b, &f, 4);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}

extern "C" int kr_set_routing_weights(kr_engine* e, int layer, const void* gate, int gate_is_f32, const float* bias,
                                      const float* e_score_corr) {
    if (int rc = check_layer(e, layer)) return rc;
    if (!e->routing_set) return kr_fail(KR_ERR_STATE, "Routing config not set");
    if (!gate) return kr_fail(KR_ERR_VALUE, "null gate pointer");
    KR_HIP(hipSetDevice(e->device));
    Layer& L = e->layers[layer];
    const int E = e->r_ne, H = e->r_hidden;
    if (H % 128 != 0) return kr_fail(KR_ERR_VALUE, "router hidden dim %d must be a multiple of 128", H);
    L.gate_host.resize((size_t)E * H);
    bool exact = true;
    if (gate_is_f32) {
        memcpy(L.gate_host.data(), gate, (size_t)E * H * 4);
        for (size_t i = 0; i < (size_t)E * H && exact; i++) {
            uint32_t b; memcpy(&b, &L.gate_host[i], 4);
            exact = (b & 0xFFFFu) == 0;
        }
    } else {
        const uint16_t* g = (const uint16_t*)gate;
        for (size_t i = 0; i < (size_t)E * H; i++) { uint32_t b = (uint32_t)g[i] << 16; memcpy(&L.gate_host[i], &b, 4); }
    }
    L.gate_bf16_exact = exact;
    const int neb = (E + 3) / 4;
    if (exact) {
        const int nc = H / 128;
        std::vector<uint16_t> cm((size_t)neb * nc * 64 * 8, 0);
        for (int eb = 0; eb < neb; eb++) for (int c = 0; c < nc; c++) for (int lane = 0; lane < 64; lane++) {
            const int ex = eb * 4 + lane / 16, j = lane % 16;
            if (ex >= E) continue;
            for (int u = 0; u < 8; u++) {
                uint32_t b; memcpy(&b, &L.gate_host[(size_t)ex * H + 16 * (8 * c + u) + j], 4);
                cm[(((size_t)eb * nc + c) * 64 + lane) * 8 + u] = (uint16_t)(b >> 16);
            }
        }
        if (L.gate_cm.ensure(cm.size() * 2)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        KR_HIP(hipMemcpy(L.gate_cm.p, cm.data(), cm.size() * 2, hipMemcpyHostToDevice));
    } else {
        const int nc = H / 64;
        std::vector<float> cm((size_t)neb * nc * 64 * 4, 0.0f);
        for (int eb = 0; eb < neb; eb++) for (int c = 0; c < nc; c++) for (int lane = 0; lane < 64; lane++) {
            const int ex = eb * 4 + lane / 16, j = lane % 16;
            if (ex >= E) continue;
            for (int u = 0; u < 4; u++) cm[(((size_t)eb * nc + c) * 64 + lane) * 4 + u] = L.gate_host[(size_t)ex * H + 16 * (4 * c + u) + j];
        }
        if (L.gate_cm.ensure(cm.size() * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        KR_HIP(hipMemcpy(L.gate_cm.p, cm.data(), cm.size() * 4, hipMemcpyHostToDevice));
    }
    L.gate_rm.release(); L.gate_row.release();
    L.has_bias = bias != nullptr; L.has_esc = e_score_corr != nullptr;
    if (bias) { if (L.bias.ensure((size_t)E * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed"); KR_HIP(hipMemcpy(L.bias.p, bias, (size_t)E * 4, hipMemcpyHostToDevice)); }
    if (e_score_corr) { if (L.esc.ensure((size_t)E * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed"); KR_HIP(hipMemcpy(L.esc.p, e_score_corr, (size_t)E * 4, hipMemcpyHostToDevice)); }
    L.routing_present = true;
    return KR_OK;
}

// bench_decode_synthetic's router gate (decode.rs:5181 fill_random_f32(route_data, rng, 0.02) with Xorshift64, decode.rs:4356-4376):
// x ^= x<<13; x ^= x>>7; x ^= x<<17; value = (x as i64 / i64::MAX) as f32 * amp.  round_bf16 != 0 truncates to bf16 the way a checkpoint's
// gate is stored (the gate is then kept as bf16 in HBM).  One stream per layer: seed + layer (seed 0 -> 0xDEADBEEF like the reference).
extern "C" int kr_set_routing_weights_synthetic(kr_engine* e, int layer, uint64_t seed, float amp, int round_bf16) {
    if (int rc = check_layer(e, layer)) return rc;
    if (!e->routing_set) return kr_fail(KR_ERR_STATE, "Routing config not set");
    const size_t n = (size_t)e->r_ne * e->r_hidden;
    std::vector<float> g(n);
    uint64_t x = seed + (uint64_t)layer * 0x9E3779B97F4A7C15ull; if (x == 0) x = 0xDEADBEEFull;
    for (size_t i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        float v = (float)((double)(int64_t)x / 9223372036854775807.0) * amp;
        if (round_bf16) { uint32_t b; memcpy(&b, &v, 4); b &= 0xFFFF0000u; memcpy(&v, &b, 4); }
        g[i] = v;
    }
    return kr_set_routing_weights(e, layer, g.data(), 1, nullptr, nullptr);
}

static int build_gate_rm(kr_engine* e, Layer& L) {
    if (L.gate_rm.p) return KR_OK;
    if (!L.gate_bf16_exact) return kr_fail(KR_ERR_VALUE, "engine routing rule needs a bf16 gate (moe.rs:2990); this layer's gate is not bf16-exact");
    const int E = e->r_ne, H = e->r_hidden, neb = (E + 63) / 64;
    std::vector<uint16_t> rm((size_t)neb * (H / 8) * 64 * 8, 0);
    for (int eb = 0; eb < neb; eb++) for (int c = 0; c < H / 8; c++) for (int lane = 0; lane < 64; lane++) {
        const int ex = eb * 64 + lane;
        if (ex >= E) continue;
        for (int u = 0; u < 8; u++) {
            uint32_t b; memcpy(&b, &L.gate_host[(size_t)ex * H + 8 * c + u], 4);
            rm[(((size_t)eb * (H / 8) + c) * 64 + lane) * 8 + u] = (uint16_t)(b >> 16);
        }
    }
    if (L.gate_rm.ensure(rm.size() * 2)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    KR_HIP(hipMemcpy(L.gate_rm.p, rm.data(), rm.size() * 2, hipMemcpyHostToDevice));
    return KR_OK;
}

int kr_ensure_gate_row(kr_engine* e, int layer) {
    Layer& L = e->layers[layer];
    if (L.gate_row.p || L.gate_host.empty()) return KR_OK;
    const size_t n = L.gate_host.size();
    if (L.gate_bf16_exact) {
        std::vector<uint16_t> g(n);
        for (size_t i = 0; i < n; i++) { uint32_t b; memcpy(&b, &L.gate_host[i], 4); g[i] = (uint16_t)(b >> 16); }
        if (L.gate_row.ensure(n * 2)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        KR_HIP(hipMemcpy(L.gate_row.p, g.data(), n * 2, hipMemcpyHostToDevice));
    } else {
        if (L.gate_row.ensure(n * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        KR_HIP(hipMemcpy(L.gate_row.p, L.gate_host.data(), n * 4, hipMemcpyHostToDevice));
    }
    return KR_OK;
}

// routes m tokens; results stay on the device in e->r_ids / e->r_w (and e->r_logits)
static int route_device(kr_engine* e, Layer& L, const void* d_x, int m, int rule, hipStream_t st) {
    const int E = e->r_ne, H = e->r_hidden, k = e->r_topk;
    if (e->r_logits.ensure((size_t)m * E * 4) || e->r_ids.ensure((size_t)m * k * 4) || e->r_w.ensure((size_t)m * k * 4))
        return kr_fail(KR_ERR_HIP, "hipMalloc of routing scratch failed");
    const int gptoss = e->cfg.swiglu_limit > 0.0f;
    if (rule == KR_ROUTE_RULE_DECODE) {
        const float* bias = L.has_bias ? (const float*)L.bias.p : nullptr;
        bool done = false;
        if (m >= 32) {      // batches: the same chains on the f32 MFMA
            if (int rc = kr_ensure_gate_row(e, (int)(&L - e->layers.data()))) return rc;
            // kr_moe_set_gemm_mode(e, 1): the tolerance form of the batch logits (bf16 MFMA on x = hi + lo) -- the form KR_GEMM_FAST prompt passes run
            if (e->gemm_fast) done = L.gate_row.p && 0 == kr_launch_route_logits_fast(L.gate_row.p, L.gate_bf16_exact, (const float*)d_x, bias, (float*)e->r_logits.p, m, E, H, st);
            if (!done) done = L.gate_row.p && 0 == kr_launch_route_logits_mfma(L.gate_row.p, L.gate_bf16_exact, (const float*)d_x, bias, (float*)e->r_logits.p, m, E, H, st);
        }
        if (!done) kr_launch_route_logits_decode(L.gate_cm.p, L.gate_bf16_exact, (const float*)d_x, bias, (float*)e->r_logits.p, m, E, H, st);
    } else {
        if (int rc = build_gate_rm(e, L)) return rc;
        kr_launch_route_logits_engine(L.gate_rm.p, (const uint16_t*)d_x, (float*)e->r_logits.p, m, E, H, st);
    }
    kr_launch_route_select((const float*)e->r_logits.p, L.has_esc ? (const float*)L.esc.p : nullptr, (int32_t*)e->r_ids.p,
                           (float*)e->r_w.p, m, E, k, e->r_scoring, e->r_norm, rule, gptoss, st);
    KR_HIP(hipGetLastError());
    return KR_OK;
}

static int copy_out(void* dst, const void* dev_src, size_t bytes, hipStream_t st, bool* need_sync) {
    if (!dst) return KR_OK;
    if (is_device_ptr(dst)) { KR_HIP(hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToDevice, st)); }
    else { KR_HIP(hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToHost, st)); *need_sync = true; }
    return KR_OK;
}

extern "C" int kr_route_topk(kr_engine* e, int layer, const void* x, int m, int rule, int32_t* ids_out, float* w_out,
                             float* logits_out, void* stream) {
    if (int rc = check_layer(e, layer)) return rc;
    if (!e->routing_set) return kr_fail(KR_ERR_STATE, "Routing config not set");
    Layer& L = e->layers[layer];
    if (!L.routing_present) return kr_fail(KR_ERR_STATE, "Routing weights not set for layer %d", layer);
    if (!x || m <= 0) return kr_fail(KR_ERR_VALUE, "bad arguments");
    if (rule != KR_ROUTE_RULE_DECODE && rule != KR_ROUTE_RULE_ENGINE) return kr_fail(KR_ERR_VALUE, "unknown routing rule %d", rule);
    std::lock_guard<std::mutex> lk(e->mu);
    KR_HIP(hipSetDevice(e->device));
    hipStream_t st = kr_pick_stream(e, stream);
    const size_t xb = (size_t)m * e->r_hidden * (rule == KR_ROUTE_RULE_DECODE ? 4 : 2);
    const void* d_x = x;
    if (!is_device_ptr(x)) {
        if (e->r_x.ensure(xb)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        KR_HIP(hipMemcpyAsync(e->r_x.p, x, xb, hipMemcpyHostToDevice, st));
        d_x = e->r_x.p;
    }
    if (int rc = route_device(e, L, d_x, m, rule, st)) return rc;
    bool need_sync = false;
    if (int rc = copy_out(ids_out, e->r_ids.p, (size_t)m * e->r_topk * 4, st, &need_sync)) return rc;
    if (int rc = copy_out(w_out, e->r_w.p, (size_t)m * e->r_topk * 4, st, &need_sync)) return rc;
    if (int rc = copy_out(logits_out, e->r_logits.p, (size_t)m * e->r_ne * 4, st, &need_sync)) return rc;
    if (need_sync) KR_HIP(hipStreamSynchronize(st));
    return KR_OK;
}

extern "C" int kr_forward_moe_routed(kr_engine* e, int layer, const void* act_bf16, void* out_bf16, void* stream) {
    if (int rc = check_layer(e, layer)) return rc;
    Layer& L = e->layers[layer];
    if (!L.w13.allocated() && !L.gguf) return kr_fail(KR_ERR_STATE, "Model not loaded");
    if (!e->routing_set) return kr_fail(KR_ERR_STATE, "Routing config not set");
    if (!L.routing_present) return kr_fail(KR_ERR_STATE, "Routing weights not set for layer %d", layer);
    if (!act_bf16 || !out_bf16) return kr_fail(KR_ERR_VALUE, "null pointer argument");
    // ONE resolved stream for the router and the experts (re-encoding a resolved legacy stream 0 as the ABI's NULL would move the experts
    // onto the engine stream, unordered against the router), and the lock held across both: r_ids / r_w are engine scratch
    hipStream_t st = kr_pick_stream(e, stream);
    std::lock_guard<std::mutex> lk(e->mu);
    KR_HIP(hipSetDevice(e->device));
    const void* d_act = act_bf16;
    if (!is_device_ptr(act_bf16)) {
        if (e->r_x.ensure((size_t)e->r_hidden * 2)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        KR_HIP(hipMemcpyAsync(e->r_x.p, act_bf16, (size_t)e->r_hidden * 2, hipMemcpyHostToDevice, st));
        d_act = e->r_x.p;
    }
    if (int rc = route_device(e, L, d_act, 1, KR_ROUTE_RULE_ENGINE, st)) return rc;
    return moe_forward_locked(e, layer, d_act, (const int32_t*)e->r_ids.p, (const float*)e->r_w.p, out_bf16, 1, e->r_topk, KR_OUT_BF16, 0, st);
}

// ---- profiling hooks used by bench.py (HIP events on the launch stream, per kernel kind) ----
// kinds: 0 = kr_moe_w13_kernel, 1 = kr_moe_w2_kernel, 2 = kr_moe_combine_kernel
extern "C" int kr_set_profiling(kr_engine* e, int enable) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    e->prof = enable != 0;
    for (int i = 0; i < 8; i++) { e->prof_ms[i] = 0; e->prof_n[i] = 0; }
    return KR_OK;
}
extern "C" int kr_get_profile(kr_engine* e, int kind, double* total_ms, long* launches) {
    if (!e || kind < 0 || kind >= 8) return kr_fail(KR_ERR_VALUE, "bad profile kind");
    if (total_ms) *total_ms = e->prof_ms[kind];
    if (launches) *launches = e->prof_n[kind];
    return KR_OK;
}

// ------------------------------------------------------------------------------------------------
// prefill: GpuPrefillManager.forward (python/krasis/gpu_prefill.py:4374-4484) on int8 MFMA, numerics == kr_moe_forward
// ------------------------------------------------------------------------------------------------
int kr_ensure_wsum(kr_engine* e, MatSet& ms, hipStream_t st) {
    if (ms.wsum.p || !ms.allocated()) return KR_OK;
    if (ms.bits != 4 && ms.bits != 8) return kr_fail(KR_ERR_VALUE, "prefill MFMA path needs INT4-g128 or INT8-g128 weights (got %d-bit)", ms.bits);
    if (ms.wsum.ensure(ms.s_stride * ms.count)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    if (!ms.wsum_counted) { e->weight_bytes += ms.s_stride * ms.count; ms.wsum_counted = true; }
    kr_launch_pf_wsum(ms.view(), ms.count, (uint32_t*)ms.wsum.p, st);
    return KR_OK;
}

// tokens per pass of the expert path are bounded by (token, slot) PAIRS, the unit the scratch and the GEMM row tiles are sized in: a
// top-10 batch walks 8192 tokens at a time, the top-1 rows of the expert-parallel dispatch 81 920 -- so every expert's weights are
// streamed once per ~80 k pairs either way
#define KR_PF_PAIRS 81920

// ---- native GGUF layers (moe_forward_gguf, moe.rs:990): Q4_K / Q8_0 blocks on the int8-MFMA grouped GEMM (kr_gguf_prefill.hip); layers
//      whose block types have no MFMA form (Q4_0 / Q5_0 / Q6_K) walk the batch through the streaming kernels.  The caller holds e->mu.
static int gg_ensure_ws(kr_engine* e, GgufSet& g, hipStream_t st) {
    if (g.ws.p || !g.allocated()) return KR_OK;
    g.ws_stride = kr_gpf_ws_bytes(g.type, g.K, g.N);
    if (g.ws.ensure(g.ws_stride * g.count)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    kr_launch_gpf_wsum(g.view(), g.count, g.ws.p, g.ws_stride, st);
    (void)e;
    return KR_OK;
}

// KR_GEMM_FAST / kr_moe_set_gemm_mode(1) on a native Q4_K layer: the tolerance GEMM's copy of (gate | up) and of down, built once
static int gg_ensure_fast(kr_engine* e, GgufSet& a, GgufSet* b, hipStream_t st) {     // b: second matrix appended on N (up after gate), or null
    if (a.fq.p || !a.allocated()) return KR_OK;
    const int N = a.N + (b ? b->N : 0);
    if (a.type == GG_Q8_0) {      // INT8 lane tiles + one f16 scale per 32-wide block, no offsets
        const int ngp = (a.K / 128 + 1) / 2;
        a.fN = N; a.fq_stride = kr_mat_q_bytes(a.K, N, 8); a.fqs_stride = (size_t)(N / 8) * ngp * 8 * 8 * 2;
        if (a.fq.ensure(a.fq_stride * a.count) || a.fqs.ensure(a.fqs_stride * a.count)) return kr_fail(KR_ERR_HIP, "hipMalloc of the Q8_0 tolerance copy failed");
        kr_launch_gq8_repack(a.view(), a.count, a.fq.p, a.fq_stride, a.fqs.p, a.fqs_stride, 0, st);
        if (b) kr_launch_gq8_repack(b->view(), b->count, a.fq.p, a.fq_stride, a.fqs.p, a.fqs_stride, a.N / 8, st);
        e->weight_bytes += (a.fq_stride + a.fqs_stride) * a.count;
        return KR_OK;
    }
    a.fN = N; a.fq_stride = kr_mat_q_bytes(a.K, N, 4); a.fqs_stride = (size_t)(N / 8) * (a.K / 256) * 8 * 8 * 2;
    if (a.fq.ensure(a.fq_stride * a.count) || a.fqs.ensure(a.fqs_stride * a.count) || a.fqo.ensure(a.fqs_stride * a.count)) return kr_fail(KR_ERR_HIP, "hipMalloc of the Q4_K tolerance copy failed");
    kr_launch_gq_repack(a.view(), a.count, a.fq.p, a.fq_stride, a.fqs.p, a.fqo.p, a.fqs_stride, 0, N / 8, st);
    if (b) kr_launch_gq_repack(b->view(), b->count, a.fq.p, a.fq_stride, a.fqs.p, a.fqo.p, a.fqs_stride, a.N / 8, N / 8, st);
    e->weight_bytes += (a.fq_stride + 2 * a.fqs_stride) * a.count;
    return KR_OK;
}
static bool gg_fast_ok(const Layer& L, int H, bool use_shared) {
    // Q4_K over k ranges of whole super-blocks, Q8_0 over whole 128-k groups (V2-Lite's down projection: K = 1408 = 11 groups); gate and up of one type
    auto q4 = [](const GgufSet& g, int K) { return ((g.type == GG_Q4_K && K % 256 == 0) || (g.type == GG_Q8_0 && K % 128 == 0)) && g.N % 8 == 0; };
    bool ok = q4(L.g_gate, H) && q4(L.g_up, H) && L.g_gate.type == L.g_up.type && q4(L.g_down, L.inter) && L.inter % 32 == 0 && L.inter <= 2048;
    if (use_shared) ok = ok && q4(L.gs_gate, H) && q4(L.gs_up, H) && L.gs_gate.type == L.gs_up.type && q4(L.gs_down, L.shared_inter) && L.shared_inter % 32 == 0 && L.shared_inter <= 2048;
    return ok;
}
// Everything moe_prefill_gguf / moe_prefill_gguf_fast would build lazily for one layer, built NOW on `st`.  The whole-model prompt pass calls this for every
// MoE layer on its main stream before the start event (kr_decode_prefill.cpp prefill_impl): with several chunks in flight on several streams, a lazy build
// queued on chunk c's stream is not ordered before chunk c + 1's GEMMs on another stream.
int kr_moe_prefill_prepare(kr_engine* e, int layer, int fast, int routed_only, hipStream_t st) {
    Layer& L = e->layers[layer];
    if (!L.gguf) return KR_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    const bool use_shared = L.gguf_shared && !routed_only;
    if (fast && gg_fast_ok(L, e->cfg.hidden_size, use_shared)) {
        if (int rc = gg_ensure_fast(e, L.g_gate, &L.g_up, st)) return rc;
        if (int rc = gg_ensure_fast(e, L.g_down, nullptr, st)) return rc;
        if (use_shared) { if (int rc = gg_ensure_fast(e, L.gs_gate, &L.gs_up, st)) return rc; if (int rc = gg_ensure_fast(e, L.gs_down, nullptr, st)) return rc; }
        return KR_OK;
    }
    bool mfma = kr_gpf_type_supported(L.g_gate.type, e->cfg.hidden_size) && kr_gpf_type_supported(L.g_down.type, L.inter);
    if (use_shared) mfma = mfma && kr_gpf_type_supported(L.gs_gate.type, e->cfg.hidden_size) && kr_gpf_type_supported(L.gs_down.type, L.shared_inter);
    if (!mfma) return KR_OK;
    for (GgufSet* g : {&L.g_gate, &L.g_up, &L.g_down}) if (int rc = gg_ensure_ws(e, *g, st)) return rc;
    if (use_shared) for (GgufSet* g : {&L.gs_gate, &L.gs_up, &L.gs_down}) if (int rc = gg_ensure_ws(e, *g, st)) return rc;
    return KR_OK;
}
// the prompt pass of a native Q4_K layer in the tolerance form: f16 rows (+ their per-32 sums) x nibbles de-quantized in registers with the
// sub-block scale folded in, offsets as extra k-columns, libm SiLU like expert_forward_gguf (gguf_kernels.rs:690-756).  Same sort / combine as the exact path.
static int moe_prefill_gguf_fast(kr_engine* e, Layer& L, const void* x_bf16, const int32_t* ids, const float* wts, void* out, int M, int topk,
                                 int out_dtype, int routed_only, int set, hipStream_t st) {
    kr_engine::PfSet& P = e->pf[set % KR_PF_MAX_DEPTH];
    const int H = e->cfg.hidden_size, I = L.inter, E = e->cfg.n_routed_experts;
    const bool use_shared = L.gguf_shared && !routed_only;
    const int SI = L.shared_inter;
    if (int rc = gg_ensure_fast(e, L.g_gate, &L.g_up, st)) return rc;
    if (int rc = gg_ensure_fast(e, L.g_down, nullptr, st)) return rc;
    if (use_shared) { if (int rc = gg_ensure_fast(e, L.gs_gate, &L.gs_up, st)) return rc; if (int rc = gg_ensure_fast(e, L.gs_down, nullptr, st)) return rc; }
    const int pairs = e->pf_pairs > 0 ? e->pf_pairs : KR_PF_PAIRS;
    const int CHmax = pairs / topk > 64 ? pairs / topk : 64;
    const int CH = M < CHmax ? M : CHmax;
    const size_t np = (size_t)CH * topk;
    const int max_tiles = (int)(np / 64) + E + 1;
    const size_t n_i32 = 3 * (size_t)E + 3 * (size_t)max_tiles + 4 + 2 * np;
    if ((size_t)CH * H * 2 >= (1ull << 32) || np * I * 2 >= (1ull << 32))
        return kr_fail(KR_ERR_VALUE, "tolerance GEMM: %zu rows x %d values per pass exceed 4 GiB of f16 activations; lower kr_moe_set_prefill_pairs", np, H > I ? H : I);
    if (P.i32.ensure(n_i32 * 4) || P.xf.ensure((size_t)CH * H * 2) || P.xfm.ensure((size_t)CH * 4) || P.xs.ensure((size_t)CH * (H / 32) * 2) || P.gu.ensure(np * 2 * I * 4) ||
        P.hf.ensure(np * I * 2) || P.hfm.ensure(np * 4) || P.hs.ensure(np * (I / 32) * 2) || P.eo.ensure(np * H * 4))
        return kr_fail(KR_ERR_HIP, "hipMalloc of prefill scratch failed");
    if (use_shared && (P.sgu.ensure((size_t)CH * 2 * SI * 4) || P.shf.ensure((size_t)CH * SI * 2) || P.shfm.ensure((size_t)CH * 4) || P.shs.ensure((size_t)CH * (SI / 32) * 2) ||
                       P.seo.ensure((size_t)CH * H * 4)))
        return kr_fail(KR_ERR_HIP, "hipMalloc of prefill scratch failed");
    int* ib = (int*)P.i32.p;
    KrPfSort so{};
    so.counts = ib; so.offsets = ib + E; so.cursor = ib + 2 * E; ib += 3 * E;
    so.tile_expert = ib; so.tile_row0 = ib + max_tiles; so.tile_rows = ib + 2 * max_tiles; ib += 3 * max_tiles;
    so.n_tiles = ib; ib += 4; so.row_pair = ib; so.pair_row = ib + np;
    const size_t ob = out_dtype == KR_OUT_BF16 ? 2 : 4;
    const KrMatDev w13 = L.g_gate.fast_view(), w2 = L.g_down.fast_view();
    for (int m0 = 0; m0 < M; m0 += CH) {
        const int mc = M - m0 < CH ? M - m0 : CH;
        const uint16_t* xc = (const uint16_t*)x_bf16 + (size_t)m0 * H;
        const int32_t* idc = ids + (size_t)m0 * topk; const float* wc = wts + (size_t)m0 * topk;
        const int tiles_bound = (mc * topk) / 64 + E + 1;
        int run = (int)(((long)mc * topk / E + 63) / 64);
        run = run < 1 ? 1 : (run > 4 ? 4 : run);
        kr_launch_pf_sort(idc, mc, topk, E, so, st);
        kr_launch_pfh_rows_bf16(xc, mc, H, H, (uint16_t*)P.xf.p, (float*)P.xfm.p, st, (uint16_t*)P.xs.p);
        kr_launch_pfh_w13_act(w13, (const uint16_t*)P.xf.p, (const float*)P.xfm.p, &so, topk, 1, tiles_bound, 0, (float*)P.gu.p, mc * topk, 3 /* libm SiLU */, 0.0f, 0.0f,
                              (uint16_t*)P.hf.p, (float*)P.hfm.p, st, run, (const uint16_t*)P.xs.p, (uint16_t*)P.hs.p);
        kr_launch_pfh_gemm(w2, (const uint16_t*)P.hf.p, (const float*)P.hfm.p, &so, topk, 0, tiles_bound, 0, (float*)P.eo.p, H, st, 0, 2 /* f16 rows */, run, (const uint16_t*)P.hs.p);
        if (use_shared) {
            kr_launch_pfh_gemm(L.gs_gate.fast_view(), (const uint16_t*)P.xf.p, (const float*)P.xfm.p, nullptr, topk, 0, 0, mc, (float*)P.sgu.p, 2 * SI, st, 0, 0, 1, (const uint16_t*)P.xs.p);
            kr_launch_pfh_act((const float*)P.sgu.p, mc, SI, 2 * SI, 3, 0.0f, 0.0f, (uint16_t*)P.shf.p, (float*)P.shfm.p, st, (uint16_t*)P.shs.p);
            kr_launch_pfh_gemm(L.gs_down.fast_view(), (const uint16_t*)P.shf.p, (const float*)P.shfm.p, nullptr, topk, 0, 0, mc, (float*)P.seo.p, H, st, 0, 0, 1, (const uint16_t*)P.shs.p);
        }
        kr_launch_pf_combine_f16rows((const uint16_t*)P.eo.p, (const float*)P.hfm.p, so.pair_row, wc, mc, topk, H, use_shared ? (const float*)P.seo.p : nullptr, e->cfg.routed_scaling_factor,
                                     (char*)out + (size_t)m0 * H * ob, out_dtype == KR_OUT_BF16, st);
    }
    KR_HIP(hipGetLastError());
    return KR_OK;
}

static int moe_prefill_gguf(kr_engine* e, Layer& L, int layer, const void* x_bf16, const int32_t* ids, const float* wts, void* out, int M, int topk,
                            int out_dtype, int routed_only, int set, hipStream_t st) {
    const int H = e->cfg.hidden_size, I = L.inter, E = e->cfg.n_routed_experts;
    const bool use_shared = L.gguf_shared && !routed_only;
    const int SI = L.shared_inter;
    bool mfma = kr_gpf_type_supported(L.g_gate.type, H) && kr_gpf_type_supported(L.g_down.type, I);
    if (use_shared) mfma = mfma && kr_gpf_type_supported(L.gs_gate.type, H) && kr_gpf_type_supported(L.gs_down.type, SI);
    const size_t ob = out_dtype == KR_OUT_BF16 ? 2 : 4;
    if (!mfma) {   // streaming kernels, 16 384 tokens per pass (grid z limit 65 535)
        for (int m0 = 0; m0 < M; m0 += 16384) {
            const int mc = M - m0 < 16384 ? M - m0 : 16384;
            if (int rc = moe_forward_locked(e, layer, (const uint16_t*)x_bf16 + (size_t)m0 * H, ids + (size_t)m0 * topk, wts + (size_t)m0 * topk,
                                            (char*)out + (size_t)m0 * H * ob, mc, topk, out_dtype, routed_only, st)) return rc;
        }
        return KR_OK;
    }
    kr_engine::PfSet& P = e->pf[set % KR_PF_MAX_DEPTH];
    for (GgufSet* g : {&L.g_gate, &L.g_up, &L.g_down}) if (int rc = gg_ensure_ws(e, *g, st)) return rc;
    if (use_shared) for (GgufSet* g : {&L.gs_gate, &L.gs_up, &L.gs_down}) if (int rc = gg_ensure_ws(e, *g, st)) return rc;
    const int pairs = e->pf_pairs > 0 ? e->pf_pairs : KR_PF_PAIRS;
    const int CHmax = pairs / topk > 64 ? pairs / topk : 64;
    const int CH = M < CHmax ? M : CHmax;
    const size_t np = (size_t)CH * topk;
    const int max_tiles = (int)(np / 64) + E + 1;
    const size_t n_i32 = 3 * (size_t)E + 3 * (size_t)max_tiles + 4 + 2 * np;
    if (P.i32.ensure(n_i32 * 4) || P.xh.ensure((size_t)CH * H) || P.xl.ensure((size_t)CH * H) || P.xs.ensure((size_t)CH * (H / 32) * 4) || P.xm.ensure((size_t)CH * (H / 32) * 4) ||
        P.gu.ensure(np * 2 * I * 4) || P.hh.ensure(np * I) || P.hl.ensure(np * I) || P.hs.ensure(np * (I / 32) * 4) || P.hm.ensure(np * (I / 32) * 4) || P.eo.ensure(np * H * 4))
        return kr_fail(KR_ERR_HIP, "hipMalloc of prefill scratch failed");
    if (use_shared && (P.sgu.ensure((size_t)CH * 2 * SI * 4) || P.shh.ensure((size_t)CH * SI) || P.shl.ensure((size_t)CH * SI) || P.shs.ensure((size_t)CH * (SI / 32) * 4) ||
                       P.shm.ensure((size_t)CH * (SI / 32) * 4) || P.seo.ensure((size_t)CH * H * 4)))
        return kr_fail(KR_ERR_HIP, "hipMalloc of prefill scratch failed");
    int* ib = (int*)P.i32.p;
    KrPfSort so{};
    so.counts = ib; so.offsets = ib + E; so.cursor = ib + 2 * E; ib += 3 * E;
    so.tile_expert = ib; so.tile_row0 = ib + max_tiles; so.tile_rows = ib + 2 * max_tiles; ib += 3 * max_tiles;
    so.n_tiles = ib; ib += 4; so.row_pair = ib; so.pair_row = ib + np;
    for (int m0 = 0; m0 < M; m0 += CH) {
        const int mc = M - m0 < CH ? M - m0 : CH;
        const uint16_t* xc = (const uint16_t*)x_bf16 + (size_t)m0 * H;
        const int32_t* idc = ids + (size_t)m0 * topk; const float* wc = wts + (size_t)m0 * topk;
        const int tiles_bound = (mc * topk) / 64 + E + 1;
        kr_launch_pf_sort(idc, mc, topk, E, so, st);
        kr_launch_gpf_quant_x(xc, mc, H, (int8_t*)P.xh.p, (int8_t*)P.xl.p, (float*)P.xs.p, (float*)P.xm.p, st);
        kr_launch_gpf_gemm(L.g_gate.view(), L.g_gate.ws.p, L.g_gate.ws_stride, (const int8_t*)P.xh.p, (const int8_t*)P.xl.p, (const float*)P.xs.p, (const float*)P.xm.p, &so,
                           topk, 1, tiles_bound, 0, (float*)P.gu.p, 2 * I, 0, st);
        kr_launch_gpf_gemm(L.g_up.view(), L.g_up.ws.p, L.g_up.ws_stride, (const int8_t*)P.xh.p, (const int8_t*)P.xl.p, (const float*)P.xs.p, (const float*)P.xm.p, &so,
                           topk, 1, tiles_bound, 0, (float*)P.gu.p, 2 * I, I, st);
        kr_launch_gpf_act((const float*)P.gu.p, mc * topk, I, 2 * I, (int8_t*)P.hh.p, (int8_t*)P.hl.p, (float*)P.hs.p, (float*)P.hm.p, st);
        kr_launch_gpf_gemm(L.g_down.view(), L.g_down.ws.p, L.g_down.ws_stride, (const int8_t*)P.hh.p, (const int8_t*)P.hl.p, (const float*)P.hs.p, (const float*)P.hm.p, &so,
                           topk, 0, tiles_bound, 0, (float*)P.eo.p, H, 0, st);
        if (use_shared) {
            kr_launch_gpf_gemm(L.gs_gate.view(), L.gs_gate.ws.p, L.gs_gate.ws_stride, (const int8_t*)P.xh.p, (const int8_t*)P.xl.p, (const float*)P.xs.p, (const float*)P.xm.p,
                               nullptr, topk, 0, 0, mc, (float*)P.sgu.p, 2 * SI, 0, st);
            kr_launch_gpf_gemm(L.gs_up.view(), L.gs_up.ws.p, L.gs_up.ws_stride, (const int8_t*)P.xh.p, (const int8_t*)P.xl.p, (const float*)P.xs.p, (const float*)P.xm.p,
                               nullptr, topk, 0, 0, mc, (float*)P.sgu.p, 2 * SI, SI, st);
            kr_launch_gpf_act((const float*)P.sgu.p, mc, SI, 2 * SI, (int8_t*)P.shh.p, (int8_t*)P.shl.p, (float*)P.shs.p, (float*)P.shm.p, st);
            kr_launch_gpf_gemm(L.gs_down.view(), L.gs_down.ws.p, L.gs_down.ws_stride, (const int8_t*)P.shh.p, (const int8_t*)P.shl.p, (const float*)P.shs.p,
                               (const float*)P.shm.p, nullptr, topk, 0, 0, mc, (float*)P.seo.p, H, 0, st);
        }
        kr_launch_pf_combine((const float*)P.eo.p, so.pair_row, wc, mc, topk, H, use_shared ? (const float*)P.seo.p : nullptr, e->cfg.routed_scaling_factor,
                             (char*)out + (size_t)m0 * H * ob, out_dtype == KR_OUT_BF16, st);
    }
    KR_HIP(hipGetLastError());
    return KR_OK;
}

// kr_moe_prefill in the tolerance form (kr_moe_set_gemm_mode(e, 1)): same launches, the activations travel as f16 rows and the GEMMs run on the
// f16 matrix cores with f32 accumulation over the whole k range (kr_prefill_h.hip).  Sort, combine, routing weights and skip rules are the exact path's.
static int moe_prefill_fast(kr_engine* e, Layer& L, const void* x_bf16, const int32_t* ids, const float* wts, void* out, int M, int topk,
                            int out_dtype, int routed_only, int set, hipStream_t st) {
    kr_engine::PfSet& P = e->pf[set % KR_PF_MAX_DEPTH];
    const int H = e->cfg.hidden_size, I = L.inter, E = e->cfg.n_routed_experts;
    const bool use_shared = L.shared_present && !routed_only;
    const int SI = L.shared_inter;
    const int pairs = e->pf_pairs > 0 ? e->pf_pairs : KR_PF_PAIRS;
    const int CHmax = pairs / topk > 64 ? pairs / topk : 64;
    const int CH = M < CHmax ? M : CHmax;
    const size_t np = (size_t)CH * topk;
    const int max_tiles = (int)(np / 64) + E + 1;
    const size_t n_i32 = 3 * (size_t)E + 3 * (size_t)max_tiles + 4 + 2 * np;
    // the tolerance GEMM addresses its A rows with 32-bit byte offsets (kr_prefill_h.hip): a pass holds at most 4 GiB of f16 rows
    if ((size_t)CH * H * 2 >= (1ull << 32) || np * I * 2 >= (1ull << 32))
        return kr_fail(KR_ERR_VALUE, "tolerance GEMM: %zu rows x %d values per pass exceed 4 GiB of f16 activations; lower kr_moe_set_prefill_pairs", np, H > I ? H : I);
    if (P.i32.ensure(n_i32 * 4) || P.xf.ensure((size_t)CH * H * 2) || P.xfm.ensure((size_t)CH * 4) || P.gu.ensure(np * 2 * I * 4) || P.hf.ensure(np * I * 2) ||
        P.hfm.ensure(np * 4) || P.eo.ensure(np * H * 4))
        return kr_fail(KR_ERR_HIP, "hipMalloc of prefill scratch failed");
    if (use_shared && (P.sgu.ensure((size_t)CH * 2 * SI * 4) || P.shf.ensure((size_t)CH * SI * 2) || P.shfm.ensure((size_t)CH * 4) || P.seo.ensure((size_t)CH * H * 4)))
        return kr_fail(KR_ERR_HIP, "hipMalloc of prefill scratch failed");
    int* ib = (int*)P.i32.p;
    KrPfSort so{};
    so.counts = ib; so.offsets = ib + E; so.cursor = ib + 2 * E; ib += 3 * E;
    so.tile_expert = ib; so.tile_row0 = ib + max_tiles; so.tile_rows = ib + 2 * max_tiles; ib += 3 * max_tiles;
    so.n_tiles = ib; ib += 4; so.row_pair = ib; so.pair_row = ib + np;
    const int act_mode = e->cfg.swiglu_limit > 0.0f ? KR_ACT_GPTOSS : KR_ACT_SILU_FUSED;
    const size_t ob = out_dtype == KR_OUT_BF16 ? 2 : 4;
    for (int m0 = 0; m0 < M; m0 += CH) {
        const int mc = M - m0 < CH ? M - m0 : CH;
        const uint16_t* xc = (const uint16_t*)x_bf16 + (size_t)m0 * H;
        const int32_t* idc = ids + (size_t)m0 * topk; const float* wc = wts + (size_t)m0 * topk;
        const int tiles_bound = (mc * topk) / 64 + E + 1;
        int run = (int)(((long)mc * topk / E + 63) / 64);           // row tiles of an average expert: they share an XCD (one L2 fill of its weights)
        run = run < 1 ? 1 : (run > 4 ? 4 : run);
        kr_launch_pf_sort(idc, mc, topk, E, so, st);
        kr_launch_pfh_rows_bf16(xc, mc, H, H, (uint16_t*)P.xf.p, (float*)P.xfm.p, st);
        kr_launch_pfh_w13_act(L.w13.view(), (const uint16_t*)P.xf.p, (const float*)P.xfm.p, &so, topk, 1, tiles_bound, 0, (float*)P.gu.p, mc * topk, act_mode, e->cfg.swiglu_limit,
                              e->cfg.activation_alpha, (uint16_t*)P.hf.p, (float*)P.hfm.p, st, run);
        kr_launch_pfh_gemm(L.w2.view(), (const uint16_t*)P.hf.p, (const float*)P.hfm.p, &so, topk, 0, tiles_bound, 0, (float*)P.eo.p, H, st, 0, 2 /* f16 rows */, run);
        if (use_shared) {
            kr_launch_pfh_gemm(L.sw13.view(), (const uint16_t*)P.xf.p, (const float*)P.xfm.p, nullptr, topk, 0, 0, mc, (float*)P.sgu.p, 2 * SI, st);
            kr_launch_pfh_act((const float*)P.sgu.p, mc, SI, 2 * SI, act_mode, e->cfg.swiglu_limit, e->cfg.activation_alpha, (uint16_t*)P.shf.p, (float*)P.shfm.p, st);
            kr_launch_pfh_gemm(L.sw2.view(), (const uint16_t*)P.shf.p, (const float*)P.shfm.p, nullptr, topk, 0, 0, mc, (float*)P.seo.p, H, st);
        }
        kr_launch_pf_combine_f16rows((const uint16_t*)P.eo.p, (const float*)P.hfm.p, so.pair_row, wc, mc, topk, H, use_shared ? (const float*)P.seo.p : nullptr, e->cfg.routed_scaling_factor,
                                     (char*)out + (size_t)m0 * H * ob, out_dtype == KR_OUT_BF16, st);
    }
    KR_HIP(hipGetLastError());
    return KR_OK;
}

int kr_moe_prefill_set(kr_engine* e, int layer, const void* x_bf16, const int32_t* ids, const float* wts, void* out, int M, int topk,
                       int out_dtype, int routed_only, int set, hipStream_t st) {
    if (int rc = check_layer(e, layer)) return rc;
    if (!x_bf16 || !ids || !wts || !out) return kr_fail(KR_ERR_VALUE, "null pointer argument");
    if (M <= 0) return kr_fail(KR_ERR_VALUE, "M must be > 0");
    if (topk <= 0 || topk > KR_MAX_TOPK) return kr_fail(KR_ERR_VALUE, "topk %d exceeds MAX_TOPK %d", topk, KR_MAX_TOPK);
    Layer& L = e->layers[layer];
    if (!L.w13.allocated() && !L.gguf) return kr_fail(KR_ERR_STATE, "Model not loaded -- call load() first (layer %d has no experts)", layer);
    if (!is_device_ptr(x_bf16) || !is_device_ptr(ids) || !is_device_ptr(wts) || !is_device_ptr(out))
        return kr_fail(KR_ERR_VALUE, "kr_moe_prefill expects device pointers (hidden/topk tensors live in HBM during prefill)");
    std::lock_guard<std::mutex> lk(e->mu);
    KR_HIP(hipSetDevice(e->device));
    const bool fast = e->gemm_fast || (set & KR_PF_SET_FAST);
    set &= 0xFF;
    if (L.gguf) {      // native GGUF blocks: the exact int8-MFMA form, or (tolerance mode, Q4_K layers) f16 MFMA on the re-tiled copy
        if (fast && gg_fast_ok(L, e->cfg.hidden_size, L.gguf_shared && !routed_only)) return moe_prefill_gguf_fast(e, L, x_bf16, ids, wts, out, M, topk, out_dtype, routed_only, set, st);
        return moe_prefill_gguf(e, L, layer, x_bf16, ids, wts, out, M, topk, out_dtype, routed_only, set, st);
    }
    if (e->cfg.hidden_size % 128 || L.inter % 128) return kr_fail(KR_ERR_VALUE, "prefill path needs dims divisible by 128");
    if (fast && (!L.shared_present || routed_only || L.shared_inter % 128 == 0))
        return moe_prefill_fast(e, L, x_bf16, ids, wts, out, M, topk, out_dtype, routed_only, set, st);
    kr_engine::PfSet& P = e->pf[set % KR_PF_MAX_DEPTH];
    const int H = e->cfg.hidden_size, I = L.inter, E = e->cfg.n_routed_experts;
    const bool use_shared = L.shared_present && !routed_only;
    const int SI = L.shared_inter;
    if (int rc = kr_ensure_wsum(e, L.w13, st)) return rc;
    if (int rc = kr_ensure_wsum(e, L.w2, st)) return rc;
    if (use_shared) { if (int rc = kr_ensure_wsum(e, L.sw13, st)) return rc; if (int rc = kr_ensure_wsum(e, L.sw2, st)) return rc; }
    const int pairs = e->pf_pairs > 0 ? e->pf_pairs : KR_PF_PAIRS;
    const int CHmax = pairs / topk > 64 ? pairs / topk : 64;
    const int CH = M < CHmax ? M : CHmax;
    const size_t np = (size_t)CH * topk;
    const int max_tiles = (int)(np / 64) + E + 1;
    const size_t n_i32 = 3 * (size_t)E + 3 * (size_t)max_tiles + 4 + 2 * np;
    if (P.i32.ensure(n_i32 * 4) || P.xh.ensure((size_t)CH * H) || P.xl.ensure((size_t)CH * H) || P.xs.ensure((size_t)CH * (H / 128) * 4) ||
        P.gu.ensure(np * 2 * I * 4) || P.hh.ensure(np * I) || P.hl.ensure(np * I) || P.hs.ensure(np * (I / 128) * 4) || P.eo.ensure(np * H * 4))
        return kr_fail(KR_ERR_HIP, "hipMalloc of prefill scratch failed");
    if (use_shared && (P.sgu.ensure((size_t)CH * 2 * SI * 4) || P.shh.ensure((size_t)CH * SI) || P.shl.ensure((size_t)CH * SI) ||
                       P.shs.ensure((size_t)CH * (SI / 128) * 4) || P.seo.ensure((size_t)CH * H * 4)))
        return kr_fail(KR_ERR_HIP, "hipMalloc of prefill scratch failed");
    int* ib = (int*)P.i32.p;
    KrPfSort so{};
    so.counts = ib; so.offsets = ib + E; so.cursor = ib + 2 * E; ib += 3 * E;
    so.tile_expert = ib; so.tile_row0 = ib + max_tiles; so.tile_rows = ib + 2 * max_tiles; ib += 3 * max_tiles;
    so.n_tiles = ib; ib += 4; so.row_pair = ib; so.pair_row = ib + np;
    const int act_mode = e->cfg.swiglu_limit > 0.0f ? KR_ACT_GPTOSS : KR_ACT_SILU_FUSED;
    const size_t ob = out_dtype == KR_OUT_BF16 ? 2 : 4;
    for (int m0 = 0; m0 < M; m0 += CH) {
        const int mc = M - m0 < CH ? M - m0 : CH;
        const uint16_t* xc = (const uint16_t*)x_bf16 + (size_t)m0 * H;
        const int32_t* idc = ids + (size_t)m0 * topk; const float* wc = wts + (size_t)m0 * topk;
        const int tiles_bound = (mc * topk) / 64 + E + 1;
        kr_launch_pf_sort(idc, mc, topk, E, so, st);
        kr_launch_pf_quant_x(xc, mc, H, (int8_t*)P.xh.p, (int8_t*)P.xl.p, (float*)P.xs.p, st);
        const int var_rows = (long)mc * topk < 48L * E;     // fewer than ~48 rows per expert: most 64-row tiles have an empty second half
        kr_launch_pf_gemm(L.w13.view(), (const uint32_t*)L.w13.wsum.p, (const int8_t*)P.xh.p, (const int8_t*)P.xl.p, (const float*)P.xs.p, &so, topk, 1,
                          tiles_bound, 0, (float*)P.gu.p, 2 * I, st, 0, 0, var_rows);
        kr_launch_pf_act((const float*)P.gu.p, mc * topk, I, 2 * I, act_mode, e->cfg.swiglu_limit, e->cfg.activation_alpha, (int8_t*)P.hh.p, (int8_t*)P.hl.p,
                         (float*)P.hs.p, st);
        kr_launch_pf_gemm(L.w2.view(), (const uint32_t*)L.w2.wsum.p, (const int8_t*)P.hh.p, (const int8_t*)P.hl.p, (const float*)P.hs.p, &so, topk, 0,
                          tiles_bound, 0, (float*)P.eo.p, H, st, 0, 0, var_rows);
        if (use_shared) {
            kr_launch_pf_gemm(L.sw13.view(), (const uint32_t*)L.sw13.wsum.p, (const int8_t*)P.xh.p, (const int8_t*)P.xl.p, (const float*)P.xs.p, nullptr, topk, 0,
                              0, mc, (float*)P.sgu.p, 2 * SI, st);
            kr_launch_pf_act((const float*)P.sgu.p, mc, SI, 2 * SI, act_mode, e->cfg.swiglu_limit, e->cfg.activation_alpha, (int8_t*)P.shh.p, (int8_t*)P.shl.p,
                             (float*)P.shs.p, st);
            kr_launch_pf_gemm(L.sw2.view(), (const uint32_t*)L.sw2.wsum.p, (const int8_t*)P.shh.p, (const int8_t*)P.shl.p, (const float*)P.shs.p, nullptr, topk, 0,
                              0, mc, (float*)P.seo.p, H, st);
        }
        kr_launch_pf_combine((const float*)P.eo.p, so.pair_row, wc, mc, topk, H, use_shared ? (const float*)P.seo.p : nullptr, e->cfg.routed_scaling_factor,
                             (char*)out + (size_t)m0 * H * ob, out_dtype == KR_OUT_BF16, st);
    }
    KR_HIP(hipGetLastError());
    return KR_OK;
}

// Expert-parallel receive side (kr_ep.cpp): `n` rows, each for ONE local expert `lid[r]` (all valid), weight 1, no shared expert.  The rows are
// sorted by expert like a prompt pass with topk = 1, but the w2 GEMM writes row r's result straight to out[r] (caller's order) in the dtype the
// rows travel back in: no expert-output buffer, no combine pass, no conversion pass.  The bf16 rounding is the one kr_launch_ep_rows_bf16 applied.
int kr_moe_prefill_rows(kr_engine* e, int layer, const void* rows_bf16, const int32_t* lid, void* out, int n, int out_bf16, int set, hipStream_t st) {
    if (int rc = check_layer(e, layer)) return rc;
    Layer& L = e->layers[layer];
    if (L.gguf || !L.w13.allocated() || e->cfg.hidden_size % 128 || L.inter % 128) return -1;     // caller falls back to kr_moe_prefill_set
    std::lock_guard<std::mutex> lk(e->mu);
    KR_HIP(hipSetDevice(e->device));
    kr_engine::PfSet& P = e->pf[set % KR_PF_MAX_DEPTH];
    const int H = e->cfg.hidden_size, I = L.inter, E = e->cfg.n_routed_experts;
    if (int rc = kr_ensure_wsum(e, L.w13, st)) return rc;
    if (int rc = kr_ensure_wsum(e, L.w2, st)) return rc;
    const int pairs = e->pf_pairs > 0 ? e->pf_pairs : KR_PF_PAIRS;
    const int CH = n < pairs ? n : pairs;
    const size_t np = (size_t)CH;
    const int max_tiles = (int)(np / 64) + E + 1;
    const size_t n_i32 = 3 * (size_t)E + 3 * (size_t)max_tiles + 4 + 2 * np;
    if (P.i32.ensure(n_i32 * 4) || P.xh.ensure((size_t)CH * H) || P.xl.ensure((size_t)CH * H) || P.xs.ensure((size_t)CH * (H / 128) * 4) ||
        P.gu.ensure(np * 2 * I * 4) || P.hh.ensure(np * I) || P.hl.ensure(np * I) || P.hs.ensure(np * (I / 128) * 4))
        return kr_fail(KR_ERR_HIP, "hipMalloc of prefill scratch failed");
    int* ib = (int*)P.i32.p;
    KrPfSort so{};
    so.counts = ib; so.offsets = ib + E; so.cursor = ib + 2 * E; ib += 3 * E;
    so.tile_expert = ib; so.tile_row0 = ib + max_tiles; so.tile_rows = ib + 2 * max_tiles; ib += 3 * max_tiles;
    so.n_tiles = ib; ib += 4; so.row_pair = ib; so.pair_row = ib + np;
    const int act_mode = e->cfg.swiglu_limit > 0.0f ? KR_ACT_GPTOSS : KR_ACT_SILU_FUSED;
    const size_t ob = out_bf16 ? 2 : 4;
    for (int m0 = 0; m0 < n; m0 += CH) {
        const int mc = n - m0 < CH ? n - m0 : CH;
        const int tiles_bound = mc / 64 + E + 1;
        kr_launch_pf_sort(lid + m0, mc, 1, E, so, st);
        kr_launch_pf_quant_x((const uint16_t*)rows_bf16 + (size_t)m0 * H, mc, H, (int8_t*)P.xh.p, (int8_t*)P.xl.p, (float*)P.xs.p, st);
        const int var_rows = (long)mc < 48L * E;
        kr_launch_pf_gemm(L.w13.view(), (const uint32_t*)L.w13.wsum.p, (const int8_t*)P.xh.p, (const int8_t*)P.xl.p, (const float*)P.xs.p, &so, 1, 1,
                          tiles_bound, 0, (float*)P.gu.p, 2 * I, st, 0, 0, var_rows);
        kr_launch_pf_act((const float*)P.gu.p, mc, I, 2 * I, act_mode, e->cfg.swiglu_limit, e->cfg.activation_alpha, (int8_t*)P.hh.p, (int8_t*)P.hl.p, (float*)P.hs.p, st);
        kr_launch_pf_gemm(L.w2.view(), (const uint32_t*)L.w2.wsum.p, (const int8_t*)P.hh.p, (const int8_t*)P.hl.p, (const float*)P.hs.p, &so, 1, 0,
                          tiles_bound, 0, (float*)((char*)out + (size_t)m0 * H * ob), H, st, 1, out_bf16, var_rows);
    }
    KR_HIP(hipGetLastError());
    return KR_OK;
}

extern "C" int kr_moe_prefill(kr_engine* e, int layer, const void* x_bf16, const int32_t* ids, const float* wts, void* out, int M, int topk,
                              int out_dtype, int routed_only, void* stream) {
    if (int rc = check_layer(e, layer)) return rc;
    return kr_moe_prefill_set(e, layer, x_bf16, ids, wts, out, M, topk, out_dtype, routed_only, 0, kr_pick_stream(e, stream));
}

// test / tuning hook: (token, slot) pairs per pass of kr_moe_prefill (0 = default 81 920)
extern "C" int kr_moe_set_prefill_pairs(kr_engine* e, int pairs) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (pairs < 0) return kr_fail(KR_ERR_VALUE, "pairs must be >= 0, got %d", pairs);
    e->pf_pairs = pairs;
    return KR_OK;
}

// numerics of the prompt-pass expert GEMMs (kr_moe_prefill, kr_decode_prefill): 0 = the CPU engine's arithmetic, bit-identical to kr_moe_forward
// (default); 1 = tolerance form (f16 activations, f32 accumulation over the whole k range -- the dataflow of the reference's GPU prompt pass)
extern "C" int kr_moe_set_gemm_mode(kr_engine* e, int fast) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (fast != 0 && fast != 1 && fast != 3 && fast != 5) return kr_fail(KR_ERR_VALUE, "gemm mode %d unknown (0 = exact, 1 = fast, 3 = fast on the register-staged kernels only, 5 = fast with the ring kernel for every shape it takes)", fast);
    std::lock_guard<std::mutex> lk(e->mu);
    e->gemm_fast = fast & 1;
    if (fast) kr_pfr_set_enabled(fast == 1 ? 1 : (fast == 5 ? 2 : 0));      // 3: A/B and test hook, process-wide -- the LDS-ring GEMM (kr_prefill_ring.hip) off; results are bit-identical either way
    return KR_OK;
}

// Expert-parallel combine (krasis_amd/ep.py): rows returned by the owning ranks are summed per token in routing order with the routing
// weights, exactly as the single-GPU combine does (moe.rs:661-667).  eo_rows f32 [n_rows,H]; pair_row i32 [M,k] (-1 = skipped slot).
extern "C" int kr_combine_rows(kr_engine* e, const float* eo_rows, const int32_t* pair_row, const float* wts, void* out, int M, int topk,
                               int out_dtype, void* stream) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (!is_device_ptr(eo_rows) || !is_device_ptr(pair_row) || !is_device_ptr(wts) || !is_device_ptr(out)) return kr_fail(KR_ERR_VALUE, "kr_combine_rows expects device pointers");
    if (e->cfg.hidden_size % 4) return kr_fail(KR_ERR_VALUE, "kr_combine_rows needs hidden_size %% 4 == 0 (got %d)", e->cfg.hidden_size);
    KR_HIP(hipSetDevice(e->device));
    kr_launch_pf_combine(eo_rows, pair_row, wts, M, topk, e->cfg.hidden_size, nullptr, 1.0f, out, out_dtype == KR_OUT_BF16, kr_pick_stream(e, stream));
    KR_HIP(hipGetLastError());
    return KR_OK;
}
