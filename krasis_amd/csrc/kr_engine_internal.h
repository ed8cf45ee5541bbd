// kr_engine_internal.h -- engine state shared by kr_engine.cpp and kr_decode.cpp (not part of the C ABI)
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/krasis_hip.h"
#include "kr_kernels.h"
#include "kr_gguf.h"

#define KR_PF_MAX_DEPTH 8
int kr_fail(int code, const char* fmt, ...);
#define KR_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) return kr_fail(KR_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)

// number of device allocations made by the library so far (process-wide).  bench.py reads it on both sides of every timed region: an allocation
// (hipFree + hipMalloc synchronise the device and can take seconds for tens of GB) inside a timed region is a measurement defect.
inline std::atomic<long>& kr_alloc_count() { static std::atomic<long> n{0}; return n; }

struct DevBuf {     // owning device allocation: released by the destructor, so a new scratch member cannot be forgotten in a release list
    void* p = nullptr; size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
    ~DevBuf() { release(); }
    int ensure(size_t n) {
        if (n <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        kr_alloc_count().fetch_add(1, std::memory_order_relaxed);      // every device allocation of the library passes here: kr_debug_alloc_count()
        if (hipMalloc(&p, n) != hipSuccess) return 1;
        bytes = n; return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct MatSet {            // all experts of one layer for one projection, contiguous in HBM
    DevBuf q, s;
    DevBuf wsum;           // prefill: per (group, column) sum of (nibble-8), i16 pairs in the scale-pair layout (built on first use)
    bool wsum_counted = false;   // its bytes are added to weight_bytes once, however often it is rebuilt
    int K = 0, N = 0, bits = 0, count = 0;
    size_t q_stride = 0, s_stride = 0;
    bool allocated() const { return q.p != nullptr; }
    KrMatDev view() const {
        KrMatDev m{};
        m.q = q.p; m.s = (const uint32_t*)s.p; m.K = K; m.N = N;
        m.ng = K / 128; m.ngp = (m.ng + 1) / 2; m.bits = bits; m.n_fma = (N / 8) * 8;
        m.q_stride = q_stride; m.s_stride = s_stride;
        return m;
    }
};

struct GgufSet {          // native GGUF experts of one layer for one projection
    DevBuf q, h; int type = 0, K = 0, N = 0, count = 0; size_t q_stride = 0, h_stride = 0;
    DevBuf ws; size_t ws_stride = 0;   // prompt pass: per (row, sub-block) sums of the quants (kr_gguf_prefill.hip), built on first use
    // KR_GEMM_FAST on Q4_K / Q8_0 layers: operand form of the tolerance GEMM (kr_gq_repack_kernel / kr_gq8_repack_kernel), built on first use.  gate and up share ONE copy (N = 2 I,
    // held by the gate set) so that the gate | up GEMM is one launch like the INT4 path's w13.
    DevBuf fq, fqs, fqo; size_t fq_stride = 0, fqs_stride = 0; int fN = 0;
    KrMatDev fast_view() const {
        KrMatDev m{};
        m.q = fq.p; m.K = K; m.N = fN; m.ng = K / 128; m.ngp = (m.ng + 1) / 2; m.bits = type == GG_Q8_0 ? 8 : 4; m.n_fma = (fN / 8) * 8; m.q_stride = fq_stride;
        m.qs = (const uint16_t*)fqs.p; m.qo = (const uint16_t*)fqo.p; m.qs_stride = fqs_stride;       // Q8_0: no offset table (qo == nullptr)
        return m;
    }
    bool allocated() const { return q.p != nullptr; }
    GgMat view() const { GgMat m{}; m.q = q.p; m.h = h.p; m.type = type; m.K = K; m.N = N; m.q_stride = q_stride; m.h_stride = h_stride; return m; }
};

struct Layer {
    GgufSet g_gate, g_up, g_down, gs_gate, gs_up, gs_down;   // native GGUF blocks (has_gguf)
    bool gguf = false, gguf_shared = false;
    MatSet w13, w2;        // routed experts
    MatSet sw13, sw2;      // shared expert (count == 1)
    std::vector<uint8_t> present;
    bool shared_present = false;
    int inter = 0, shared_inter = 0;
    // routing (set_routing_weights)
    std::vector<float> gate_host;          // [E,H] f32 copy (bf16 inputs widen exactly)
    bool gate_bf16_exact = false;          // every value representable in bf16 -> stored as bf16 in HBM
    DevBuf gate_cm;                        // chain-major layout for rule DECODE
    DevBuf gate_rm;                        // row-lane layout for rule ENGINE (built on first use)
    DevBuf gate_row;                       // plain row-major [E][H] (bf16 when exact, else f32) for the MFMA logits of the prompt pass (built on first use)
    DevBuf bias, esc; bool has_bias = false, has_esc = false, routing_present = false;
};

struct kr_engine {
    int device = 0;
    kr_model_config cfg{};
    hipStream_t stream = nullptr;
    std::vector<Layer> layers;
    size_t weight_bytes = 0;
    // scratch
    DevBuf gu, eo, st_act, st_ids, st_w, st_out, ptr_table;
    // routing config
    bool routing_set = false; int r_scoring = 1, r_norm = 1, r_topk = 0, r_ne = 0, r_hidden = 0;
    DevBuf r_logits, r_ids, r_w, r_x;
    // prefill scratch (kr_moe_prefill)
    int pf_pairs = 0;          // kr_moe_set_prefill_pairs
    int gemm_fast = 0;         // kr_moe_set_gemm_mode: prompt-pass expert GEMMs in the tolerance form (kr_prefill_h.hip)
    struct PfSet { DevBuf i32, xh, xl, xs, xm, gu, hh, hl, hs, hm, eo, sgu, shh, shl, shs, shm, seo, xf, xfm, hf, hfm, shf, shfm; } pf[KR_PF_MAX_DEPTH];   // one set per chunk in flight of the prompt pass
    // per-kernel profiling (kr_set_profiling): HIP events around each launch, accumulated per kernel kind
    bool prof = false; hipEvent_t pev[4] = {nullptr, nullptr, nullptr, nullptr}; double prof_ms[8] = {0}; long prof_n[8] = {0};
    std::mutex mu;
    uint64_t ep_generation = 0;         // bumped by kr_ep_init* / kr_ep_destroy: a decode store drops its captured graph and its eager warm-up count when it changes
    struct kr_ep_state* ep = nullptr;   // expert parallelism (kr_ep.cpp): communicator, exchange buffers
};
extern "C" int kr_ep_destroy(kr_engine* e);
// kr_ep.cpp, for the decode graph's expert-parallel step: ranks of the group (1 without expert parallelism), this rank's expert slice [lo, hi) in
// GLOBAL ids and the offset to subtract for the engine-local index (0 when the engine holds the whole model), in-place f32 sum over the ranks
int kr_ep_world(const kr_engine* e);
void kr_ep_slice(const kr_engine* e, int* lo, int* hi, int* sub);
int kr_ep_allreduce_on(kr_engine* e, float* buf_dev, size_t n, hipStream_t st);
bool kr_ep_decode_active(const kr_engine* e);   // world > 1, or a one-rank RCCL communicator (bring-up): the decode step takes its expert-parallel form
bool kr_ep_is_rccl(const kr_engine* e);          // the transport is an RCCL communicator (its collectives are stream operations: capturable in a hipGraph)
int kr_ep_rank(const kr_engine* e);
// kr_moe_prefill_ep with an explicit exchange-buffer set (one per prompt-pass chunk in flight, 0 .. KR_PF_MAX_DEPTH - 1)
int kr_moe_prefill_ep_set(kr_engine* e, int layer, const void* x_bf16, const int32_t* ids, const float* wts, void* out, int M, int topk, int out_dtype,
                          int routed_only, int set, void* stream);

bool is_device_ptr(const void* p);
kr_engine* kr_engine_new_bare(int device);   // device + stream only (a decode store created before its MoE engine)
// ABI stream convention: NULL = the engine's own stream, (void*)1 = the legacy default (null) stream, else a hipStream_t
static inline hipStream_t kr_pick_stream(kr_engine* e, void* stream);
int matset_alloc(kr_engine* e, MatSet& ms, int K, int N, int bits, int count);
int upload_mat(kr_engine* e, MatSet& ms, int idx, const void* w, const uint16_t* sc);
int download_mat(kr_engine* e, MatSet& ms, int idx, void* w, uint16_t* sc);
// kr_moe_prefill with an explicit scratch set (0/1) and stream; all pointers device.  out f32 or bf16 per out_dtype.
int kr_ensure_gate_row(kr_engine* e, int layer);   // kr_engine.cpp: uploads Layer::gate_row on first use (synchronous copy: call before enqueueing the pass)
int kr_moe_prefill_rows(kr_engine* e, int layer, const void* rows_bf16, const int32_t* lid, void* out, int n, int out_bf16, int set, hipStream_t st);
#define KR_PF_SET_FAST 0x100   // or-ed into `set`: this call takes the tolerance GEMMs whatever kr_moe_set_gemm_mode says (the decode store's KR_GEMM_FAST)
int kr_moe_prefill_set(kr_engine* e, int layer, const void* x_bf16, const int32_t* ids, const float* wts, void* out, int M, int topk,
                       int out_dtype, int routed_only, int set, hipStream_t st);

static inline hipStream_t kr_pick_stream(kr_engine* e, void* stream) {
    if (!stream) return e->stream;
    if (stream == (void*)1) return (hipStream_t)0;
    return (hipStream_t)stream;
}
