// kr_ep.cpp -- expert parallelism inside the library, over RCCL (north_star: "experts shard by expert-parallel RCCL all-to-all over
// xGMI across the 8 GPUs of one node").
//
// Reference dataflow (python/krasis/gpu_prefill.py:353-359, :4140-4148, :4467-4469; python/krasis/model.py:3131-3241): rank r owns the contiguous
// expert slice [r * floor(E/R), ...) (last rank takes the remainder), tokens and routing are replicated, non-local ids are masked, every rank
// returns a partial sum and rank 0 adds them through a pinned-host bounce.  Here each rank owns a SHARD OF THE TOKENS and every (token, slot)
// row travels once to the rank that owns its expert and back -- xGMI is a full mesh, every peer pair has its own link, so the exchange is an
// all-to-all of point-to-point transfers (ncclSend / ncclRecv in one group), not a ring:
//
//   owner sort (kr_ep_dest + the prompt-pass token sort, "experts" = destination ranks)  ->  send counts all-gathered (world x world i32)  ->
//   rows gathered in destination order  ->  dispatch: rows bf16 [n, H] + local expert ids i32 [n]  ->  expert GEMMs on the received rows
//   (kr_moe_prefill_set, top-1 rows, f32)  ->  return: rows f32 (exact: the result equals single-GPU execution bit for bit) or bf16 (half the
//   bytes; one extra rounding per row)  ->  combine in routing order with the routing weights (moe.rs:661-667) [+ rsf * . + shared expert].
//
// The split sizes of ncclSend / ncclRecv are host integers, so ONE 4 * world^2-byte device-to-host copy per call is waited for (the only host
// synchronisation; it covers the sort kernels, not the previous layer's GEMMs when the caller alternates streams).  RCCL is bound at
// kr_ep_init by dlopen("librccl.so.1") -- the runtime a framework already loaded is reused, and single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <memory>
#include <vector>

#include "kr_engine_internal.h"
#include "kr_prefill.h"

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return KR_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return kr_fail(KR_ERR_STATE, "RCCL not available: %s", dlerror());
#define KR_SYM(field, name) do { *(void**)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return kr_fail(KR_ERR_STATE, "librccl has no symbol %s", name); } while (0)
    KR_SYM(GetUniqueId, "ncclGetUniqueId"); KR_SYM(CommInitRank, "ncclCommInitRank"); KR_SYM(CommDestroy, "ncclCommDestroy");
    KR_SYM(AllGather, "ncclAllGather"); KR_SYM(Send, "ncclSend"); KR_SYM(Recv, "ncclRecv"); KR_SYM(GroupStart, "ncclGroupStart");
    KR_SYM(GroupEnd, "ncclGroupEnd"); KR_SYM(GetErrorString, "ncclGetErrorString");
#undef KR_SYM
    g_rccl.h = h;
    return KR_OK;
}
#define KR_NCCL(call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) return kr_fail(KR_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r__)); } while (0)
}  // namespace

struct kr_ep_state {
    int world = 1, rank = 0, E_total = 0, per = 0, ret_bf16 = 0, full = 0;   // full: the engine holds ALL experts of the model (a replica that can also decode): local id = global id
    ncclComm_t comm = nullptr;
    DevBuf dest, lid, i32, rows, row_lid, rrows, rlid, eo, eo16, back, ones, cnt_all, shared_out, neg_ids;
    int* cnt_host = nullptr;          // pinned [world * world]
    hipEvent_t ev = nullptr;
};

extern "C" int kr_ep_unique_id(void* id_out128) {
    if (!id_out128) return kr_fail(KR_ERR_VALUE, "null argument");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    KR_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out128, &id, sizeof id);
    return KR_OK;
}

// world ranks, this process is `rank`; the engine holds the experts of its slice as local experts 0 .. n_local-1 (cfg.n_routed_experts = n_local);
// n_experts_total = experts of the whole model.  id128 = kr_ep_unique_id of rank 0, carried to the other ranks by the host's own bootstrap
// (world == 1: may be NULL, no communicator is created).  return_bf16 != 0: expert rows come back as bf16 instead of f32.
extern "C" int kr_ep_init(kr_engine* e, int world, int rank, int n_experts_total, const void* id128, int return_bf16) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return kr_fail(KR_ERR_VALUE, "bad world / rank (%d / %d; at most 64 ranks)", world, rank);
    if (n_experts_total < world) return kr_fail(KR_ERR_VALUE, "%d experts cannot be split over %d ranks", n_experts_total, world);
    const int per = n_experts_total / world, n_local = rank == world - 1 ? n_experts_total - per * (world - 1) : per;
    if (e->cfg.n_routed_experts < n_local)      // an engine that holds the WHOLE model (>= n_experts_total experts) serves its slice under the global ids; otherwise experts 0 .. n_local-1 are the slice
        return kr_fail(KR_ERR_VALUE, "rank %d of %d owns %d of %d experts but the engine holds only %d", rank, world, n_local, n_experts_total, e->cfg.n_routed_experts);
    if (e->ep) return kr_fail(KR_ERR_STATE, "expert parallelism is already initialised");
    KR_HIP(hipSetDevice(e->device));
    std::unique_ptr<kr_ep_state> s(new kr_ep_state);
    s->world = world; s->rank = rank; s->E_total = n_experts_total; s->per = per; s->ret_bf16 = return_bf16 != 0;
    s->full = e->cfg.n_routed_experts >= n_experts_total && world > 0 ? 1 : 0;
    if (world > 1) {
        if (!id128) return kr_fail(KR_ERR_VALUE, "kr_ep_init needs the unique id of rank 0 when world > 1");
        if (int rc = load_rccl()) return rc;
        ncclUniqueId id; memcpy(&id, id128, sizeof id);
        KR_NCCL(g_rccl.CommInitRank(&s->comm, world, id, rank));
    }
    KR_HIP(hipHostMalloc((void**)&s->cnt_host, sizeof(int) * world * world, hipHostMallocDefault));
    KR_HIP(hipEventCreateWithFlags(&s->ev, hipEventDisableTiming));
    e->ep = s.release();
    return KR_OK;
}

extern "C" int kr_ep_destroy(kr_engine* e) {
    if (!e || !e->ep) return KR_OK;
    kr_ep_state* s = e->ep;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    if (s->comm) (void)g_rccl.CommDestroy(s->comm);
    for (DevBuf* b : {&s->dest, &s->lid, &s->i32, &s->rows, &s->row_lid, &s->rrows, &s->rlid, &s->eo, &s->eo16, &s->back, &s->ones, &s->cnt_all, &s->shared_out, &s->neg_ids}) b->release();
    if (s->cnt_host) (void)hipHostFree(s->cnt_host);
    if (s->ev) (void)hipEventDestroy(s->ev);
    delete s; e->ep = nullptr;
    return KR_OK;
}

// x bf16 [M, H] (this rank's token shard), ids i32 [M, topk] GLOBAL expert ids (-1 = skip), w f32 [M, topk]; out bf16 / f32 [M, H] =
// the single-GPU kr_moe_prefill result of the same tokens.  routed_only == 0 adds rsf * routed + shared with this rank's shared expert.
extern "C" int kr_moe_prefill_ep(kr_engine* e, int layer, const void* x_bf16, const int32_t* ids, const float* wts, void* out, int M, int topk, int out_dtype,
                                 int routed_only, void* stream) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (!e->ep) return kr_fail(KR_ERR_STATE, "call kr_ep_init first");
    if (layer < 0 || layer >= (int)e->layers.size()) return kr_fail(KR_ERR_VALUE, "moe_layer_idx %d out of range", layer);
    if (!x_bf16 || !ids || !wts || !out || M <= 0) return kr_fail(KR_ERR_VALUE, "bad arguments");
    if (topk <= 0 || topk > KR_MAX_TOPK) return kr_fail(KR_ERR_VALUE, "topk %d exceeds MAX_TOPK %d", topk, KR_MAX_TOPK);
    if (!is_device_ptr(x_bf16) || !is_device_ptr(ids) || !is_device_ptr(wts) || !is_device_ptr(out)) return kr_fail(KR_ERR_VALUE, "kr_moe_prefill_ep expects device pointers");
    kr_ep_state* s = e->ep;
    Layer& L = e->layers[layer];
    KR_HIP(hipSetDevice(e->device));
    hipStream_t st = kr_pick_stream(e, stream);
    const int H = e->cfg.hidden_size, W = s->world, np = M * topk;
    const bool use_shared = !routed_only && (L.shared_present || L.gguf_shared);
    // ---- owner sort: destination rank of every pair, rows grouped by destination (the prompt-pass sort with "experts" = ranks)
    const int max_tiles = np / 64 + W + 1;
    const size_t n_i32 = 3 * (size_t)W + 3 * (size_t)max_tiles + 4 + 2 * (size_t)np;
    if (s->dest.ensure((size_t)np * 4) || s->lid.ensure((size_t)np * 4) || s->i32.ensure(n_i32 * 4) || s->rows.ensure((size_t)np * H * 2) || s->row_lid.ensure((size_t)np * 4) ||
        s->cnt_all.ensure((size_t)W * W * 4))
        return kr_fail(KR_ERR_HIP, "hipMalloc of the expert-parallel scratch failed");
    int* ib = (int*)s->i32.p;
    KrPfSort so{};
    so.counts = ib; so.offsets = ib + W; so.cursor = ib + 2 * W; ib += 3 * W;
    so.tile_expert = ib; so.tile_row0 = ib + max_tiles; so.tile_rows = ib + 2 * max_tiles; ib += 3 * max_tiles;
    so.n_tiles = ib; ib += 4; so.row_pair = ib; so.pair_row = ib + np;
    kr_launch_ep_dest(ids, np, s->E_total, s->per, W, s->full, (int32_t*)s->dest.p, (int32_t*)s->lid.p, st);
    kr_launch_ep_sort((const int32_t*)s->dest.p, np, W, so, st);
    kr_launch_ep_gather((const uint16_t*)x_bf16, so.row_pair, (const int32_t*)s->lid.p, topk, H, so.n_tiles + 1, np, (uint16_t*)s->rows.p, (int32_t*)s->row_lid.p, st);
    // ---- send counts of every rank: cnt[src][dst]
    // The host needs the split sizes to post the sends / receives: one small DtoH and an event wait per call.  A world of one posts nothing:
    // it takes every pair slot as a row (rows past the last routed pair carry local id -1 and belong to no expert tile) and never waits.
    std::vector<size_t> soff(W + 1, 0), roff(W + 1, 0);
    if (W > 1) {
        KR_NCCL(g_rccl.AllGather(so.counts, s->cnt_all.p, W, ncclInt32, s->comm, st));
        KR_HIP(hipMemcpyAsync(s->cnt_host, s->cnt_all.p, (size_t)W * W * 4, hipMemcpyDeviceToHost, st));
        KR_HIP(hipEventRecord(s->ev, st));
        KR_HIP(hipEventSynchronize(s->ev));
        for (int r = 0; r < W; r++) { soff[r + 1] = soff[r] + (size_t)s->cnt_host[s->rank * W + r]; roff[r + 1] = roff[r] + (size_t)s->cnt_host[r * W + s->rank]; }
    } else { soff[1] = roff[1] = (size_t)np; }
    const size_t n_send = soff[W], n_recv = roff[W];
    // ---- dispatch
    const uint16_t* rrows = (const uint16_t*)s->rows.p; const int32_t* rlid = (const int32_t*)s->row_lid.p;
    if (W > 1) {
        if (s->rrows.ensure((n_recv ? n_recv : 1) * H * 2) || s->rlid.ensure((n_recv ? n_recv : 1) * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc of the receive buffers failed");
        KR_NCCL(g_rccl.GroupStart());
        for (int r = 0; r < W; r++) {
            const size_t sc = soff[r + 1] - soff[r], rc = roff[r + 1] - roff[r];
            if (sc) { KR_NCCL(g_rccl.Send((const uint16_t*)s->rows.p + soff[r] * H, sc * H, ncclBfloat16, r, s->comm, st));
                      KR_NCCL(g_rccl.Send((const int32_t*)s->row_lid.p + soff[r], sc, ncclInt32, r, s->comm, st)); }
            if (rc) { KR_NCCL(g_rccl.Recv((uint16_t*)s->rrows.p + roff[r] * H, rc * H, ncclBfloat16, r, s->comm, st));
                      KR_NCCL(g_rccl.Recv((int32_t*)s->rlid.p + roff[r], rc, ncclInt32, r, s->comm, st)); }
        }
        KR_NCCL(g_rccl.GroupEnd());
        rrows = (const uint16_t*)s->rrows.p; rlid = (const int32_t*)s->rlid.p;
    }
    // ---- experts on the received rows (top-1 rows, weight 1, f32)
    const size_t n_ones = n_recv > (size_t)M ? n_recv : (size_t)M;
    if (s->eo.ensure((n_recv ? n_recv : 1) * (size_t)H * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc of the expert row buffer failed");
    if (s->ones.bytes < n_ones * 4) {
        if (s->ones.ensure(n_ones * 4 * 2)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        KR_HIP(hipMemsetD32Async((hipDeviceptr_t)s->ones.p, 0x3F800000, s->ones.bytes / 4, st));      // f32 1.0
    }
    // the w2 GEMM writes every row at its place, in the dtype it travels back in (kr_moe_prefill_rows); native-GGUF layers take the generic
    // prompt-pass entry (combine with weight 1 = a copy) and a conversion pass
    const void* back = s->eo.p;
    if (n_recv) {
        const int rc = kr_moe_prefill_rows(e, layer, rrows, rlid, s->eo.p, (int)n_recv, s->ret_bf16 ? 1 : 0, 0, st);
        if (rc > 0) return rc;
        if (rc < 0) {
            if (int rc2 = kr_moe_prefill_set(e, layer, rrows, rlid, (const float*)s->ones.p, s->eo.p, (int)n_recv, 1, KR_OUT_F32, 1, 0, st)) return rc2;
            if (s->ret_bf16) {
                if (s->eo16.ensure(n_recv * (size_t)H * 2)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
                kr_launch_ep_rows_bf16((const float*)s->eo.p, (uint16_t*)s->eo16.p, n_recv * (size_t)H, st);
                back = s->eo16.p;
            }
        }
    }
    if (W > 1) {
        const size_t esz = s->ret_bf16 ? 2 : 4;
        if (s->back.ensure((n_send ? n_send : 1) * (size_t)H * esz)) return kr_fail(KR_ERR_HIP, "hipMalloc of the return buffer failed");
        const ncclDataType_t dt = s->ret_bf16 ? ncclBfloat16 : ncclFloat32;
        KR_NCCL(g_rccl.GroupStart());
        for (int r = 0; r < W; r++) {
            const size_t sc = roff[r + 1] - roff[r], rc = soff[r + 1] - soff[r];
            if (sc) KR_NCCL(g_rccl.Send((const char*)back + roff[r] * H * esz, sc * H, dt, r, s->comm, st));
            if (rc) KR_NCCL(g_rccl.Recv((char*)s->back.p + soff[r] * H * esz, rc * H, dt, r, s->comm, st));
        }
        KR_NCCL(g_rccl.GroupEnd());
        back = s->back.p;
    }
    // ---- shared expert of this rank's tokens (replicated weights), then the combine in routing order
    const float* shared_eo = nullptr;
    if (use_shared) {   // one all-skipped slot per token: kr_moe_prefill then returns rsf * 0 + shared = the shared expert's rows
        if (s->shared_out.ensure((size_t)M * H * 4) || s->neg_ids.ensure((size_t)M * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        KR_HIP(hipMemsetAsync(s->neg_ids.p, 0xFF, (size_t)M * 4, st));
        if (int rc = kr_moe_prefill_set(e, layer, x_bf16, (const int32_t*)s->neg_ids.p, (const float*)s->ones.p, s->shared_out.p, M, 1, KR_OUT_F32, 0, 1, st)) return rc;
        shared_eo = (const float*)s->shared_out.p;
    }
    if (s->ret_bf16) kr_launch_pf_combine_bf16rows((const uint16_t*)back, so.pair_row, wts, M, topk, H, shared_eo, e->cfg.routed_scaling_factor, out, out_dtype == KR_OUT_BF16, st);
    else kr_launch_pf_combine((const float*)back, so.pair_row, wts, M, topk, H, shared_eo, e->cfg.routed_scaling_factor, out, out_dtype == KR_OUT_BF16, st);
    KR_HIP(hipGetLastError());
    return KR_OK;
}
