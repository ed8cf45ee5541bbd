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
//
// TRANSPORT.  The exchange is two primitives behind a small table (KrEpTransport): `gather_counts` (every rank contributes `world` ints, every rank
// receives all world x world) and `exchange` (a personalised all-to-all of row blocks described by per-peer offsets).  Two implementations:
//   * RCCL       -- one process per GPU: ncclAllGather + grouped ncclSend / ncclRecv on the caller's stream (kr_ep_init);
//   * loopback   -- W virtual ranks = W engines of ONE process on one device (kr_ep_loopback_create / kr_ep_init_loopback): every rank runs in its own
//                   host thread, publishes its buffers in a shared group object, the peers pull them with hipMemcpyAsync after an event wait, and
//                   host barriers bracket each step.  It exists so that the offset arithmetic below (split sizes, per-peer offsets, the receive-side
//                   scatter) runs at W = 2, 3 (remainder slice), 8 on a single-GPU box: tests/test_ep_gpu.py asserts bit-identity with one engine.
// A rank whose token shard is EMPTY (M == 0) still takes part in every collective with zero-sized blocks -- returning early would leave its peers
// blocked in the all-gather.  Failures after the first collective of a call abort the communicator (ncclCommAbort) so that the peers get an error
// instead of waiting forever.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

// the handful of RCCL declarations this file needs (the library itself is bound with dlopen at kr_ep_init): no <rccl/rccl.h> at build time
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclHalf = 6,
               ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8, ncclBfloat16 = 9 } ncclDataType_t;
}

#include "kr_engine_internal.h"
#include "kr_prefill.h"

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, int /* ncclRedOp_t: ncclSum = 0 */, ncclComm_t, hipStream_t) = nullptr;
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
    KR_SYM(AllGather, "ncclAllGather"); KR_SYM(AllReduce, "ncclAllReduce"); KR_SYM(Send, "ncclSend"); KR_SYM(Recv, "ncclRecv"); KR_SYM(GroupStart, "ncclGroupStart");
    KR_SYM(GroupEnd, "ncclGroupEnd"); KR_SYM(GetErrorString, "ncclGetErrorString"); KR_SYM(CommAbort, "ncclCommAbort"); KR_SYM(CommCount, "ncclCommCount");
#undef KR_SYM
    g_rccl.h = h;
    return KR_OK;
}
#define KR_NCCL(call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) return kr_fail(KR_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r__)); } while (0)
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// loopback group: W virtual ranks in one process (one host thread each)
// ---------------------------------------------------------------------------------------------------------------------------------
struct kr_ep_loop_group {
    int world = 0;
    int attached = 0;      // engines initialised on this group and not yet destroyed (guarded by mu): kr_ep_loopback_destroy refuses while > 0
    std::mutex mu; std::condition_variable cv; int arrived = 0; long gen = 0; bool failed = false;
    int failed_rank = -1; std::string failed_why;      // first failure: which rank gave up and why (the peers' error text names it)
    struct Slot { const void* const* send = nullptr; const size_t* off = nullptr; const void* buf = nullptr; hipEvent_t ready = nullptr; };
    Slot slots[64];
    // generation barrier over the W rank threads; false = a rank failed (everybody gives up instead of waiting)
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const long g = gen;
        if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g || failed; });
        return !failed;
    }
    void fail(int rank = -1, const char* why = nullptr) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed_rank = rank; failed_why = why ? why : ""; }
        failed = true; cv.notify_all();
    }
    std::string why() { std::lock_guard<std::mutex> lk(mu); return "rank " + std::to_string(failed_rank) + (failed_why.empty() ? std::string() : ": " + failed_why); }
};

struct kr_ep_state {
    int world = 1, rank = 0, E_total = 0, per = 0, ret_bf16 = 0, full = 0;   // full: the engine holds ALL experts of the model (a replica that can also decode): local id = global id
    ncclComm_t comm = nullptr;           // RCCL transport
    kr_ep_loop_group* loop = nullptr;    // loopback transport
    bool broken = false;                 // a collective failed: every later call is refused
    // exchange buffers of one prompt-pass chunk in flight (kr_decode_prefill runs up to KR_PF_MAX_DEPTH chunks on as many streams: every chunk brings
    // its own set, so that the rows of chunk c + 1 can be sorted / gathered / sent while chunk c's experts still run)
    struct Bufs {
        DevBuf dest, lid, i32, rows, row_lid, rrows, rlid, eo, eo16, back, ones, cnt_all, shared_out, neg_ids;
        int* cnt_host = nullptr;          // pinned [world * world]
        hipEvent_t ev = nullptr;
    };
    Bufs sets[KR_PF_MAX_DEPTH];
    DevBuf red;                           // loopback all-reduce staging
    int ensure_set(int i) {               // pinned split-size block + event of set i (created on first use)
        Bufs& b = sets[i];
        if (!b.cnt_host) KR_HIP(hipHostMalloc((void**)&b.cnt_host, sizeof(int) * world * world, hipHostMallocDefault));
        if (!b.ev) KR_HIP(hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
        return KR_OK;
    }
};

namespace {
// after a failure past the first collective of a call the peers must not be left waiting: tear the communicator down
int ep_abort(kr_ep_state* s, int rc) {
    s->broken = true;
    if (s->comm && g_rccl.CommAbort) { (void)g_rccl.CommAbort(s->comm); s->comm = nullptr; }
    if (s->loop) s->loop->fail(s->rank, kr_last_error());
    return rc;
}

// ---- transport primitive 1: every rank contributes `world` ints, every rank receives world x world (row r = rank r's contribution)
int ep_gather_counts(kr_ep_state* s, const int* mine_dev, int* all_dev, hipStream_t st) {
    const int W = s->world;
    if (s->comm) { KR_NCCL(g_rccl.AllGather(mine_dev, all_dev, W, ncclInt32, s->comm, st)); return KR_OK; }
    kr_ep_loop_group* g = s->loop;
    g->slots[s->rank].buf = mine_dev;
    KR_HIP(hipEventRecord(g->slots[s->rank].ready, st));
    if (!g->barrier()) return kr_fail(KR_ERR_STATE, "loopback expert parallelism: a peer rank failed (%s)", g->why().c_str());
    for (int p = 0; p < W; p++) {
        KR_HIP(hipStreamWaitEvent(st, g->slots[p].ready, 0));
        KR_HIP(hipMemcpyAsync(all_dev + (size_t)p * W, g->slots[p].buf, (size_t)W * 4, hipMemcpyDeviceToDevice, st));
    }
    KR_HIP(hipStreamSynchronize(st));
    if (!g->barrier()) return kr_fail(KR_ERR_STATE, "loopback expert parallelism: a peer rank failed (%s)", g->why().c_str());     // every pull is done: the sources may be reused
    return KR_OK;
}

// ---- transport primitive 2: personalised all-to-all of row blocks.  Buffer b holds rows of row_bytes[b] bytes; rows [soff[r], soff[r+1]) of the
// send buffers go to rank r, rows [roff[r], roff[r+1]) of the receive buffers come from rank r (soff / roff in rows, shared by the n buffers).
int ep_exchange(kr_ep_state* s, int nbuf, const void* const* send, void* const* recv, const size_t* row_bytes, const size_t* soff, const size_t* roff, hipStream_t st) {
    const int W = s->world;
    if (s->comm) {
        KR_NCCL(g_rccl.GroupStart());
        for (int r = 0; r < W; r++) {
            const size_t sc = soff[r + 1] - soff[r], rc = roff[r + 1] - roff[r];
            for (int b = 0; b < nbuf; b++) {
                if (sc) KR_NCCL(g_rccl.Send((const char*)send[b] + soff[r] * row_bytes[b], sc * row_bytes[b], ncclInt8, r, s->comm, st));
                if (rc) KR_NCCL(g_rccl.Recv((char*)recv[b] + roff[r] * row_bytes[b], rc * row_bytes[b], ncclInt8, r, s->comm, st));
            }
        }
        KR_NCCL(g_rccl.GroupEnd());
        return KR_OK;
    }
    kr_ep_loop_group* g = s->loop;
    g->slots[s->rank].send = send; g->slots[s->rank].off = soff;
    KR_HIP(hipEventRecord(g->slots[s->rank].ready, st));
    if (!g->barrier()) return kr_fail(KR_ERR_STATE, "loopback expert parallelism: a peer rank failed (%s)", g->why().c_str());
    for (int p = 0; p < W; p++) {
        const size_t* po = g->slots[p].off;
        const size_t n = po[s->rank + 1] - po[s->rank];
        if (n != roff[p + 1] - roff[p]) return kr_fail(KR_ERR_STATE, "loopback exchange: rank %d sends %zu rows to rank %d, which expects %zu", p, n, s->rank, roff[p + 1] - roff[p]);
        if (!n) continue;
        KR_HIP(hipStreamWaitEvent(st, g->slots[p].ready, 0));
        for (int b = 0; b < nbuf; b++)
            KR_HIP(hipMemcpyAsync((char*)recv[b] + roff[p] * row_bytes[b], (const char*)g->slots[p].send[b] + po[s->rank] * row_bytes[b], n * row_bytes[b], hipMemcpyDeviceToDevice, st));
    }
    KR_HIP(hipStreamSynchronize(st));
    if (!g->barrier()) return kr_fail(KR_ERR_STATE, "loopback expert parallelism: a peer rank failed (%s)", g->why().c_str());
    return KR_OK;
}

// ---- transport primitive 3: in-place sum of n f32 over the ranks, in rank order (decode: every element is non-zero on exactly one rank, so the
// order does not matter there -- the sum is exact)
int ep_allreduce_f32(kr_engine* e, float* buf, size_t n, hipStream_t st) {
    kr_ep_state* s = e->ep;
    const int W = s->world;
    if (s->comm) { KR_NCCL(g_rccl.AllReduce(buf, buf, n, ncclFloat32, 0, s->comm, st)); return KR_OK; }
    if (W == 1) return KR_OK;
    kr_ep_loop_group* g = s->loop;
    if (s->red.ensure((size_t)W * n * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc of the all-reduce staging failed");
    g->slots[s->rank].buf = buf;
    KR_HIP(hipEventRecord(g->slots[s->rank].ready, st));
    if (!g->barrier()) return kr_fail(KR_ERR_STATE, "loopback expert parallelism: a peer rank failed (%s)", g->why().c_str());
    for (int p = 0; p < W; p++) {
        KR_HIP(hipStreamWaitEvent(st, g->slots[p].ready, 0));
        KR_HIP(hipMemcpyAsync((float*)s->red.p + (size_t)p * n, g->slots[p].buf, n * 4, hipMemcpyDeviceToDevice, st));
    }
    KR_HIP(hipStreamSynchronize(st));
    if (!g->barrier()) return kr_fail(KR_ERR_STATE, "loopback expert parallelism: a peer rank failed (%s)", g->why().c_str());     // all pulls done before anybody overwrites its buffer
    kr_launch_ep_sum_f32((const float*)s->red.p, W, n, buf, st);
    return KR_OK;
}
}  // namespace

extern "C" int kr_ep_unique_id(void* id_out128) {
    if (!id_out128) return kr_fail(KR_ERR_VALUE, "null argument");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    KR_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out128, &id, sizeof id);
    return KR_OK;
}

static int ep_init_common(kr_engine* e, int world, int rank, int n_experts_total, int return_bf16, std::unique_ptr<kr_ep_state>& s) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return kr_fail(KR_ERR_VALUE, "bad world / rank (%d / %d; at most 64 ranks)", world, rank);
    if (n_experts_total < world) return kr_fail(KR_ERR_VALUE, "%d experts cannot be split over %d ranks", n_experts_total, world);
    const int per = n_experts_total / world, n_local = rank == world - 1 ? n_experts_total - per * (world - 1) : per;
    if (e->cfg.n_routed_experts < n_local)      // an engine that holds the WHOLE model (>= n_experts_total experts) serves its slice under the global ids; otherwise experts 0 .. n_local-1 are the slice
        return kr_fail(KR_ERR_VALUE, "rank %d of %d owns %d of %d experts but the engine holds only %d", rank, world, n_local, n_experts_total, e->cfg.n_routed_experts);
    if (e->ep) return kr_fail(KR_ERR_STATE, "expert parallelism is already initialised");
    KR_HIP(hipSetDevice(e->device));
    s.reset(new kr_ep_state);
    s->world = world; s->rank = rank; s->E_total = n_experts_total; s->per = per; s->ret_bf16 = return_bf16 != 0;
    s->full = e->cfg.n_routed_experts >= n_experts_total && world > 0 ? 1 : 0;
    return KR_OK;
}
static int ep_init_finish(kr_engine* e, std::unique_ptr<kr_ep_state>& s) {
    if (int rc = s->ensure_set(0)) return rc;
    e->ep = s.release(); e->ep_generation++;
    return KR_OK;
}

// world ranks, this process is `rank`; the engine holds the experts of its slice as local experts 0 .. n_local-1 (cfg.n_routed_experts = n_local);
// n_experts_total = experts of the whole model.  id128 = kr_ep_unique_id of rank 0, carried to the other ranks by the host's own bootstrap
// (world == 1: may be NULL, no communicator is created).  return_bf16 != 0: expert rows come back as bf16 instead of f32.
extern "C" int kr_ep_init(kr_engine* e, int world, int rank, int n_experts_total, const void* id128, int return_bf16) {
    std::unique_ptr<kr_ep_state> s;
    if (int rc = ep_init_common(e, world, rank, n_experts_total, return_bf16, s)) return rc;
    if (world > 1 || id128) {      // world == 1 WITH an id: a one-rank RCCL communicator -- every collective then goes through librccl (bring-up aid: the RCCL
                                   // transport, its grouped self send / recv and the captured all-reduce can be exercised on a single-GPU box)
        if (!id128) return kr_fail(KR_ERR_VALUE, "kr_ep_init needs the unique id of rank 0 when world > 1");
        if (int rc = load_rccl()) return rc;
        ncclUniqueId id; memcpy(&id, id128, sizeof id);
        KR_NCCL(g_rccl.CommInitRank(&s->comm, world, id, rank));
    }
    return ep_init_finish(e, s);
}

// ranks the communicator of this engine spans, as RCCL counts them (ncclCommCount); 1 without a communicator, the group size under loopback
extern "C" int kr_ep_comm_ranks(kr_engine* e, int* n_out) {
    if (!e || !e->ep || !n_out) return kr_fail(KR_ERR_VALUE, "kr_ep_comm_ranks: no expert-parallel state");
    *n_out = e->ep->loop ? e->ep->loop->world : 1;
    if (e->ep->comm) KR_NCCL(g_rccl.CommCount(e->ep->comm, n_out));
    return KR_OK;
}

// ---- loopback transport: W engines of this process (usually on one device), one host thread per rank while a collective call is in flight
extern "C" int kr_ep_loopback_create(int world, kr_ep_loop_group** out) {
    if (!out || world < 1 || world > 64) return kr_fail(KR_ERR_VALUE, "kr_ep_loopback_create: world %d out of range [1, 64]", world);
    std::unique_ptr<kr_ep_loop_group> g(new kr_ep_loop_group);
    g->world = world;
    for (int r = 0; r < world; r++) KR_HIP(hipEventCreateWithFlags(&g->slots[r].ready, hipEventDisableTiming));
    *out = g.release();
    return KR_OK;
}
// Engines attached with kr_ep_init_loopback hold a pointer to the group: destroying it under them would leave that pointer dangling, so the
// call is refused (non-zero, the group stays valid) until every attached engine went through kr_ep_destroy / kr_engine_destroy.
extern "C" int kr_ep_loopback_destroy(kr_ep_loop_group* g) {
    if (!g) return KR_OK;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->attached > 0) return kr_fail(KR_ERR_STATE, "kr_ep_loopback_destroy: %d engine(s) are still attached to the group (kr_ep_destroy them first)", g->attached);
    }
    for (int r = 0; r < g->world; r++) if (g->slots[r].ready) (void)hipEventDestroy(g->slots[r].ready);
    delete g;
    return KR_OK;
}
extern "C" int kr_ep_init_loopback(kr_engine* e, kr_ep_loop_group* group, int rank, int n_experts_total, int return_bf16) {
    if (!group) return kr_fail(KR_ERR_VALUE, "null loopback group");
    std::unique_ptr<kr_ep_state> s;
    if (int rc = ep_init_common(e, group->world, rank, n_experts_total, return_bf16, s)) return rc;
    s->loop = group;
    if (int rc = ep_init_finish(e, s)) return rc;
    { std::lock_guard<std::mutex> lk(group->mu); group->attached++; }
    return KR_OK;
}

extern "C" int kr_ep_destroy(kr_engine* e) {
    if (!e || !e->ep) return KR_OK;
    kr_ep_state* s = e->ep;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    if (s->comm) (void)g_rccl.CommDestroy(s->comm);
    if (s->loop) { std::lock_guard<std::mutex> lk(s->loop->mu); s->loop->attached--; }
    for (auto& b : s->sets) {      // the DevBufs release themselves with the state
        if (b.cnt_host) (void)hipHostFree(b.cnt_host);
        if (b.ev) (void)hipEventDestroy(b.ev);
    }
    delete s; e->ep = nullptr; e->ep_generation++;
    return KR_OK;
}

// every rank passes a value; max over the ranks comes back (a collective: all ranks must call).  Used by kr_decode_prefill to learn how many
// chunks the longest prompt shard has, so that ranks with fewer chunks keep taking part in the exchanges with empty shards.
extern "C" int kr_ep_max_int(kr_engine* e, int value, int* max_out, void* stream) {
    if (!e || !e->ep || !max_out) return kr_fail(KR_ERR_VALUE, "kr_ep_max_int: no expert-parallel state");
    kr_ep_state* s = e->ep;
    *max_out = value;
    if (s->world == 1) return KR_OK;
    if (s->broken) return kr_fail(KR_ERR_STATE, "expert parallelism: an earlier collective failed");
    KR_HIP(hipSetDevice(e->device));
    hipStream_t st = kr_pick_stream(e, stream);
    const int W = s->world;
    kr_ep_state::Bufs& B = s->sets[0];
    if (B.i32.ensure((size_t)W * 4) || B.cnt_all.ensure((size_t)W * W * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    for (int r = 0; r < W; r++) B.cnt_host[r] = value;
    KR_HIP(hipMemcpyAsync(B.i32.p, B.cnt_host, (size_t)W * 4, hipMemcpyHostToDevice, st));
    if (int rc = ep_gather_counts(s, (const int*)B.i32.p, (int*)B.cnt_all.p, st)) return ep_abort(s, rc);
    KR_HIP(hipMemcpyAsync(B.cnt_host, B.cnt_all.p, (size_t)W * W * 4, hipMemcpyDeviceToHost, st));
    KR_HIP(hipStreamSynchronize(st));
    for (int r = 0; r < W; r++) if (B.cnt_host[r * W] > *max_out) *max_out = B.cnt_host[r * W];
    return KR_OK;
}

// x bf16 [M, H] (this rank's token shard), ids i32 [M, topk] GLOBAL expert ids (-1 = skip), w f32 [M, topk]; out bf16 / f32 [M, H] =
// the single-GPU kr_moe_prefill result of the same tokens.  routed_only == 0 adds rsf * routed + shared with this rank's shared expert.
// M == 0 (an empty shard; pointers may be NULL) takes part in the exchanges with empty blocks and computes the rows its peers send.
// A collective call: every rank of the group calls with the same layer, in the same order.
int kr_moe_prefill_ep_set(kr_engine* e, int layer, const void* x_bf16, const int32_t* ids, const float* wts, void* out, int M, int topk, int out_dtype,
                          int routed_only, int set, void* stream) {
    if (!e) return kr_fail(KR_ERR_VALUE, "null engine");
    if (!e->ep) return kr_fail(KR_ERR_STATE, "call kr_ep_init first");
    kr_ep_state* s = e->ep;
    if (s->broken) return kr_fail(KR_ERR_STATE, "expert parallelism: an earlier collective failed");
    if (set < 0 || set >= KR_PF_MAX_DEPTH) return kr_fail(KR_ERR_VALUE, "exchange buffer set %d out of range", set);
    if (int rc = s->ensure_set(set)) return rc;
    kr_ep_state::Bufs& B = s->sets[set];
    // ---- everything that can fail for a local reason is checked BEFORE the first collective (a rank that bails out later would strand its peers)
    if (layer < 0 || layer >= (int)e->layers.size()) return kr_fail(KR_ERR_VALUE, "moe_layer_idx %d out of range", layer);
    if (M < 0 || (M > 0 && (!x_bf16 || !ids || !wts || !out))) return kr_fail(KR_ERR_VALUE, "bad arguments");
    if (topk <= 0 || topk > KR_MAX_TOPK) return kr_fail(KR_ERR_VALUE, "topk %d exceeds MAX_TOPK %d", topk, KR_MAX_TOPK);
    if (M > 0 && (!is_device_ptr(x_bf16) || !is_device_ptr(ids) || !is_device_ptr(wts) || !is_device_ptr(out))) return kr_fail(KR_ERR_VALUE, "kr_moe_prefill_ep expects device pointers");
    Layer& L = e->layers[layer];
    KR_HIP(hipSetDevice(e->device));
    hipStream_t st = kr_pick_stream(e, stream);
    const int H = e->cfg.hidden_size, W = s->world, np = M * topk;
    const bool use_shared = M > 0 && !routed_only && (L.shared_present || L.gguf_shared);
    // ---- owner sort: destination rank of every pair, rows grouped by destination (the prompt-pass sort with "experts" = ranks)
    const int max_tiles = np / 64 + W + 1;
    const size_t n_i32 = 3 * (size_t)W + 3 * (size_t)max_tiles + 4 + 2 * (size_t)np;
    const size_t np1 = np ? (size_t)np : 1;
    if (B.dest.ensure(np1 * 4) || B.lid.ensure(np1 * 4) || B.i32.ensure(n_i32 * 4) || B.rows.ensure(np1 * H * 2) || B.row_lid.ensure(np1 * 4) ||
        B.cnt_all.ensure((size_t)W * W * 4) || (use_shared && (B.shared_out.ensure((size_t)M * H * 4) || B.neg_ids.ensure((size_t)M * 4))))
        return kr_fail(KR_ERR_HIP, "hipMalloc of the expert-parallel scratch failed");
    int* ib = (int*)B.i32.p;
    KrPfSort so{};
    so.counts = ib; so.offsets = ib + W; so.cursor = ib + 2 * W; ib += 3 * W;
    so.tile_expert = ib; so.tile_row0 = ib + max_tiles; so.tile_rows = ib + 2 * max_tiles; ib += 3 * max_tiles;
    so.n_tiles = ib; ib += 4; so.row_pair = ib; so.pair_row = ib + np;
    if (np) {
        kr_launch_ep_dest(ids, np, s->E_total, s->per, W, s->full, (int32_t*)B.dest.p, (int32_t*)B.lid.p, st);
        kr_launch_ep_sort((const int32_t*)B.dest.p, np, W, so, st);
        kr_launch_ep_gather((const uint16_t*)x_bf16, so.row_pair, (const int32_t*)B.lid.p, topk, H, so.n_tiles + 1, np, (uint16_t*)B.rows.p, (int32_t*)B.row_lid.p, st);
    } else KR_HIP(hipMemsetAsync(so.counts, 0, (size_t)W * 4, st));
    // ---- send counts of every rank: cnt[src][dst]
    // The host needs the split sizes to post the sends / receives: one small DtoH and an event wait per call.  A world of one posts nothing:
    // it takes every pair slot as a row (rows past the last routed pair carry local id -1 and belong to no expert tile) and never waits.
    std::vector<size_t> soff(W + 1, 0), roff(W + 1, 0);
    const bool xchg = W > 1 || s->comm != nullptr;      // a one-rank communicator still runs the exchange (through RCCL, to itself)
    // Round 6: the shared expert of this rank's own tokens (replicated weights) depends on nothing the exchange moves.  It is launched BETWEEN the request for the split
    // sizes (count all-gather + DtoH + event) and the host's wait for them: the stream has ~0.2 - 0.4 ms of GEMMs queued while the 4 W^2 bytes travel, so the wait no longer
    // idles the GPU between the counts and the row exchange of the same layer.  Same launches, same results, another order.
    const float* shared_eo = nullptr;
    auto run_shared = [&]() -> int {   // one all-skipped slot per token: kr_moe_prefill then returns rsf * 0 + shared = the shared expert's rows
        if (!use_shared) return KR_OK;
        if (B.ones.bytes < (size_t)M * 4) {
            if (B.ones.ensure((size_t)M * 4 * 2)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
            if (hipMemsetD32Async((hipDeviceptr_t)B.ones.p, 0x3F800000, B.ones.bytes / 4, st) != hipSuccess) return kr_fail(KR_ERR_HIP, "hipMemsetD32Async failed");
        }
        KR_HIP(hipMemsetAsync(B.neg_ids.p, 0xFF, (size_t)M * 4, st));
        if (int rc = kr_moe_prefill_set(e, layer, x_bf16, (const int32_t*)B.neg_ids.p, (const float*)B.ones.p, B.shared_out.p, M, 1, KR_OUT_F32, 0, set, st)) return rc;
        shared_eo = (const float*)B.shared_out.p;
        return KR_OK;
    };
    if (xchg) {
        if (int rc = ep_gather_counts(s, so.counts, (int*)B.cnt_all.p, st)) return ep_abort(s, rc);
        if (hipMemcpyAsync(B.cnt_host, B.cnt_all.p, (size_t)W * W * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipEventRecord(B.ev, st) != hipSuccess)
            return ep_abort(s, kr_fail(KR_ERR_HIP, "expert parallelism: reading the split sizes failed"));
        if (int rc = run_shared()) return ep_abort(s, rc);      // queued behind the event: runs while the host waits
        if (hipEventSynchronize(B.ev) != hipSuccess) return ep_abort(s, kr_fail(KR_ERR_HIP, "expert parallelism: reading the split sizes failed"));
        for (int r = 0; r < W; r++) { soff[r + 1] = soff[r] + (size_t)B.cnt_host[s->rank * W + r]; roff[r + 1] = roff[r] + (size_t)B.cnt_host[r * W + s->rank]; }
    } else { soff[1] = roff[1] = (size_t)np; if (int rc = run_shared()) return rc; }
    const size_t n_send = soff[W], n_recv = roff[W];
    // ---- buffers whose size depends on what the peers send: a failure here aborts the communicator (the peers are already inside the call)
    {
        const size_t nr1 = n_recv ? n_recv : 1, esz = s->ret_bf16 ? 2 : 4;
        const size_t n_ones = n_recv > (size_t)M ? n_recv : (size_t)(M ? M : 1);
        bool bad = B.eo.ensure(nr1 * (size_t)H * 4);
        if (xchg) bad = bad || B.rrows.ensure(nr1 * H * 2) || B.rlid.ensure(nr1 * 4) || B.back.ensure((n_send ? n_send : 1) * (size_t)H * esz);
        if (!bad && B.ones.bytes < n_ones * 4) {
            bad = B.ones.ensure(n_ones * 4 * 2);
            if (!bad) bad = hipMemsetD32Async((hipDeviceptr_t)B.ones.p, 0x3F800000, B.ones.bytes / 4, st) != hipSuccess;      // f32 1.0
        }
        if (bad) return ep_abort(s, kr_fail(KR_ERR_HIP, "hipMalloc of the expert-parallel exchange buffers failed"));
    }
    // ---- dispatch
    const uint16_t* rrows = (const uint16_t*)B.rows.p; const int32_t* rlid = (const int32_t*)B.row_lid.p;
    if (xchg) {
        const void* sb[2] = {B.rows.p, B.row_lid.p}; void* rb[2] = {B.rrows.p, B.rlid.p}; const size_t rbts[2] = {(size_t)H * 2, 4};
        if (int rc = ep_exchange(s, 2, sb, rb, rbts, soff.data(), roff.data(), st)) return ep_abort(s, rc);
        rrows = (const uint16_t*)B.rrows.p; rlid = (const int32_t*)B.rlid.p;
    }
    // ---- experts on the received rows (top-1 rows, weight 1, f32)
    // the w2 GEMM writes every row at its place, in the dtype it travels back in (kr_moe_prefill_rows); native-GGUF layers take the generic
    // prompt-pass entry (combine with weight 1 = a copy) and a conversion pass
    const void* back = B.eo.p;
    if (n_recv) {
        int rc = kr_moe_prefill_rows(e, layer, rrows, rlid, B.eo.p, (int)n_recv, s->ret_bf16 ? 1 : 0, set, st);
        if (rc < 0) {
            rc = kr_moe_prefill_set(e, layer, rrows, rlid, (const float*)B.ones.p, B.eo.p, (int)n_recv, 1, KR_OUT_F32, 1, set, st);
            if (!rc && s->ret_bf16) {
                if (B.eo16.ensure(n_recv * (size_t)H * 2)) rc = kr_fail(KR_ERR_HIP, "hipMalloc failed");
                else { kr_launch_ep_rows_bf16((const float*)B.eo.p, (uint16_t*)B.eo16.p, n_recv * (size_t)H, st); back = B.eo16.p; }
            }
        }
        if (rc > 0) return xchg ? ep_abort(s, rc) : rc;
    }
    if (xchg) {
        const size_t esz = s->ret_bf16 ? 2 : 4;
        const void* sb[1] = {back}; void* rb[1] = {B.back.p}; const size_t rbts[1] = {(size_t)H * esz};
        if (int rc = ep_exchange(s, 1, sb, rb, rbts, roff.data(), soff.data(), st)) return ep_abort(s, rc);
        back = B.back.p;
    }
    if (!M) { KR_HIP(hipGetLastError()); return KR_OK; }
    // ---- the combine in routing order (+ the shared expert's rows computed above)
    if (s->ret_bf16) kr_launch_pf_combine_bf16rows((const uint16_t*)back, so.pair_row, wts, M, topk, H, shared_eo, e->cfg.routed_scaling_factor, out, out_dtype == KR_OUT_BF16, st);
    else kr_launch_pf_combine((const float*)back, so.pair_row, wts, M, topk, H, shared_eo, e->cfg.routed_scaling_factor, out, out_dtype == KR_OUT_BF16, st);
    KR_HIP(hipGetLastError());
    return KR_OK;
}

extern "C" int kr_moe_prefill_ep(kr_engine* e, int layer, const void* x_bf16, const int32_t* ids, const float* wts, void* out, int M, int topk, int out_dtype,
                                 int routed_only, void* stream) {
    return kr_moe_prefill_ep_set(e, layer, x_bf16, ids, wts, out, M, topk, out_dtype, routed_only, 0, stream);
}

int kr_ep_world(const kr_engine* e) { return e && e->ep ? e->ep->world : 1; }
// expert-parallel DECODE is in force: more than one rank, or a one-rank RCCL communicator (bring-up: the collective path on one GPU)
bool kr_ep_decode_active(const kr_engine* e) { return e && e->ep && (e->ep->world > 1 || e->ep->comm != nullptr); }
bool kr_ep_is_rccl(const kr_engine* e) { return e && e->ep && e->ep->comm != nullptr; }
int kr_ep_rank(const kr_engine* e) { return e && e->ep ? e->ep->rank : 0; }
void kr_ep_slice(const kr_engine* e, int* lo, int* hi, int* sub) {
    const kr_ep_state* s = e->ep;
    *lo = s->rank * s->per; *hi = s->rank == s->world - 1 ? s->E_total : (s->rank + 1) * s->per; *sub = s->full ? 0 : *lo;
}
int kr_ep_allreduce_on(kr_engine* e, float* buf_dev, size_t n, hipStream_t st) {
    if (e->ep->broken) return kr_fail(KR_ERR_STATE, "expert parallelism: an earlier collective failed");
    if (int rc = ep_allreduce_f32(e, buf_dev, n, st)) return ep_abort(e->ep, rc);
    return KR_OK;
}

// in-place sum over the ranks of n f32 on the device (collective); the decode graph's expert-parallel step and tests use it
extern "C" int kr_ep_allreduce_f32(kr_engine* e, float* buf_dev, size_t n, void* stream) {
    if (!e || !e->ep) return kr_fail(KR_ERR_STATE, "call kr_ep_init first");
    if (e->ep->broken) return kr_fail(KR_ERR_STATE, "expert parallelism: an earlier collective failed");
    KR_HIP(hipSetDevice(e->device));
    hipStream_t st = kr_pick_stream(e, stream);
    if (int rc = ep_allreduce_f32(e, buf_dev, n, st)) return ep_abort(e->ep, rc);
    return KR_OK;
}
