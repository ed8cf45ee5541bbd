// kr_decode_standalone.cpp -- C-ABI of the reference's stand-alone CpuDecodeStore operators (src/decode.rs:253-1117) and of the cancellable,
// streaming generation loop (decode.rs:3611).  Every pointer argument may be a HOST pointer (what the reference's Python callers pass:
// tensor.data_ptr() of CPU tensors) or a DEVICE pointer; host buffers are staged through store-owned device buffers and copied back before
// the call returns.  The arithmetic runs on the GPU only (kr_standalone.hip for the scalar-loop operators, the decode graph's own kernels for
// the ones the reference shares with decode_step: dispatch_matmul, fused_add_rmsnorm_avx2, linear_attention_recurrent_avx2, moe_route).
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

#include "kr_decode_internal.h"
#include "kr_kernels.h"
#include "kr_prefill_ops.h"
#include "kr_router.h"
#include "kr_standalone.h"
#include "../../include/krasis_hip.h"

struct kr_standalone_state {      // per store; created on first use (kr_decode_destroy releases it through kr_standalone_release)
    std::vector<DevBuf> pool;     // staging buffers, one per staged argument of a call
    struct Route { DevBuf gate_cm, bias, esc; int E = 0, H = 0; int bf16_exact = 0; bool has_bias = false, has_esc = false; };
    std::vector<Route> routes;
    DevBuf logits, ids, w, gexp, conv_out, mid, act;
    std::atomic<int> cancel{0};
    double last_elapsed_s = 0.0;
    std::mutex mu;
};

static kr_standalone_state* sa(kr_decode_store* s) {
    if (!s->standalone) s->standalone = new kr_standalone_state();
    return s->standalone;
}
void kr_standalone_release(kr_decode_store* s) {
    if (!s->standalone) return;
    for (auto& b : s->standalone->pool) b.release();
    for (auto& r : s->standalone->routes) { r.gate_cm.release(); r.bias.release(); r.esc.release(); }
    for (DevBuf* b : {&s->standalone->logits, &s->standalone->ids, &s->standalone->w, &s->standalone->gexp, &s->standalone->conv_out, &s->standalone->mid, &s->standalone->act}) b->release();
    delete s->standalone; s->standalone = nullptr;
}

namespace {
// host or device arguments of one call
struct Stage {
    kr_decode_store* s; kr_standalone_state* S; hipStream_t st; size_t next = 0; bool failed = false;
    struct Back { void* host; void* dev; size_t bytes; };
    std::vector<Back> backs;
    Stage(kr_decode_store* s_) : s(s_), S(sa(s_)), st(s_->eng->stream) {}
    void* slot(size_t bytes) {
        if (next >= S->pool.size()) S->pool.resize(next + 1);
        DevBuf& b = S->pool[next++];
        if (b.ensure(bytes ? bytes : 4)) { failed = true; return nullptr; }
        return b.p;
    }
    const void* in(const void* p, size_t bytes) {
        if (!p || is_device_ptr(p)) return p;
        void* d = slot(bytes);
        if (d && hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, st) != hipSuccess) failed = true;
        return d;
    }
    void* out(void* p, size_t bytes, bool also_in = false) {
        if (!p || is_device_ptr(p)) return p;
        void* d = slot(bytes);
        if (d && also_in && hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, st) != hipSuccess) failed = true;
        if (d) backs.push_back({p, d, bytes});
        return d;
    }
    int finish() {
        if (failed) return kr_fail(KR_ERR_HIP, "staging a stand-alone operator argument failed (hipMalloc / hipMemcpy)");
        KR_HIP(hipGetLastError());
        for (auto& b : backs) KR_HIP(hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, st));
        KR_HIP(hipStreamSynchronize(st));
        return KR_OK;
    }
};
int chk(kr_decode_store* s) {
    if (!s || !s->eng) return kr_fail(KR_ERR_VALUE, "null decode store");
    KR_HIP(hipSetDevice(s->eng->device));
    return KR_OK;
}
int chk_w(kr_decode_store* s, int wid) {
    if (wid < 0 || wid >= (int)s->weights.size()) return kr_fail(KR_ERR_VALUE, "weight_id %d out of range (%zu)", wid, s->weights.size());   // decode.rs:335
    return KR_OK;
}
}  // namespace

// ---- matmul / matmul_batch (decode.rs:328, 364): f32 input quantised to INT16 per 128-group once, dispatch_matmul per weight
extern "C" int kr_decode_matmul(kr_decode_store* s, int weight_id, const float* input, float* output) {
    if (int rc = chk(s)) return rc;
    if (int rc = chk_w(s, weight_id)) return rc;
    if (!input || !output) return kr_fail(KR_ERR_VALUE, "null pointer argument");
    std::lock_guard<std::mutex> lk(sa(s)->mu);
    DWeight& W = *s->weights[weight_id];
    Stage g(s);
    const float* x = (const float*)g.in(input, (size_t)W.cols * 4); float* y = (float*)g.out(output, (size_t)W.rows * 4);
    if (!g.failed) kr_launch_matvec(mv(s, weight_id), x, 1, y, g.st);
    return g.finish();
}

extern "C" int kr_decode_matmul_batch(kr_decode_store* s, const int* weight_ids, int n, const float* input, float* const* outputs) {
    if (int rc = chk(s)) return rc;
    if (n < 0 || (n > 0 && (!weight_ids || !outputs || !input))) return kr_fail(KR_ERR_VALUE, "null pointer argument");
    if (n == 0) return KR_OK;                                                               // decode.rs:374
    for (int i = 0; i < n; i++) if (int rc = chk_w(s, weight_ids[i])) return rc;
    const int K = s->weights[weight_ids[0]]->cols;
    for (int i = 0; i < n; i++) if (s->weights[weight_ids[i]]->cols != K) return kr_fail(KR_ERR_VALUE, "All weights in batch must have same K");   // decode.rs:395
    std::lock_guard<std::mutex> lk(sa(s)->mu);
    Stage g(s);
    const float* x = (const float*)g.in(input, (size_t)K * 4);
    for (int i = 0; i < n && !g.failed; i++) {
        float* y = (float*)g.out(outputs[i], (size_t)s->weights[weight_ids[i]]->rows * 4);
        if (!g.failed) kr_launch_matvec(mv(s, weight_ids[i]), x, 1, y, g.st);
    }
    return g.finish();
}

// ---- fused_add_rmsnorm / fused_add_rmsnorm_id (decode.rs:406, 447): in place on hidden and residual; weight pointer, or a stored norm id
extern "C" int kr_decode_fused_add_rmsnorm(kr_decode_store* s, float* hidden, float* residual, const float* weight, int norm_id, float eps, int size, int first_call) {
    if (int rc = chk(s)) return rc;
    if (!hidden || !residual || size <= 0) return kr_fail(KR_ERR_VALUE, "bad arguments");
    if (!weight) {
        if (norm_id < 0 || norm_id >= (int)s->norms.size()) return kr_fail(KR_ERR_VALUE, "norm_id %d out of range (%zu)", norm_id, s->norms.size());   // decode.rs:456
        if (s->norm_len[norm_id] < size) return kr_fail(KR_ERR_VALUE, "norm weight %d holds %d values, size is %d", norm_id, s->norm_len[norm_id], size);
    }
    std::lock_guard<std::mutex> lk(sa(s)->mu);
    Stage g(s);
    float* h = (float*)g.out(hidden, (size_t)size * 4, true); float* r = (float*)g.out(residual, (size_t)size * 4, !first_call);
    const float* w = weight ? (const float*)g.in(weight, (size_t)size * 4) : (const float*)s->norms[norm_id]->p;
    if (!g.failed) { KrNormSrc src{}; kr_launch_fused_add_rmsnorm(src, h, r, r, w, size, eps, first_call ? 1 : 0, s->norm_bias_one ? 1 : 0, g.st); }
    return g.finish();
}

extern "C" int kr_decode_rmsnorm(kr_decode_store* s, const float* input, const float* weight, float eps, float* output, int size) {
    if (int rc = chk(s)) return rc;
    if (!input || !weight || !output || size <= 0) return kr_fail(KR_ERR_VALUE, "bad arguments");
    std::lock_guard<std::mutex> lk(sa(s)->mu);
    Stage g(s);
    const float* x = (const float*)g.in(input, (size_t)size * 4); const float* w = (const float*)g.in(weight, (size_t)size * 4);
    float* y = (float*)g.out(output, (size_t)size * 4);
    if (!g.failed) kr_launch_op_rmsnorm(x, w, y, size, eps, s->norm_bias_one ? 1 : 0, g.st);
    return g.finish();
}

extern "C" int kr_decode_silu_mul(kr_decode_store* s, const float* gate, const float* up, float* output, int size) {
    if (int rc = chk(s)) return rc;
    if (!gate || !up || !output || size <= 0) return kr_fail(KR_ERR_VALUE, "bad arguments");
    std::lock_guard<std::mutex> lk(sa(s)->mu);
    Stage g(s);
    const float* a = (const float*)g.in(gate, (size_t)size * 4); const float* b = (const float*)g.in(up, (size_t)size * 4);
    float* y = (float*)g.out(output, (size_t)size * 4);
    if (!g.failed) kr_launch_op_silu_mul(a, b, y, size, g.st);
    return g.finish();
}

// ---- fused_shared_expert (decode.rs:542): gate_up matvec -> SiLU(gate) * up with libm's exp -> down matvec
extern "C" int kr_decode_fused_shared_expert(kr_decode_store* s, int gate_up_wid, int down_wid, const float* input, float* output) {
    if (int rc = chk(s)) return rc;
    if (int rc = chk_w(s, gate_up_wid)) return rc;
    if (int rc = chk_w(s, down_wid)) return rc;
    if (!input || !output) return kr_fail(KR_ERR_VALUE, "null pointer argument");
    DWeight& GU = *s->weights[gate_up_wid]; DWeight& D = *s->weights[down_wid];
    const int inter = GU.rows / 2;
    if (D.cols != inter) return kr_fail(KR_ERR_VALUE, "down weight K %d != intermediate %d", D.cols, inter);
    kr_standalone_state* S = sa(s);
    std::lock_guard<std::mutex> lk(S->mu);
    if (S->mid.ensure((size_t)GU.rows * 4) || S->act.ensure((size_t)inter * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    Stage g(s);
    const float* x = (const float*)g.in(input, (size_t)GU.cols * 4); float* y = (float*)g.out(output, (size_t)D.rows * 4);
    if (!g.failed) {
        kr_launch_matvec(mv(s, gate_up_wid), x, 1, (float*)S->mid.p, g.st);
        kr_launch_op_silu_mul((const float*)S->mid.p, (const float*)S->mid.p + inter, (float*)S->act.p, inter, g.st);
        kr_launch_matvec(mv(s, down_wid), S->act.p, 1, y, g.st);
    }
    return g.finish();
}

// ---- linear_attention_recurrent (decode.rs:609): state [nv,dk,dv] updated in place, output [nv,dv]
extern "C" int kr_decode_linear_attention_recurrent(kr_decode_store* s, float* state, const float* q, const float* k, const float* v, const float* g_, const float* beta,
                                                    float* output, int nv, int dk, int dv) {
    if (int rc = chk(s)) return rc;
    if (!state || !q || !k || !v || !g_ || !beta || !output || nv <= 0) return kr_fail(KR_ERR_VALUE, "bad arguments");
    kr_standalone_state* S = sa(s);
    std::lock_guard<std::mutex> lk(S->mu);
    if (S->gexp.ensure((size_t)nv * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    Stage g(s);
    float* st_d = (float*)g.out(state, (size_t)nv * dk * dv * 4, true);
    const float* qd = (const float*)g.in(q, (size_t)nv * dk * 4); const float* kd = (const float*)g.in(k, (size_t)nv * dk * 4);
    const float* vd = (const float*)g.in(v, (size_t)nv * dv * 4); const float* gd = (const float*)g.in(g_, (size_t)nv * 4); const float* bd = (const float*)g.in(beta, (size_t)nv * 4);
    float* od = (float*)g.out(output, (size_t)nv * dv * 4);
    if (!g.failed) {
        kr_launch_op_exp(gd, (float*)S->gexp.p, nv, g.st);
        if (kr_launch_pfm_la_recur(st_d, qd, kd, vd, (const float*)S->gexp.p, bd, od, nv, dk, dv, 1, g.st))
            return kr_fail(KR_ERR_VALUE, "unsupported linear-attention geometry (dk %d must be 64 or 128, dv %d a multiple of 8 in [dk, 256])", dk, dv);
    }
    return g.finish();
}

extern "C" int kr_decode_gated_rmsnorm_silu(kr_decode_store* s, const float* x, const float* z, const float* norm_weight, float* output, float eps, int nv, int dv) {
    if (int rc = chk(s)) return rc;
    if (!x || !z || !norm_weight || !output || nv <= 0 || dv <= 0) return kr_fail(KR_ERR_VALUE, "bad arguments");
    std::lock_guard<std::mutex> lk(sa(s)->mu);
    const size_t n = (size_t)nv * dv * 4;
    Stage g(s);
    const float* xd = (const float*)g.in(x, n); const float* zd = (const float*)g.in(z, n); const float* wd = (const float*)g.in(norm_weight, n);
    float* od = (float*)g.out(output, n);
    if (!g.failed) kr_launch_op_gated_rmsnorm_silu(xd, zd, wd, od, nv, dv, eps, g.st);
    return g.finish();
}

// ---- linear_attention_conv (decode.rs:713): conv_state updated in place; q, k (head-expanded, L2-normalised), v, z, g (raw), beta out
extern "C" int kr_decode_linear_attention_conv(kr_decode_store* s, const float* qkvz, const float* ba, float* conv_state, const float* conv_weight, const float* a_log,
                                               const float* dt_bias, float scale, float* q_out, float* k_out, float* v_out, float* z_out, float* g_out, float* beta_out,
                                               int nk, int nv, int dk, int dv, int hr, int kernel_dim) {
    if (int rc = chk(s)) return rc;
    if (!qkvz || !ba || !conv_state || !conv_weight || !a_log || !dt_bias || !q_out || !k_out || !v_out || !z_out || !g_out || !beta_out) return kr_fail(KR_ERR_VALUE, "null pointer argument");
    if (nk <= 0 || hr <= 0 || nv != nk * hr || dk <= 0 || dv <= 0 || kernel_dim < 1) return kr_fail(KR_ERR_VALUE, "bad linear-attention geometry (nv must equal nk * hr)");
    kr_standalone_state* S = sa(s);
    std::lock_guard<std::mutex> lk(S->mu);
    const size_t conv_dim = (size_t)2 * nk * dk + (size_t)nv * dv, group_dim = (size_t)2 * dk + (size_t)2 * dv * hr;
    if (S->conv_out.ensure(conv_dim * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    Stage g(s);
    KrOpLaConvArgs a{};
    a.qkvz = (const float*)g.in(qkvz, (size_t)nk * group_dim * 4); a.ba = (const float*)g.in(ba, (size_t)nk * 2 * hr * 4);
    a.conv_state = (float*)g.out(conv_state, conv_dim * kernel_dim * 4, true); a.conv_w = (const float*)g.in(conv_weight, conv_dim * kernel_dim * 4);
    a.a_log = (const float*)g.in(a_log, (size_t)nv * 4); a.dt_bias = (const float*)g.in(dt_bias, (size_t)nv * 4); a.scale = scale;
    a.q = (float*)g.out(q_out, (size_t)nv * dk * 4); a.k = (float*)g.out(k_out, (size_t)nv * dk * 4); a.v = (float*)g.out(v_out, (size_t)nv * dv * 4);
    a.z = (float*)g.out(z_out, (size_t)nv * dv * 4); a.g = (float*)g.out(g_out, (size_t)nv * 4); a.beta = (float*)g.out(beta_out, (size_t)nv * 4);
    a.conv_out = (float*)S->conv_out.p; a.nk = nk; a.nv = nv; a.dk = dk; a.dv = dv; a.hr = hr; a.kernel_dim = kernel_dim;
    if (!g.failed) kr_launch_op_la_conv(a, g.st);
    return g.finish();
}

// ---- store_route_weight / moe_route (decode.rs:895, 955): the gate is kept in the chain-major layout of the decode router (kr_router.hip)
extern "C" int kr_decode_store_route_weight(kr_decode_store* s, const float* gate, int num_experts, int hidden_dim, const float* bias, const float* e_score_corr, int* route_id_out) {
    if (int rc = chk(s)) return rc;
    if (!gate || !route_id_out || num_experts <= 0) return kr_fail(KR_ERR_VALUE, "bad arguments");
    if (hidden_dim % 128 != 0) return kr_fail(KR_ERR_VALUE, "router hidden dim %d must be a multiple of 128", hidden_dim);
    kr_standalone_state* S = sa(s);
    std::lock_guard<std::mutex> lk(S->mu);
    const int E = num_experts, H = hidden_dim, neb = (E + 3) / 4;
    bool exact = true;
    for (size_t i = 0; i < (size_t)E * H && exact; i++) { uint32_t b; memcpy(&b, &gate[i], 4); exact = (b & 0xFFFFu) == 0; }
    S->routes.emplace_back();
    kr_standalone_state::Route& R = S->routes.back();
    R.E = E; R.H = H; R.bf16_exact = exact ? 1 : 0;
    if (exact) {      // same layouts as kr_set_routing_weights (kr_engine.cpp)
        const int nc = H / 128;
        std::vector<uint16_t> cm((size_t)neb * nc * 64 * 8, 0);
        for (int eb = 0; eb < neb; eb++) for (int c = 0; c < nc; c++) for (int lane = 0; lane < 64; lane++) {
            const int ex = eb * 4 + lane / 16, j = lane % 16;
            if (ex >= E) continue;
            for (int u = 0; u < 8; u++) { uint32_t b; memcpy(&b, &gate[(size_t)ex * H + 16 * (8 * c + u) + j], 4); cm[(((size_t)eb * nc + c) * 64 + lane) * 8 + u] = (uint16_t)(b >> 16); }
        }
        if (R.gate_cm.ensure(cm.size() * 2)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        KR_HIP(hipMemcpy(R.gate_cm.p, cm.data(), cm.size() * 2, hipMemcpyHostToDevice));
    } else {
        const int nc = H / 64;
        std::vector<float> cm((size_t)neb * nc * 64 * 4, 0.0f);
        for (int eb = 0; eb < neb; eb++) for (int c = 0; c < nc; c++) for (int lane = 0; lane < 64; lane++) {
            const int ex = eb * 4 + lane / 16, j = lane % 16;
            if (ex >= E) continue;
            for (int u = 0; u < 4; u++) cm[(((size_t)eb * nc + c) * 64 + lane) * 4 + u] = gate[(size_t)ex * H + 16 * (4 * c + u) + j];
        }
        if (R.gate_cm.ensure(cm.size() * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        KR_HIP(hipMemcpy(R.gate_cm.p, cm.data(), cm.size() * 4, hipMemcpyHostToDevice));
    }
    R.has_bias = bias != nullptr; R.has_esc = e_score_corr != nullptr;
    if (bias) { if (R.bias.ensure((size_t)E * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed"); KR_HIP(hipMemcpy(R.bias.p, bias, (size_t)E * 4, hipMemcpyHostToDevice)); }
    if (e_score_corr) { if (R.esc.ensure((size_t)E * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed"); KR_HIP(hipMemcpy(R.esc.p, e_score_corr, (size_t)E * 4, hipMemcpyHostToDevice)); }
    *route_id_out = (int)S->routes.size() - 1;
    return KR_OK;
}

extern "C" int kr_decode_moe_route(kr_decode_store* s, int route_id, const float* hidden, int32_t* topk_ids_out, float* topk_weights_out, int topk, int scoring_func, int norm_topk_prob) {
    if (int rc = chk(s)) return rc;
    kr_standalone_state* S = sa(s);
    if (route_id < 0 || route_id >= (int)S->routes.size()) return kr_fail(KR_ERR_VALUE, "route_id %d out of range (%zu)", route_id, S->routes.size());   // decode.rs:966
    if (scoring_func < 0 || scoring_func > 2) return kr_fail(KR_ERR_VALUE, "Unknown scoring_func: %d", scoring_func);                               // decode.rs:1080
    if (!hidden || !topk_ids_out || !topk_weights_out) return kr_fail(KR_ERR_VALUE, "null pointer argument");
    std::lock_guard<std::mutex> lk(S->mu);
    kr_standalone_state::Route& R = S->routes[route_id];
    if (topk <= 0 || topk > KR_MAX_TOPK || topk > R.E) return kr_fail(KR_ERR_VALUE, "bad topk %d for %d experts", topk, R.E);
    if (S->logits.ensure((size_t)R.E * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    Stage g(s);
    const float* x = (const float*)g.in(hidden, (size_t)R.H * 4);
    int32_t* ids = (int32_t*)g.out(topk_ids_out, (size_t)topk * 4); float* w = (float*)g.out(topk_weights_out, (size_t)topk * 4);
    if (!g.failed) {
        kr_launch_route_logits_decode(R.gate_cm.p, R.bf16_exact, x, R.has_bias ? (const float*)R.bias.p : nullptr, (float*)S->logits.p, 1, R.E, R.H, g.st);
        kr_launch_route_select((const float*)S->logits.p, R.has_esc ? (const float*)R.esc.p : nullptr, ids, w, 1, R.E, topk, scoring_func, norm_topk_prob ? 1 : 0,
                               KR_ROUTE_RULE_DECODE, 0, g.st);
    }
    return g.finish();
}

extern "C" int kr_decode_num_route_weights(kr_decode_store* s) { return s && s->standalone ? (int)s->standalone->routes.size() : 0; }

// decode.rs:1107: packed words * 4 + scales * 2 of one stored weight
extern "C" size_t kr_decode_weight_bytes(kr_decode_store* s, int weight_id) {
    if (!s || weight_id < 0 || weight_id >= (int)s->weights.size()) return 0;
    DWeight& W = *s->weights[weight_id];
    const size_t K = (size_t)W.cols, N = (size_t)W.rows;
    return (W.ms.bits == 4 ? K / 8 * N * 4 : K / 4 * N * 4) + K / (size_t)s->group_size * N * 2;
}

// ---- cancel / reset_cancel / last_decode_elapsed_s (decode.rs:253-265)
extern "C" int kr_decode_cancel(kr_decode_store* s) { if (int rc = chk(s)) return rc; sa(s)->cancel.store(1, std::memory_order_release); return KR_OK; }
extern "C" int kr_decode_reset_cancel(kr_decode_store* s) { if (int rc = chk(s)) return rc; sa(s)->cancel.store(0, std::memory_order_release); return KR_OK; }
extern "C" double kr_decode_last_elapsed_s(kr_decode_store* s) { return s && s->standalone ? s->standalone->last_elapsed_s : 0.0; }
int kr_standalone_cancelled(kr_decode_store* s) { return s->standalone ? s->standalone->cancel.load(std::memory_order_acquire) : 0; }
void kr_standalone_set_elapsed(kr_decode_store* s, double sec) { sa(s)->last_elapsed_s = sec; }
