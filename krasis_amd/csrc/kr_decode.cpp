// kr_decode.cpp -- the decode graph ("CpuDecodeStore" in the reference, src/decode.rs:193-3602) resident on the GPU.
//
// Same construction API as the reference (store_weight_f32 / store_norm_weight / configure_decode / add_decode_*_layer /
// set_decode_layer_moe / set_decode_rope / set_decode_state / decode_step / generate_batch), same numerics (bit-exact
// against the oracle restatement), but every weight, KV page, recurrent state and scratch buffer lives in HBM and one
// decode_step is a fixed sequence of kernel launches that reads (token, position) from device memory, i.e. it is
// hipGraph-capturable and replayable per token.
#include <algorithm>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/krasis_hip.h"
#include "kr_decode_ops.h"
#include "kr_engine_internal.h"
#include "kr_kernels.h"
#include "kr_prefill.h"
#include "kr_router.h"
#include "kr_sampler.h"
#include <chrono>

#include "kr_decode_internal.h"
#include "kr_decode_fast.h"

static void prof_mark(kr_decode_store* s, int kind, hipStream_t st) {
    if (!s->prof) return;
    if (s->ev_used + 2 > s->ev_pool.size()) { for (int i = 0; i < 64; i++) { hipEvent_t e; (void)hipEventCreate(&e); s->ev_pool.push_back(e); } }
    if (kind >= 0) s->ev_kind.push_back(kind);
    (void)hipEventRecord(s->ev_pool[s->ev_used++], st);
}
#define PROF(kind, stmt) do { prof_mark(s, kind, st); stmt; prof_mark(s, -1, st); } while (0)

static int chk_store(kr_decode_store* s) { return s ? KR_OK : kr_fail(KR_ERR_VALUE, "null decode store"); }
static int chk_wid(kr_decode_store* s, int id, const char* what) {
    if (id < 0 || id >= (int)s->weights.size()) return kr_fail(KR_ERR_VALUE, "%s: weight id %d out of range", what, id);
    return KR_OK;
}

extern "C" int kr_decode_create(kr_engine* eng, int group_size, int norm_bias_one, kr_decode_store** out) {
    if (!out) return kr_fail(KR_ERR_VALUE, "null argument");
    if (group_size != 128) return kr_fail(KR_ERR_VALUE, "group_size %d unsupported", group_size);
    bool own = false;
    if (!eng) {   // CpuDecodeStore::new comes before set_moe_store in the reference (decode.rs:229, :2250): run on a bare engine of the current device
        int ndev = 0, dev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return kr_fail(KR_ERR_HIP, "no HIP device available: libkrasis_hip.so has no CPU fallback"); }
        KR_HIP(hipGetDevice(&dev));
        eng = kr_engine_new_bare(dev); own = true;
        if (!eng) return kr_fail(KR_ERR_HIP, "could not create a stream on device %d", dev);
    }
    KR_HIP(hipSetDevice(eng->device));
    std::unique_ptr<kr_decode_store> s(new kr_decode_store);
    s->eng = eng; s->device = eng->device; s->own_eng = own; s->group_size = group_size; s->norm_bias_one = norm_bias_one != 0;
    if (s->step_dev.ensure(sizeof(KrStep))) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    *out = s.release();
    return KR_OK;
}

// the same with the device named explicitly (a store created before its engine on a box with several GPUs: ADVICE r2 -- the bare engine used to
// land on whatever device the calling thread had current, and set_moe_store then refused the engine of the device the caller meant)
extern "C" int kr_decode_create_on(int device_ordinal, int group_size, int norm_bias_one, kr_decode_store** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return kr_fail(KR_ERR_HIP, "no HIP device available: libkrasis_hip.so has no CPU fallback"); }
    if (device_ordinal < 0 || device_ordinal >= ndev) return kr_fail(KR_ERR_VALUE, "device %d out of range (%d devices)", device_ordinal, ndev);
    KR_HIP(hipSetDevice(device_ordinal));
    return kr_decode_create(nullptr, group_size, norm_bias_one, out);
}

extern "C" void kr_decode_destroy(kr_decode_store* s) {
    if (!s) return;
    // the engine may already be gone (a garbage collector finalises an engine and its store in any order): only the store's own copy of the
    // device ordinal is used, and the device is drained instead of the engine's stream
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    if (s->graph_exec) (void)hipGraphExecDestroy(s->graph_exec);
    kr_standalone_release(s);
    for (auto& w : s->weights) { w->ms.q.release(); w->ms.s.release(); w->ms.wsum.release(); }
    for (auto& n : s->norms) n->release();
    for (auto& l : s->layers)
        for (DevBuf* b : {&l.conv_w, &l.a_log, &l.dt_bias, &l.la_norm_w, &l.conv_state, &l.recur_state, &l.q_norm, &l.k_norm, &l.kv_k, &l.kv_v,
                           &l.w_kc, &l.w_vc, &l.kv_a_norm, &l.q_a_norm, &l.mla_cos, &l.mla_sin}) b->release();
    for (DevBuf* b : {&s->embedding, &s->rope_cos, &s->rope_sin, &s->hid, &s->res, &s->proj_a, &s->proj_b, &s->qbuf, &s->kbuf, &s->vbuf, &s->zbuf,
                      &s->gbuf, &s->betabuf, &s->gatebuf, &s->latbuf, &s->recur_out, &s->attn_out, &s->logits, &s->gate_val, &s->tok, &s->step_dev,
                      &s->hid2, &s->res2, &s->f_qk, &s->r_counter, &s->gqa_scores, &s->fd_o, &s->fd_ml, &s->argmax_scratch, &s->img_in, &s->img_post, &s->img_post_bf16, &s->img_attn, &s->smp_seen, &s->smp_keys, &s->smp_temp, &s->smp_probs, &s->smp_rng, &s->pf_scratch, &s->pf_scores, &s->pf_vlogits, &s->pf_nll, &s->pf_tokens, &s->moe_gu, &s->moe_eo, &s->r_logits, &s->r_ids, &s->r_w, &s->dense_gu}) b->release();
    for (hipEvent_t ev : s->pf_events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : s->gen_ev) if (ev) (void)hipEventDestroy(ev);
    if (s->gen_ring) (void)hipHostFree(s->gen_ring);
    for (hipStream_t ps : s->pf_side) { (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps); }
    if (s->own_eng) kr_engine_destroy(s->eng);
    delete s;
    (void)hipGetLastError();      // a failure while tearing down must not surface as the "last error" of an unrelated later call
}

// ---- host-side weight quantizers (decode.rs:46-178): f32 [N,K] -> transposed INT4/INT8 + bf16 scales ----
static inline uint16_t bf16_rne(float f) { uint32_t b; memcpy(&b, &f, 4); b += 0x7FFFu + ((b >> 16) & 1u); return (uint16_t)(b >> 16); }
static inline int32_t sat_i32(float x) { if (x != x) return 0; if (x >= 2147483648.0f) return INT32_MAX; if (x <= -2147483648.0f) return INT32_MIN; return (int32_t)x; }

static void quantize_f32_transposed(const float* w, int rows, int cols, int bits, std::vector<uint32_t>& packed, std::vector<int8_t>& data8,
                                    std::vector<uint16_t>& scales) {
    const int gs = 128, ng = cols / gs;
    scales.assign((size_t)ng * rows, 0);
    if (bits == 4) packed.assign((size_t)(cols / 8) * rows, 0); else data8.assign((size_t)cols * rows, 0);
    const float qmax = bits == 4 ? 7.0f : 127.0f; const int lo = bits == 4 ? -8 : -128, hi = bits == 4 ? 7 : 127;
    for (int r = 0; r < rows; r++) {
        const float* row = w + (size_t)r * cols;
        for (int g = 0; g < ng; g++) {
            float mx = 0.0f;
            for (int i = 0; i < gs; i++) { const float a = fabsf(row[g * gs + i]); if (a > mx) mx = a; }
            const float scale = mx > 0.0f ? mx / qmax : 1.0f, inv = mx > 0.0f ? qmax / mx : 0.0f;
            scales[(size_t)g * rows + r] = bf16_rne(scale);
            if (bits == 4) {
                for (int p = 0; p < gs / 8; p++) {
                    uint32_t word = 0;
                    for (int j = 0; j < 8; j++) {
                        int q = sat_i32(roundf(row[g * gs + p * 8 + j] * inv)); q = q < lo ? lo : (q > hi ? hi : q);
                        word |= (uint32_t)(q + 8) << (j * 4);
                    }
                    packed[(size_t)(g * (gs / 8) + p) * rows + r] = word;
                }
            } else {
                for (int i = 0; i < gs; i++) {
                    int q = sat_i32(roundf(row[g * gs + i] * inv)); q = q < lo ? lo : (q > hi ? hi : q);
                    data8[(size_t)(g * gs + i) * rows + r] = (int8_t)q;
                }
            }
        }
    }
}

static int new_weight(kr_decode_store* s, int rows, int cols, int bits, DWeight** out) {
    if (cols % 128 != 0) return kr_fail(KR_ERR_VALUE, "cols %d must be divisible by group_size 128", cols);
    if (bits != 4 && bits != 8) return kr_fail(KR_ERR_VALUE, "num_bits must be 4 or 8, got %d", bits);
    std::unique_ptr<DWeight> w(new DWeight);
    w->rows = rows; w->cols = cols;
    if (int rc = matset_alloc(s->eng, w->ms, cols, rows, bits, 1)) return rc;
    s->weight_bytes += w->ms.q_stride + w->ms.s_stride;
    *out = w.get();
    s->weights.push_back(std::move(w));
    return KR_OK;
}

extern "C" int kr_decode_store_weight_f32(kr_decode_store* s, const float* w, int rows, int cols, int bits, int* id_out) {
    if (int rc = chk_store(s)) return rc;
    if (!w || !id_out) return kr_fail(KR_ERR_VALUE, "null argument");
    KR_HIP(hipSetDevice(s->eng->device));
    DWeight* dw;
    if (int rc = new_weight(s, rows, cols, bits, &dw)) return rc;
    std::vector<uint32_t> packed; std::vector<int8_t> d8; std::vector<uint16_t> sc;
    quantize_f32_transposed(w, rows, cols, bits, packed, d8, sc);
    if (int rc = upload_mat(s->eng, dw->ms, 0, bits == 4 ? (const void*)packed.data() : (const void*)d8.data(), sc.data())) return rc;
    *id_out = (int)s->weights.size() - 1;
    return KR_OK;
}

extern "C" int kr_decode_store_weight_synthetic(kr_decode_store* s, int rows, int cols, int bits, uint64_t seed, int* id_out) {
    if (int rc = chk_store(s)) return rc;
    KR_HIP(hipSetDevice(s->eng->device));
    DWeight* dw;
    if (int rc = new_weight(s, rows, cols, bits, &dw)) return rc;
    kr_launch_fill_synth(dw->ms.q.p, dw->ms.q_stride, (uint32_t*)dw->ms.s.p, dw->ms.s_stride / 4, seed, s->eng->stream);
    KR_HIP(hipStreamSynchronize(s->eng->stream));
    *id_out = (int)s->weights.size() - 1;
    return KR_OK;
}

extern "C" int kr_decode_download_weight(kr_decode_store* s, int wid, void* packed_t, uint16_t* scales_t) {
    if (int rc = chk_store(s)) return rc;
    if (int rc = chk_wid(s, wid, "download_weight")) return rc;
    KR_HIP(hipSetDevice(s->eng->device));
    return download_mat(s->eng, s->weights[wid]->ms, 0, packed_t, scales_t);
}

static int upload_f32(DevBuf& b, const float* src, size_t n) {
    if (b.ensure(n * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    KR_HIP(hipMemcpy(b.p, src, n * 4, hipMemcpyHostToDevice));
    return KR_OK;
}

extern "C" int kr_decode_store_norm_weight(kr_decode_store* s, const float* w, int n, int* id_out) {
    if (int rc = chk_store(s)) return rc;
    if (!w || !id_out) return kr_fail(KR_ERR_VALUE, "null argument");
    KR_HIP(hipSetDevice(s->eng->device));
    std::unique_ptr<DevBuf> b(new DevBuf);
    if (int rc = upload_f32(*b, w, n)) return rc;
    s->norms.push_back(std::move(b)); s->norm_len.push_back(n);
    *id_out = (int)s->norms.size() - 1;
    return KR_OK;
}

extern "C" int kr_decode_configure(kr_decode_store* s, int hidden, int n_layers, float eps, int final_norm_id, int lm_head_wid, int vocab,
                                   int topk, int scoring, int norm_topk_prob, float rsf, const float* embedding, uint64_t synth_seed) {
    if (int rc = chk_store(s)) return rc;
    if (hidden % 128 != 0) return kr_fail(KR_ERR_VALUE, "hidden %d must be a multiple of 128", hidden);
    if (int rc = chk_wid(s, lm_head_wid, "configure_decode lm_head")) return rc;
    if (final_norm_id < 0 || final_norm_id >= (int)s->norms.size()) return kr_fail(KR_ERR_VALUE, "final norm id out of range");
    KR_HIP(hipSetDevice(s->eng->device));
    s->hidden = hidden; s->n_layers = n_layers; s->eps = eps; s->final_norm = final_norm_id; s->lm_head = lm_head_wid; s->vocab = vocab;
    s->topk = topk; s->scoring = scoring; s->norm_topk = norm_topk_prob; s->rsf = rsf;
    s->layers.clear(); s->layers.reserve(n_layers);
    const size_t ne = (size_t)vocab * hidden;
    if (embedding) { if (int rc = upload_f32(s->embedding, embedding, ne)) return rc; }
    else {
        // synthetic embedding +-0.1 (decode.rs:5364): generated on the GPU
        if (s->embedding.ensure(ne * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
        kr_launch_fill_uniform_f32((float*)s->embedding.p, ne, 0.1f, synth_seed, s->eng->stream);
        KR_HIP(hipStreamSynchronize(s->eng->stream));
    }
    if (s->hid.ensure((size_t)hidden * 4) || s->res.ensure((size_t)hidden * 4) || s->logits.ensure((size_t)vocab * 4) ||
        s->gate_val.ensure(64) || s->tok.ensure(64) || s->hid2.ensure((size_t)hidden * 4) || s->res2.ensure((size_t)hidden * 4) || s->r_counter.ensure(64) || s->argmax_scratch.ensure(1024))
        return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    KR_HIP(hipMemset(s->r_counter.p, 0, 64));
    KR_HIP(hipMemset(s->argmax_scratch.p, 0, 1024));
    KR_HIP(hipMemset(s->gate_val.p, 0, 64));
    s->configured = true; s->graph_ok = false;
    return KR_OK;
}

static int need_cfg(kr_decode_store* s) {
    if (int rc = chk_store(s)) return rc;
    if (!s->configured) return kr_fail(KR_ERR_STATE, "Call configure_decode first");
    return KR_OK;
}

extern "C" int kr_decode_add_la_layer(kr_decode_store* s, int input_norm_id, int post_attn_norm_id, int qkvz_wid, int ba_wid, int out_wid,
                                      const float* conv_weight, const float* a_log, const float* dt_bias, const float* norm_weight, int nk, int nv,
                                      int dk, int dv, int kernel_dim, float scale) {
    if (int rc = need_cfg(s)) return rc;
    for (int id : {qkvz_wid, ba_wid, out_wid}) if (int rc = chk_wid(s, id, "add_decode_la_layer")) return rc;
    if (kernel_dim != 4) return kr_fail(KR_ERR_VALUE, "linear-attention conv kernel_dim %d unsupported (4 only)", kernel_dim);
    if (dk != 128 && dk != 64) return kr_fail(KR_ERR_VALUE, "linear_key_head_dim %d unsupported (64/128)", dk);
    if (dv % 64 != 0 || dv > 256 || nv % nk != 0) return kr_fail(KR_ERR_VALUE, "unsupported linear-attention head geometry");
    KR_HIP(hipSetDevice(s->eng->device));
    DLayer L; L.input_norm = input_norm_id; L.post_norm = post_attn_norm_id; L.attn = ATTN_LA;
    L.qkvz_wid = qkvz_wid; L.ba_wid = ba_wid; L.out_wid = out_wid; L.nk = nk; L.nv = nv; L.dk = dk; L.dv = dv; L.kd = kernel_dim; L.la_scale = scale;
    const int conv_dim = 2 * nk * dk + nv * dv;
    if (int rc = upload_f32(L.conv_w, conv_weight, (size_t)conv_dim * kernel_dim)) return rc;
    if (int rc = upload_f32(L.a_log, a_log, nv)) return rc;
    if (int rc = upload_f32(L.dt_bias, dt_bias, nv)) return rc;
    if (int rc = upload_f32(L.la_norm_w, norm_weight, (size_t)nv * dv)) return rc;
    if (L.conv_state.ensure((size_t)conv_dim * kernel_dim * 4) || L.recur_state.ensure((size_t)nv * dk * dv * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    KR_HIP(hipMemset(L.conv_state.p, 0, (size_t)conv_dim * kernel_dim * 4));
    KR_HIP(hipMemset(L.recur_state.p, 0, (size_t)nv * dk * dv * 4));
    s->layers.push_back(std::move(L)); s->graph_ok = false;
    return KR_OK;
}

extern "C" int kr_decode_add_gqa_layer(kr_decode_store* s, int input_norm_id, int post_attn_norm_id, int q_wid, int k_wid, int v_wid, int o_wid,
                                       const float* q_norm, int q_norm_len, const float* k_norm, int k_norm_len, int gated, int num_heads,
                                       int num_kv_heads, int head_dim, float sm_scale) {
    if (int rc = need_cfg(s)) return rc;
    for (int id : {q_wid, k_wid, v_wid, o_wid}) if (int rc = chk_wid(s, id, "add_decode_gqa_layer")) return rc;
    if (head_dim > 256 || head_dim % 8 != 0) return kr_fail(KR_ERR_VALUE, "head_dim %d unsupported", head_dim);
    KR_HIP(hipSetDevice(s->eng->device));
    DLayer L; L.input_norm = input_norm_id; L.post_norm = post_attn_norm_id; L.attn = ATTN_GQA;
    L.q_wid = q_wid; L.k_wid = k_wid; L.v_wid = v_wid; L.o_wid = o_wid; L.gated = gated; L.nh = num_heads; L.nkv = num_kv_heads; L.hd = head_dim;
    L.sm_scale = sm_scale; L.q_norm_len = q_norm ? q_norm_len : 0; L.k_norm_len = k_norm ? k_norm_len : 0;
    if (q_norm) if (int rc = upload_f32(L.q_norm, q_norm, q_norm_len)) return rc;
    if (k_norm) if (int rc = upload_f32(L.k_norm, k_norm, k_norm_len)) return rc;
    s->layers.push_back(std::move(L)); s->graph_ok = false;
    return KR_OK;
}

// add_decode_mla_layer (decode.rs:2131): w_kc / w_vc arrive as bf16 [nh, nd, klr] / [nh, vhd, klr] and are widened to f32 like the
// reference does (:2157-2160); q_proj_wid < 0 selects the LoRA query path (q_a_proj -> q_a_norm -> q_b_proj).
extern "C" int kr_decode_add_mla_layer(kr_decode_store* s, int input_norm_id, int post_attn_norm_id, int kv_a_proj_wid, int o_proj_wid,
                                       int q_proj_wid, int q_a_proj_wid, int q_b_proj_wid, const uint16_t* w_kc_bf16, size_t w_kc_len,
                                       const uint16_t* w_vc_bf16, size_t w_vc_len, const float* kv_a_norm, int kv_a_norm_len,
                                       const float* q_a_norm, int q_a_norm_len, const float* rope_cos, const float* rope_sin,
                                       int rope_max_seq, int num_heads, int kv_lora_rank, int qk_nope_dim, int qk_rope_dim, int v_head_dim,
                                       float sm_scale) {
    if (int rc = need_cfg(s)) return rc;
    if (!w_kc_bf16 || !w_vc_bf16 || !kv_a_norm || !rope_cos || !rope_sin) return kr_fail(KR_ERR_VALUE, "add_decode_mla_layer: null argument");
    for (int id : {kv_a_proj_wid, o_proj_wid}) if (int rc = chk_wid(s, id, "add_decode_mla_layer")) return rc;
    if (q_proj_wid >= 0) { if (int rc = chk_wid(s, q_proj_wid, "add_decode_mla_layer")) return rc; }
    else for (int id : {q_a_proj_wid, q_b_proj_wid}) if (int rc = chk_wid(s, id, "add_decode_mla_layer (q LoRA)")) return rc;
    const int nh = num_heads, klr = kv_lora_rank, nd = qk_nope_dim, rd = qk_rope_dim, vhd = v_head_dim;
    if (klr % 64 || klr > 576 || nd > 640 || nd % 8 || rd % 16 || rd > 256 || nh <= 0 || vhd <= 0)
        return kr_fail(KR_ERR_VALUE, "MLA geometry unsupported (kv_lora_rank %d, nope %d, rope %d)", klr, nd, rd);
    if (w_kc_len != (size_t)nh * nd * klr || w_vc_len != (size_t)nh * vhd * klr || kv_a_norm_len != klr)
        return kr_fail(KR_ERR_VALUE, "MLA weight lengths do not match the geometry");
    if (s->weights[kv_a_proj_wid]->rows != klr + rd) return kr_fail(KR_ERR_VALUE, "kv_a_proj rows %d != kv_lora_rank + rope", s->weights[kv_a_proj_wid]->rows);
    KR_HIP(hipSetDevice(s->eng->device));
    DLayer L; L.input_norm = input_norm_id; L.post_norm = post_attn_norm_id; L.attn = ATTN_MLA;
    L.kva_wid = kv_a_proj_wid; L.o_wid = o_proj_wid; L.mq_wid = q_proj_wid; L.mqa_wid = q_a_proj_wid; L.mqb_wid = q_b_proj_wid;
    L.nh = nh; L.klr = klr; L.nd = nd; L.rd = rd; L.vhd = vhd; L.sm_scale = sm_scale; L.mla_rope_seq = rope_max_seq;
    std::vector<float> tmp(w_kc_len > w_vc_len ? w_kc_len : w_vc_len);
    auto widen = [&](const uint16_t* src, size_t n) { for (size_t i = 0; i < n; i++) { uint32_t b = (uint32_t)src[i] << 16; memcpy(&tmp[i], &b, 4); } };
    widen(w_kc_bf16, w_kc_len); if (int rc = upload_f32(L.w_kc, tmp.data(), w_kc_len)) return rc;
    widen(w_vc_bf16, w_vc_len); if (int rc = upload_f32(L.w_vc, tmp.data(), w_vc_len)) return rc;
    if (int rc = upload_f32(L.kv_a_norm, kv_a_norm, klr)) return rc;
    if (q_a_norm && q_a_norm_len > 0) { if (int rc = upload_f32(L.q_a_norm, q_a_norm, q_a_norm_len)) return rc; L.q_a_norm_len = q_a_norm_len; }
    if (int rc = upload_f32(L.mla_cos, rope_cos, (size_t)rope_max_seq * (rd / 2))) return rc;
    if (int rc = upload_f32(L.mla_sin, rope_sin, (size_t)rope_max_seq * (rd / 2))) return rc;
    s->weight_bytes += (w_kc_len + w_vc_len) * 4;
    s->layers.push_back(std::move(L)); s->graph_ok = false;
    return KR_OK;
}

// set_moe_store (decode.rs:2250-2266): bind the engine that owns the routed experts and routers.  May come before or after the builder calls.
extern "C" int kr_decode_set_moe_store(kr_decode_store* s, kr_engine* eng) {
    if (int rc = chk_store(s)) return rc;
    if (!eng) return kr_fail(KR_ERR_VALUE, "null engine");
    if (eng == s->eng) return KR_OK;
    if (!s->own_eng) return kr_fail(KR_ERR_STATE, "MoE store already set");
    if (eng->device != s->eng->device) return kr_fail(KR_ERR_VALUE, "engine lives on device %d, the decode store on device %d", eng->device, s->eng->device);
    KR_HIP(hipSetDevice(eng->device));
    KR_HIP(hipDeviceSynchronize());
    for (auto& L : s->layers) if (L.mlp == MLP_MOE && L.moe_layer >= (int)eng->layers.size())
        return kr_fail(KR_ERR_VALUE, "moe_layer_idx %d out of range (engine has %zu MoE layers)", L.moe_layer, eng->layers.size());
    kr_engine_destroy(s->eng);
    s->eng = eng; s->own_eng = false; s->graph_ok = false; s->last_stream = nullptr;
    return KR_OK;
}

extern "C" int kr_decode_set_layer_moe(kr_decode_store* s, int layer, int moe_layer_idx, int shared_gate_up_wid, int shared_down_wid, int shared_gate_wid) {
    if (int rc = need_cfg(s)) return rc;
    if (layer < 0 || layer >= (int)s->layers.size()) return kr_fail(KR_ERR_VALUE, "layer %d out of range", layer);
    if (moe_layer_idx < 0 || (!s->own_eng && moe_layer_idx >= (int)s->eng->layers.size())) return kr_fail(KR_ERR_VALUE, "moe_layer_idx %d out of range", moe_layer_idx);
    if ((shared_gate_up_wid >= 0) != (shared_down_wid >= 0)) return kr_fail(KR_ERR_VALUE, "shared gate_up and down weights must be set together");
    for (int id : {shared_gate_up_wid, shared_down_wid, shared_gate_wid}) if (id >= 0) if (int rc = chk_wid(s, id, "set_decode_layer_moe")) return rc;
    DLayer& L = s->layers[layer];
    L.mlp = MLP_MOE; L.moe_layer = moe_layer_idx; L.sgu_wid = shared_gate_up_wid; L.sd_wid = shared_down_wid; L.sg_wid = shared_gate_wid;
    s->graph_ok = false;
    return KR_OK;
}

extern "C" int kr_decode_set_layer_dense(kr_decode_store* s, int layer, int gate_wid, int up_wid, int down_wid) {
    if (int rc = need_cfg(s)) return rc;
    if (layer < 0 || layer >= (int)s->layers.size()) return kr_fail(KR_ERR_VALUE, "layer %d out of range", layer);
    for (int id : {gate_wid, up_wid, down_wid}) if (int rc = chk_wid(s, id, "set_decode_layer_dense")) return rc;
    DLayer& L = s->layers[layer];
    L.mlp = MLP_DENSE; L.gate_wid = gate_wid; L.up_wid = up_wid; L.down_wid = down_wid;
    s->graph_ok = false;
    return KR_OK;
}

extern "C" int kr_decode_set_rope(kr_decode_store* s, const float* cos_t, const float* sin_t, int half_dim, int max_seq) {
    if (int rc = need_cfg(s)) return rc;
    KR_HIP(hipSetDevice(s->eng->device));
    if (int rc = upload_f32(s->rope_cos, cos_t, (size_t)half_dim * max_seq)) return rc;
    if (int rc = upload_f32(s->rope_sin, sin_t, (size_t)half_dim * max_seq)) return rc;
    s->rope_half = half_dim; s->max_rope_seq = max_seq; s->graph_ok = false;
    return KR_OK;
}

static size_t maxz(size_t a, size_t b) { return a > b ? a : b; }

// finalize_decode (decode.rs:2471): size scratch, check ids
extern "C" int kr_decode_finalize(kr_decode_store* s) {
    if (int rc = need_cfg(s)) return rc;
    if ((int)s->layers.size() != s->n_layers) return kr_fail(KR_ERR_STATE, "configured %d layers but %zu were added", s->n_layers, s->layers.size());
    KR_HIP(hipSetDevice(s->eng->device));
    size_t pa = 0, pb = 0, qb = 0, kb = 0, vb = 0, zb = 0, ro = 0, ao = 0, gb = 0, dg = 0, lb = 0, fq = 64;
    for (auto& L : s->layers) {
        if (L.attn == ATTN_LA) {
            pa = maxz(pa, s->weights[L.qkvz_wid]->rows); pb = maxz(pb, s->weights[L.ba_wid]->rows);
            qb = maxz(qb, (size_t)L.nv * L.dk); kb = maxz(kb, (size_t)L.nv * L.dk); vb = maxz(vb, (size_t)L.nv * L.dv); zb = maxz(zb, (size_t)L.nv * L.dv);
            ro = maxz(ro, (size_t)L.nv * L.dv); ao = maxz(ao, s->weights[L.out_wid]->cols); gb = maxz(gb, L.nv); fq = maxz(fq, (size_t)L.nk * 2 * L.dk);
        } else if (L.attn == ATTN_GQA) {
            pa = maxz(pa, s->weights[L.q_wid]->rows); kb = maxz(kb, s->weights[L.k_wid]->rows); vb = maxz(vb, s->weights[L.v_wid]->rows);
            qb = maxz(qb, (size_t)L.nh * L.hd); zb = maxz(zb, (size_t)L.nh * L.hd); ao = maxz(ao, s->weights[L.o_wid]->cols);
        }
        else if (L.attn == ATTN_MLA) {
            const size_t hd = (size_t)L.nd + L.rd;
            pa = maxz(pa, (size_t)L.nh * hd); kb = maxz(kb, (size_t)L.klr + L.rd); qb = maxz(qb, (size_t)L.nh * L.klr); zb = maxz(zb, (size_t)L.nh * L.rd);
            lb = maxz(lb, (size_t)L.nh * L.klr); ao = maxz(ao, maxz((size_t)L.nh * L.vhd, s->weights[L.o_wid]->cols));
            if (L.mq_wid < 0) pb = maxz(pb, maxz(s->weights[L.mqa_wid]->rows, s->weights[L.mqb_wid]->cols));
        }
        if (L.mlp == MLP_DENSE) dg = maxz(dg, 2 * (size_t)s->weights[L.down_wid]->cols);
    }
    ao = maxz(ao, (size_t)s->hidden);
    {   // activation images: K = hidden for the norm outputs, K = widest attention output for img_attn
        const size_t kh = ((size_t)s->hidden + 127) / 128 * 128, ka = (ao + 127) / 128 * 128;
        if (s->img_in.ensure(kr_act_image_bytes((int)kh)) || s->img_post.ensure(kr_act_image_bytes((int)kh)) || s->img_post_bf16.ensure(kr_act_image_bytes((int)kh)) ||
            s->img_attn.ensure(kr_act_image_bytes((int)ka)))
            return kr_fail(KR_ERR_HIP, "hipMalloc of the activation images failed");
    }
    if (s->proj_a.ensure(maxz(pa, 64) * 4) || s->proj_b.ensure(maxz(pb, 64) * 4) || s->qbuf.ensure(maxz(qb, 64) * 4) || s->kbuf.ensure(maxz(kb, 64) * 4) ||
        s->vbuf.ensure(maxz(vb, 64) * 4) || s->zbuf.ensure(maxz(zb, 64) * 4) || s->recur_out.ensure(maxz(ro, 64) * 4) ||
        s->attn_out.ensure(maxz(ao, 64) * 4) || s->gbuf.ensure(maxz(gb, 64) * 4) || s->betabuf.ensure(maxz(gb, 64) * 4) ||
        s->gatebuf.ensure(maxz(zb, 64) * 4) || s->latbuf.ensure(maxz(lb, 64) * 4) || s->f_qk.ensure(fq * 4))
        return kr_fail(KR_ERR_HIP, "hipMalloc of decode scratch failed");
    if (dg) { if (s->dense_gu.ensure(dg * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc failed"); KR_HIP(hipMemset(s->dense_gu.p, 0, s->dense_gu.bytes)); }
    KR_HIP(hipMemset(s->proj_a.p, 0, s->proj_a.bytes));
    KR_HIP(hipMemset(s->proj_b.p, 0, s->proj_b.bytes));
    KR_HIP(hipMemset(s->attn_out.p, 0, s->attn_out.bytes));
    s->graph_ok = false;
    return KR_OK;
}

// set_decode_state (decode.rs:2640): host pointers per layer (NULL where not applicable); contents are copied to HBM
extern "C" int kr_decode_set_state(kr_decode_store* s, int seq_len, int kv_max_seq, const uint16_t* const* kv_k, const uint16_t* const* kv_v,
                                   const float* const* conv_state, const float* const* recur_state) {
    if (int rc = need_cfg(s)) return rc;
    (void)seq_len;
    KR_HIP(hipSetDevice(s->eng->device));
    KR_HIP(hipDeviceSynchronize());          // steps still in flight on the engine's (non-blocking) streams write these buffers
    s->kv_max_seq = kv_max_seq;
    for (size_t i = 0; i < s->layers.size(); i++) {
        DLayer& L = s->layers[i];
        if (L.attn == ATTN_GQA) {
            const size_t n = (size_t)kv_max_seq * L.nkv * L.hd * (s->kv_fp8 ? 1 : 2);
            if (L.kv_k.ensure(n) || L.kv_v.ensure(n)) return kr_fail(KR_ERR_HIP, "hipMalloc of KV cache failed");
            if (kv_k && kv_k[i]) KR_HIP(hipMemcpy(L.kv_k.p, kv_k[i], n, hipMemcpyHostToDevice)); else KR_HIP(hipMemset(L.kv_k.p, 0, n));
            if (kv_v && kv_v[i]) KR_HIP(hipMemcpy(L.kv_v.p, kv_v[i], n, hipMemcpyHostToDevice)); else KR_HIP(hipMemset(L.kv_v.p, 0, n));
        } else if (L.attn == ATTN_MLA) {   // kv_k[i] = compressed-KV cache, kv_v[i] = k_pe cache (set_decode_state's mla_ckv_ptrs / mla_kpe_ptrs)
            const size_t esz = s->kv_fp8 ? 1 : 2, nc = (size_t)kv_max_seq * L.klr * esz, np = (size_t)kv_max_seq * L.rd * esz;
            if (L.kv_k.ensure(nc) || L.kv_v.ensure(np)) return kr_fail(KR_ERR_HIP, "hipMalloc of MLA cache failed");
            if (kv_k && kv_k[i]) KR_HIP(hipMemcpy(L.kv_k.p, kv_k[i], nc, hipMemcpyHostToDevice)); else KR_HIP(hipMemset(L.kv_k.p, 0, nc));
            if (kv_v && kv_v[i]) KR_HIP(hipMemcpy(L.kv_v.p, kv_v[i], np, hipMemcpyHostToDevice)); else KR_HIP(hipMemset(L.kv_v.p, 0, np));
        } else if (L.attn == ATTN_LA) {
            const size_t cn = (size_t)(2 * L.nk * L.dk + L.nv * L.dv) * L.kd * 4, rn = (size_t)L.nv * L.dk * L.dv * 4;
            if (conv_state && conv_state[i]) KR_HIP(hipMemcpy(L.conv_state.p, conv_state[i], cn, hipMemcpyHostToDevice));
            if (recur_state && recur_state[i]) KR_HIP(hipMemcpy(L.recur_state.p, recur_state[i], rn, hipMemcpyHostToDevice));
        }
    }
    s->graph_ok = false;
    return KR_OK;
}

// fresh request: zero KV / latent caches, conv and recurrent states (what the perplexity harness does per window: new SequenceKVState +
// layer.attention.reset_state(), perplexity/measure_ppl.py:199-206)
extern "C" int kr_decode_reset_state(kr_decode_store* s, int kv_max_seq) {
    if (int rc = need_cfg(s)) return rc;
    if (kv_max_seq <= 0) return kr_fail(KR_ERR_VALUE, "kv_max_seq must be positive, got %d", kv_max_seq);
    if (int rc = kr_decode_set_state(s, 0, kv_max_seq, nullptr, nullptr, nullptr, nullptr)) return rc;
    for (auto& L : s->layers) {
        if (L.attn != ATTN_LA) continue;
        KR_HIP(hipMemset(L.conv_state.p, 0, (size_t)(2 * L.nk * L.dk + L.nv * L.dv) * L.kd * 4));
        KR_HIP(hipMemset(L.recur_state.p, 0, (size_t)L.nv * L.dk * L.dv * 4));
    }
    return KR_OK;
}

// synthetic per-request state with the distributions of bench_decode_synthetic (decode.rs:5421-5424, 4402-4411)
extern "C" int kr_decode_fill_state_synthetic(kr_decode_store* s, int kv_max_seq, uint64_t seed) {
    if (int rc = need_cfg(s)) return rc;
    KR_HIP(hipSetDevice(s->eng->device));
    s->kv_max_seq = kv_max_seq;
    for (size_t i = 0; i < s->layers.size(); i++) {
        DLayer& L = s->layers[i];
        if (L.attn == ATTN_GQA || L.attn == ATTN_MLA) {
            // GQA: K / V [kv_max_seq, nkv*hd]; MLA: compressed KV [kv_max_seq, klr] / rope keys [kv_max_seq, rd].  FP16 pattern of
            // bench_decode_synthetic (decode.rs:4402-4411) or its E4M3 twin when the caches hold the reference's GPU dtype
            const size_t nk_ = L.attn == ATTN_GQA ? (size_t)kv_max_seq * L.nkv * L.hd : (size_t)kv_max_seq * L.klr;
            const size_t nv_ = L.attn == ATTN_GQA ? nk_ : (size_t)kv_max_seq * L.rd;
            const size_t esz = s->kv_fp8 ? 1 : 2;
            if (L.kv_k.ensure(nk_ * esz) || L.kv_v.ensure(nv_ * esz)) return kr_fail(KR_ERR_HIP, "hipMalloc of KV cache failed");
            if (s->kv_fp8) {
                kr_launch_fill_e4m3_kv((uint8_t*)L.kv_k.p, nk_, seed + i * 4 + 0, s->eng->stream);
                kr_launch_fill_e4m3_kv((uint8_t*)L.kv_v.p, nv_, seed + i * 4 + 1, s->eng->stream);
            } else {
                kr_launch_fill_fp16_kv((uint16_t*)L.kv_k.p, nk_, seed + i * 4 + 0, s->eng->stream);
                kr_launch_fill_fp16_kv((uint16_t*)L.kv_v.p, nv_, seed + i * 4 + 1, s->eng->stream);
            }
        } else if (L.attn == ATTN_LA) {
            kr_launch_fill_uniform_f32((float*)L.conv_state.p, (size_t)(2 * L.nk * L.dk + L.nv * L.dv) * L.kd, 0.1f, seed + i * 4 + 2, s->eng->stream);
            kr_launch_fill_uniform_f32((float*)L.recur_state.p, (size_t)L.nv * L.dk * L.dv, 0.01f, seed + i * 4 + 3, s->eng->stream);
        }
    }
    KR_HIP(hipStreamSynchronize(s->eng->stream));
    s->graph_ok = false;
    return KR_OK;
}

extern "C" int kr_decode_get_state(kr_decode_store* s, int layer, uint16_t* kv_k, uint16_t* kv_v, float* conv_state, float* recur_state) {
    if (int rc = need_cfg(s)) return rc;
    if (layer < 0 || layer >= (int)s->layers.size()) return kr_fail(KR_ERR_VALUE, "layer out of range");
    KR_HIP(hipSetDevice(s->eng->device));
    KR_HIP(hipStreamSynchronize(s->eng->stream));
    DLayer& L = s->layers[layer];
    if (L.attn == ATTN_GQA) {
        const size_t n = (size_t)s->kv_max_seq * L.nkv * L.hd * (s->kv_fp8 ? 1 : 2);
        if (kv_k) KR_HIP(hipMemcpy(kv_k, L.kv_k.p, n, hipMemcpyDeviceToHost));
        if (kv_v) KR_HIP(hipMemcpy(kv_v, L.kv_v.p, n, hipMemcpyDeviceToHost));
    } else if (L.attn == ATTN_MLA) {
        const size_t esz = s->kv_fp8 ? 1 : 2;
        if (kv_k) KR_HIP(hipMemcpy(kv_k, L.kv_k.p, (size_t)s->kv_max_seq * L.klr * esz, hipMemcpyDeviceToHost));
        if (kv_v) KR_HIP(hipMemcpy(kv_v, L.kv_v.p, (size_t)s->kv_max_seq * L.rd * esz, hipMemcpyDeviceToHost));
    } else if (L.attn == ATTN_LA) {
        if (conv_state) KR_HIP(hipMemcpy(conv_state, L.conv_state.p, (size_t)(2 * L.nk * L.dk + L.nv * L.dv) * L.kd * 4, hipMemcpyDeviceToHost));
        if (recur_state) KR_HIP(hipMemcpy(recur_state, L.recur_state.p, (size_t)L.nv * L.dk * L.dv * 4, hipMemcpyDeviceToHost));
    }
    return KR_OK;
}

// ------------------------------------------------------------------------------------------------
// one decode step: the launch sequence of decode_step (decode.rs:2690-3520)
// ------------------------------------------------------------------------------------------------

static int enqueue_step(kr_decode_store* s, hipStream_t st) {
    kr_engine* e = s->eng;
    const int H = s->hidden;
    float* hid = (float*)s->hid.p; float* res = (float*)s->res.p;
    const KrStep* step = (const KrStep*)s->step_dev.p;
    // the value the next fused add+RMSNorm adds to the residual: embedding row, attention output, or the previous MoE epilogue
    KrNormSrc src{}; src.mode = 1; src.emb = (const float*)s->embedding.p; src.step = step;
    const KrNormSrc from_hidden{};  // mode 0
    bool first = true;
    float* res_cur = res;   // where the residual stream currently lives: a launch whose workgroups all read the residual writes the OTHER buffer
    float* const res_b = (float*)s->res2.p;
    auto other = [&](float* r) { return r == res ? res_b : res; };
    // KR_DECODE_FAST (kr_decode_fast.hip): tolerance-mode kernels for the pieces whose geometry they cover; everything else stays exact
    // expert-parallel decode: router, attention, norms and the shared expert are replicated; every rank runs the routed experts of ITS slice
    // and the k expert rows are summed over the ranks (each row is non-zero on exactly one rank, so the sum is exact: logits equal single-engine decode
    // bit for bit) before the combine in routing order.  Reference analogue: python/krasis/gpu_prefill.py:3700-3790 (local subset + partial-sum reduce).
    // Under KR_DECODE_FAST the expert-parallel form is lighter: every rank's down launch leaves its PARTIAL combine (rsf * sum over its own slots; the layer's
    // shared expert is evaluated by ONE rank, layer mod world, and added there) in the hidden buffer and ONE all-reduce of [hidden] f32 per MoE layer sums the
    // partials -- 1 / k of the bytes of the exact form, and no combine left for the next norm.  The sum order over the ranks is RCCL's: tolerance mode.
    const bool ep_dec = kr_ep_decode_active(e);
    const bool fast = s->decode_fast && s->use_images && H % 128 == 0 && H <= 4096 && s->f_qk.p;
    for (size_t li = 0; li < s->layers.size(); li++) {
        DLayer& L = s->layers[li];
        // INT16 activation images: built once by the kernel that produces an activation, copied by every workgroup of the
        // matvec launch that consumes it (instead of being re-quantised per workgroup).  INT4 consumers only.
        const bool img_ok = s->use_images && H % 128 == 0;
        // a launch takes one weight width: the matrices that share an activation image must agree (INT4 or INT8)
        auto img_w = [&](int wid, int like = -1) { return wid >= 0 && (like < 0 || s->weights[wid]->ms.bits == s->weights[like]->ms.bits); };
        bool in_img = false;
        if (img_ok) {
            if (L.attn == ATTN_LA) in_img = img_w(L.qkvz_wid) && img_w(L.ba_wid, L.qkvz_wid);
            else if (L.attn == ATTN_GQA) in_img = img_w(L.q_wid) && img_w(L.k_wid, L.q_wid) && img_w(L.v_wid, L.q_wid);
        }
        // KR_DECODE_FAST on MLA layers (round 4): the input add + RMSNorm folded into the first projection launch (kv_a | q, or kv_a | q_a on the LoRA path) and
        // the o projection on the K-split tree-sum matvec; the attention launches themselves (absorb, scores, weighted sum, w_vc) keep the reference's order
        const int mla_w2 = L.attn == ATTN_MLA ? (L.mq_wid >= 0 ? L.mq_wid : L.mqa_wid) : -1;
        const bool mla_in = fast && L.attn == ATTN_MLA && img_w(L.kva_wid) && img_w(mla_w2, L.kva_wid);
        // FAST: the input add+RMSNorm folded into the first projection launch (every workgroup rebuilds the normalised vector with tree sums)
        bool did_in = false, la_conv_done = false;
        if (fast && (in_img || mla_in) && src.mode != 2) {
            KrFdmArgs fa{};
            fa.mode = 1; fa.hid_in = hid; fa.res_in = res_cur; fa.res_out = other(res_cur); fa.norm_w = (const float*)s->norms[L.input_norm]->p;
            fa.first = first ? 1 : 0; fa.eps = s->eps; fa.bias_one = s->norm_bias_one;
            if (src.mode == 1) { fa.emb = src.emb; fa.step = src.step; }
            int total = 0;
            auto add = [&](int wid, float* y) { fa.mm.m[fa.mm.n] = mv(s, wid); fa.mm.y[fa.mm.n] = y; total += (fa.mm.m[fa.mm.n].N + 7) / 8; fa.mm.tile_end[fa.mm.n] = total; fa.mm.n++; };
            if (L.attn == ATTN_LA) {
                add(L.ba_wid, (float*)s->proj_b.p); add(L.qkvz_wid, (float*)s->proj_a.p);   // the ba tiles first: their lanes carry the libm gate epilogue, the first workgroups start earliest
                const int hr = L.nv / L.nk;
                if (L.dv == 128 && (L.dk == 128 || L.dk == 64) && L.nv == L.nk * hr && L.kd == 4 && img_w(L.out_wid)) {
                    fa.conv_state = (float*)L.conv_state.p; fa.conv_w = (const float*)L.conv_w.p; fa.qk_out = (float*)s->f_qk.p; fa.v_out = (float*)s->vbuf.p; fa.z_out = (float*)s->zbuf.p;
                    fa.nk = L.nk; fa.dk = L.dk; fa.hr = hr; fa.dv = L.dv; fa.conv_mi = 1; fa.gate_mi = 0;
                    fa.a_log = (const float*)L.a_log.p; fa.dt_bias = (const float*)L.dt_bias.p; fa.ge_out = (float*)s->gbuf.p; fa.beta_out = (float*)s->betabuf.p;
                }
            } else if (L.attn == ATTN_MLA) { add(L.kva_wid, (float*)s->kbuf.p); add(mla_w2, L.mq_wid >= 0 ? (float*)s->proj_a.p : (float*)s->proj_b.p); }
            else { add(L.q_wid, (float*)s->proj_a.p); add(L.k_wid, (float*)s->kbuf.p); add(L.v_wid, (float*)s->vbuf.p); }
            prof_mark(s, PK_MATVEC, st);
            did_in = 0 == kr_launch_fdm(fa, st);
            prof_mark(s, -1, st);
            if (did_in) { la_conv_done = fa.conv_state != nullptr; res_cur = other(res_cur); }
        }
        if (!did_in) {
            PROF(PK_RMSNORM, kr_launch_fused_add_rmsnorm(src, hid, res_cur, res, (const float*)s->norms[L.input_norm]->p, H, s->eps, first ? 1 : 0, s->norm_bias_one, st,
                                                         in_img ? s->img_in.p : nullptr));
            res_cur = res;
        }
        first = false; src = from_hidden;
        const void* xin = in_img ? (const void*)s->img_in.p : (const void*)hid; const int xin_kind = in_img ? 2 : 1;
        // FAST out / o projection from the attention output's INT16 image: K split over the waves of a workgroup, tree sums
        auto out_proj = [&](int wid) {
            if (fast) {
                KrFdmArgs fo{}; fo.mode = 0; fo.img = s->img_attn.p; fo.mm.n = 1; fo.mm.m[0] = mv(s, wid); fo.mm.y[0] = hid; fo.mm.tile_end[0] = (fo.mm.m[0].N + 7) / 8;
                prof_mark(s, PK_OUT_PROJ, st);
                const int rc = kr_launch_fdm(fo, st);
                prof_mark(s, -1, st);
                if (rc == 0) return;
            }
            PROF(PK_MATVEC, kr_launch_matvec(mv(s, wid), s->img_attn.p, 2, hid, st));
        };
        if (L.attn == ATTN_LA && la_conv_done) {
            KrFlaArgs a{};
            a.qk = (const float*)s->f_qk.p; a.v = (const float*)s->vbuf.p; a.z = (const float*)s->zbuf.p; a.ge = (const float*)s->gbuf.p; a.beta = (const float*)s->betabuf.p;
            a.scale = L.la_scale; a.state = (float*)L.recur_state.p;
            a.norm_w = (const float*)L.la_norm_w.p; a.out = (float*)s->attn_out.p; a.img_out = s->img_attn.p; a.img_k = s->weights[L.out_wid]->ms.view().ng * 128;
            a.nk = L.nk; a.nv = L.nv; a.hr = L.nv / L.nk; a.dk = L.dk; a.dv = L.dv; a.eps = s->eps;
            prof_mark(s, PK_LA_RECUR, st);
            if (kr_launch_fla(a, st)) return kr_fail(KR_ERR_VALUE, "unsupported linear-attention geometry (fast mode)");
            prof_mark(s, -1, st);
            out_proj(L.out_wid);
        } else if (L.attn == ATTN_LA) {
            bool la_heads = false;      // exact step: conv + gates ran as the in-projection's epilogue, the recurrence runs one workgroup per value head
            if (!did_in) {
                const KrMatDev mats[2] = {mv(s, L.qkvz_wid), mv(s, L.ba_wid)};
                float* ys[2] = {(float*)s->proj_a.p, (float*)s->proj_b.p};
                if (s->opt_la_heads && s->fuse_la && xin_kind == 2 && mats[0].bits == mats[1].bits && s->f_qk.p && L.kd == 4 && L.nv == L.nk * (L.nv / L.nk) &&
                    (L.dk == 128 || L.dk == 64) && (L.dv == 128 || L.dv == 64)) {
                    KrCoLa la{};
                    la.conv_state = (float*)L.conv_state.p; la.conv_w = (const float*)L.conv_w.p; la.qk_out = (float*)s->f_qk.p; la.v_out = (float*)s->vbuf.p; la.z_out = (float*)s->zbuf.p;
                    la.nk = L.nk; la.dk = L.dk; la.hr = L.nv / L.nk; la.dv = L.dv; la.conv_mi = 0;
                    prof_mark(s, PK_MATVEC, st);
                    la_heads = 0 == kr_launch_multi_matvec_la(mats, ys, 2, xin, la, st);
                    prof_mark(s, -1, st);
                }
                if (la_heads) {}
                else if (mats[0].bits == mats[1].bits) PROF(PK_MATVEC, kr_launch_multi_matvec(mats, ys, 2, xin, xin_kind, st));
                else { PROF(PK_MATVEC, kr_launch_matvec(mats[0], hid, 1, ys[0], st)); PROF(PK_MATVEC, kr_launch_matvec(mats[1], hid, 1, ys[1], st)); }
            }
            KrLaArgs a{};
            a.qkvz = (const float*)s->proj_a.p; a.ba = (const float*)s->proj_b.p; a.conv_state = (float*)L.conv_state.p;
            a.conv_w = (const float*)L.conv_w.p; a.a_log = (const float*)L.a_log.p; a.dt_bias = (const float*)L.dt_bias.p; a.scale = L.la_scale;
            a.q = (float*)s->qbuf.p; a.k = (float*)s->kbuf.p; a.v = (float*)s->vbuf.p; a.z = (float*)s->zbuf.p; a.g = (float*)s->gbuf.p; a.beta = (float*)s->betabuf.p;
            a.nk = L.nk; a.nv = L.nv; a.dk = L.dk; a.dv = L.dv; a.hr = L.nv / L.nk;
            void* la_img = (img_ok && img_w(L.out_wid) && L.dv == 128) ? s->img_attn.p : nullptr;
            const int nt_la = a.hr * L.dv;
            const bool la_fused = s->fuse_la && nt_la <= 256 && nt_la % 64 == 0 && (L.dv == 128 || L.dv == 64) && L.nv == L.nk * a.hr && (L.dk == 128 || L.dk == 64);
            if (la_heads) {
                a.q = (float*)s->f_qk.p;      // [nk][2 dk] conv + SiLU outputs of q | k (normalised inside the launch)
                prof_mark(s, PK_LA_RECUR, st);
                if (kr_launch_la_step_heads(a, (float*)L.recur_state.p, (const float*)L.la_norm_w.p, (float*)s->attn_out.p, s->eps, st, la_img))
                    return kr_fail(KR_ERR_VALUE, "unsupported linear-attention geometry");
                prof_mark(s, -1, st);
            } else if (la_fused) {
                prof_mark(s, PK_LA_RECUR, st);
                (void)kr_launch_la_step(a, (float*)L.recur_state.p, (const float*)L.la_norm_w.p, (float*)s->attn_out.p, s->eps, st, la_img);
                prof_mark(s, -1, st);
            } else {
                PROF(PK_LA_CONV, kr_launch_la_conv(a, st));
                prof_mark(s, PK_LA_RECUR, st);
                if (kr_launch_la_recurrent_gnorm((float*)L.recur_state.p, a.q, a.k, a.v, a.g, a.beta, a.z, (const float*)L.la_norm_w.p, (float*)s->attn_out.p,
                                                 L.nv, L.dk, L.dv, s->eps, st, la_img))
                    return kr_fail(KR_ERR_VALUE, "unsupported linear-attention geometry");
                prof_mark(s, -1, st);
            }
            if (img_ok && img_w(L.out_wid) && L.dv == 128) out_proj(L.out_wid);
            else PROF(PK_MATVEC, kr_launch_matvec(mv(s, L.out_wid), s->attn_out.p, 1, hid, st));
        } else if (L.attn == ATTN_GQA) {
            if (!L.kv_k.p) return kr_fail(KR_ERR_STATE, "set_decode_state was not called (no KV cache for layer %zu)", li);
            if (!did_in) {
                const KrMatDev mats[3] = {mv(s, L.q_wid), mv(s, L.k_wid), mv(s, L.v_wid)};
                float* ys[3] = {(float*)s->proj_a.p, (float*)s->kbuf.p, (float*)s->vbuf.p};
                if (mats[0].bits == mats[1].bits && mats[0].bits == mats[2].bits) PROF(PK_MATVEC, kr_launch_multi_matvec(mats, ys, 3, xin, xin_kind, st));
                else for (int i = 0; i < 3; i++) PROF(PK_MATVEC, kr_launch_matvec(mats[i], hid, 1, ys[i], st));
            }
            KrGqaArgs a{};
            a.step = step; a.q_in = (const float*)s->proj_a.p; a.k_in = (const float*)s->kbuf.p; a.v_in = (const float*)s->vbuf.p;
            a.q_norm = L.q_norm_len ? (const float*)L.q_norm.p : nullptr; a.k_norm = L.k_norm_len ? (const float*)L.k_norm.p : nullptr;
            a.q_norm_per_head = L.q_norm_len == L.nh * L.hd; a.k_norm_per_head = L.k_norm_len == L.nkv * L.hd;
            a.rope_cos = (const float*)s->rope_cos.p; a.rope_sin = (const float*)s->rope_sin.p; a.rope_half = s->rope_half;
            a.k_cache = L.kv_k.p; a.v_cache = L.kv_v.p; a.kv_fp8 = s->kv_fp8; a.q_out = (float*)s->qbuf.p; a.gate = (float*)s->gatebuf.p;
            a.attn_out = (float*)s->attn_out.p; a.gated = L.gated; a.nh = L.nh; a.nkv = L.nkv; a.hd = L.hd; a.eps = s->eps; a.sm_scale = L.sm_scale;
            const bool o_img = img_ok && img_w(L.o_wid) && L.hd % 128 == 0 && s->weights[L.o_wid]->cols == L.nh * L.hd;
            a.img_out = o_img ? s->img_attn.p : nullptr;
            a.sc_g = s->kv_max_seq > s->gqa_split_min ? (float*)s->gqa_scores.p : nullptr;
            a.force_stream = s->opt_gqa_stream; a.tree_norm = fast ? 1 : 0;
            if (s->attn_fast && a.sc_g) { a.fd_o = (float*)s->fd_o.p; a.fd_ml = (float*)s->fd_ml.p; }
            PROF(PK_GQA, { if (!(fast && s->opt_gqa_fused && !a.sc_g && kr_launch_fgqa(a, s->kv_max_seq, st) == 0)) kr_launch_gqa(a, s->kv_max_seq, st); });      // KR_DECODE_FAST, short cache: one launch
            if (o_img) out_proj(L.o_wid);
            else PROF(PK_MATVEC, kr_launch_matvec(mv(s, L.o_wid), s->attn_out.p, 1, hid, st));
        }
        else if (L.attn == ATTN_MLA) {
            if (!L.kv_k.p) return kr_fail(KR_ERR_STATE, "set_decode_state was not called (no MLA cache for layer %zu)", li);
            if (L.mla_rope_seq < s->kv_max_seq) return kr_fail(KR_ERR_VALUE, "MLA rope table (%d) shorter than kv_max_seq (%d)", L.mla_rope_seq, s->kv_max_seq);
            float* kv_out = (float*)s->kbuf.p; float* q_full = (float*)s->proj_a.p;
            if (did_in) {          // KR_DECODE_FAST: kv_a | q (or kv_a | q_a) came out of the folded norm + projection launch above
                if (L.mq_wid < 0) {
                    if (L.q_a_norm_len) PROF(PK_RMSNORM, kr_launch_rmsnorm_seq((float*)s->proj_b.p, (const float*)L.q_a_norm.p, s->weights[L.mqa_wid]->rows, s->eps, st));
                    PROF(PK_MATVEC, kr_launch_matvec(mv(s, L.mqb_wid), s->proj_b.p, 1, q_full, st));
                }
            } else
            if (L.mq_wid >= 0) {   // direct query projection shares the activation with kv_a_proj: one launch
                const KrMatDev mats[2] = {mv(s, L.kva_wid), mv(s, L.mq_wid)};
                float* ys[2] = {kv_out, q_full};
                if (mats[0].bits == mats[1].bits) PROF(PK_MATVEC, kr_launch_multi_matvec(mats, ys, 2, hid, 1, st));
                else { PROF(PK_MATVEC, kr_launch_matvec(mats[0], hid, 1, ys[0], st)); PROF(PK_MATVEC, kr_launch_matvec(mats[1], hid, 1, ys[1], st)); }
            } else {               // LoRA: q_a_proj -> sequential RMSNorm -> q_b_proj (decode.rs:3036-3079)
                const KrMatDev mats[2] = {mv(s, L.kva_wid), mv(s, L.mqa_wid)};
                float* ys[2] = {kv_out, (float*)s->proj_b.p};
                if (mats[0].bits == mats[1].bits) PROF(PK_MATVEC, kr_launch_multi_matvec(mats, ys, 2, hid, 1, st));
                else { PROF(PK_MATVEC, kr_launch_matvec(mats[0], hid, 1, ys[0], st)); PROF(PK_MATVEC, kr_launch_matvec(mats[1], hid, 1, ys[1], st)); }
                if (L.q_a_norm_len) PROF(PK_RMSNORM, kr_launch_rmsnorm_seq((float*)s->proj_b.p, (const float*)L.q_a_norm.p, s->weights[L.mqa_wid]->rows, s->eps, st));
                PROF(PK_MATVEC, kr_launch_matvec(mv(s, L.mqb_wid), s->proj_b.p, 1, q_full, st));
            }
            KrMlaArgs a{};
            a.step = step; a.kv_out = kv_out; a.q_full = q_full; a.kv_a_norm = (const float*)L.kv_a_norm.p; a.w_kc = (const float*)L.w_kc.p;
            a.w_vc = (const float*)L.w_vc.p; a.rope_cos = (const float*)L.mla_cos.p; a.rope_sin = (const float*)L.mla_sin.p;
            a.ckv_cache = L.kv_k.p; a.kpe_cache = L.kv_v.p; a.kv_fp8 = s->kv_fp8; a.q_abs = (float*)s->qbuf.p; a.q_pe = (float*)s->zbuf.p;
            a.attn_lat = (float*)s->latbuf.p; a.v_proj = (float*)s->attn_out.p;
            a.nh = L.nh; a.klr = L.klr; a.nd = L.nd; a.rd = L.rd; a.vhd = L.vhd; a.eps = s->eps; a.sm_scale = L.sm_scale;
            a.sc_g = s->kv_max_seq > s->mla_split_min ? (float*)s->gqa_scores.p : nullptr;
            if (s->attn_fast && a.sc_g) { a.fast = 1; a.fd_o = (float*)s->fd_o.p; a.fd_ml = (float*)s->fd_ml.p; }
            a.decode_fast = fast ? 1 : 0; a.decode_fused = fast && s->opt_gqa_fused ? 1 : 0;
            PROF(PK_GQA, kr_launch_mla(a, s->kv_max_seq, st));
            bool o_done = false;
            if (fast && s->weights[L.o_wid]->cols == L.nh * L.vhd) {      // o projection straight from the f32 w_vc output: every workgroup quantises it, K split over the waves
                KrFdmArgs fo{}; fo.mode = 2; fo.hid_in = (const float*)s->attn_out.p; fo.mm.n = 1; fo.mm.m[0] = mv(s, L.o_wid); fo.mm.y[0] = hid; fo.mm.tile_end[0] = (fo.mm.m[0].N + 7) / 8;
                prof_mark(s, PK_OUT_PROJ, st);
                o_done = 0 == kr_launch_fdm(fo, st);
                prof_mark(s, -1, st);
            }
            if (!o_done) PROF(PK_MATVEC, kr_launch_matvec(mv(s, L.o_wid), s->attn_out.p, 1, hid, st));
        }
        // post-attention fused add+RMSNorm: folded into the router launch of MoE layers, its own launch otherwise
        const float* act = hid;   // normalised hidden the MLP block reads
        bool routed = false, moe_done = false;
        if (L.mlp == MLP_MOE) {
            if (s->own_eng || L.moe_layer >= (int)e->layers.size()) return kr_fail(KR_ERR_STATE, "set_moe_store was not called (MoE layer %d has no engine)", L.moe_layer);
            Layer& EL = e->layers[L.moe_layer];
            if (!EL.routing_present) return kr_fail(KR_ERR_STATE, "Routing weights not set for layer %d", L.moe_layer);
            if (!EL.w13.allocated() && !EL.gguf) return kr_fail(KR_ERR_STATE, "Model not loaded (MoE layer %d has no experts)", L.moe_layer);
            if (EL.gguf && ep_dec) return kr_fail(KR_ERR_STATE, "expert-parallel decode on native GGUF layers is not implemented (MoE layer %d)", L.moe_layer);
            if (fast && !EL.gguf) {   // norm + gate GEMV | select + gate|up + silu*up | down + combine: three launches, the MoE output lands in `hid`
                const bool has_shared = L.sgu_wid >= 0, has_gate = has_shared && L.sg_wid >= 0;
                KrFmoeArgs fa{}; KrMoeArgs& a = fa.m;
                a.shared_decode = 1; a.act_img = s->img_post.p; a.act_img_bf16 = s->img_post_bf16.p;
                a.ids = (const int32_t*)s->r_ids.p; a.wts = (const float*)s->r_w.p;
                a.B = 1; a.topk = s->topk; a.n_slots = s->topk + (has_shared ? 1 : 0); a.E = e->r_ne; a.H = H; a.I = EL.inter;
                a.w13 = EL.w13.view(); a.w2 = EL.w2.view();
                if (has_shared) { a.sw13 = mv(s, L.sgu_wid); a.sw2 = mv(s, L.sd_wid); a.I_shared = s->weights[L.sgu_wid]->rows / 2; }
                const int imax = has_shared && a.I_shared > a.I ? a.I_shared : a.I;
                a.gu_ld = 2 * imax; a.gu = (float*)s->moe_gu.p; a.eo = (float*)s->moe_eo.p;
                a.rsf = s->rsf; a.swiglu_limit = e->cfg.swiglu_limit; a.alpha = e->cfg.activation_alpha;
                a.act_mode = e->cfg.swiglu_limit > 0.0f ? KR_ACT_GPTOSS : KR_ACT_SILU_FUSED;
                bool ok = true;
                if (has_gate) {
                    if (a.w13.bits == a.sw13.bits && mv(s, L.sg_wid).bits == a.w13.bits) { a.sgate = mv(s, L.sg_wid); a.gate_out = (float*)s->gate_val.p; }
                    else ok = false;
                }
                fa.logits = (const float*)s->r_logits.p; fa.esc = EL.has_esc ? (const float*)EL.esc.p : nullptr; fa.scoring = s->scoring; fa.norm_topk = s->norm_topk;
                fa.hid_out = hid;
                if (ep_dec) {
                    kr_ep_slice(e, &a.e_lo, &a.e_hi, &a.e_sub);
                    fa.shared_skip = has_shared && kr_ep_rank(e) != (int)(li % (size_t)kr_ep_world(e)) ? 1 : 0;
                }
                if (ok && kr_fmoe_check(fa) == 0) {
                    KrFrtArgs ra{};
                    ra.gate_cm = EL.gate_cm.p; ra.gate_bf16 = EL.gate_bf16_exact; ra.bias = EL.has_bias ? (const float*)EL.bias.p : nullptr; ra.logits = (float*)s->r_logits.p;
                    ra.E = e->r_ne; ra.H = H; ra.hid_in = hid; ra.res_in = res_cur; ra.norm_w = (const float*)s->norms[L.post_norm]->p; ra.hid_out = (float*)s->hid2.p;
                    ra.res_out = other(res_cur); ra.eps = s->eps; ra.bias_one = s->norm_bias_one; ra.img_f32 = s->img_post.p; ra.img_bf16 = s->img_post_bf16.p;
                    prof_mark(s, PK_ROUTE_LOGITS, st);
                    const int rc = kr_launch_frt(ra, st);
                    prof_mark(s, -1, st);
                    if (rc == 0) {
                        // kr_fmoe_check above covers every refusal of these two launchers; a non-zero return here would leave the layer half-run
                        prof_mark(s, PK_MOE_W13, st); const int r13 = kr_launch_fw13(fa, st); prof_mark(s, -1, st);
                        prof_mark(s, PK_MOE_W2, st); const int r2 = kr_launch_fw2(fa, st); prof_mark(s, -1, st);
                        if (r13 || r2) return kr_fail(KR_ERR_STATE, "internal: KR_DECODE_FAST expert launch refused after kr_fmoe_check accepted layer %zu", li);
                        if (ep_dec) if (int rc = kr_ep_allreduce_on(e, hid, (size_t)H, st)) return rc;
                        res_cur = other(res_cur); src = from_hidden; moe_done = true;
                    }
                }
            }
            // KR_DECODE_FAST on a native-GGUF layer: the same three launches as for transposed experts -- the routed slots of the gate|up and down launches walk the
            // GGUF blocks (kr_gguf_dev.h: the block kernels' products, a row's blocks split over two waves), the shared expert keeps its transposed form
            if (!moe_done && fast && EL.gguf && src.mode != 2) {
                const bool has_shared = L.sgu_wid >= 0, has_gate = has_shared && L.sg_wid >= 0;
                KrFmoeArgs fa{}; KrMoeArgs& a = fa.m;
                fa.gguf = 1; fa.ggate = EL.g_gate.view(); fa.gup = EL.g_up.view(); fa.gdown = EL.g_down.view(); fa.act_f32 = (const float*)s->hid2.p;
                a.shared_decode = 1; a.act_img = s->img_post.p; a.act_img_bf16 = s->img_post_bf16.p;
                a.ids = (const int32_t*)s->r_ids.p; a.wts = (const float*)s->r_w.p;
                a.B = 1; a.topk = s->topk; a.n_slots = s->topk + (has_shared ? 1 : 0); a.E = e->cfg.n_routed_experts; a.H = H; a.I = EL.inter;
                if (has_shared) { a.sw13 = mv(s, L.sgu_wid); a.sw2 = mv(s, L.sd_wid); a.I_shared = s->weights[L.sgu_wid]->rows / 2; }
                const int imax = has_shared && a.I_shared > a.I ? a.I_shared : a.I;
                a.gu_ld = 2 * imax; a.gu = (float*)s->moe_gu.p; a.eo = (float*)s->moe_eo.p;
                a.rsf = s->rsf; a.swiglu_limit = e->cfg.swiglu_limit; a.alpha = e->cfg.activation_alpha;
                a.act_mode = e->cfg.swiglu_limit > 0.0f ? KR_ACT_GPTOSS : KR_ACT_SILU_FUSED;
                bool ok = e->r_ne == e->cfg.n_routed_experts;
                if (has_gate) {
                    if (mv(s, L.sg_wid).bits == a.sw13.bits) { a.sgate = mv(s, L.sg_wid); a.gate_out = (float*)s->gate_val.p; }
                    else ok = false;
                }
                fa.logits = (const float*)s->r_logits.p; fa.esc = EL.has_esc ? (const float*)EL.esc.p : nullptr; fa.scoring = s->scoring; fa.norm_topk = s->norm_topk;
                fa.hid_out = hid;
                if (ok && kr_fmoe_check(fa) == 0) {
                    KrFrtArgs ra{};
                    ra.gate_cm = EL.gate_cm.p; ra.gate_bf16 = EL.gate_bf16_exact; ra.bias = EL.has_bias ? (const float*)EL.bias.p : nullptr; ra.logits = (float*)s->r_logits.p;
                    ra.E = e->r_ne; ra.H = H; ra.hid_in = hid; ra.res_in = res_cur; ra.norm_w = (const float*)s->norms[L.post_norm]->p; ra.hid_out = (float*)s->hid2.p;
                    ra.res_out = other(res_cur); ra.eps = s->eps; ra.bias_one = s->norm_bias_one; ra.img_f32 = s->img_post.p; ra.img_bf16 = s->img_post_bf16.p;
                    prof_mark(s, PK_ROUTE_LOGITS, st);
                    const int rc = kr_launch_frt(ra, st);
                    prof_mark(s, -1, st);
                    if (rc == 0) {
                        prof_mark(s, PK_MOE_W13, st); const int r13 = kr_launch_fw13(fa, st); prof_mark(s, -1, st);
                        prof_mark(s, PK_MOE_W2, st); const int r2 = kr_launch_fw2(fa, st); prof_mark(s, -1, st);
                        if (r13 || r2) return kr_fail(KR_ERR_STATE, "internal: KR_DECODE_FAST expert launch refused after kr_fmoe_check accepted GGUF layer %zu", li);
                        res_cur = other(res_cur); src = from_hidden; moe_done = true;
                    }
                }
            }
            // ... and, for GGUF geometries those launches do not cover: the mode's norm + gate GEMV launch and the stand-alone select in front of the (exact) block kernels
            if (!moe_done && fast && EL.gguf && src.mode != 2) {
                KrFrtArgs ra{};
                ra.gate_cm = EL.gate_cm.p; ra.gate_bf16 = EL.gate_bf16_exact; ra.bias = EL.has_bias ? (const float*)EL.bias.p : nullptr; ra.logits = (float*)s->r_logits.p;
                ra.E = e->r_ne; ra.H = H; ra.hid_in = hid; ra.res_in = res_cur; ra.norm_w = (const float*)s->norms[L.post_norm]->p; ra.hid_out = (float*)s->hid2.p;
                ra.res_out = other(res_cur); ra.eps = s->eps; ra.bias_one = s->norm_bias_one; ra.img_f32 = s->img_post.p; ra.img_bf16 = s->img_post_bf16.p;
                prof_mark(s, PK_ROUTE_LOGITS, st);
                const int rc = kr_launch_frt(ra, st);
                prof_mark(s, -1, st);
                if (rc == 0) {
                    PROF(PK_ROUTE_SELECT, kr_launch_route_select((const float*)s->r_logits.p, EL.has_esc ? (const float*)EL.esc.p : nullptr, (int32_t*)s->r_ids.p, (float*)s->r_w.p, 1, e->r_ne, s->topk,
                                                                 s->scoring, s->norm_topk, KR_ROUTE_RULE_DECODE, 0, st));
                    routed = true; act = (const float*)s->hid2.p; res_cur = other(res_cur);
                }
            }
            if (!moe_done && !routed && s->fuse_router) {
                prof_mark(s, PK_ROUTE_LOGITS, st);
                routed = 0 == kr_launch_route_fused_decode(EL.gate_cm.p, EL.gate_bf16_exact, EL.has_bias ? (const float*)EL.bias.p : nullptr, (float*)s->r_logits.p,
                                                           (unsigned*)s->r_counter.p, EL.has_esc ? (const float*)EL.esc.p : nullptr, (int32_t*)s->r_ids.p,
                                                           (float*)s->r_w.p, e->r_ne, H, s->topk, s->scoring, s->norm_topk, nullptr, hid, res_cur,
                                                           (const float*)s->norms[L.post_norm]->p, (float*)s->hid2.p, other(res_cur), s->eps,
                                                           s->norm_bias_one, st, img_ok ? s->img_post.p : nullptr, img_ok ? s->img_post_bf16.p : nullptr);
                prof_mark(s, -1, st);
                if (routed) { act = (const float*)s->hid2.p; res_cur = other(res_cur); }
            }
        }
        if (moe_done) continue;
        // KR_DECODE_FAST on a dense MLP layer (V2-Lite's first, decode.rs:1693-1741; round 6): the post-attention add + RMSNorm folded into the gate | up projection launch
        // (every workgroup rebuilds the normalised vector with tree sums, as the attention in-projections of the mode do); the down projection keeps the exact kernel with
        // its silu * up + INT16 prologue -- its K (10 944) is beyond what a folded prologue holds
        if (fast && !routed && L.mlp == MLP_DENSE && src.mode == 0 && s->opt_dense_fast) {
            const KrMatDev g = mv(s, L.gate_wid), u = mv(s, L.up_wid);
            const int K = s->weights[L.down_wid]->cols;
            if (g.bits == u.bits && (g.bits == 4 || g.bits == 8)) {
                KrFdmArgs fd{};
                fd.mode = 1; fd.hid_in = hid; fd.res_in = res_cur; fd.res_out = other(res_cur); fd.norm_w = (const float*)s->norms[L.post_norm]->p; fd.eps = s->eps; fd.bias_one = s->norm_bias_one;
                fd.mm.n = 2; fd.mm.m[0] = g; fd.mm.y[0] = (float*)s->dense_gu.p; fd.mm.tile_end[0] = (g.N + 7) / 8;
                fd.mm.m[1] = u; fd.mm.y[1] = (float*)s->dense_gu.p + K; fd.mm.tile_end[1] = fd.mm.tile_end[0] + (u.N + 7) / 8;
                prof_mark(s, PK_MATVEC, st);
                const bool done = 0 == kr_launch_fdm(fd, st);
                prof_mark(s, -1, st);
                if (done) {
                    res_cur = other(res_cur);
                    PROF(PK_MATVEC, kr_launch_matvec(mv(s, L.down_wid), s->dense_gu.p, 1, hid, st, KR_ACT_SILU_MUL));
                    continue;
                }
            }
        }
        if (!routed) PROF(PK_RMSNORM, kr_launch_fused_add_rmsnorm(src, hid, res_cur, res_cur, (const float*)s->norms[L.post_norm]->p, H, s->eps, 0, s->norm_bias_one, st));
        if (L.mlp == MLP_MOE) {
            Layer& EL = e->layers[L.moe_layer];
            const int E = e->r_ne, k = s->topk;
            if (!routed) {
                PROF(PK_ROUTE_LOGITS, kr_launch_route_logits_decode(EL.gate_cm.p, EL.gate_bf16_exact, hid, EL.has_bias ? (const float*)EL.bias.p : nullptr, (float*)s->r_logits.p, 1, E, H, st));
                PROF(PK_ROUTE_SELECT, kr_launch_route_select((const float*)s->r_logits.p, EL.has_esc ? (const float*)EL.esc.p : nullptr, (int32_t*)s->r_ids.p, (float*)s->r_w.p, 1, E, k,
                                       s->scoring, s->norm_topk, KR_ROUTE_RULE_DECODE, 0, st));
            }
            const bool has_shared = L.sgu_wid >= 0;
            const bool has_gate = has_shared && L.sg_wid >= 0;
            if (EL.gguf) {
                // NATIVE GGUF EXPERTS in the decode step.  The reference's Rust decode_step always calls moe_forward_unified (decode.rs:3334), its
                // native-GGUF layers decode through the per-layer Python loop -> moe_forward_gguf (moe.rs:990-1110, tests/test_gguf_native.py:47-57).  Here the same
                // arithmetic sits inside the captured step: router as above; the k routed experts through the block kernels of kr_gguf.hip on bf16(hidden)
                // (per-32 INT16 activations, exact integer dots, the AVX2 kernel's 8 lane chains + hsum per row: bit-equal to kr_moe_forward = the oracle's
                // moe_forward_gguf); the decode store's shared expert (INT4 / INT8 transposed, f32 hidden, decode.rs:3356-3378) as slot k of the same row
                // buffer; weighted sum in routing order, rsf and shared * sigmoid(gate) in the next fused add+RMSNorm, as for transposed experts.
                GgMoeArgs g{};
                g.B = 1; g.topk = k; g.n_slots = k; g.H = H; g.E = e->cfg.n_routed_experts; g.act_f32 = act; g.ids = (const int32_t*)s->r_ids.p;
                g.gate = EL.g_gate.view(); g.up = EL.g_up.view(); g.down = EL.g_down.view();
                const int sI = has_shared ? s->weights[L.sgu_wid]->rows / 2 : 0;
                g.I_max = EL.inter; g.gu_ld = 2 * (sI > EL.inter ? sI : EL.inter);       // row pitch of the shared gate|up scratch below
                g.gu = (float*)s->moe_gu.p; g.eo = (float*)s->moe_eo.p;
                prof_mark(s, PK_MOE_W13, st);
                kr_launch_gguf_moe(g, st);
                prof_mark(s, -1, st);
                if (has_shared) {
                    KrMoeArgs a{};
                    a.act_f32 = act; a.shared_decode = 1; if (routed && img_ok) { a.act_img = s->img_post.p; a.act_img_bf16 = s->img_post_bf16.p; }
                    a.B = 1; a.topk = 0; a.n_slots = 1; a.E = 0; a.H = H; a.I = sI; a.I_shared = sI;
                    a.sw13 = mv(s, L.sgu_wid); a.sw2 = mv(s, L.sd_wid); a.w13 = a.sw13; a.w2 = a.sw2;      // launch geometry follows w13 / w2 (bits, K); slot 0 is the shared slot
                    a.gu_ld = g.gu_ld; a.gu = g.gu + (size_t)k * g.gu_ld; a.eo = g.eo + (size_t)k * H;
                    a.rsf = s->rsf; a.swiglu_limit = e->cfg.swiglu_limit; a.alpha = e->cfg.activation_alpha;
                    a.act_mode = e->cfg.swiglu_limit > 0.0f ? KR_ACT_GPTOSS : KR_ACT_SILU_FUSED;
                    const bool fuse_gate = has_gate && mv(s, L.sg_wid).bits == a.sw13.bits;      // the sigmoid-gate row rides in the shared slot's gate|up launch
                    if (fuse_gate) { a.sgate = mv(s, L.sg_wid); a.gate_out = (float*)s->gate_val.p; }
                    PROF(PK_MOE_W13, kr_launch_moe_w13(a, st));
                    if (has_gate && !fuse_gate) PROF(PK_SHARED_GATE, kr_launch_matvec(mv(s, L.sg_wid), act, 1, (float*)s->gate_val.p, st));
                    PROF(PK_MOE_W2, kr_launch_moe_w2(a, st));
                }
                if (fast) {      // KR_DECODE_FAST: the combine as its own small launch into `hid`, so that the next layer's folded norm + projection launch applies
                    PROF(PK_MOE_COMBINE, kr_launch_moe_combine_decode(g.eo, g.ids, (const float*)s->r_w.p, k, has_shared ? 1 : 0, has_gate ? (const float*)s->gate_val.p : nullptr, s->rsf, hid, H, st));
                    src = from_hidden;
                    continue;
                }
                src = KrNormSrc{}; src.mode = 2; src.eo = g.eo; src.ids = g.ids; src.wts = (const float*)s->r_w.p; src.topk = k; src.has_shared = has_shared ? 1 : 0;
                src.gate_val = has_gate ? (const float*)s->gate_val.p : nullptr; src.rsf = s->rsf;
                continue;
            }
            KrMoeArgs a{};
            a.act = nullptr; a.act_f32 = act; a.shared_decode = 1;
            if (routed && img_ok) { a.act_img = s->img_post.p; a.act_img_bf16 = s->img_post_bf16.p; }
            a.ids = (const int32_t*)s->r_ids.p; a.wts = (const float*)s->r_w.p;
            a.B = 1; a.topk = k; a.n_slots = k + (has_shared ? 1 : 0); a.E = E; a.H = H; a.I = EL.inter;
            a.w13 = EL.w13.view(); a.w2 = EL.w2.view();
            if (has_shared) { a.sw13 = mv(s, L.sgu_wid); a.sw2 = mv(s, L.sd_wid); a.I_shared = s->weights[L.sgu_wid]->rows / 2; }
            const int imax = has_shared && a.I_shared > a.I ? a.I_shared : a.I;
            a.gu_ld = 2 * imax;
            a.gu = (float*)s->moe_gu.p; a.eo = (float*)s->moe_eo.p; a.out = nullptr; a.out_bf16 = 0;
            a.rsf = s->rsf; a.swiglu_limit = e->cfg.swiglu_limit; a.alpha = e->cfg.activation_alpha;
            a.act_mode = e->cfg.swiglu_limit > 0.0f ? KR_ACT_GPTOSS : KR_ACT_SILU_FUSED;
            // the shared expert's sigmoid-gate row rides in the shared slot's w13 launch when routed, shared and gate weights have one width
            const bool fuse_gate = has_gate && a.w13.bits == a.sw13.bits && mv(s, L.sg_wid).bits == a.w13.bits;
            if (fuse_gate) { a.sgate = mv(s, L.sg_wid); a.gate_out = (float*)s->gate_val.p; }
            if (ep_dec) {
                kr_ep_slice(e, &a.e_lo, &a.e_hi, &a.e_sub);
                KR_HIP(hipMemsetAsync(a.eo, 0, (size_t)k * H * 4, st));          // rows of experts other ranks own stay 0
            }
            PROF(PK_MOE_W13, kr_launch_moe_w13(a, st));
            if (has_gate && !fuse_gate) PROF(PK_SHARED_GATE, kr_launch_matvec(mv(s, L.sg_wid), act, 1, (float*)s->gate_val.p, st));
            if (!ep_dec && s->opt_w2_combine) {      // stage 2 + the routing-order combine in one launch: the MoE output lands in `hid`, the next norm launch reads it like any hidden vector
                prof_mark(s, PK_MOE_W2, st);
                const bool done = 0 == kr_launch_moe_w2c(a, has_gate ? (const float*)s->gate_val.p : nullptr, hid, st);
                prof_mark(s, -1, st);
                if (done) { src = from_hidden; continue; }
            }
            PROF(PK_MOE_W2, kr_launch_moe_w2(a, st));
            if (ep_dec) if (int rc = kr_ep_allreduce_on(e, a.eo, (size_t)k * H, st)) return rc;    // the shared expert's row (slot k) is replicated, not reduced
            // epilogue (weighted sum, rsf, shared * sigmoid(gate)) is folded into the next fused add+RMSNorm
            src = KrNormSrc{}; src.mode = 2; src.eo = a.eo; src.ids = a.ids; src.wts = a.wts; src.topk = k; src.has_shared = has_shared ? 1 : 0;
            src.gate_val = has_gate ? (const float*)s->gate_val.p : nullptr; src.rsf = s->rsf;
        } else if (L.mlp == MLP_DENSE) {
            // gate / up into [0,K) and [K,2K) of dense_gu (K = padded intermediate), then down with the fused silu*up + quant prologue
            const int K = s->weights[L.down_wid]->cols;
            const KrMatDev mats[2] = {mv(s, L.gate_wid), mv(s, L.up_wid)};
            float* ys[2] = {(float*)s->dense_gu.p, (float*)s->dense_gu.p + K};
            if (mats[0].bits == mats[1].bits) PROF(PK_MATVEC, kr_launch_multi_matvec(mats, ys, 2, hid, 1, st));
            else { PROF(PK_MATVEC, kr_launch_matvec(mats[0], hid, 1, ys[0], st)); PROF(PK_MATVEC, kr_launch_matvec(mats[1], hid, 1, ys[1], st)); }
            PROF(PK_MATVEC, kr_launch_matvec(mv(s, L.down_wid), s->dense_gu.p, 1, hid, st, KR_ACT_SILU_MUL));
        }
    }
    const bool lm_img = false;   // the vocabulary projection is wide enough that per-wave tiles beat the cooperative form (measured)
    bool lm_done = false;
    if (fast && s->opt_lm_fused && src.mode == 0 && !first) {
        // KR_DECODE_FAST: final norm + vocabulary projection as ONE launch of the projection kernel (the norm folded into every workgroup, tree sums) --
        // one launch and one boundary less per token
        const KrMatDev lm = mv(s, s->lm_head);
        if (lm.bits == 4 || lm.bits == 8) {
            KrFdmArgs fl{};
            fl.mode = 1; fl.hid_in = hid; fl.res_in = res_cur; fl.res_out = other(res_cur); fl.norm_w = (const float*)s->norms[s->final_norm]->p;
            fl.eps = s->eps; fl.bias_one = s->norm_bias_one;
            fl.mm.n = 1; fl.mm.m[0] = lm; fl.mm.y[0] = (float*)s->logits.p; fl.mm.tile_end[0] = (lm.N + 7) / 8;
            prof_mark(s, PK_LM_HEAD, st);
            lm_done = 0 == kr_launch_fdm(fl, st);
            prof_mark(s, -1, st);
        }
    }
    if (!lm_done) {
        PROF(PK_RMSNORM, kr_launch_fused_add_rmsnorm(src, hid, res_cur, res, (const float*)s->norms[s->final_norm]->p, H, s->eps, first ? 1 : 0, s->norm_bias_one, st,
                                                     lm_img ? s->img_in.p : nullptr));
        if (lm_img) PROF(PK_LM_HEAD, kr_launch_matvec(mv(s, s->lm_head), s->img_in.p, 2, (float*)s->logits.p, st));
        else PROF(PK_LM_HEAD, kr_launch_matvec(mv(s, s->lm_head), hid, 1, (float*)s->logits.p, st));
    }
    PROF(PK_ARGMAX, kr_launch_argmax((const float*)s->logits.p, s->vocab, (int*)s->tok.p, (float*)s->argmax_scratch.p, st));
    KR_HIP(hipGetLastError());
    return KR_OK;
}

// token == KR_TOKEN_FROM_DEVICE: the step's token is whatever the previous step's sampler left in s->tok (always a valid id)
#define KR_TOKEN_FROM_DEVICE (-2)
static int run_step(kr_decode_store* s, int token, int pos, hipStream_t st) {
    if (token != KR_TOKEN_FROM_DEVICE && (token < 0 || token >= s->vocab)) return kr_fail(KR_ERR_VALUE, "token id %d out of range (vocab %d)", token, s->vocab);
    if (pos < 0) return kr_fail(KR_ERR_VALUE, "position %d must be >= 0", pos);
    if (s->kv_max_seq > 0 && pos >= s->kv_max_seq) return kr_fail(KR_ERR_VALUE, "position %d >= kv_max_seq %d", pos, s->kv_max_seq);
    if (s->max_rope_seq > 0 && pos >= s->max_rope_seq) return kr_fail(KR_ERR_VALUE, "position %d >= rope table length %d", pos, s->max_rope_seq);
    for (const DLayer& L : s->layers)            // outside capture: the attention kernel's LDS window (scores + one stage of cache rows)
        if (L.hd > 0 && L.q_wid >= 0) {
            if (s->kv_max_seq > s->gqa_split_min && s->gqa_scores.ensure((size_t)L.nh * s->kv_max_seq * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc of the attention score scratch failed");
            if (s->attn_fast && s->kv_max_seq > s->gqa_split_min) {
                const size_t nch = std::max(((size_t)s->kv_max_seq + 255) / 256, kr_fd_flash_chunks(s->kv_max_seq));
                if (s->fd_o.ensure((size_t)L.nh * L.hd * nch * 4) || s->fd_ml.ensure((size_t)L.nh * nch * 8)) return kr_fail(KR_ERR_HIP, "hipMalloc of the split-KV partials failed");
                if ((L.hd == 64 || L.hd == 128 || L.hd == 256) && kr_fd_flash_prepare(L.hd, s->kv_fp8)) return kr_fail(KR_ERR_HIP, "LDS window of the flash-decode kernel refused");
            }
            const int pr = kr_gqa_attn_prepare(s->kv_max_seq, L.hd, s->kv_fp8);
            if (pr) return kr_fail(pr == -1 ? KR_ERR_VALUE : KR_ERR_HIP, "GQA decode attention: kv_max_seq %d with head_dim %d does not fit the 160 KiB LDS window", s->kv_max_seq, L.hd);
        }
    for (const DLayer& L : s->layers)
        if (L.attn == ATTN_MLA) {
            KrMlaArgs pa{}; pa.klr = L.klr; pa.rd = L.rd; pa.kv_fp8 = s->kv_fp8; kr_mla_attn_prepare(pa, s->kv_max_seq);
            if (s->kv_max_seq > s->mla_split_min && s->gqa_scores.ensure((size_t)L.nh * s->kv_max_seq * 4)) return kr_fail(KR_ERR_HIP, "hipMalloc of the attention score scratch failed");
            if (s->attn_fast && s->kv_max_seq > s->mla_split_min) {
                const size_t nch = std::max(((size_t)s->kv_max_seq + 255) / 256, kr_mla_flash_decode_chunks(s->kv_max_seq));
                if (s->fd_o.ensure((size_t)L.nh * L.klr * nch * 4) || s->fd_ml.ensure((size_t)L.nh * nch * 8)) return kr_fail(KR_ERR_HIP, "hipMalloc of the split-KV partials failed");
            }
        }
    if (token == KR_TOKEN_FROM_DEVICE) kr_launch_set_step_dev((KrStep*)s->step_dev.p, (const int*)s->tok.p, pos, st);
    else kr_launch_set_step((KrStep*)s->step_dev.p, token, pos, st);   // by-value kernel arguments: no host slot shared between queued steps
    s->last_stream = st;
    // expert-parallel decode enqueues every step by default (the loopback transport's hand-off is host-driven and cannot be captured); over RCCL the
    // all-reduce is a stream operation like any kernel: kr_decode_set_option("ep_graph", 1) captures it with the step.  The first two steps of a store run
    // eagerly even then -- RCCL sets up its channels / proxy connections on the first collectives, which must not happen inside a capture.
    const bool ep_active = kr_ep_decode_active(s->eng);
    if (s->eng->ep_generation != s->ep_generation_seen) {      // communicator created or destroyed since the last step: nothing captured or warmed up before survives it
        s->ep_generation_seen = s->eng->ep_generation; s->graph_ok = false; s->ep_eager_steps = 0;
    }
    if (ep_active && s->ep_eager_steps < 2) s->ep_eager_steps++;
    else if (s->use_graph && (!ep_active || (s->opt_ep_graph && kr_ep_is_rccl(s->eng)))) {
        if (!s->graph_ok) {
            if (s->graph_exec) { (void)hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
            hipGraph_t g = nullptr;
            KR_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue_step(s, st);
            hipError_t ce = hipStreamEndCapture(st, &g);
            if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
            if (ce != hipSuccess) return kr_fail(KR_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
            KR_HIP(hipGraphInstantiate(&s->graph_exec, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
            s->graph_ok = true;
        }
        KR_HIP(hipGraphLaunch(s->graph_exec, st));
        return KR_OK;
    }
    return enqueue_step(s, st);
}

// KV element type of the GQA caches: 0 = FP16 (reference CPU decode, decode.rs:4423-4478), 1 = FP8-E4M3 (reference GPU cache, kv_cache.py:38-135).
// Call before set_decode_state; existing caches are dropped.
extern "C" int kr_decode_set_kv_dtype(kr_decode_store* s, int kv_dtype) {
    if (int rc = chk_store(s)) return rc;
    if (kv_dtype != 0 && kv_dtype != 1) return kr_fail(KR_ERR_VALUE, "kv_dtype %d unknown (0 = FP16, 1 = FP8-E4M3)", kv_dtype);
    if (s->kv_fp8 != kv_dtype) for (auto& L : s->layers) if (L.attn == ATTN_GQA || L.attn == ATTN_MLA) { L.kv_k.release(); L.kv_v.release(); }
    s->kv_fp8 = kv_dtype; s->graph_ok = false;
    return KR_OK;
}

// attention numerics of LONG caches: KR_ATTN_EXACT (default) keeps the reference's sequential softmax sum and p.v order (bit-identical to
// decode.rs:4194-4281); KR_ATTN_FAST splits the cache over many workgroups and merges the partials (log-sum-exp) -- same products, another
// summation order; logits within ~1e-4 relative.  Router ids, every matvec and the short-cache kernels are unaffected.
extern "C" int kr_decode_set_attention_mode(kr_decode_store* s, int mode) {
    if (int rc = chk_store(s)) return rc;
    if (mode < 0 || mode > 7) return kr_fail(KR_ERR_VALUE, "numerics mode %d unknown (bit 0 = KR_ATTN_FAST, bit 1 = KR_GEMM_FAST, bit 2 = KR_DECODE_FAST)", mode);
    s->attn_fast = mode & 1; s->gemm_fast = (mode >> 1) & 1; s->decode_fast = (mode >> 2) & 1; s->graph_ok = false;
    return KR_OK;
}

// test / tuning hooks by name (they used to be environment variables read on the launch path): "gqa_stream" = long-cache GQA decode attention streams
// its score row from HBM even when it fits LDS; "pfm_timing" = kr_decode_prefill prints host-enqueue vs GPU-drain time to stderr
extern "C" int kr_decode_set_option(kr_decode_store* s, const char* name, int value) {
    if (int rc = chk_store(s)) return rc;
    if (!name) return kr_fail(KR_ERR_VALUE, "null option name");
    if (!strcmp(name, "gqa_stream")) { s->opt_gqa_stream = value != 0; s->graph_ok = false; return KR_OK; }
    if (!strcmp(name, "w2_combine")) { s->opt_w2_combine = value != 0; s->graph_ok = false; return KR_OK; }    // exact step: 0 = down projection per slot, combine inside the next norm launch (A/B and test hook)
    if (!strcmp(name, "la_heads")) { s->opt_la_heads = value != 0; s->graph_ok = false; return KR_OK; }        // exact step, linear-attention layers: 0 = in-projection + the one-launch conv / recurrence per KEY head (A/B and test hook)
    if (!strcmp(name, "lm_fused")) { s->opt_lm_fused = value != 0; s->graph_ok = false; return KR_OK; }        // KR_DECODE_FAST: 0 = final norm and vocabulary projection as two launches (A/B and test hook)
    if (!strcmp(name, "gqa_fused")) { s->opt_gqa_fused = value != 0; s->graph_ok = false; return KR_OK; }      // KR_DECODE_FAST, short caches: 0 = prep + attention as two launches (A/B and test hook)
    if (!strcmp(name, "gemm_ring")) { kr_pfr_set_enabled(value); return KR_OK; }                              // KR_GEMM_FAST: 0 = register-staged tolerance GEMMs only, 1 = ring form for big problems (default), 2 = for every shape it takes (process-wide A/B and test hook; same bits)
    if (!strcmp(name, "la_conv_fused")) { s->opt_la_conv_fused = value != 0; return KR_OK; }                // KR_ATTN_FAST prompt pass: 0 = the stand-alone conv launch in front of the delta-rule prep (A/B and test hook; same bits)
    if (!strcmp(name, "dense_fast")) { s->opt_dense_fast = value != 0; s->graph_ok = false; return KR_OK; }       // KR_DECODE_FAST: 0 = a dense MLP layer keeps the exact norm + gate | up launches (A/B and test hook)
    if (!strcmp(name, "norm_rows")) { s->opt_norm_rows = value != 0; return KR_OK; }                        // KR_GEMM_FAST prompt pass: 0 = the f16 row image of a norm's output by its own launch (A/B and test hook; same bits)
    if (!strcmp(name, "pfm_timing")) { s->opt_pfm_timing = value != 0; return KR_OK; }
    if (!strcmp(name, "generate_lookahead")) { s->opt_gen_lookahead = value != 0; return KR_OK; }
    if (!strcmp(name, "ep_graph")) { s->opt_ep_graph = value != 0; s->graph_ok = false; return KR_OK; }
    return kr_fail(KR_ERR_VALUE, "unknown option '%s'", name);
}

extern "C" int kr_decode_set_use_graph(kr_decode_store* s, int enable) {
    if (int rc = chk_store(s)) return rc;
    s->use_graph = enable != 0; return KR_OK;
}

static int decode_step_on(kr_decode_store* s, int token_id, int position, float* logits_out, hipStream_t st) {
    // all scratch the MoE block needs must exist before capture (no allocation inside a capture)
    {
        kr_engine* e = s->eng; size_t gu = 0, eo = 0;
        for (auto& L : s->layers) if (L.mlp == MLP_MOE) {
            if (s->own_eng || L.moe_layer >= (int)e->layers.size()) return kr_fail(KR_ERR_STATE, "set_moe_store was not called (MoE layer %d has no engine)", L.moe_layer);
            Layer& EL = e->layers[L.moe_layer];
            const int sI = L.sgu_wid >= 0 ? s->weights[L.sgu_wid]->rows / 2 : 0;
            const int imax = sI > EL.inter ? sI : EL.inter; const int ns = s->topk + (L.sgu_wid >= 0 ? 1 : 0);
            gu = maxz(gu, (size_t)ns * 2 * imax * 4); eo = maxz(eo, (size_t)ns * s->hidden * 4);
        }
        if (gu && (gu > s->moe_gu.bytes || eo > s->moe_eo.bytes || (size_t)e->r_ne * 4 > s->r_logits.bytes)) s->graph_ok = false;
        if (gu && (s->moe_gu.ensure(gu) || s->moe_eo.ensure(eo) || s->r_logits.ensure((size_t)e->r_ne * 4) || s->r_ids.ensure(256) || s->r_w.ensure(256)))
            return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    }
    if (int rc = run_step(s, token_id, position, st)) return rc;
    if (logits_out) {
        if (is_device_ptr(logits_out)) KR_HIP(hipMemcpyAsync(logits_out, s->logits.p, (size_t)s->vocab * 4, hipMemcpyDeviceToDevice, st));
        else { KR_HIP(hipMemcpyAsync(logits_out, s->logits.p, (size_t)s->vocab * 4, hipMemcpyDeviceToHost, st)); KR_HIP(hipStreamSynchronize(st)); }
    }
    return KR_OK;
}

extern "C" int kr_decode_step(kr_decode_store* s, int token_id, int position, float* logits_out, void* stream) {
    if (int rc = need_cfg(s)) return rc;
    if (token_id < 0) return kr_fail(KR_ERR_VALUE, "token id %d out of range (vocab %d)", token_id, s->vocab);
    KR_HIP(hipSetDevice(s->eng->device));
    return decode_step_on(s, token_id, position, logits_out, kr_pick_stream(s->eng, stream));
}

// ---- sampling state (generate_batch, decode.rs:3525-3600): seen-token bitmap, xorshift64 state, sort scratch -- all on the device
static int sampler_prepare(kr_decode_store* s) {
    const size_t vocab = (size_t)s->vocab;
    if (s->smp_temp_bytes == 0) s->smp_temp_bytes = kr_sampler_temp_bytes(s->vocab);
    if (s->smp_seen.ensure(((vocab + 31) / 32) * 4 + 64) || s->smp_keys.ensure(2 * vocab * 8) || s->smp_temp.ensure(s->smp_temp_bytes + 256) ||
        s->smp_probs.ensure(vocab * 4) || s->smp_rng.ensure(64))
        return kr_fail(KR_ERR_HIP, "hipMalloc of sampler scratch failed");
    return KR_OK;
}

// sample_from_logits (decode.rs:3718) applied to the logits of the last step; temperature 0 = greedy (first maximum).
// rng_seed != 0 re-seeds the xorshift64 state (the reference seeds from the wall clock; 0xDEADBEEF if that is 0).
// test aid (no reference counterpart): the token ids kr_decode_sample would draw from, in its order -- the top_k largest of `logits` (host, f32), value descending, equal values
// by ascending id (decode.rs:3740-3760 sorts by value only; the oracle fixes the same tie rule).  top_k <= 0 or >= vocab: the whole vocabulary.
extern "C" int kr_sample_order(const float* logits_host, int vocab, int top_k, int32_t* ids_out_host) {
    if (!logits_host || !ids_out_host || vocab <= 0) return kr_fail(KR_ERR_VALUE, "kr_sample_order: null pointer or empty vocabulary");
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count == 0) return kr_fail(KR_ERR_HIP, "no HIP device");
    const int k = (top_k > 0 && top_k < vocab) ? top_k : vocab;
    DevBuf lg, keys, tmp;
    const size_t tb = kr_sampler_temp_bytes(vocab);
    if (lg.ensure((size_t)vocab * 4) || keys.ensure((size_t)vocab * 16) || tmp.ensure(tb ? tb : 16)) return kr_fail(KR_ERR_HIP, "hipMalloc failed");
    KR_HIP(hipMemcpy(lg.p, logits_host, (size_t)vocab * 4, hipMemcpyHostToDevice));
    uint64_t* kin = (uint64_t*)keys.p;
    if (kr_launch_sample_order((float*)lg.p, vocab, top_k, nullptr, kin, kin + vocab, tmp.p, tb, nullptr)) return kr_fail(KR_ERR_HIP, "sampler order launch failed");
    KR_HIP(hipDeviceSynchronize());
    std::vector<uint64_t> h((size_t)k);
    KR_HIP(hipMemcpy(h.data(), kin + vocab, (size_t)k * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < k; i++) ids_out_host[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)h[i]);
    return KR_OK;
}

extern "C" int kr_decode_sample(kr_decode_store* s, float temperature, int top_k, float top_p, float presence_penalty, uint64_t rng_seed,
                                int reset_seen, int* token_out, void* stream) {
    if (int rc = need_cfg(s)) return rc;
    if (temperature < 0.0f) return kr_fail(KR_ERR_VALUE, "temperature must be >= 0");
    KR_HIP(hipSetDevice(s->eng->device));
    hipStream_t st = kr_pick_stream(s->eng, stream);
    if (int rc = sampler_prepare(s)) return rc;
    if (reset_seen) KR_HIP(hipMemsetAsync(s->smp_seen.p, 0, s->smp_seen.bytes, st));
    if (rng_seed) KR_HIP(hipMemcpyAsync(s->smp_rng.p, &rng_seed, 8, hipMemcpyHostToDevice, st));
    if (temperature == 0.0f) {
        if (presence_penalty != 0.0f) return kr_fail(KR_ERR_VALUE, "presence_penalty with greedy sampling goes through kr_decode_generate");
        kr_launch_argmax((const float*)s->logits.p, s->vocab, (int*)s->tok.p, (float*)s->argmax_scratch.p, st);
    } else {
        uint64_t* keys = (uint64_t*)s->smp_keys.p;
        if (kr_launch_sample((float*)s->logits.p, s->vocab, temperature, top_k, top_p, presence_penalty, (uint32_t*)s->smp_seen.p, keys, keys + s->vocab,
                             s->smp_temp.p, s->smp_temp_bytes, (float*)s->smp_probs.p, (uint64_t*)s->smp_rng.p, (int*)s->tok.p, st))
            return kr_fail(KR_ERR_HIP, "device radix sort failed");
    }
    if (token_out) { KR_HIP(hipMemcpyAsync(token_out, s->tok.p, 4, hipMemcpyDeviceToHost, st)); KR_HIP(hipStreamSynchronize(st)); }
    return KR_OK;
}

// generate_batch (decode.rs:3525): decode_step -> presence penalty on seen tokens -> sample_from_logits; the sampled token is appended
// BEFORE the stop test (decode.rs:3587-3591), so a stop id is the last element of the result.
// generate_batch (decode.rs:3525) and generate_stream (decode.rs:3611) share one loop.  on_token != nullptr is the streaming form: the cancel
// flag is polled before every step (a cancelled run reports the last token with reason 3), and the callback gets (token, finish_reason:
// 0 none, 1 stop, 2 length, 3 cancelled); returning 0 ends the run.  Both forms record the wall time of the loop (last_decode_elapsed_s).
static int generate_core(kr_decode_store* s, int first_token, int start_pos, int max_tokens, float temperature, int top_k, float top_p,
                         const int* stop_ids, int n_stop, float presence_penalty, uint64_t rng_seed, int* tokens_out, int* n_out, void* stream,
                         kr_token_cb on_token, void* user) {
    if (int rc = need_cfg(s)) return rc;
    if (!n_out || (!tokens_out && !on_token)) return kr_fail(KR_ERR_VALUE, "null output pointer");
    if (temperature < 0.0f) return kr_fail(KR_ERR_VALUE, "temperature must be >= 0");
    KR_HIP(hipSetDevice(s->eng->device));
    hipStream_t st = kr_pick_stream(s->eng, stream);
    const bool sampled = temperature != 0.0f, penal = presence_penalty != 0.0f;
    if (sampled || penal) {
        if (int rc = sampler_prepare(s)) return rc;
        KR_HIP(hipMemsetAsync(s->smp_seen.p, 0, s->smp_seen.bytes, st));
        if (first_token >= 0 && first_token < s->vocab) kr_launch_mark_seen((uint32_t*)s->smp_seen.p, nullptr, first_token, st);
        if (rng_seed == 0) { rng_seed = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count(); if (rng_seed == 0) rng_seed = 0xDEADBEEFull; }
        KR_HIP(hipMemcpyAsync(s->smp_rng.p, &rng_seed, 8, hipMemcpyHostToDevice, st));
    }
    int tok = first_token, n = 0;
    const auto t_start = std::chrono::steady_clock::now();
    auto stamp = [&]() { kr_standalone_set_elapsed(s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count()); };
    // one decode step + the sampler; token < 0: the step consumes the token the previous sampler left on the device
    auto step_and_sample = [&](int token, int pos) -> int {
        if (int rc = decode_step_on(s, token, pos, nullptr, st)) return rc;     // graph replay ends with the greedy argmax into s->tok
        if (sampled) {
            uint64_t* keys = (uint64_t*)s->smp_keys.p;
            if (kr_launch_sample((float*)s->logits.p, s->vocab, temperature, top_k, top_p, presence_penalty, (uint32_t*)s->smp_seen.p, keys, keys + s->vocab,
                                 s->smp_temp.p, s->smp_temp_bytes, (float*)s->smp_probs.p, (uint64_t*)s->smp_rng.p, (int*)s->tok.p, st))
                return kr_fail(KR_ERR_HIP, "device radix sort failed");
        } else if (penal) {   // greedy with a presence penalty: penalise, then first maximum
            kr_launch_penalty((float*)s->logits.p, s->vocab, presence_penalty, (const uint32_t*)s->smp_seen.p, st);
            kr_launch_argmax((const float*)s->logits.p, s->vocab, (int*)s->tok.p, (float*)s->argmax_scratch.p, st);
            kr_launch_mark_seen((uint32_t*)s->smp_seen.p, (const int*)s->tok.p, 0, st);
        }
        return KR_OK;
    };
    if (s->opt_gen_lookahead && !on_token && max_tokens > 0) {
        // LOOK-AHEAD form (kr_decode_set_option "generate_lookahead", off by default).  The reference's loop reads every sampled token on the host before it
        // queues the next step (decode.rs:3560-3591), which idles the GPU for one host round trip per token.  Here the sampler's output stays on the
        // device and feeds step i + 1 directly (kr_set_step_dev_kernel); the host reads token i from a pinned ring WHILE step i + 1 runs.  Same tokens
        // as the plain loop.  One stated difference: when token i turns out to be a stop id, step i + 1 has already been queued -- the KV / recurrent
        // state then includes the stop token (the plain loop leaves it out), and the sampler's RNG state / seen-token bitmap / device token slot have advanced one
        // sample past where the plain loop leaves them.  Callers that continue decoding from the state after a stop keep the default.  A step whose position is past
        // the cache or the rope table is never queued ahead: the loop reads the token first, as the plain loop does (which returns its tokens when that token stops).
        if (s->gen_ring_n < max_tokens) {
            if (s->gen_ring) (void)hipHostFree(s->gen_ring);
            s->gen_ring = nullptr; s->gen_ring_n = 0;
            KR_HIP(hipHostMalloc((void**)&s->gen_ring, sizeof(int) * (size_t)max_tokens, hipHostMallocDefault));
            s->gen_ring_n = max_tokens;
        }
        for (auto& ev : s->gen_ev) if (!ev) KR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        if (int rc = step_and_sample(tok, start_pos)) { stamp(); return rc; }
        for (int i = 0; i < max_tokens; i++) {
            KR_HIP(hipMemcpyAsync(&s->gen_ring[i], s->tok.p, 4, hipMemcpyDeviceToHost, st));
            KR_HIP(hipEventRecord(s->gen_ev[i & 1], st));
            const int npos = start_pos + i + 1;
            const bool in_range = !((s->kv_max_seq > 0 && npos >= s->kv_max_seq) || (s->max_rope_seq > 0 && npos >= s->max_rope_seq));
            const bool ahead = i + 1 < max_tokens && in_range;
            if (ahead) if (int rc = step_and_sample(KR_TOKEN_FROM_DEVICE, npos)) { stamp(); return rc; }
            KR_HIP(hipEventSynchronize(s->gen_ev[i & 1]));
            const int next = s->gen_ring[i];
            tokens_out[n++] = next;
            bool stop = false;
            for (int j = 0; j < n_stop; j++) stop |= (stop_ids[j] == next);
            if (stop) break;
            if (i + 1 < max_tokens && !ahead) if (int rc = step_and_sample(next, npos)) { stamp(); return rc; }      // out of range: the plain loop's error, after the plain loop's check of the stop ids
        }
        KR_HIP(hipStreamSynchronize(st));
        stamp();
        *n_out = n;
        return KR_OK;
    }
    for (int i = 0; i < max_tokens; i++) {
        if (on_token && kr_standalone_cancelled(s)) { on_token(tok, 3, user); break; }
        if (int rc = step_and_sample(tok, start_pos + i)) { stamp(); return rc; }
        int next = 0;
        KR_HIP(hipMemcpyAsync(&next, s->tok.p, 4, hipMemcpyDeviceToHost, st));
        KR_HIP(hipStreamSynchronize(st));
        if (tokens_out) tokens_out[n] = next;
        n++; tok = next;
        bool stop = false;
        for (int j = 0; j < n_stop; j++) stop |= (stop_ids[j] == next);
        if (on_token) {
            const int reason = stop ? 1 : (n >= max_tokens ? 2 : 0);
            const int cont = on_token(next, reason, user);
            if (reason || !cont) break;
        } else if (stop) break;
    }
    stamp();
    *n_out = n;
    return KR_OK;
}

extern "C" int kr_decode_generate(kr_decode_store* s, int first_token, int start_pos, int max_tokens, float temperature, int top_k, float top_p,
                                  const int* stop_ids, int n_stop, float presence_penalty, uint64_t rng_seed, int* tokens_out, int* n_out, void* stream) {
    if (!tokens_out) return kr_fail(KR_ERR_VALUE, "null output pointer");
    return generate_core(s, first_token, start_pos, max_tokens, temperature, top_k, top_p, stop_ids, n_stop, presence_penalty, rng_seed, tokens_out, n_out, stream, nullptr, nullptr);
}

extern "C" int kr_decode_generate_stream(kr_decode_store* s, int first_token, int start_pos, int max_tokens, float temperature, int top_k, float top_p,
                                         const int* stop_ids, int n_stop, float presence_penalty, uint64_t rng_seed, kr_token_cb on_token, void* user, int* n_out,
                                         void* stream) {
    if (!on_token) return kr_fail(KR_ERR_VALUE, "generate_stream needs a token callback");
    return generate_core(s, first_token, start_pos, max_tokens, temperature, top_k, top_p, stop_ids, n_stop, presence_penalty, rng_seed, nullptr, n_out, stream, on_token, user);
}

extern "C" int kr_decode_generate_greedy(kr_decode_store* s, int first_token, int start_pos, int max_tokens, const int* stop_ids, int n_stop,
                                         int* tokens_out, int* n_out, void* stream) {
    return kr_decode_generate(s, first_token, start_pos, max_tokens, 0.0f, 0, 1.0f, stop_ids, n_stop, 0.0f, 0, tokens_out, n_out, stream);
}

extern "C" int kr_decode_last_token(kr_decode_store* s, int* tok) {
    if (int rc = need_cfg(s)) return rc;
    KR_HIP(hipSetDevice(s->eng->device));
    // the engine / caller streams are non-blocking: a legacy-stream hipMemcpy would not wait for the step that writes s->tok
    hipStream_t st = s->last_stream ? s->last_stream : s->eng->stream;
    KR_HIP(hipMemcpyAsync(tok, s->tok.p, 4, hipMemcpyDeviceToHost, st));
    KR_HIP(hipStreamSynchronize(st));
    return KR_OK;
}

// debug/parity: copy a named scratch buffer (0 hidden, 1 residual) to the host
extern "C" int kr_decode_read_buffer(kr_decode_store* s, int which, float* out, int n) {
    if (int rc = need_cfg(s)) return rc;
    KR_HIP(hipSetDevice(s->eng->device));
    KR_HIP(hipStreamSynchronize(s->eng->stream));
    // 0 hidden, 1 residual, 2 router ids (int32 bits), 3 router weights, 4 router logits of the LAST MoE layer of the last step, 5 second residual buffer
    DevBuf* b = which == 0 ? &s->hid : which == 1 ? &s->res : which == 2 ? &s->r_ids : which == 3 ? &s->r_w : which == 4 ? &s->r_logits : which == 5 ? &s->res2 : nullptr;
    if (!b || !b->p || (size_t)n * 4 > b->bytes) return kr_fail(KR_ERR_VALUE, "read_buffer: buffer %d unknown or shorter than %d words", which, n);
    if (s->last_stream) KR_HIP(hipStreamSynchronize(s->last_stream));
    KR_HIP(hipMemcpy(out, b->p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return KR_OK;
}

extern "C" size_t kr_decode_device_bytes(const kr_decode_store* s) { return s ? s->weight_bytes + s->embedding.bytes : 0; }

// One un-graphed decode step with HIP events around every launch (on the launch stream); returns per-kind totals for that step.
extern "C" int kr_decode_profile_step(kr_decode_store* s, int token_id, int position, double* ms_by_kind, long* launches_by_kind, int n_kinds) {
    if (int rc = need_cfg(s)) return rc;
    if (n_kinds < PK_COUNT) return kr_fail(KR_ERR_VALUE, "need %d kinds", PK_COUNT);
    KR_HIP(hipSetDevice(s->eng->device));
    hipStream_t st = s->eng->stream;
    const bool saved = s->use_graph;
    s->use_graph = false; s->prof = true; s->ev_used = 0; s->ev_kind.clear();
    const int rc = kr_decode_step(s, token_id, position, nullptr, st);
    s->use_graph = saved; s->prof = false;
    if (rc) return rc;
    KR_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < n_kinds; i++) { ms_by_kind[i] = 0; launches_by_kind[i] = 0; }
    for (size_t i = 0; i < s->ev_kind.size(); i++) {
        float ms = 0; KR_HIP(hipEventElapsedTime(&ms, s->ev_pool[2 * i], s->ev_pool[2 * i + 1]));
        ms_by_kind[s->ev_kind[i]] += ms; launches_by_kind[s->ev_kind[i]]++;
    }
    return KR_OK;
}
