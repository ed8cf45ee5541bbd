// kr_decode_internal.h -- the decode store's state, shared by kr_decode.cpp (single-token graph) and kr_decode_prefill.cpp (batched prompt)
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <vector>

#include "kr_decode_ops.h"
#include "kr_engine_internal.h"

struct DWeight { MatSet ms; int rows = 0, cols = 0; };

enum { ATTN_NONE = 0, ATTN_LA = 1, ATTN_GQA = 2, ATTN_MLA = 3 };
enum { MLP_NONE = 0, MLP_MOE = 1, MLP_DENSE = 2 };

struct DLayer {
    int input_norm = -1, post_norm = -1;
    int attn = ATTN_NONE, mlp = MLP_NONE;
    // LA
    int qkvz_wid = -1, ba_wid = -1, out_wid = -1, nk = 0, nv = 0, dk = 0, dv = 0, kd = 4; float la_scale = 1.0f;
    DevBuf conv_w, a_log, dt_bias, la_norm_w, conv_state, recur_state;
    // GQA
    int q_wid = -1, k_wid = -1, v_wid = -1, o_wid = -1, gated = 0, nh = 0, nkv = 0, hd = 0; float sm_scale = 1.0f;
    DevBuf q_norm, k_norm, kv_k, kv_v; int q_norm_len = 0, k_norm_len = 0;
    // MLA (kv_k = compressed-KV cache [max_seq, klr], kv_v = k_pe cache [max_seq, rd], both FP16)
    int kva_wid = -1, mq_wid = -1, mqa_wid = -1, mqb_wid = -1, klr = 0, nd = 0, rd = 0, vhd = 0, q_a_norm_len = 0;
    DevBuf w_kc, w_vc, kv_a_norm, q_a_norm, mla_cos, mla_sin; int mla_rope_seq = 0;
    // MLP
    int moe_layer = -1, sgu_wid = -1, sd_wid = -1, sg_wid = -1;
    int gate_wid = -1, up_wid = -1, down_wid = -1;
};

struct kr_standalone_state;   // kr_decode_standalone.cpp: staging buffers, stand-alone router gates, cancel flag, elapsed time
struct kr_decode_store {
    kr_standalone_state* standalone = nullptr;
    kr_engine* eng = nullptr; int device = 0; bool own_eng = false;   // own_eng: a bare engine made by kr_decode_create(NULL, ...), replaced by kr_decode_set_moe_store
    int group_size = 128; bool norm_bias_one = false;
    std::vector<std::unique_ptr<DWeight>> weights;
    std::vector<std::unique_ptr<DevBuf>> norms; std::vector<int> norm_len;
    bool configured = false;
    int hidden = 0, n_layers = 0, vocab = 0, topk = 0, scoring = 1, norm_topk = 1, final_norm = -1, lm_head = -1;
    float eps = 1e-6f, rsf = 1.0f;
    DevBuf embedding;
    std::vector<DLayer> layers;
    DevBuf rope_cos, rope_sin; int rope_half = 0, max_rope_seq = 0;
    int kv_max_seq = 0;
    // scratch
    DevBuf hid, res, proj_a, proj_b, qbuf, kbuf, vbuf, zbuf, gbuf, betabuf, gatebuf, latbuf, recur_out, attn_out, logits, gate_val, tok;
    DevBuf dense_gu;  // [gate(K) | up(K)] of the dense MLP; only [0,inter) of each half is ever written, the padding stays 0
    DevBuf hid2, res2, r_counter, argmax_scratch;
    int kv_fp8 = 0;            // GQA KV element type: 0 FP16 (reference CPU decode), 1 FP8-E4M3 (reference GPU cache dtype)
    DevBuf img_in, img_post, img_post_bf16, img_attn; bool use_images = true;   // pre-built INT16 activation images (input norm, post-attention norm f32 / bf16, attention output)
    int opt_gqa_stream = 0, opt_pfm_timing = 0, opt_norm_rows = 1, opt_la_conv_fused = 1, opt_gqa_fused = 1, opt_lm_fused = 1, opt_la_heads = 1, opt_w2_combine = 1, opt_dense_fast = 1;   // kr_decode_set_option: test / tuning hooks (no environment lookups on launch paths)
    int opt_gen_lookahead = 0;                    // kr_decode_set_option("generate_lookahead"): generate_batch feeds the sampled token back ON THE DEVICE and queues step i + 1 before the host has read token i
    int opt_ep_graph = 0;                         // kr_decode_set_option("ep_graph"): expert-parallel decode over RCCL replays a captured graph (the all-reduce is captured with the kernels)
    uint64_t ep_generation_seen = 0;               // kr_engine::ep_generation at the last step: a new / destroyed communicator invalidates the graph and restarts the warm-up
    int ep_eager_steps = 0;                        // expert-parallel decode: steps enqueued eagerly so far (RCCL warms up outside any capture)
    int* gen_ring = nullptr; int gen_ring_n = 0; hipEvent_t gen_ev[2] = {nullptr, nullptr};   // pinned token ring + events of the look-ahead loop
    int decode_fast = 0; DevBuf f_qk;         // KR_DECODE_FAST: decode steps on the tolerance-mode kernels (kr_decode_fast.hip); f_qk = conv outputs [nk][q(dk) | k(dk)]
    int gemm_fast = 0;                        // KR_GEMM_FAST: prompt-pass GEMMs in the tolerance form (kr_prefill_h.hip)
    int attn_fast = 0; DevBuf fd_o, fd_ml;   // KR_ATTN_FAST: split-KV softmax + p.v with a log-sum-exp merge for long caches (tolerance mode)
    DevBuf gqa_scores; int gqa_split_min = 1024, mla_split_min = 512;   // caches longer than this split decode attention into a scores launch + softmax / p.v launch
    bool fuse_la = true;       // conv + recurrence + gated norm of a linear-attention layer in one launch (kr_la_step_kernel)
    bool fuse_router = true;   // hid2/res2: outputs of the fused norm+router launch (its inputs stay readable for every workgroup)
    DevBuf smp_seen, smp_keys, smp_temp, smp_probs, smp_rng; size_t smp_temp_bytes = 0;   // sampler: seen bitmap, sort keys / scratch, probabilities, xorshift64 state
    DevBuf pf_scores;          // kr_decode_prefill: attention scores [chunk*nh rows][context] f32
    DevBuf pf_vlogits, pf_nll; // kr_decode_prefill_nll: [chunk, vocab] logits per arena; per-position negative log-likelihoods
    DevBuf pf_tokens; int pf_chunk = 0; int pf_depth = 0; std::vector<hipStream_t> pf_side; std::vector<hipEvent_t> pf_events;   // prompt pass: token ids, chunk size, second stream
    DevBuf pf_scratch;         // kr_decode_prefill: one arena for the chunk buffers
    DevBuf moe_gu, moe_eo, r_logits, r_ids, r_w;  // store-owned so a captured graph never sees them reallocated
    DevBuf step_dev; hipStream_t last_stream = nullptr;   // stream of the most recent step / prompt pass (kr_decode_last_token waits on it)
    size_t weight_bytes = 0;
    // captured graph of one decode step
    hipGraphExec_t graph_exec = nullptr; bool graph_ok = false; bool use_graph = true;
    // profiling pass (kr_decode_profile_step): HIP events around every launch, accumulated per kernel kind
    bool prof = false; std::vector<hipEvent_t> ev_pool; size_t ev_used = 0; std::vector<int> ev_kind;
};

enum { PK_EMBED = 0, PK_RMSNORM, PK_MATVEC, PK_LA_CONV, PK_LA_RECUR, PK_GATED_NORM, PK_GQA, PK_ROUTE_LOGITS, PK_ROUTE_SELECT, PK_MOE_W13,
       PK_MOE_W2, PK_MOE_COMBINE, PK_LM_HEAD, PK_ARGMAX, PK_SHARED_GATE, PK_OUT_PROJ /* KR_DECODE_FAST: the out / o projection from the attention image (its own kernel instantiation) */, PK_COUNT };


static inline KrMatDev mv(kr_decode_store* s, int wid) { return s->weights[wid]->ms.view(); }
int kr_ensure_wsum(kr_engine* e, MatSet& ms, hipStream_t st);
int kr_moe_prefill_prepare(kr_engine* e, int layer, int fast, int routed_only, hipStream_t st);   // kr_engine.cpp: the lazily derived data of a native-GGUF layer, built on `st` now
void kr_standalone_release(kr_decode_store* s);
int kr_standalone_cancelled(kr_decode_store* s);
void kr_standalone_set_elapsed(kr_decode_store* s, double sec);   // kr_engine.cpp: per (group, column) nibble sums for the int8-MFMA GEMM
