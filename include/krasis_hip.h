/*
 * krasis_hip.h -- C ABI of libkrasis_hip.so: the MI355X (gfx950) quantized-MoE hot path.
 *
 * This is the drop-in boundary.  Each entry point names the reference
 * interface it replaces (paths relative to the brontoguana/krasis checkout).  The reference's
 * own FFI for this path is PyO3 (Rust <-> Python, raw `usize` pointers); a maintainer binds
 * these symbols from Rust with a plain `extern "C"` block (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; kr_last_error() gives the text.
 *     The Python mirror maps codes to the reference's exception classes
 *     (KR_ERR_STATE -> RuntimeError, KR_ERR_VALUE -> ValueError, KR_ERR_IO -> IOError).
 *   - "dev" pointers are HBM addresses on the engine's device, "host" pointers are CPU addresses.
 *     Entry points that take activations accept either: host pointers are staged through the
 *     engine's pinned bounce buffers (the reference passes host addresses, moe.rs:2843-2853).
 *   - `stream` is a hipStream_t cast to void*; NULL = the engine's own (non-blocking) stream, (void*)1 = the legacy default
 *     stream (what a framework reports as stream 0).  Device-pointer outputs are ready when that stream reaches the call.
 *   - no torch types, no Python callbacks; every entry point may be called without the GIL.
 */
#ifndef KRASIS_HIP_H
#define KRASIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KR_OK 0
#define KR_ERR_STATE 1 /* PyRuntimeError in the reference ("Model not loaded -- call load() first", moe.rs:1783) */
#define KR_ERR_VALUE 2 /* PyValueError (shape/size errors, moe.rs:1790-1803) */
#define KR_ERR_IO 3    /* PyIOError */
#define KR_ERR_HIP 4   /* HIP runtime failure */

/* expert weight formats (SURVEY §8 path matrix) */
#define KR_FMT_INT4_G128 4   /* quantize_int4, marlin.rs:145; CPU transposed layout weights/mod.rs:329 */
#define KR_FMT_INT8_G128 8   /* quantize_int8, marlin.rs:65 */
/* raw GGUF block types use the GGML ids (gguf.rs:15-31): Q4_0=2 Q5_0=6 Q8_0=8 Q4_K=12 Q6_K=14 */
#define KR_GGML_Q4_0 2
#define KR_GGML_Q5_0 6
#define KR_GGML_Q8_0 8
#define KR_GGML_Q4_K 12
#define KR_GGML_Q6_K 14

#define KR_MAX_TOPK 32 /* moe.rs:2899 */

/* output dtype of kr_moe_forward */
#define KR_OUT_F32 0  /* KrasisEngine.moe_forward returns f32 bytes (moe.rs:1863) */
#define KR_OUT_BF16 1 /* forward_moe_direct / sync_forward return bf16 RNE (moe.rs:2947) */

typedef struct kr_engine kr_engine;

const char* kr_last_error(void);
int kr_version(void);
/* measurement aid (no reference counterpart): device allocations the library has made so far, process-wide.  bench.py reads it on both sides of
   every timed region and refuses a measurement that allocated (an allocation synchronises the device). */
long kr_alloc_count_total(void);

/* ---- engine lifetime: KrasisEngine::new / load (moe.rs:1435-1760) ---- */
typedef struct {
    int hidden_size;            /* ModelConfig, weights/mod.rs:51-70 */
    int moe_intermediate_size;
    int n_routed_experts;
    int num_experts_per_tok;
    int num_moe_layers;
    int n_shared_experts;       /* shared intermediate = n_shared * moe_intermediate */
    int group_size;             /* 128 (marlin.rs:12) */
    float routed_scaling_factor;
    float swiglu_limit;         /* 0 = SiLU; 7.0 = GPT-OSS (weights/mod.rs:159-166) */
    float activation_alpha;
} kr_model_config;

int kr_engine_create(int device_ordinal, const kr_model_config* cfg, kr_engine** out);
void kr_engine_destroy(kr_engine* e);
int kr_engine_get_config(const kr_engine* e, kr_model_config* out);
size_t kr_engine_device_bytes(const kr_engine* e);

/* ---- weight upload.  Host arrays are in the REFERENCE layouts; the library re-tiles them into its
 *      own HBM layout.  expert = -1 uploads the layer's shared expert. ---- */
/* UnifiedExpertWeights CPU transposed format (weights/mod.rs:287-322):
 *   bits 4: w13 u32 [H/8, 2I], w2 u32 [I/8, H];  bits 8: w13 i8 [H, 2I], w2 i8 [I, H]
 *   scales bf16 [H/gs, 2I] and [I/gs, H].  `inter` is this expert's intermediate size. */
int kr_upload_expert_unified(kr_engine* e, int layer, int expert, int inter,
                             const void* w13, const uint16_t* w13_scales, int w13_bits,
                             const void* w2, const uint16_t* w2_scales, int w2_bits);
/* GgufExpertWeights (weights/mod.rs:252-266): raw blocks, gate/up [inter, hidden], down [hidden, inter] */
int kr_upload_expert_gguf(kr_engine* e, int layer, int expert, int inter,
                          const uint8_t* gate, const uint8_t* up, int gate_up_type,
                          const uint8_t* down, int down_type);
/* load_from_hf (weights/mod.rs:1181 -> load_and_quantize_expert -> weights/marlin.rs:65,145): BF16 tensors of one expert in the HF
 * checkpoint layout -- gate, up [inter, hidden], down [hidden, inter], row-major, host or device pointers -- quantized on the GPU with
 * the reference's rule (scale = bf16(amax/7 | amax/127), q = clamp(round(v * (1/scale)))) and stored in the resident layout.
 * expert = -1: the shared expert (inter = n_shared_experts * moe_intermediate_size). */
int kr_upload_expert_bf16(kr_engine* e, int layer, int expert, int inter, const uint16_t* gate, const uint16_t* up, const uint16_t* down,
                          int w13_bits, int w2_bits);
/* Fill a layer's routed experts (and shared, if configured) with device-generated pseudo-random INT4/INT8
 * words and bf16 scales in [0.005,0.05] -- the distribution of bench_decode_synthetic (decode.rs:4379-4392),
 * generated by a counter hash on the GPU instead of one serial xorshift stream. */
int kr_fill_layer_synthetic(kr_engine* e, int layer, int bits, uint64_t seed);
/* the native-GGUF twin: every routed expert of the layer as Q4_K (12) / Q8_0 (8) blocks with the block distribution SURVEY 8d defines
 * (raw quant / scale bytes, d = f16((0.005 + u * 0.045) / 63), dmin = f16(8 d); Q8_0: d = f16((0.005 + u * 0.045) / 127)) */
int kr_fill_layer_synthetic_gguf(kr_engine* e, int layer, int gate_up_type, int down_type, uint64_t seed);
/* Read one expert back in the reference layout (inverse re-tiling); used by parity tests at full size. */
int kr_download_expert_unified(kr_engine* e, int layer, int expert, void* w13, uint16_t* w13_scales,
                               void* w2, uint16_t* w2_scales);

/* ---- the reference's Marlin GPU layout: marlin_repack / marlin_repack_int8 (src/weights/marlin.rs:256-491, :587-758), undone
 *      in the reference by inverse_marlin_repack / inverse_scale_permute (python/krasis/triton_moe.py:71-170).  Host-side conversions only -- no
 *      CDNA kernel consumes this layout.  row-major: packed [N, K/8] u32 (INT4, nibble = q + 8, LSB = lowest k) or [N, K] i8 (INT8), scales bf16
 *      [N, K/gs];  Marlin: packed [K/16, 2N] (INT4) / [K/16, 4N] (INT8) u32, scales bf16 [K/gs, N] permuted.  N % 64 == 0, K % 16 == 0. ---- */
int kr_marlin_repack(const void* rowmajor, const uint16_t* scales, int N, int K, int group_size, int bits, uint32_t* out_packed, uint16_t* out_scales);
int kr_marlin_unpack(const uint32_t* marlin_packed, const uint16_t* marlin_scales, int N, int K, int group_size, int bits, void* out_rowmajor, uint16_t* out_scales);
/* one expert in the Marlin GPU format of UnifiedExpertWeights::from_expert_weights_marlin_* (weights/mod.rs:506-640): w13 = repack(gate rows || up
 * rows), w2 = repack(down) with N padded per marlin_w2_padded_n (weights/mod.rs:942-949); what the reference's disk cache holds (:856-934) and
 * what KrasisEngine.get_expert_w13_packed / _scales / get_expert_w2_packed / _scales return (moe.rs:1972-2090) */
int kr_upload_expert_marlin(kr_engine* e, int layer, int expert, int inter, const uint32_t* w13_packed, const uint16_t* w13_scales,
                            const uint32_t* w2_packed, const uint16_t* w2_scales, int bits);
int kr_download_expert_marlin(kr_engine* e, int layer, int expert, uint32_t* w13_packed, uint16_t* w13_scales, uint32_t* w2_packed, uint16_t* w2_scales);

/* ---- MoE forward: moe_forward_unified / moe_forward_gguf (moe.rs:572, 990) behind
 *      KrasisEngine.moe_forward (moe.rs:1775), forward_moe_direct (moe.rs:2843), submit/sync_forward (:2723).
 *   act  bf16 [batch, hidden]; ids i32 [batch, topk] (-1 = skip, moe.rs:2904); w f32 [batch, topk]
 *   out  f32 or bf16 [batch, hidden].  routed_only != 0 omits shared expert and rsf (gpu_prefill.py:4467). */
int kr_moe_forward(kr_engine* e, int layer, const void* act_bf16, const int32_t* ids, const float* weights,
                   void* out, int batch, int topk, int out_dtype, int routed_only, void* stream);

/* ---- prefill: GpuPrefillManager.forward(moe_layer_idx, hidden[M,H] bf16, topk_ids[M,k] i32, topk_weights[M,k] f32, routed_only)
 *      (python/krasis/gpu_prefill.py:4374-4394).  All pointers are device pointers.  Token sort + int8-MFMA grouped GEMM; every row
 *      carries the arithmetic of expert_forward_unified (moe.rs:184), so the result equals kr_moe_forward on the same batch bit for bit.
 *      out = rsf*routed + shared unless routed_only (gpu_prefill.py:4467-4484).
 *      Native-GGUF layers (kr_upload_expert_gguf): Q4_K / Q8_0 blocks run on the same matrix cores -- raw super-blocks staged in LDS, one
 *      int8 MFMA per 32-wide sub-block and activation digit, the per-sub-block scale / min epilogue of gguf_kernels.rs:271-432 -- with ONE
 *      f32 chain per output instead of the AVX2 kernel's eight lane chains: equal to kr_moe_forward within ~1e-6 relative (stated in
 *      tests/test_gguf_gpu.py), not bit for bit; Q4_0 / Q5_0 / Q6_K layers walk the batch through the bit-exact streaming kernels. ---- */
int kr_moe_prefill(kr_engine* e, int layer, const void* x_bf16, const int32_t* ids, const float* weights, void* out, int M, int topk,
                   int out_dtype, int routed_only, void* stream);

/* ---- routing: set_routing_config / set_routing_weights / forward_moe_routed (moe.rs:2959-3246) ---- */
#define KR_SCORE_SIGMOID 0 /* decode.rs scoring_func codes (decode.rs:1955-1969) */
#define KR_SCORE_SOFTMAX 1
#define KR_SCORE_TOPK_SOFTMAX 2
#define KR_ROUTE_RULE_ENGINE 0 /* bf16 gate x bf16 act, sequential f32, iterative argmax (moe.rs:3088-3236) */
#define KR_ROUTE_RULE_DECODE 1 /* f32 gate x f32 hidden, 2x8-lane fma, heap top-k (decode.rs:1385-1535,4088) */
int kr_set_routing_config(kr_engine* e, int scoring, int norm_topk_prob, int topk, int n_experts, int hidden);
/* gate: bf16 [E,H] when gate_is_f32 == 0 else f32 [E,H]; bias / e_score_corr f32 [E] or NULL (host pointers) */
int kr_set_routing_weights(kr_engine* e, int layer, const void* gate, int gate_is_f32, const float* bias,
                           const float* e_score_corr);
/* bench_decode_synthetic's router gate (decode.rs:5181: fill_random_f32(route_data, rng, 0.02) over Xorshift64, decode.rs:4356-4376), generated
 * by the library's host side; round_bf16 != 0 truncates each value to bf16 like a checkpoint's gate tensor (stored as bf16 in HBM) */
int kr_set_routing_weights_synthetic(kr_engine* e, int layer, uint64_t seed, float amp, int round_bf16);
/* x: bf16 [m,H] (rule ENGINE) or f32 [m,H] (rule DECODE); ids_out i32 [m,topk]; w_out f32 [m,topk] */
int kr_route_topk(kr_engine* e, int layer, const void* x, int m, int rule, int32_t* ids_out, float* w_out,
                  float* logits_out /* optional f32 [m,E] */, void* stream);
/* forward_moe_routed (moe.rs:3050): act bf16 [1,H] -> out bf16 [1,H], router + experts */
int kr_forward_moe_routed(kr_engine* e, int layer, const void* act_bf16, void* out_bf16, void* stream);

/* ---- reduce_sum_bf16 (moe.rs:2505): N-way bf16 sum, f32 accumulate in input order, RNE ---- */
int kr_reduce_sum_bf16(kr_engine* e, const void* const* inputs, int n_inputs, void* out, size_t n, void* stream);

/* test / tuning hook: (token, slot) pairs per pass of kr_moe_prefill (0 = default 81 920: 8192 tokens of a top-10 model, 81 920 top-1 rows
 * of the expert-parallel dispatch); larger batches are walked in passes of pairs / topk tokens */
int kr_moe_set_prefill_pairs(kr_engine* e, int pairs);
/* numerics of the prompt-pass expert GEMMs of kr_moe_prefill: 0 (default) = the CPU engine's arithmetic (INT16 activation digits, one f32 fma per
 * 128-group: bit-identical to kr_moe_forward, moe.rs:184); 1 = tolerance form: f16 activations x weights de-quantized in registers, f32 accumulation
 * over the whole k range -- the dataflow of the reference's GPU prompt pass (gpu_prefill.py:64-239, Marlin).  Native GGUF layers stay exact.
 * 3 = the tolerance form on the register-staged kernels only (the LDS-ring kernel of kr_prefill_ring.hip off, process-wide): A/B and test hook,
 * results are bit-identical to mode 1; 5 = the ring kernel for every shape it takes, not only problems that fill the chip (test hook, same bits). */
int kr_moe_set_gemm_mode(kr_engine* e, int fast);
/* expert-parallel combine: out[t] = sum_s w[t][s] * eo_rows[pair_row[t][s]] in routing order (moe.rs:661-667); pair_row -1 = skip */
int kr_combine_rows(kr_engine* e, const float* eo_rows, const int32_t* pair_row, const float* weights, void* out, int M, int topk,
                    int out_dtype, void* stream);

/* ---- expert parallelism over RCCL inside the library (reference dataflow python/krasis/gpu_prefill.py:353-359,4140-4148 +
 *      python/krasis/model.py:3131-3241).  One process per GPU; rank r's engine is configured with ITS experts only (contiguous slice
 *      [r * floor(E/R), ...), the last rank takes the remainder) as local experts 0..n_local-1.  kr_moe_prefill_ep takes this rank's SHARD of
 *      the tokens with GLOBAL expert ids: every (token, slot) row travels once to the rank that owns its expert (ncclSend / ncclRecv groups
 *      -- xGMI is a full mesh, no ring), runs through that rank's expert GEMMs and returns; the source rank combines its rows in routing
 *      order.  With f32 return rows the result equals single-GPU kr_moe_prefill bit for bit; return_bf16 halves the return bytes for one
 *      extra rounding per row.  Bootstrap: rank 0 calls kr_ep_unique_id, the host carries the 128 bytes to the other ranks. ---- */
int kr_ep_unique_id(void* id_out128);
int kr_ep_init(kr_engine* e, int world, int rank, int n_experts_total, const void* id128 /* NULL when world == 1 */, int return_bf16);
int kr_ep_destroy(kr_engine* e);
int kr_moe_prefill_ep(kr_engine* e, int layer, const void* x_bf16, const int32_t* ids_global, const float* weights, void* out, int M, int topk,
                      int out_dtype, int routed_only, void* stream);   /* COLLECTIVE: every rank calls, same layer order; M == 0 (empty shard, NULL pointers) is valid and still takes part */
int kr_ep_comm_ranks(kr_engine* e, int* n_out);                        /* ranks of this engine's communicator as RCCL counts them (ncclCommCount) */
int kr_ep_max_int(kr_engine* e, int value, int* max_out, void* stream);                /* collective: max of `value` over the ranks (kr_decode_prefill pads ranks with fewer chunks) */
int kr_ep_allreduce_f32(kr_engine* e, float* buf_dev, size_t n, void* stream);         /* collective: in-place f32 sum over the ranks (expert-parallel decode step) */
/* Expert-parallel DECODE: a decode store whose engine has expert parallelism initialised (world > 1) runs kr_decode_step as a collective -- router,
 * attention, norms and the shared expert replicated on every rank, the routed experts of the rank's slice only, one f32 all-reduce of the k expert rows
 * ([k, hidden]; each row is non-zero on exactly one rank, so the result equals single-engine decode bit for bit) per MoE layer before the combine in
 * routing order.  The steps are enqueued eagerly (no hipGraph replay) and KR_DECODE_FAST is ignored on such stores. */
/* loopback transport: W virtual ranks = W engines of ONE process (normally on one device), every rank driven by its own host thread while a
 * collective call is in flight; the exchanges are hipMemcpyAsync pulls between the engines' buffers bracketed by host barriers.  It runs the
 * SAME split-size / offset / scatter code as the RCCL transport, so a single-GPU box can check world sizes 2, 3 (remainder slice), 8
 * against single-engine execution (tests/test_ep_gpu.py).  Not a performance path. */
typedef struct kr_ep_loop_group kr_ep_loop_group;
int kr_ep_loopback_create(int world, kr_ep_loop_group** out);
int kr_ep_loopback_destroy(kr_ep_loop_group* g);   /* refused (non-zero, group intact) while engines initialised on it have not been through kr_ep_destroy */
int kr_ep_init_loopback(kr_engine* e, kr_ep_loop_group* group, int rank, int n_experts_total, int return_bf16);

int kr_synchronize(kr_engine* e);

/* ================================================================================================
 * Decode graph: CpuDecodeStore (src/decode.rs:193-3602) with every weight, KV page and recurrent state in HBM.
 * Construction mirrors the reference's builder methods; pointer arguments keep their meaning (HOST f32/u16 buffers,
 * copied to the device).  decode_step is a fixed launch sequence replayed as a hipGraph.
 * ================================================================================================ */
typedef struct kr_decode_store kr_decode_store;
/* CpuDecodeStore(group_size=128, parallel=True, norm_bias_one=False)  decode.rs:229.  e may be NULL: the reference constructs the store first
 * and binds the engine LAST (set_moe_store, decode.rs:2250; decode_setup.py:1010); the store then runs on the current HIP device until
 * kr_decode_set_moe_store hands it the engine that owns the routed experts and routers (same device). */
int kr_decode_create(kr_engine* e, int group_size, int norm_bias_one, kr_decode_store** out);
int kr_decode_create_on(int device_ordinal, int group_size, int norm_bias_one, kr_decode_store** out);   /* the same, bare engine on the NAMED device (multi-GPU hosts) */
int kr_decode_set_moe_store(kr_decode_store* s, kr_engine* e);
void kr_decode_destroy(kr_decode_store* s);
/* store_weight_f32(ptr, rows, cols, bits) -> id (decode.rs:280): f32 [rows, cols] -> quantize_f32_to_transposed_int4/8
 * (decode.rs:46,115; scale = amax/7, q = round(v * 7/amax)) -> HBM */
int kr_decode_store_weight_f32(kr_decode_store* s, const float* w, int rows, int cols, int bits, int* id_out);
/* fake_transposed_weight (decode.rs:4480): random packed words + bf16 scales in [0.005,0.05], generated on the GPU */
int kr_decode_store_weight_synthetic(kr_decode_store* s, int rows, int cols, int bits, uint64_t seed, int* id_out);
/* read a stored weight back in the reference layout: packed [cols/8, rows] u32 (or [cols, rows] i8), scales [cols/128, rows] */
int kr_decode_download_weight(kr_decode_store* s, int wid, void* packed_t, uint16_t* scales_t);
int kr_decode_store_norm_weight(kr_decode_store* s, const float* w, int n, int* id_out);            /* decode.rs:430 */
/* configure_decode (decode.rs:1955): scoring 0 sigmoid / 1 softmax / 2 topk-softmax; embedding f32 [vocab, hidden] on the
 * host, or NULL for a GPU-generated +-0.1 table (decode.rs:5364) from synth_seed */
int kr_decode_configure(kr_decode_store* s, int hidden, int n_layers, float eps, int final_norm_id, int lm_head_wid, int vocab,
                        int topk, int scoring, int norm_topk_prob, float routed_scaling_factor, const float* embedding,
                        uint64_t synth_seed);
int kr_decode_add_la_layer(kr_decode_store* s, int input_norm_id, int post_attn_norm_id, int in_proj_qkvz_wid, int in_proj_ba_wid,
                           int out_proj_wid, const float* conv_weight, const float* a_log, const float* dt_bias,
                           const float* norm_weight, int nk, int nv, int dk, int dv, int kernel_dim, float scale); /* decode.rs:2036 */
int kr_decode_add_gqa_layer(kr_decode_store* s, int input_norm_id, int post_attn_norm_id, int q_wid, int k_wid, int v_wid, int o_wid,
                            const float* q_norm, int q_norm_len, const float* k_norm, int k_norm_len, int gated, int num_heads,
                            int num_kv_heads, int head_dim, float sm_scale);                                       /* decode.rs:2082 */
/* add_decode_mla_layer (decode.rs:2131): w_kc / w_vc are bf16 [nh, nope, klr] / [nh, vhd, klr] on the host (widened to f32 on upload,
 * decode.rs:2157-2160); q_proj_wid < 0 selects the LoRA query path (q_a_proj -> q_a_norm -> q_b_proj); rope tables f32
 * [rope_max_seq, qk_rope_dim/2].  The layer's caches are FP16 [kv_max_seq, kv_lora_rank] (compressed KV) and [kv_max_seq, qk_rope_dim]
 * (k_pe); kr_decode_set_state / kr_decode_get_state address them through the kv_k / kv_v slots of the layer. */
int kr_decode_add_mla_layer(kr_decode_store* s, int input_norm_id, int post_attn_norm_id, int kv_a_proj_wid, int o_proj_wid,
                            int q_proj_wid, int q_a_proj_wid, int q_b_proj_wid, const uint16_t* w_kc_bf16, size_t w_kc_len,
                            const uint16_t* w_vc_bf16, size_t w_vc_len, const float* kv_a_norm, int kv_a_norm_len,
                            const float* q_a_norm, int q_a_norm_len, const float* rope_cos, const float* rope_sin, int rope_max_seq,
                            int num_heads, int kv_lora_rank, int qk_nope_dim, int qk_rope_dim, int v_head_dim, float sm_scale);
/* set_decode_layer_moe (decode.rs:2195): routed experts + router of engine MoE layer `moe_layer_idx`; shared expert weights
 * are decode-store weights (ids, -1 = none) */
int kr_decode_set_layer_moe(kr_decode_store* s, int layer, int moe_layer_idx, int shared_gate_up_wid, int shared_down_wid,
                            int shared_gate_wid);
int kr_decode_set_layer_dense(kr_decode_store* s, int layer, int gate_wid, int up_wid, int down_wid);              /* decode.rs:2215 */
int kr_decode_set_rope(kr_decode_store* s, const float* cos_table, const float* sin_table, int half_dim, int max_seq);
int kr_decode_finalize(kr_decode_store* s);
/* element type of the GQA KV caches and of the MLA compressed-KV / rope caches: KR_KV_FP16 = the reference's CPU-decode cache (decode.rs:4423-4478, default), KR_KV_FP8_E4M3 = the
 * reference's GPU cache dtype (python/krasis/kv_cache.py:38-135, torch.float8_e4m3fn: RNE, no saturation).  Call before set_decode_state;
 * kv_k / kv_v buffers of kr_decode_set_state / kr_decode_get_state then hold 1-byte elements. */
#define KR_KV_FP16 0
#define KR_KV_FP8_E4M3 1
int kr_decode_set_kv_dtype(kr_decode_store* s, int kv_dtype);
/* numerics of attention over LONG caches / long prompts (north_star: fp tolerance outside the router ids).  KR_ATTN_EXACT (default): the
 * reference's sequential softmax sum and p.v order, bit-identical to decode.rs:4194-4281.  KR_ATTN_FAST: decode caches longer than 1024
 * positions are split over (256-position chunk x KV head) workgroups and merged by log-sum-exp (same products and exponentials, another
 * summation order: logits within ~1e-4 relative, tests state 5e-4).  Call before the first step (a captured graph is rebuilt). */
#define KR_ATTN_EXACT 0
#define KR_ATTN_FAST 1
#define KR_GEMM_FAST 2   /* or-ed into the mode: every GEMM of kr_decode_prefill (projections, shared expert, routed experts, lm_head of the scoring pass) in the tolerance form of kr_moe_set_gemm_mode; decode steps are unaffected */
#define KR_DECODE_FAST 4 /* or-ed into the mode: decode steps run the tolerance-mode kernels of kr_decode_fast.hip -- the reference's products (INT16 activation digits, exact integer group sums, its sigmoid / libm functions), but every f32 reduction as a lane / wave / workgroup TREE instead of the reference's sequential chain, the norms folded into the launches that consume them, top-k + silu*up + the expert combine inside the expert launches (6 launches per linear-attention MoE layer).  Router ids are those of the exact kernels for identical logits, with one stated exception: exact ties fall back to the reference's heap order, but with plain softmax scoring + renormalised weights the selection runs on the LOGITS, so two distinct logits among the leading k + 1 whose f32 softmax scores round to the same value (a few ulp apart) are ordered by value here and by the heap's tie rule in the exact kernel; logits agree with the exact mode to the tolerance stated in tests/test_decode_fast_gpu.py.  MLA layers: projections, absorption, latent norm and w_vc in this form, the attention launch itself in the exact order.  Native-GGUF MoE layers (Q4_K / Q8_0 / Q4_0): the routed slots of the two expert launches walk the GGUF blocks (the block kernels' products, a row's blocks split over two waves).  Geometries the kernels do not cover (the down projection of a dense MLP, GPT-OSS activation, E > 512, hidden > 4096, Q5_0 / Q6_K blocks) keep the exact kernels for that layer -- never a CPU path.  The prompt pass is unaffected. */
int kr_decode_set_attention_mode(kr_decode_store* s, int mode);
int kr_decode_set_option(kr_decode_store* s, const char* name, int value);   /* test / tuning hooks by name: "gqa_stream", "pfm_timing" (see kr_decode.cpp) */                                                                        /* decode.rs:2471 */
/* Whole-model prompt pass.  Replaces the reference's GPU prefill (python/krasis/model.py forward_prefill_layer_grouped / server_prefill,
 * layer.py:242-461, attention.py:496-687, linear_attention.py:695-845 -- third-party kernels) AND the GPU->CPU state hand-off
 * (decode_setup.py:232-278): tokens[0..n) (host ints) at positions start_pos.. are run through every layer in chunks; afterwards the
 * logits of the LAST token (optional host/device f32 [vocab]), the greedy sample (kr_decode_last_token), the FP16 KV caches and the
 * conv / recurrent states are bit-identical to n successive kr_decode_step calls (decode.rs:2690-3520).  GEMM-shaped work runs on the
 * int8-MFMA grouped GEMM with the exact INT16-digit arithmetic.  INT4 / INT8-g128 weights; linear-attention, GQA and MLA layers. */
int kr_decode_prefill(kr_decode_store* s, const int32_t* tokens, int n_tokens, int start_pos, float* logits_out, void* stream);
/* Scoring prompt pass = the reference's model.forward(..., return_all_logits=True) + cross_entropy(logits[:-1], tokens[1:], reduction="none")
 * (perplexity/measure_ppl.py:212-227): as kr_decode_prefill, plus final norm + lm_head for EVERY position (one [chunk, vocab] GEMM per
 * chunk, never the [n, vocab] tensor) and nll_out[i] = logsumexp(logits_i) - logits_i[tokens[i+1]] for i in [0, n_tokens-1), f32, host or
 * device.  The logits are the decode step's bit for bit; the log-sum-exp sums f32 expf terms in double and rounds once (within one f32 ulp of a
 * float64 restatement; tests hold it to 4e-6 absolute at nll < 16). */
int kr_decode_prefill_nll(kr_decode_store* s, const int32_t* tokens, int n_tokens, int start_pos, float* nll_out, float* logits_out, void* stream);
/* fresh request state: zero caches / conv / recurrent states sized for kv_max_seq (measure_ppl.py:199-206: new SequenceKVState per window +
 * linear-attention reset_state()) */
int kr_decode_reset_state(kr_decode_store* s, int kv_max_seq);
/* tokens per chunk of the prompt pass (0 = default 1024).  Chunks rotate over `depth` HIP streams and scratch arenas: layer l of
 * chunk c+1 runs concurrently with layer l+1 of chunk c (per-layer events carry the state / KV dependencies). */
int kr_decode_set_prefill_chunk(kr_decode_store* s, int chunk);
int kr_decode_set_prefill_depth(kr_decode_store* s, int depth);   /* chunks in flight = streams = scratch arenas, 1..8 (0 = default 3) */
/* set_decode_state (decode.rs:2640): per-layer host pointers (NULL = not applicable / zero-init) */
int kr_decode_set_state(kr_decode_store* s, int seq_len, int kv_max_seq, const uint16_t* const* kv_k, const uint16_t* const* kv_v,
                        const float* const* conv_state, const float* const* recur_state);
int kr_decode_fill_state_synthetic(kr_decode_store* s, int kv_max_seq, uint64_t seed);
int kr_decode_get_state(kr_decode_store* s, int layer, uint16_t* kv_k, uint16_t* kv_v, float* conv_state, float* recur_state);
/* decode_step(token, pos, out_ptr) (decode.rs:2690): logits f32 [vocab] to a host or device pointer (NULL = keep on device) */
int kr_decode_step(kr_decode_store* s, int token_id, int position, float* logits_out, void* stream);
/* generate_batch (decode.rs:3525-3600) with the full sampler sample_from_logits (decode.rs:3718-3811): per step decode_step, presence
 * penalty on already seen tokens, 1/temperature, top-k (k largest, sorted descending; ties by ascending token id), libm softmax with
 * sequential sums in that order, top-p prefix, renormalisation, xorshift64 draw (x ^= x<<13; x ^= x>>7; x ^= x<<17; r = x / u64::MAX).
 * temperature == 0 is greedy (first maximum).  The sampled token is appended before the stop test, so a stop id is the last element.
 * rng_seed 0 = wall clock like the reference.  Everything runs on the GPU; one 4-byte DtoH per token returns the id. */
int kr_decode_generate(kr_decode_store* s, int first_token, int start_pos, int max_tokens, float temperature, int top_k, float top_p,
                       const int* stop_ids, int n_stop, float presence_penalty, uint64_t rng_seed, int* tokens_out, int* n_out, void* stream);
/* sample_from_logits on the logits of the last decode_step / prefill (modifies them in place like the reference) */
int kr_decode_sample(kr_decode_store* s, float temperature, int top_k, float top_p, float presence_penalty, uint64_t rng_seed, int reset_seen,
                     int* token_out, void* stream);
/* test aid (no reference counterpart): the token ids kr_decode_sample draws from, in its order -- the top_k largest of `logits` (host pointers), value descending,
   equal values by ascending id (decode.rs:3740-3760 sorts by value only; oracle.sample_from_logits fixes the same rule); top_k <= 0: the whole vocabulary */
int kr_sample_order(const float* logits_host, int vocab, int top_k, int32_t* ids_out_host);
/* generate_batch (decode.rs:3525), greedy sampling only in this round */
int kr_decode_generate_greedy(kr_decode_store* s, int first_token, int start_pos, int max_tokens, const int* stop_ids, int n_stop,
                              int* tokens_out, int* n_out, void* stream);
/* generate_stream (decode.rs:3611): the cancellable loop of the reference's Rust server.  on_token(token, finish_reason, user) is called once per
 * generated token -- finish_reason 0 none, 1 "stop" (a stop id, reported last), 2 "length" (max_tokens reached), 3 "cancelled" (cancel flag seen
 * before the step; token = the last token) -- and returns non-zero to continue.  Text decoding (tokenizers::Tokenizer in the reference) stays
 * with the host language.  *n_out = tokens generated (the return value of the reference method). */
typedef int (*kr_token_cb)(int token, int finish_reason, void* user);
int kr_decode_generate_stream(kr_decode_store* s, int first_token, int start_pos, int max_tokens, float temperature, int top_k, float top_p,
                              const int* stop_ids, int n_stop, float presence_penalty, uint64_t rng_seed, kr_token_cb on_token, void* user, int* n_out,
                              void* stream);
/* cancel / reset_cancel / last_decode_elapsed_s (decode.rs:253-265): the flag is read by generate_stream before every step; the elapsed time
 * is the wall time of the last generate / generate_stream loop */
int kr_decode_cancel(kr_decode_store* s);
int kr_decode_reset_cancel(kr_decode_store* s);
double kr_decode_last_elapsed_s(kr_decode_store* s);

/* ---- stand-alone CpuDecodeStore operators (decode.rs:328-1086).  Every pointer may be a host or a device pointer; host buffers are staged and
 * the call returns after the results are back.  Bit-identical to the reference methods (tests/test_standalone_ops_gpu.py against the oracle's
 * kro_op_* restatements); several differ from the decode graph's arithmetic exactly as they do in the reference (scalar loops, libm exp). */
int kr_decode_matmul(kr_decode_store* s, int weight_id, const float* input, float* output);                                   /* decode.rs:328 */
int kr_decode_matmul_batch(kr_decode_store* s, const int* weight_ids, int n, const float* input, float* const* outputs);      /* decode.rs:364 */
/* fused_add_rmsnorm (weight != NULL, decode.rs:406) / fused_add_rmsnorm_id (weight == NULL, stored norm `norm_id`, decode.rs:447) */
int kr_decode_fused_add_rmsnorm(kr_decode_store* s, float* hidden, float* residual, const float* weight, int norm_id, float eps, int size, int first_call);
int kr_decode_rmsnorm(kr_decode_store* s, const float* input, const float* weight, float eps, float* output, int size);      /* decode.rs:473 */
int kr_decode_silu_mul(kr_decode_store* s, const float* gate, const float* up, float* output, int size);                      /* decode.rs:511 */
int kr_decode_fused_shared_expert(kr_decode_store* s, int gate_up_wid, int down_wid, const float* input, float* output);      /* decode.rs:542 */
int kr_decode_linear_attention_recurrent(kr_decode_store* s, float* state, const float* q, const float* k, const float* v, const float* g, const float* beta,
                                         float* output, int nv, int dk, int dv);                                              /* decode.rs:609 */
int kr_decode_gated_rmsnorm_silu(kr_decode_store* s, const float* x, const float* z, const float* norm_weight, float* output, float eps, int nv, int dv); /* decode.rs:650 */
int kr_decode_linear_attention_conv(kr_decode_store* s, const float* qkvz, const float* ba, float* conv_state, const float* conv_weight, const float* a_log,
                                    const float* dt_bias, float scale, float* q_out, float* k_out, float* v_out, float* z_out, float* g_out, float* beta_out,
                                    int nk, int nv, int dk, int dv, int hr, int kernel_dim);                                  /* decode.rs:713 */
/* store_route_weight (decode.rs:895) / moe_route (decode.rs:955): f32 gate [E, H], optional bias / e_score_correction [E] (NULL = none) */
int kr_decode_store_route_weight(kr_decode_store* s, const float* gate, int num_experts, int hidden_dim, const float* bias, const float* e_score_corr, int* route_id_out);
int kr_decode_moe_route(kr_decode_store* s, int route_id, const float* hidden, int32_t* topk_ids_out, float* topk_weights_out, int topk, int scoring_func, int norm_topk_prob);
int kr_decode_num_route_weights(kr_decode_store* s);                                                                          /* decode.rs:1113 */
size_t kr_decode_weight_bytes(kr_decode_store* s, int weight_id);                                                             /* decode.rs:1107 */
int kr_decode_last_token(kr_decode_store* s, int* token);
int kr_decode_set_use_graph(kr_decode_store* s, int enable);
int kr_decode_read_buffer(kr_decode_store* s, int which /* 0 hidden, 1 residual, 2 router ids (i32 bits), 3 router weights, 4 router logits, 5 second residual (KR_DECODE_FAST) */, float* out, int n);
size_t kr_decode_device_bytes(const kr_decode_store* s);
/* measurement hook (bench.py): one un-graphed step with HIP events around every launch; per-kind totals (ms) and launch counts.
 * kinds: 0 embed 1 fused_add_rmsnorm 2 projection matvec 3 la_conv 4 la_recurrent 5 gated_rmsnorm_silu 6 gqa 7 route_logits
 *        8 route_select 9 moe_w13 10 moe_w2 11 moe_combine 12 lm_head 13 argmax 14 shared-gate matvec */
int kr_decode_profile_step(kr_decode_store* s, int token_id, int position, double* ms_by_kind, long* launches_by_kind, int n_kinds);

/* ---- measurement hooks (bench.py): HIP events around each kernel launch on the launch stream ----
 * kinds: 0 kr_moe_w13_kernel, 1 kr_moe_w2_kernel, 2 kr_moe_combine_kernel.  Enabling profiling serialises calls. */
int kr_set_profiling(kr_engine* e, int enable);
int kr_get_profile(kr_engine* e, int kind, double* total_ms, long* launches);

#ifdef __cplusplus
}
#endif
#endif
