// kr_sampler.h -- launch wrappers of kr_sampler.hip (sample_from_logits, decode.rs:3718)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
size_t kr_sampler_temp_bytes(int vocab);
// logits are modified in place (penalty, temperature) like the reference; returns non-zero on a sort failure
int kr_launch_sample(float* logits, int vocab, float temperature, int top_k, float top_p, float penalty, uint32_t* seen, uint64_t* keys_in,
                     uint64_t* keys_sorted, void* temp, size_t temp_bytes, float* probs, uint64_t* rng_state, int* out_token, hipStream_t st);
void kr_launch_mark_seen(uint32_t* seen, const int* tok_dev, int tok_host, hipStream_t st);
void kr_launch_penalty(float* logits, int vocab, float penalty, const uint32_t* seen, hipStream_t st);
// test aid behind kr_sample_order: keys_sorted[0 .. k) = the first k keys of the sampler's order (token id = 0xFFFFFFFF - low word); seen_zero is only read when a penalty is set (never here)
int kr_launch_sample_order(float* logits, int vocab, int top_k, uint32_t* seen_zero, uint64_t* keys_in, uint64_t* keys_sorted, void* temp, size_t temp_bytes, hipStream_t st);
