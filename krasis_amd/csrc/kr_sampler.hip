// kr_sampler.hip -- sample_from_logits (src/decode.rs:3718-3811) on the GPU: presence penalty, temperature, top-k, top-p, xorshift64 draw.
//
// Reference order of operations, kept exactly:
//   logits[tok] -= presence_penalty for every seen token (decode.rs:3576-3582); logits *= 1/temperature;
//   top-k = the k largest scaled logits, sorted descending; p_i = expf(l_i - l_0) (libm); sum sequential in that order; p *= 1/sum;
//   top-p: first prefix with cumulative p >= top_p (sequential); renormalise the prefix (sequential sum, multiply by 1/sum);
//   r = (next_u64 as f64 / u64::MAX as f64) as f32; first i with r < cum_i; fallback = last of the prefix.
// The reference sorts with sort_unstable_by on the VALUE only, so the order of equal logits is whatever its pdqsort happens to produce;
// here ties are broken by ascending token id (a valid outcome of that sort, and the oracle uses the same rule).
// top-k <= 4096 (every sampling configuration the reference ships): one workgroup finds the k-th largest of the 64-bit (value, ~index) keys by radix select -- the
// keys are unique, so exactly k keys are >= it -- gathers those k into LDS and sorts them there (kr_sample_select_kernel).  Only a draw over more candidates
// (top_k = 0: the whole vocabulary) still orders the full key array with rocPRIM's radix sort.  The sequential sums run on one lane.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "kr_libm.h"
#include "kr_sampler.h"

__global__ void kr_sample_prepare_kernel(float* __restrict__ logits, int vocab, float inv_temp, float penalty, const uint32_t* __restrict__ seen,
                                         uint64_t* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vocab) return;
    float v = logits[i];
    if (penalty != 0.0f && ((seen[i >> 5] >> (i & 31)) & 1u)) v -= penalty;
    v *= inv_temp;
    logits[i] = v;
    float o = v == 0.0f ? 0.0f : v;                  // -0 == +0 for partial_cmp
    uint32_t u = __float_as_uint(o);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;      // monotone float -> uint
    keys[i] = ((uint64_t)u << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
}


// ------------------------------------------------------------------------------------------
// top-k order without a full sort.  One workgroup of 1024 threads:
//   radix select, most significant byte first: histogram of the byte among the keys that match the prefix found so far, then the bucket (walking DOWN from 255)
//   in which the k-th largest lies; after the four value bytes, if every key of that value is needed the four index bytes are skipped (no tie at the boundary:
//   the common case); gather the keys >= the threshold (exactly k: the keys are unique) into LDS, bitonic sort descending, write keys_sorted[0 .. k).
// ------------------------------------------------------------------------------------------
#define KR_SEL_CAP 4096
__global__ void __launch_bounds__(1024) kr_sample_select_kernel(const uint64_t* __restrict__ keys, int n, int k, uint64_t* __restrict__ out) {
    __shared__ uint32_t hist[256];
    __shared__ uint64_t cand[KR_SEL_CAP];
    __shared__ uint32_t s_digit, s_above, s_count;
    const int t = threadIdx.x;
    uint64_t prefix = 0; int krem = k;
    for (int p = 7; p >= 0; p--) {
        if (t < 256) hist[t] = 0;
        __syncthreads();
        const uint64_t himask = p == 7 ? 0ull : (~0ull << (8 * (p + 1)));
        for (int i = t; i < n; i += 1024) { const uint64_t key = keys[i]; if ((key & himask) == prefix) atomicAdd(&hist[(uint32_t)(key >> (8 * p)) & 255u], 1u); }
        __syncthreads();
        if (t < 64) {      // lane l owns digits 255 - 4 l .. 252 - 4 l; counts from the top
            uint32_t c[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { c[j] = hist[255 - 4 * t - j]; sum += c[j]; }
            uint32_t incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if (t >= o) incl += y; }
            uint32_t above = incl - sum;          // keys in buckets above this lane's
            if (above < (uint32_t)krem && (uint32_t)krem <= incl) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if ((uint32_t)krem <= above + c[j]) { s_digit = 255 - 4 * t - j; s_above = above; s_count = c[j]; break; }
                    above += c[j];
                }
            }
        }
        __syncthreads();
        prefix |= (uint64_t)s_digit << (8 * p); krem -= (int)s_above;
        const bool all_needed = (int)s_count == krem;
        __syncthreads();
        if (all_needed) break;          // every key of this bucket is taken: the lower bytes cannot matter (after byte 4 this is "no tie at the k-th value")
    }
    // keys >= threshold: prefix with the undecided low bits zero
    if (t == 0) s_count = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) { const uint64_t key = keys[i]; if (key >= prefix) { const uint32_t pos = atomicAdd(&s_count, 1u); if (pos < KR_SEL_CAP) cand[pos] = key; } }
    __syncthreads();
    int P = 1; while (P < k) P <<= 1;
    for (int i = k + t; i < P; i += 1024) cand[i] = 0;       // padding sorts last (no real key is 0: the index part of key 0 would be token 2^32 - 1)
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = t; i < (P >> 1); i += 1024) {
                const int lo = 2 * stride * (i / stride) + (i % stride), hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = cand[lo], b = cand[hi];
                if ((a < b) == desc) { cand[lo] = b; cand[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = t; i < k; i += 1024) out[i] = cand[i];
}

// one workgroup: exps in parallel, every sum on lane 0 in sorted order
__global__ void __launch_bounds__(256) kr_sample_draw_kernel(const uint64_t* __restrict__ sorted, const float* __restrict__ logits, int k, float top_p,
                                                            uint64_t* __restrict__ rng_state, float* __restrict__ probs, uint32_t* __restrict__ seen,
                                                            int* __restrict__ out_token) {
    __shared__ float s_inv;
    const int t = threadIdx.x;
    const int i0 = (int)(0xFFFFFFFFu - (uint32_t)sorted[0]);
    const float mx = logits[i0];
    for (int i = t; i < k; i += 256) { const int idx = (int)(0xFFFFFFFFu - (uint32_t)sorted[i]); probs[i] = kr_expf(logits[idx] - mx); }
    __syncthreads();
    if (t == 0) {
        float sum = 0.0f;
        for (int i = 0; i < k; i++) sum += probs[i];
        s_inv = 1.0f / sum;
    }
    __syncthreads();
    const float inv_sum = s_inv;
    for (int i = t; i < k; i += 256) probs[i] *= inv_sum;
    __syncthreads();
    if (t != 0) return;
    int cutoff = k;
    if (top_p < 1.0f) {
        float cum = 0.0f;
        for (int i = 0; i < k; i++) { cum += probs[i]; if (cum >= top_p) { cutoff = i + 1; break; } }
    }
    if (cutoff < k) {
        float ns = 0.0f;
        for (int i = 0; i < cutoff; i++) ns += probs[i];
        const float inv_ns = 1.0f / ns;
        for (int i = 0; i < cutoff; i++) probs[i] *= inv_ns;
    }
    uint64_t x = rng_state[0];
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    rng_state[0] = x;
    const float r = (float)((double)x / 18446744073709551615.0);
    int pick = -1; float cum = 0.0f;
    for (int i = 0; i < cutoff; i++) { cum += probs[i]; if (r < cum) { pick = i; break; } }
    if (pick < 0) pick = cutoff - 1;
    const int tok = (int)(0xFFFFFFFFu - (uint32_t)sorted[pick]);
    out_token[0] = tok;
    atomicOr(&seen[tok >> 5], 1u << (tok & 31));
}

__global__ void kr_sample_mark_seen_kernel(uint32_t* seen, const int* tok_dev, int tok_host) {
    const int tok = tok_dev ? tok_dev[0] : tok_host;
    atomicOr(&seen[tok >> 5], 1u << (tok & 31));
}

__global__ void kr_sample_penalty_kernel(float* __restrict__ logits, int vocab, float penalty, const uint32_t* __restrict__ seen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < vocab && ((seen[i >> 5] >> (i & 31)) & 1u)) logits[i] -= penalty;
}
void kr_launch_penalty(float* logits, int vocab, float penalty, const uint32_t* seen, hipStream_t st) {
    hipLaunchKernelGGL(kr_sample_penalty_kernel, dim3((vocab + 255) / 256), dim3(256), 0, st, logits, vocab, penalty, seen);
}

size_t kr_sampler_temp_bytes(int vocab) {
    size_t n = 0;
    (void)hipcub::DeviceRadixSort::SortKeysDescending(nullptr, n, (const uint64_t*)nullptr, (uint64_t*)nullptr, vocab);
    (void)hipGetLastError();
    return n;
}

int kr_launch_sample(float* logits, int vocab, float temperature, int top_k, float top_p, float penalty, uint32_t* seen, uint64_t* keys_in,
                     uint64_t* keys_sorted, void* temp, size_t temp_bytes, float* probs, uint64_t* rng_state, int* out_token, hipStream_t st) {
    const float inv_temp = 1.0f / temperature;
    hipLaunchKernelGGL(kr_sample_prepare_kernel, dim3((vocab + 255) / 256), dim3(256), 0, st, logits, vocab, inv_temp, penalty, seen, keys_in);
    if (hipGetLastError() != hipSuccess) return 1;
    const int k = (top_k > 0 && top_k < vocab) ? top_k : vocab;
    if (k <= KR_SEL_CAP) hipLaunchKernelGGL(kr_sample_select_kernel, dim3(1), dim3(1024), 0, st, (const uint64_t*)keys_in, vocab, k, keys_sorted);
    else {
        if (hipcub::DeviceRadixSort::SortKeysDescending(temp, temp_bytes, keys_in, keys_sorted, vocab, 0, 64, st) != hipSuccess) return 1;
        (void)hipGetLastError();   // rocPRIM probes device attributes; a benign failed query must not surface as the next launch's error
    }
    hipLaunchKernelGGL(kr_sample_draw_kernel, dim3(1), dim3(256), 0, st, keys_sorted, logits, k, top_p, rng_state, probs, seen, out_token);
    return 0;
}
// the first k keys of the sampler's order for device logits (prepare with temperature 1, no penalty: the logits are unchanged): the same launches kr_launch_sample issues
int kr_launch_sample_order(float* logits, int vocab, int top_k, uint32_t* seen_zero, uint64_t* keys_in, uint64_t* keys_sorted, void* temp, size_t temp_bytes, hipStream_t st) {
    hipLaunchKernelGGL(kr_sample_prepare_kernel, dim3((vocab + 255) / 256), dim3(256), 0, st, logits, vocab, 1.0f, 0.0f, seen_zero, keys_in);
    const int k = (top_k > 0 && top_k < vocab) ? top_k : vocab;
    if (k <= KR_SEL_CAP) hipLaunchKernelGGL(kr_sample_select_kernel, dim3(1), dim3(1024), 0, st, (const uint64_t*)keys_in, vocab, k, keys_sorted);
    else {
        if (hipcub::DeviceRadixSort::SortKeysDescending(temp, temp_bytes, keys_in, keys_sorted, vocab, 0, 64, st) != hipSuccess) return 1;
        (void)hipGetLastError();
    }
    return hipGetLastError() != hipSuccess;
}
void kr_launch_mark_seen(uint32_t* seen, const int* tok_dev, int tok_host, hipStream_t st) {
    hipLaunchKernelGGL(kr_sample_mark_seen_kernel, dim3(1), dim3(1), 0, st, seen, tok_dev, tok_host);
}
