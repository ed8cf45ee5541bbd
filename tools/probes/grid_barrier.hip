// Probe: cost of a software grid barrier on MI355X vs. a dependent kernel-launch boundary (graph replay).
// Every spin is bounded so the probe can never hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000u) { *err = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

__global__ void barrier_loop(unsigned* cnt, unsigned* err, int iters, float* sink) {
    unsigned nb = gridDim.x;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        if (!grid_barrier(cnt, nb * (unsigned)(i + 1), err)) break;
        acc += 1.0f;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = acc;
}

// barrier + a dependent 8 KB exchange: every block writes one float4 per thread subset, then all blocks read all of it after the barrier
__global__ void barrier_exchange(unsigned* cnt, unsigned* err, int iters, float* buf, float* sink) {
    unsigned nb = gridDim.x;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        float* b = buf + (size_t)(i & 1) * 2048;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < 2048; j += nb * blockDim.x) b[j] = acc + j;
        __threadfence();
        if (!grid_barrier(cnt, nb * (unsigned)(i + 1), err)) break;
        float s = 0.f;
        for (int j = threadIdx.x; j < 2048; j += blockDim.x) s += __builtin_nontemporal_load(b + j);
        acc = s * 1e-9f;
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

int main() {
    unsigned *cnt, *err; float *sink, *buf;
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4096 * 4)); CK(hipMalloc(&buf, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int iters = 2000;
    for (int nb : {64, 256, 512, 1024}) for (int bs : {256, 1024}) {
        if ((long)nb * bs > 256L * 2048) continue;
        for (int variant = 0; variant < 2; variant++) {
            CK(hipMemsetAsync(cnt, 0, 4, s)); CK(hipMemsetAsync(err, 0, 4, s));
            void* a0[] = {&cnt, &err, &iters, &sink};
            void* a1[] = {&cnt, &err, &iters, &buf, &sink};
            CK(hipEventRecord(e0, s));
            if (variant == 0) CK(hipLaunchCooperativeKernel((void*)barrier_loop, dim3(nb), dim3(bs), a0, 0, s));
            else CK(hipLaunchCooperativeKernel((void*)barrier_exchange, dim3(nb), dim3(bs), a1, 0, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            printf("%s nb=%4d bs=%4d : %.3f us per barrier (err=%u)\n", variant ? "barrier+exchange" : "barrier         ", nb, bs, ms * 1e3 / iters, herr);
        }
    }
    // dependent tiny-kernel chain in a graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 480; i++) tiny<<<256, 256, 0, s>>>(sink);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 10; r++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph: dependent tiny kernel boundary = %.3f us\n", ms * 1e3 / 4800);
    return 0;
}
