// Probe: which ingredient of a real kernel raises the dependent-launch cost in a replayed graph above the ~1.5 us of an empty kernel?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Big { const float* p[30]; int n[12]; };
__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1.f; }
__global__ void k_bigarg(Big b, float* p) { if (b.n[3] == 12345) p[0] = b.p[7][0]; }
__global__ void k_lds(float* p) { extern __shared__ float sm[]; sm[threadIdx.x] = 1.f; __syncthreads(); if (p == nullptr) p[0] = sm[0]; }
__global__ void k_store(float* p) { if (threadIdx.x == 0) p[blockIdx.x] = 1.f; }
__global__ void k_load_store(const float* x, float* p) { float v = x[threadIdx.x]; if (threadIdx.x == 0) p[blockIdx.x] = v; }
__global__ void k_chain2(const int* idx, const float* x, float* p) { int i = idx[threadIdx.x & 7]; float v = x[i + threadIdx.x]; if (threadIdx.x == 0) p[blockIdx.x] = v; }
__global__ void k_rw_same(float* p) { float v = p[threadIdx.x + 4096]; if (threadIdx.x == 0) p[4096 + blockIdx.x] = v + 1.f; }   // reads what the previous launch wrote
__global__ void k_cold(const float* x, float* p, size_t off) { float v = x[off + threadIdx.x + blockIdx.x * 256]; if (threadIdx.x == 0) p[blockIdx.x] = v; }
__global__ void k_cold2(const int* idx, const float* x, float* p, size_t off) { int i = idx[off / 4096 + (threadIdx.x & 7)]; float v = x[off + i + threadIdx.x + blockIdx.x * 256]; if (threadIdx.x == 0) p[blockIdx.x] = v; }
// long straight-line code executed once per wave (~N dependent VALU ops) -- instruction-fetch cost of big kernels
template <int N> __global__ void k_code(float* p, float a) { float v = a; 
#pragma unroll
  for (int i = 0; i < N; i++) v = __builtin_fmaf(v, 1.0001f + i * 1e-7f, 0.5f + i); if (v == 123.f) p[0] = v; }
template <typename F> double run(hipStream_t st, F launch, int n = 400) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; i++) launch(i);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); for (int r = 0; r < 5; r++) CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / (5.0 * n);
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float *p, *x; int* idx; CK(hipMalloc(&p, 1 << 20)); CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&idx, 64)); CK(hipMemset(idx, 0, 64)); CK(hipMemset(x, 0, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    Big b{}; for (int i = 0; i < 30; i++) b.p[i] = x;
    for (int grid : {2, 256, 1024}) {
        printf("grid %4d: empty %.2f | bigarg %.2f | lds8k %.2f | store %.2f | load+store %.2f | 2-deep load chain %.2f | read-prev-write %.2f  (us per dependent launch)\n", grid,
               run(st, [&](int) { k_empty<<<grid, 256, 0, st>>>(p); }), run(st, [&](int) { k_bigarg<<<grid, 256, 0, st>>>(b, p); }),
               run(st, [&](int) { k_lds<<<grid, 256, 8192, st>>>(p); }), run(st, [&](int) { k_store<<<grid, 256, 0, st>>>(p); }),
               run(st, [&](int) { k_load_store<<<grid, 256, 0, st>>>(x, p); }), run(st, [&](int) { k_chain2<<<grid, 256, 0, st>>>(idx, x, p); }),
               run(st, [&](int) { k_rw_same<<<grid, 256, 0, st>>>(p); }));
    }
    float* big; const size_t BIG = (size_t)2 << 30; CK(hipMalloc(&big, BIG)); CK(hipMemset(big, 0, BIG));
    int* bidx; CK(hipMalloc(&bidx, BIG / 4096)); CK(hipMemset(bidx, 0, BIG / 4096));
    for (int grid : {2, 256}) {
        printf("grid %4d: cold load (each launch 4 MB further, 2 GB ring) %.2f | cold 2-deep chain %.2f | code 1000 fma %.2f | code 4000 fma %.2f\n", grid,
               run(st, [&](int i) { k_cold<<<grid, 256, 0, st>>>(big, p, ((size_t)i * (1 << 20)) % (BIG / 4 - (1 << 20))); }),
               run(st, [&](int i) { k_cold2<<<grid, 256, 0, st>>>(bidx, big, p, ((size_t)i * (1 << 20)) % (BIG / 4 - (1 << 20))); }),
               run(st, [&](int) { k_code<1000><<<grid, 256, 0, st>>>(p, 1.f); }), run(st, [&](int) { k_code<4000><<<grid, 256, 0, st>>>(p, 1.f); }));
    }
    return 0;
}
