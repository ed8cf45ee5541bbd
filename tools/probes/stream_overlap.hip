// Probe: how many HIP streams does this box run concurrently?  S streams, each N spin kernels of a few workgroups (no resource pressure): wall time vs one stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (sink && threadIdx.x == 1024) *sink = 1;
}
int main(int argc, char** argv) {
    const int N = 40; const long long ticks = 20000;      // 100 MHz wall clock: 200 us
    for (int S : {1, 2, 3, 4, 6}) {
        for (int with_events = 0; with_events < 2; with_events++) {
            hipStream_t st[8]; hipEvent_t ev[8][64];
            for (int i = 0; i < S; i++) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); for (int j = 0; j < N; j++) CK(hipEventCreateWithFlags(&ev[i][j], hipEventDisableTiming)); }
            for (int i = 0; i < S; i++) hipLaunchKernelGGL(spin, dim3(16), dim3(256), 0, st[i], 100, nullptr);
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int j = 0; j < N; j++)
                for (int i = 0; i < S; i++) {
                    if (with_events && i > 0) CK(hipStreamWaitEvent(st[i], ev[i - 1][j], 0));      // the chunk pipeline's pattern: stream i follows stream i - 1 step by step
                    hipLaunchKernelGGL(spin, dim3(16), dim3(256), 0, st[i], ticks, nullptr);
                    if (with_events) CK(hipEventRecord(ev[i][j], st[i]));
                }
            CK(hipDeviceSynchronize());
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("streams %d, %s: %d x 200 us kernels per stream: wall %.2f ms (one stream alone: %.2f ms)\n", S, with_events ? "staggered by events" : "independent", N, ms, N * 0.2);
            for (int i = 0; i < S; i++) CK(hipStreamDestroy(st[i]));
        }
    }
    return 0;
}
