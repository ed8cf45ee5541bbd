// Probe: operand layout of __builtin_amdgcn_mfma_i32_32x32x32_i8 / 16x16x64_i8 on gfx950.
// A[i][k], B[k][j] random small ints; every lane loads its 16 operand bytes under a hypothesis; result vs host reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k32(const int8_t* A, const int8_t* B, int* C, int hyp) {
    const int l = threadIdx.x; v4i a, b; int8_t* ap = (int8_t*)&a; int8_t* bp = (int8_t*)&b;
    for (int t = 0; t < 16; t++) {
        int k = hyp == 0 ? 16 * (l >> 5) + t : (t < 8 ? 8 * (l >> 5) + t : 16 + 8 * (l >> 5) + (t - 8));
        ap[t] = A[(l & 31) * 32 + k]; bp[t] = B[k * 32 + (l & 31)];
    }
    v16i c = {0}; c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31; C[row * 32 + col] = c[r]; }
}
typedef int v4ii __attribute__((ext_vector_type(4)));
__global__ void k16(const int8_t* A, const int8_t* B, int* C, int hyp) {
    const int l = threadIdx.x; v4i a, b; int8_t* ap = (int8_t*)&a; int8_t* bp = (int8_t*)&b;
    for (int t = 0; t < 16; t++) {
        int k = hyp == 0 ? 16 * (l >> 4) + t : (t < 8 ? 8 * (l >> 4) + t : 32 + 8 * (l >> 4) + (t - 8));
        ap[t] = A[(l & 15) * 64 + k]; bp[t] = B[k * 16 + (l & 15)];
    }
    v4ii c = {0, 0, 0, 0}; c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) { int row = (l >> 4) * 4 + r, col = l & 15; C[row * 16 + col] = c[r]; }
}
int main() {
    int8_t hA[32 * 64], hB[64 * 32]; for (int i = 0; i < 2048; i++) { hA[i] = rand() % 17 - 8; hB[i] = rand() % 17 - 8; }
    int8_t *dA, *dB; int* dC; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    int hC[1024];
    for (int hyp = 0; hyp < 2; hyp++) {
        k32<<<1, 64>>>(dA, dB, dC, hyp); hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { int s = 0; for (int k = 0; k < 32; k++) s += hA[i * 32 + k] * hB[k * 32 + j]; bad += s != hC[i * 32 + j]; }
        printf("32x32x32 hyp %d mismatches %d\n", hyp, bad);
        k16<<<1, 64>>>(dA, dB, dC, hyp); hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
        bad = 0; for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { int s = 0; for (int k = 0; k < 64; k++) s += hA[i * 64 + k] * hB[k * 16 + j]; bad += s != hC[i * 16 + j]; }
        printf("16x16x64 hyp %d mismatches %d\n", hyp, bad);
    }
    return 0;
}
