// LayerNorm kernels through the C ABI: generic (8-byte accesses, one wave per row) vs 16-byte / half-wave-per-row variants, on the
// encoder's row counts; GB/s = algorithmic bytes (bf16 in + bf16 out forward; dy + z in, dz + dzd out backward) / time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/rgqa.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#define RK(x) do { int r = (x); if (r) { fprintf(stderr, "rgqa error %d: %s (%s:%d)\n", r, rgqa_last_error_string(), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
        float f = ((int)(x & 0xFFFF) - 0x8000) * (1.0f / 0x8000);
        unsigned u = __float_as_uint(f); p[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 768;
    for (int M : {12356, 9216, 3140}) {
        const int nb = 12;
        const size_t bytes = (size_t)M * N * 2;
        std::vector<void*> X(nb), Y(nb), DY(nb), DZ(nb);
        float *gamma, *beta, *mean, *rstd, *dg, *db, *ws;
        CK(hipMalloc((void**)&gamma, N * 4)); CK(hipMalloc((void**)&beta, N * 4)); CK(hipMalloc((void**)&mean, M * 4)); CK(hipMalloc((void**)&rstd, M * 4));
        CK(hipMalloc((void**)&dg, N * 4)); CK(hipMalloc((void**)&db, N * 4)); CK(hipMalloc((void**)&ws, 512 * 3 * N * 4 + 4096));
        CK(hipMemset(gamma, 0, N * 4)); CK(hipMemset(beta, 0, N * 4));
        for (int i = 0; i < nb; ++i) {
            CK(hipMalloc(&X[i], bytes)); CK(hipMalloc(&Y[i], bytes)); CK(hipMalloc(&DY[i], bytes)); CK(hipMalloc(&DZ[i], bytes));
            fill_bf16<<<512, 256, 0, st>>>((unsigned short*)X[i], (size_t)M * N, 11 + i);
            fill_bf16<<<512, 256, 0, st>>>((unsigned short*)DY[i], (size_t)M * N, 77 + i);
        }
        CK(hipStreamSynchronize(st));
        for (int v = 0; v < 2; ++v) {
            rgqa_debug_set(9, v);
            double tf = 0, tb = 0;
            for (int pass = 0; pass < 2; ++pass) {
                const int iters = 48;
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) RK(rgqa_op_layernorm(X[i % nb], gamma, beta, Y[i % nb], mean, rstd, M, N, 1e-12f, 1, st));
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tf = ms * 1e3 / iters;
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) RK(rgqa_op_layernorm_bwd(DY[i % nb], X[i % nb], gamma, mean, rstd, DZ[i % nb], dg, db, ws, M, N, 1, st));
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); tb = ms * 1e3 / iters;
            }
            printf("M=%5d N=%d variant %d: fwd %6.2f us (%5.2f TB/s)  bwd+finalize %6.2f us (%5.2f TB/s of dy+z+dz)\n", M, N, v, tf, 2.0 * bytes / tf / 1e6, tb, 3.0 * bytes / tb / 1e6);
        }
        rgqa_debug_set(9, -1);
        for (int i = 0; i < nb; ++i) { CK(hipFree(X[i])); CK(hipFree(Y[i])); CK(hipFree(DY[i])); CK(hipFree(DZ[i])); }
        CK(hipFree(gamma)); CK(hipFree(beta)); CK(hipFree(mean)); CK(hipFree(rstd)); CK(hipFree(dg)); CK(hipFree(db)); CK(hipFree(ws));
    }
    return 0;
}
