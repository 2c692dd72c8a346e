// Stand-alone A/B bench of the NT GEMM kernels through the C ABI (no Python): old LDS-DMA kernel (auto tile / forced MT8) vs the
// phase-interleaved kernel, on hot (one buffer set) and cold (rotating buffer sets > Infinity Cache) operands, with a bitwise
// comparison of the outputs.  Build: tools/lab/build.sh ; run on the GPU box: tools/lab/gemm_lab [shape list]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/rgqa.h"
extern "C" int rgqa_op_linear_ex(const void* A, const void* W, const float* bias, const void* aux, void* C, void* C2, int M, int N, int K,
                                 int lda, int ldw, int ldc, int ldaux, int epilogue, float drop_p, void* stream);
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#define RK(x) do { int r = (x); if (r) { fprintf(stderr, "rgqa error %d: %s (%s:%d)\n", r, rgqa_last_error_string(), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        float f = ((int)(x & 0xFFFFFF) - 0x800000) * (1.0f / 0x800000) * scale;
        unsigned u = __float_as_uint(f); u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
        p[i] = (unsigned short)u;
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        p[i] = ((int)(x & 0xFFFFFF) - 0x800000) * (1.0f / 0x800000) * scale;
    }
}

struct Set { void *A, *W, *C, *C2, *aux; float* bias; };

int main(int argc, char** argv) {
    struct Shape { int M, N, K, epi; };
    std::vector<Shape> shapes;
    for (int i = 1; i + 3 < argc; i += 4) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3])});
    if (shapes.empty()) {
        const int M = 12356;
        shapes = {{M, 2304, 768, 0}, {M, 768, 768, 3}, {M, 3072, 768, 1}, {M, 768, 3072, 3}, {M, 768, 3072, 4}, {M, 3072, 768, 5}, {M, 768, 2304, 0},
                  {3140, 2304, 768, 0}, {3140, 768, 768, 3}, {3140, 3072, 768, 1}, {3140, 768, 3072, 3}, {8192, 8192, 8192, 0}, {4096, 4096, 4096, 0}};
    }
    const int iters = getenv("LAB_ITERS") ? atoi(getenv("LAB_ITERS")) : 40;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const size_t ab = (size_t)sh.M * sh.K * 2, wb = (size_t)sh.N * sh.K * 2, cb = (size_t)sh.M * sh.N * 2;
        size_t per = ab + wb + 3 * cb;
        int nb = (int)((size_t)700e6 / per) + 1; if (nb > 24) nb = 24; if (nb < 2) nb = 2;
        std::vector<Set> sets(nb);
        for (int i = 0; i < nb; ++i) {
            Set& s = sets[i];
            CK(hipMalloc(&s.A, ab)); CK(hipMalloc(&s.W, wb)); CK(hipMalloc(&s.C, cb)); CK(hipMalloc(&s.C2, cb)); CK(hipMalloc(&s.aux, cb)); CK(hipMalloc((void**)&s.bias, sh.N * 4));
            fill_bf16<<<1024, 256, 0, st>>>((unsigned short*)s.A, (size_t)sh.M * sh.K, 1234u + i, 1.0f);
            fill_bf16<<<1024, 256, 0, st>>>((unsigned short*)s.W, (size_t)sh.N * sh.K, 777u + i, 0.05f);
            fill_bf16<<<1024, 256, 0, st>>>((unsigned short*)s.aux, (size_t)sh.M * sh.N, 99u + i, 1.0f);
            fill_f32<<<64, 256, 0, st>>>(s.bias, sh.N, 5u + i, 0.5f);
        }
        CK(hipStreamSynchronize(st));
        const float dp = (sh.epi == 3) ? 0.1f : 0.f;
        auto run = [&](const Set& s) {
            RK(rgqa_op_linear_ex(s.A, s.W, s.bias, s.aux, s.C, sh.epi == 1 ? s.C2 : nullptr, sh.M, sh.N, sh.K, sh.K, sh.K, sh.N, sh.N, sh.epi, dp, st));
        };
        auto bench = [&](bool cold) {
            for (int i = 0; i < 3; ++i) run(sets[cold ? i % nb : 0]);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) run(sets[cold ? i % nb : 0]);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1e3 / iters;
        };
        // variants: old kernel family with the model-picked tile ("auto"), then old / phase-interleaved at each forced tile height
        const int mts[4] = {8, 7, 6, 5};
        double t_auto[2] = {1e30, 1e30}, t_old[4][2], t_p8[4][2];
        long diff[4] = {-1, -1, -1, -1};
        std::vector<unsigned short> ref(cb / 2), got(cb / 2), ref2(cb / 2), got2(cb / 2);
        auto grab = [&](std::vector<unsigned short>& c, std::vector<unsigned short>& c2) {
            CK(hipMemsetAsync(sets[0].C, 0xFF, cb, st)); if (sh.epi == 1) CK(hipMemsetAsync(sets[0].C2, 0xFF, cb, st));
            run(sets[0]); CK(hipStreamSynchronize(st));
            CK(hipMemcpy(c.data(), sets[0].C, cb, hipMemcpyDeviceToHost));
            if (sh.epi == 1) CK(hipMemcpy(c2.data(), sets[0].C2, cb, hipMemcpyDeviceToHost));
        };
        for (int rep = 0; rep < 2; ++rep) {
            rgqa_debug_set(7, 0); rgqa_debug_set(1, 0);
            for (int c = 0; c < 2; ++c) { double t = bench(c == 1); if (t < t_auto[c]) t_auto[c] = t; }
            for (int k = 0; k < 4; ++k) {
                rgqa_debug_set(1, mts[k]);
                rgqa_debug_set(7, getenv("LAB_AB") ? atoi(getenv("LAB_AB")) : 0);
                for (int c = 0; c < 2; ++c) { double t = bench(c == 1); if (rep == 0 || t < t_old[k][c]) t_old[k][c] = t; }
                if (rep == 0) grab(ref, ref2);
                rgqa_debug_set(7, 2);
                for (int c = 0; c < 2; ++c) { double t = bench(c == 1); if (rep == 0 || t < t_p8[k][c]) t_p8[k][c] = t; }
                if (rep == 0) {
                    grab(got, got2);
                    diff[k] = 0; for (size_t i = 0; i < ref.size(); ++i) diff[k] += ref[i] != got[i];
                    if (sh.epi == 1) for (size_t i = 0; i < ref2.size(); ++i) diff[k] += ref2[i] != got2[i];
                }
            }
        }
        rgqa_debug_set(7, -1); rgqa_debug_set(1, 0);
        const double fl = 2.0 * sh.M * sh.N * sh.K;
        int bk = 0; for (int k = 1; k < 4; ++k) if (t_p8[k][1] < t_p8[bk][1]) bk = k;
        printf("M=%5d N=%4d K=%4d epi=%2d | cold us: auto %6.1f |", sh.M, sh.N, sh.K, sh.epi, t_auto[1]);
        for (int k = 0; k < 4; ++k) printf(" MT%d old %6.1f p8 %6.1f |", mts[k], t_old[k][1], t_p8[k][1]);
        printf(" best p8 MT%d %6.1f us %5.0f TF (%.3f of auto) | hot: auto %6.1f p8 %6.1f | mismatches %ld %ld %ld %ld\n", mts[bk], t_p8[bk][1], fl / t_p8[bk][1] / 1e6,
               t_p8[bk][1] / t_auto[1], t_auto[0], t_p8[bk][0], diff[0], diff[1], diff[2], diff[3]);
        fflush(stdout);
        for (Set& s : sets) { CK(hipFree(s.A)); CK(hipFree(s.W)); CK(hipFree(s.C)); CK(hipFree(s.C2)); CK(hipFree(s.aux)); CK(hipFree(s.bias)); }
    }
    return 0;
}
