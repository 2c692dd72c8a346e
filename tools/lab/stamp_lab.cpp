// Diagnostic: where does a block of the phase-interleaved NT kernel spend its time?  Builds its OWN copy of gemm_nt8p.hip with
// P8_STAMPS (in-kernel s_memtime / s_memrealtime stamps, wave 0 of every block), launches it on cold operands and prints the
// distribution of: launch skew, prologue (entry -> first units landed), K loop, group re-alignment, epilogue, drain.
#define P8_STAMPS 1
// private names: the library exports the same (weak) kernel stubs, and the first registration of a stub wins
#define gemm_nt8p_kernel gemm_nt8p_kernel_stamped
#define launch_gemm_nt8p_bf16 launch_gemm_nt8p_bf16_stamped
#define gemm_nt8p_eligible gemm_nt8p_eligible_stamped
#define g_rgqa_nt8p g_rgqa_nt8p_stamped
#define launch8p launch8p_stamped
#define launch8p_mt launch8p_mt_stamped
#include "../../rgqa_amd/csrc/gemm_nt8p.hip"
#include <vector>
#include <algorithm>
#include <stdio.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        float f = ((int)(x & 0xFFFFFF) - 0x800000) * (1.0f / 0x800000) * scale;
        unsigned u = __float_as_uint(f); u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
        p[i] = (unsigned short)u;
    }
}
static double med(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
static double mx(const std::vector<double>& v) { double m = 0; for (double x : v) m = x > m ? x : m; return m; }

int main(int argc, char** argv) {
    struct Shape { int M, N, K, epi, mt; };
    std::vector<Shape> shapes = {{12356, 2304, 768, 0, 8}, {12356, 768, 3072, 3, 5}, {8192, 8192, 8192, 0, 8}};
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned long long* dstamps; CK(hipMalloc(&dstamps, (256 * 32 * 2 + 256 * 2 * 4) * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_p8_stamps), &dstamps, sizeof(dstamps)));
    g_rgqa_nt8p = 1;
    for (const Shape& sh : shapes) {
        const size_t ab = (size_t)sh.M * sh.K * 2, wb = (size_t)sh.N * sh.K * 2, cb = (size_t)sh.M * sh.N * 2;
        const int nb = 6;
        void *A[nb], *W[nb], *C[nb], *C2[nb], *X[nb]; float* bias;
        CK(hipMalloc((void**)&bias, sh.N * 4)); CK(hipMemset(bias, 0, sh.N * 4));
        for (int i = 0; i < nb; ++i) {
            CK(hipMalloc(&A[i], ab)); CK(hipMalloc(&W[i], wb)); CK(hipMalloc(&C[i], cb)); CK(hipMalloc(&C2[i], cb)); CK(hipMalloc(&X[i], cb));
            fill_bf16<<<1024, 256, 0, st>>>((unsigned short*)A[i], (size_t)sh.M * sh.K, 1234u + i, 1.0f);
            fill_bf16<<<1024, 256, 0, st>>>((unsigned short*)W[i], (size_t)sh.N * sh.K, 777u + i, 0.05f);
            fill_bf16<<<1024, 256, 0, st>>>((unsigned short*)X[i], (size_t)sh.M * sh.N, 99u + i, 1.0f);
        }
        auto run = [&](int i) {
            GemmGroup g = {};
            g.count = 1; g.drop = make_drop(sh.epi == 3 ? 0.1f : 0.f, 0x1234567ull, 0);
            GemmProblem& p = g.p[0];
            p.A = A[i]; p.B = W[i]; p.C = C[i]; p.C2 = sh.epi == 1 ? C2[i] : nullptr; p.bias = bias; p.aux = X[i]; p.M = sh.M; p.N = sh.N; p.K = sh.K;
            p.lda = sh.K; p.ldb = sh.K; p.ldc = sh.N; p.ldaux = sh.N; p.epi = sh.epi; p.drop_site = 17u;
            if (launch_gemm_nt8p_bf16(g, sh.mt, st) != 0) { fprintf(stderr, "launch failed\n"); exit(1); }
            return g.total_tiles;
        };
        int tiles = 0;
        for (int i = 0; i < 8; ++i) tiles = run(i % nb);
        CK(hipStreamSynchronize(st));
        CK(hipMemsetAsync(dstamps, 0, (256 * 32 * 2 + 256 * 2 * 4) * 8, st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        run(8 % nb);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(256 * 32 * 2 + 256 * 2 * 4);
        CK(hipMemcpy(h.data(), dstamps, h.size() * 8, hipMemcpyDeviceToHost));
        const int grid = tiles < 256 ? tiles : 256;
        unsigned long long rt0 = ~0ull, rt1 = 0;
        for (int b = 0; b < grid; ++b) { unsigned long long r = h[(b * 32 + 0) * 2 + 1]; if (r && r < rt0) rt0 = r; }
        std::vector<double> skew, prol, loop[3], align[3], epi[3], drain, total, clk;
        for (int b = 0; b < grid; ++b) {
            auto T = [&](int k) { return (double)h[(b * 32 + k) * 2 + 0]; };
            auto R = [&](int k) { return (double)h[(b * 32 + k) * 2 + 1]; };
            int n = 0; while (n < 32 && h[(b * 32 + n) * 2 + 0]) ++n;
            if (n < 5) continue;
            const double cyc_per_us = (T(n - 1) - T(0)) / ((R(n - 1) - R(0)) / 100.0);      // shader clock MHz from the two counters
            clk.push_back(cyc_per_us);
            skew.push_back((R(0) - (double)rt0) / 100.0);
            prol.push_back((T(1) - T(0)) / cyc_per_us);
            const int nt = (n - 3) / 3;
            for (int i = 0; i < nt && i < 3; ++i) {
                loop[i].push_back((T(2 + 3 * i) - T(i == 0 ? 1 : 1 + 3 * i)) / cyc_per_us);
                align[i].push_back((T(3 + 3 * i) - T(2 + 3 * i)) / cyc_per_us);
                epi[i].push_back((T(4 + 3 * i) - T(3 + 3 * i)) / cyc_per_us);
            }
            drain.push_back((T(n - 1) - T(n - 2)) / cyc_per_us);
            total.push_back((R(n - 1) - (double)rt0) / 100.0);
            if (h[(b * 32 + n - 1) * 2 + 1] > rt1) rt1 = h[(b * 32 + n - 1) * 2 + 1];
        }
        printf("M=%d N=%d K=%d epi=%d MT%d: %d tiles on %d blocks, event time %.1f us, first entry -> last exit %.1f us, shader clock med %.0f MHz\n", sh.M, sh.N, sh.K, sh.epi, sh.mt,
               tiles, grid, ms * 1e3, (rt1 - rt0) / 100.0, med(clk));
        printf("   entry skew med %.2f max %.2f | prologue med %.2f max %.2f | drain med %.2f | block end (since first entry) med %.1f max %.1f\n", med(skew), mx(skew), med(prol), mx(prol), med(drain), med(total), mx(total));
        for (int i = 0; i < 3; ++i)
            if (!loop[i].empty())
                printf("   tile %d (%zu blocks): K loop med %.2f max %.2f (%.3f us per K-tile) | re-align med %.2f | epilogue med %.2f max %.2f\n", i, loop[i].size(), med(loop[i]), mx(loop[i]),
                       med(loop[i]) / (sh.K / 64), med(align[i]), med(epi[i]), mx(epi[i]));
        for (int grp = 0; grp < 2; ++grp) {
            std::vector<double> v[4];
            for (int b = 0; b < grid; ++b)
                for (int k = 0; k < 4; ++k) v[k].push_back((double)h[256 * 32 * 2 + (b * 2 + grp) * 4 + k]);
            const double ktiles = (double)(sh.K / 64) * tiles / grid, tot = med(v[0]) + med(v[1]) + med(v[2]) + med(v[3]);
            printf("   wave %d, cycles per K-tile (4 phases): read work %.0f + read-barrier wait %.0f + MFMA issue %.0f + MFMA-barrier wait %.0f = %.0f\n", grp * 4,
                   med(v[0]) / ktiles, med(v[1]) / ktiles, med(v[2]) / ktiles, med(v[3]) / ktiles, tot / ktiles);
        }
        fflush(stdout);
        for (int i = 0; i < nb; ++i) { CK(hipFree(A[i])); CK(hipFree(W[i])); CK(hipFree(C[i])); CK(hipFree(C2[i])); CK(hipFree(X[i])); }
        CK(hipFree(bias));
    }
    return 0;
}
