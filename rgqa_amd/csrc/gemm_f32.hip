// f32 grouped GEMM (parity mode): exact-f32 FMA accumulation on the vector ALU, LDS-tiled 64x64x16,
// 256 threads x (4x4) outputs. Strided operands cover NT / NN / TN with one kernel. This is the
// "fp32 operand mode" that holds logits within 1e-3 of the reference CPU path (BASELINE config 2);
// the throughput path is gemm_mfma.hip.
#include "gemm.h"

#define FBM 64
#define FBN 64
#define FBK 16

struct F32Strides { long sam, sak, sbk, sbn; };

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmGroup g, int trans_a, int trans_b) {
    __shared__ float As[FBK][FBM + 4];
    __shared__ float Bs[FBK][FBN + 4];
    int tile = blockIdx.x, pi = 0;
    for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
        if (i < g.count && tile >= g.p[i].tile_start) pi = i;
    const GemmProblem& P = g.p[pi];
    const int local = tile - P.tile_start;
    const int m0 = (local / P.tiles_n) * FBM, n0 = (local % P.tiles_n) * FBN;
    const float* A = reinterpret_cast<const float*>(P.A);
    const float* B = reinterpret_cast<const float*>(P.B);
    // A(m,k): !trans_a -> A[m*lda + k] ; trans_a -> A[k*lda + m]
    // B(k,n): !trans_b -> B[n*ldb + k] (i.e. W[N,K], "NT") ; trans_b -> B[k*ldb + n]
    const long sam = trans_a ? 1 : P.lda, sak = trans_a ? P.lda : 1;
    const long sbn = trans_b ? 1 : P.ldb, sbk = trans_b ? P.ldb : 1;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    float acc[4][4], cs[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_cs = P.colsum_out != nullptr && (local % P.tiles_n) == 0 && tx == 0;   // bias gradient: column sums of A
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < P.K; k0 += FBK) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int q = c * 256 + tid;
            int mm, kk;
            if (trans_a) { mm = q & 63; kk = q >> 6; } else { kk = q & 15; mm = q >> 4; }
            float v = 0.f;
            if (m0 + mm < P.M && k0 + kk < P.K) v = A[(long)(m0 + mm) * sam + (long)(k0 + kk) * sak];
            As[kk][mm] = v;
            int nn, kb;
            if (trans_b) { nn = q & 63; kb = q >> 6; } else { kb = q & 15; nn = q >> 4; }
            float w = 0.f;
            if (n0 + nn < P.N && k0 + kb < P.K) w = B[(long)(k0 + kb) * sbk + (long)(n0 + nn) * sbn];
            Bs[kb][nn] = w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < FBK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cs[i] += a[i];
            }
        }
        __syncthreads();
    }
    if (do_cs) {
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ty * 4 + i;
            if (m < P.M) P.colsum_out[m] = (P.epi == EPI_ACCUM) ? P.colsum_out[m] + cs[i] : cs[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        gemm_epilogue4<float, float>(P, g.drop, m0 + ty * 4 + i, n0 + tx * 4, v);
    }
}

int launch_gemm_f32(GemmGroup& g, int trans_a, int trans_b, hipStream_t s) {
    RGQA_REQUIRE(g.count >= 1 && g.count <= GEMM_MAX_PROBLEMS, "gemm_f32: bad problem count %d", g.count);
    for (int i = 0; i < g.count; ++i) {
        const GemmProblem& p = g.p[i];
        RGQA_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.A && p.B && p.C, "gemm_f32[%d]: bad problem", i);
        RGQA_REQUIRE((p.ldc % 4) == 0 && ((uintptr_t)p.C % 16) == 0, "gemm_f32[%d]: ldc %% 4 and 16-byte aligned C required", i);
    }
    gemm_group_finalize(g, FBM, FBN);
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(g.total_tiles), dim3(256), 0, s, g, trans_a, trans_b);
    RGQA_LAUNCH_CHECK("gemm_f32_kernel");
    return RGQA_OK;
}
