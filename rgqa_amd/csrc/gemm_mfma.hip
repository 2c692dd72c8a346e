// bf16 MFMA grouped GEMM for gfx950: 128x128x64 tiles, 4 waves (2x2), v_mfma_f32_16x16x32_bf16,
// register-staged double-buffered LDS (issue-early / write-late), XOR-swizzled LDS images.
//
//   NT: C[M,N] = A[M,K] * W[N,K]^T            forward projections and (with the transposed weight copy) dgrad
//   TN: C[M,N] = A[K,M]^T * B[K,N]            wgrad (dW[n,k] = sum_m dY[m,n] X[m,k]); operands are fetched
//                                             from natural row-major tiles with ds_read_b64_tr_b16
//
// MFMA roles are swapped (a = weight-side fragment, b = activation-side fragment) so that each lane ends
// up with 4 consecutive output COLUMNS of one output row: wide (8/16 B) epilogue loads and stores.
#include "gemm.h"
#include <stdlib.h>

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    // bijective XCD-aware remap (blocks b and b+8 share an XCD): consecutive tile ids run on one XCD/L2
    int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

__device__ __forceinline__ const GemmProblem& find_problem(const GemmGroup& g, int tile, int& local) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
        if (i < g.count && tile >= g.p[i].tile_start) pi = i;
    local = tile - g.p[pi].tile_start;
    return g.p[pi];
}

__device__ __forceinline__ uint4 zero16() { return make_uint4(0, 0, 0, 0); }

__device__ __forceinline__ uint4 cvt8_f32_bf16(const float* p) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    bf16x8 r;
    r[0] = (bf16_t)a.x; r[1] = (bf16_t)a.y; r[2] = (bf16_t)a.z; r[3] = (bf16_t)a.w;
    r[4] = (bf16_t)b.x; r[5] = (bf16_t)b.y; r[6] = (bf16_t)b.z; r[7] = (bf16_t)b.w;
    return *reinterpret_cast<uint4*>(&r);
}

// ============================================================================ NT
// LDS image per operand: [128 rows][64 k] bf16 = 8 chunks of 16 B per row; chunk' = chunk ^ ((row >> 1) & 7): the 16 rows of
// a fragment read land on 16 distinct 16-B slots of the 256-B bank row (see off256 in gemm_mfma256.hip).
__device__ __forceinline__ int nt_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

template <typename OutT, bool A_F32>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_kernel(const GemmGroup g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * BM * BK * 2];  // [buf][A|W] 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int local;
    const GemmProblem& P = find_problem(g, xcd_remap(blockIdx.x, g.total_tiles), local);
    const int m0 = (local / P.tiles_n) * BM, n0 = (local % P.tiles_n) * BN;
    const int K = P.K, nkt = (K + BK - 1) / BK;

    uint4 ra[4], rw[4];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int q = c * NTHREADS + tid, row = q >> 3, ch = q & 7, k = k0 + ch * 8;
            const int m = m0 + row, n = n0 + row;
            if (m < P.M && k < K) {
                if (A_F32) ra[c] = cvt8_f32_bf16(reinterpret_cast<const float*>(P.A) + (size_t)m * P.lda + k);
                else ra[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(P.A) + (size_t)m * P.lda + k);
            } else ra[c] = zero16();
            if (n < P.N && k < K) rw[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(P.B) + (size_t)n * P.ldb + k);
            else rw[c] = zero16();
        }
    };
    auto lwrite = [&](int buf) {
        unsigned char* a = lds + buf * (2 * BM * BK * 2);
        unsigned char* w = a + BM * BK * 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int q = c * NTHREADS + tid, row = q >> 3, ch = q & 7;
            *reinterpret_cast<uint4*>(a + nt_off(row, ch)) = ra[c];
            *reinterpret_cast<uint4*>(w + nt_off(row, ch)) = rw[c];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    gload(0);
    lwrite(0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const unsigned char* a = lds + buf * (2 * BM * BK * 2);
        const unsigned char* w = a + BM * BK * 2;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 xa[4], xw[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xa[t] = *reinterpret_cast<const bf16x8*>(a + nt_off(wm * 64 + t * 16 + fr, s * 4 + fq));
                xw[t] = *reinterpret_cast<const bf16x8*>(w + nt_off(wn * 64 + t * 16 + fr, s * 4 + fq));
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xw[tn], xa[tm], acc[tm][tn], 0, 0, 0);
        }
        if (kt + 1 < nkt) lwrite(buf ^ 1);
        __syncthreads();
    }
    // epilogue: lane holds row m = ..+fr, columns n = ..+4*fq+{0..3}
    {
        const int epi = P.epi;
        float bias_r[4][4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) epi_fetch_bias(P, n0 + wn * 64 + tn * 16 + 4 * fq, bias_r[tn]);
        AuxRaw<bf16_t> aux_r[4][4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) epi_fetch_aux(P, epi, m0 + wm * 64 + tm * 16 + fr, n0 + wn * 64 + tn * 16 + 4 * fq, aux_r[tm][tn]);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
                float a4[4];
                aux_unpack(aux_r[tm][tn], a4);
                epi_finish<OutT>(P, epi, g.drop, m0 + wm * 64 + tm * 16 + fr, n0 + wn * 64 + tn * 16 + 4 * fq, bias_r[tn], a4, v);
            }
    }
}

// ============================================================================ TN
// LDS image per operand: [64 contraction rows][128 cols] bf16 = 16 chunks per 256-B row,
// chunk' = chunk ^ (((row&3)<<2) | ((row>>2)&3))   (conflict-free for ds_read_b64_tr_b16 and b128 rows)
__device__ __forceinline__ int tn_off(int row, int ch) { return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

// 8-element MFMA fragment, element jj = tile[r0 + 8*(lane>>4) + jj][c0 + (lane&15)]
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int r0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = r0 + 8 * g + q;
    const int ch = (c0 >> 3) + (p >> 1);
    const unsigned char* a1 = tile + tn_off(row, ch) + 8 * (p & 1);
    const unsigned char* a2 = tile + tn_off(row + 4, ch) + 8 * (p & 1);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a1));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a2));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <typename OutT, bool B_F32>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(const GemmGroup g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * BK * BM * 2];  // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int local;
    const GemmProblem& P = find_problem(g, xcd_remap(blockIdx.x, g.total_tiles), local);
    const int m0 = (local / P.tiles_n) * BM, n0 = (local % P.tiles_n) * BN;
    const int K = P.K, nkt = (K + BK - 1) / BK;

    uint4 ra[4], rb[4];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int q = c * NTHREADS + tid, row = q >> 4, ch = q & 15, k = k0 + row;
            const int m = m0 + ch * 8, n = n0 + ch * 8;
            if (k < K && m + 8 <= P.lda) ra[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(P.A) + (size_t)k * P.lda + m);
            else ra[c] = zero16();
            if (k < K && n + 8 <= P.ldb) {
                if (B_F32) rb[c] = cvt8_f32_bf16(reinterpret_cast<const float*>(P.B) + (size_t)k * P.ldb + n);
                else rb[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(P.B) + (size_t)k * P.ldb + n);
            } else rb[c] = zero16();
        }
    };
    auto lwrite = [&](int buf) {
        unsigned char* a = lds + buf * (2 * BK * BM * 2);
        unsigned char* b = a + BK * BM * 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int q = c * NTHREADS + tid, row = q >> 4, ch = q & 15;
            *reinterpret_cast<uint4*>(a + tn_off(row, ch)) = ra[c];
            *reinterpret_cast<uint4*>(b + tn_off(row, ch)) = rb[c];
        }
    };

    f32x4 acc[4][4], cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_cs = P.colsum_out != nullptr && (local % P.tiles_n) == 0 && wn == 0;   // see gemm_tn_dma_kernel
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;

    gload(0);
    lwrite(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const unsigned char* a = lds + buf * (2 * BK * BM * 2);
        const unsigned char* b = a + BK * BM * 2;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 xa[4], xb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xa[t] = tr_frag(a, s * 32, wm * 64 + t * 16, lane);  // j (lane) <-> C row (M index)
                xb[t] = tr_frag(b, s * 32, wn * 64 + t * 16, lane);  // i (regs) <-> C col (N index)
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb[tn], xa[tm], acc[tm][tn], 0, 0, 0);
            if (do_cs) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) cs[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xa[tm], cs[tm], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) lwrite(buf ^ 1);
        __syncthreads();
    }
    const int fr = lane & 15, fq = lane >> 4;
    if (do_cs && fq == 0) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const int m = m0 + wm * 64 + tm * 16 + fr;
            if (m < P.M) P.colsum_out[m] = (P.epi == EPI_ACCUM) ? P.colsum_out[m] + cs[tm][0] : cs[tm][0];
        }
    }
    {
        const int epi = P.epi;
        float bias_r[4][4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) epi_fetch_bias(P, n0 + wn * 64 + tn * 16 + 4 * fq, bias_r[tn]);
        AuxRaw<bf16_t> aux_r[4][4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) epi_fetch_aux(P, epi, m0 + wm * 64 + tm * 16 + fr, n0 + wn * 64 + tn * 16 + 4 * fq, aux_r[tm][tn]);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
                float a4[4];
                aux_unpack(aux_r[tm][tn], a4);
                epi_finish<OutT>(P, epi, g.drop, m0 + wm * 64 + tm * 16 + fr, n0 + wn * 64 + tn * 16 + 4 * fq, bias_r[tn], a4, v);
            }
    }
}

// ============================================================================ host side
void gemm_group_finalize(GemmGroup& g, int bm, int bn) {
    int t = 0;
    for (int i = 0; i < g.count; ++i) {
        g.p[i].tile_start = t;
        g.p[i].tiles_n = cdiv(g.p[i].N, bn);
        t += cdiv(g.p[i].M, bm) * g.p[i].tiles_n;
    }
    g.total_tiles = t;
}

static int check_group(const GemmGroup& g, bool tn) {
    RGQA_REQUIRE(g.count >= 1 && g.count <= GEMM_MAX_PROBLEMS, "gemm: bad problem count %d", g.count);
    for (int i = 0; i < g.count; ++i) {
        const GemmProblem& p = g.p[i];
        RGQA_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm[%d]: empty problem %dx%dx%d", i, p.M, p.N, p.K);
        RGQA_REQUIRE(p.A && p.B && p.C, "gemm[%d]: null operand", i);
        RGQA_REQUIRE((p.lda % 8) == 0 && (p.ldb % 8) == 0 && (p.ldc % 4) == 0, "gemm[%d]: lda/ldb must be multiples of 8, ldc of 4 (%d %d %d)", i, p.lda, p.ldb, p.ldc);
        if (!tn) RGQA_REQUIRE((p.K % 8) == 0, "gemm[%d]: NT needs K %% 8 == 0 (K=%d)", i, p.K);
        RGQA_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0 && ((uintptr_t)p.C % 16) == 0, "gemm[%d]: operands must be 16-byte aligned", i);
        if (epi_needs_aux(p.epi))
            RGQA_REQUIRE(p.aux && (p.ldaux % 4) == 0, "gemm[%d]: epilogue %d needs aux", i, p.epi);
    }
    return RGQA_OK;
}

int g_rgqa_force_gemm128 = 0;
bool gemm_nt256_eligible(const GemmGroup& g, int out_f32);
int launch_gemm_nt256_bf16(GemmGroup& g, hipStream_t s);
int launch_gemm_nt256_f32out(GemmGroup& g, hipStream_t s);
int launch_gemm_nt256_any(GemmGroup& g, int out_f32, hipStream_t s);

int launch_gemm_nt_bf16(GemmGroup& g, int out_f32, hipStream_t s) {
    int r = check_group(g, false);
    if (r) return r;
    if (!g_rgqa_force_gemm128 && gemm_nt256_eligible(g, out_f32)) return launch_gemm_nt256_any(g, out_f32, s);
    for (int i = 0; i < g.count; ++i) RGQA_REQUIRE(g.p[i].ln_tk == nullptr, "gemm: a fused LayerNorm runs on the LDS-DMA kernels only - group not eligible / 128 x 128 kernels forced");
    RGQA_REQUIRE(!g.b_kn, "gemm: a [K, N]-operand group (dgrad on the weight as it lies) runs on the LDS-DMA kernels only - not eligible / 128 x 128 kernels forced");
    static const bool log_fb = getenv("RGQA_GEMM_LOG_FALLBACK") != nullptr;       // which launches still take the 128x128 register-staged kernel
    if (log_fb && !g_rgqa_force_gemm128)
        for (int i = 0; i < g.count; ++i)
            fprintf(stderr, "rgqa: NT fallback M=%d N=%d K=%d lda=%d ldb=%d ldc=%d ldaux=%d epi=%d a_f32=%d out_f32=%d\n", g.p[i].M, g.p[i].N, g.p[i].K, g.p[i].lda, g.p[i].ldb,
                    g.p[i].ldc, g.p[i].ldaux, g.p[i].epi, g.a_f32, out_f32);
    gemm_group_finalize(g, BM, BN);
    dim3 grid(g.total_tiles), block(NTHREADS);
    if (g.a_f32) {
        if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<float, true>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, true>), grid, block, 0, s, g);
    } else {
        if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<float, false>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, false>), grid, block, 0, s, g);
    }
    RGQA_LAUNCH_CHECK("gemm_nt_kernel");
    return RGQA_OK;
}

bool gemm_tn_dma_eligible(const GemmGroup& g);
int launch_gemm_tn_dma_bf16(GemmGroup& g, hipStream_t s);

int launch_gemm_tn_bf16(GemmGroup& g, int out_f32, hipStream_t s) {
    int r = check_group(g, true);
    if (r) return r;
    if (!g_rgqa_force_gemm128 && out_f32 && gemm_tn_dma_eligible(g)) return launch_gemm_tn_dma_bf16(g, s);
    gemm_group_finalize(g, BM, BN);
    dim3 grid(g.total_tiles), block(NTHREADS);
    if (g.a_f32) {  // for TN the f32 flag applies to the B operand (the raw f32 RoI features in visn_fc's wgrad)
        if (out_f32) hipLaunchKernelGGL((gemm_tn_kernel<float, true>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, true>), grid, block, 0, s, g);
    } else {
        if (out_f32) hipLaunchKernelGGL((gemm_tn_kernel<float, false>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, false>), grid, block, 0, s, g);
    }
    RGQA_LAUNCH_CHECK("gemm_tn_kernel");
    return RGQA_OK;
}
