// GEMMs of the bf16x3 precision (rgqa.h RGQA_PRECISION_BF16X3): operands in the split-f32 layout (common.h `sf32`), every product
// formed on the bf16 matrix pipe as hi*hi + hi*lo + lo*hi with f32 accumulation - the fast path that keeps the logits within the
// reference's 1e-3 bound (lxrt/modeling.py:309-346 computes in f32 throughout).
//
//   NT  C[M,N] = A[M,K] W[N,K]^T (+ epilogue)      forward projections and (on the transposed weight copy) dgrad: the LDS-DMA kernels of
//                                                   gemm_nt256.h with X3 = true; C split f32 (or f32: the logits, dL/dfeats)
//   TN  C[M,N] (f32) (+)= A[K,M]^T B[K,N]          weight gradients: gemm_tn_x3_kernel below
#include <string.h>
#include <type_traits>
#include "gemm.h"
#include "gemm_nt256.h"

// every problem: K % 32 == 0, rows of A / W / C / aux whole 128-byte lines (ld % 32 == 0, bases 128-byte aligned), N % 8 == 0
static bool x3_aligned(const void* p, int ld) { return p == nullptr || ((((uintptr_t)p) & 127) == 0 && (ld % 32) == 0); }

// ---------------------------------------------------------------------------------------------------------------- split-K for skinny problems (round 6)
// The [CLS]-row GEMMs of the LXMERT engines (tail FFN, pooler, answer head: M = B rows) and the BUTD engine's per-sample GEMMs - above all the GRU's
// per-token h W_hh^T (M = B, K = 1024) and its dgrad (K = 3072) - are 3-12 tiles on a 256-CU chip, each walking a long contraction alone: 46 us for the
// GRU's forward product, 78 us for the tail FFN2, in split f32 twice the K-steps of the bf16 kernels.  As the bf16 launcher does since round 4
// (gemm_mfma256.hip): the contraction is cut into S = K / 256 <= 12 slices that run as the S problems of ONE grouped launch of the f32-result kernel,
// and one pass folds the slices in slice order (bit-reproducible) and applies the epilogue - bias, activation, dropout, residual, the second output and the
// bf16 images.  Chosen by call site (the caller hands scratch over), never by M; more than 256 rows run as row groups of 256.
#define X3_SPLITK_ROWS 256
extern int g_rgqa_nt_splitk;     // rgqa_debug_set key 7
template <typename OutT>
__global__ __launch_bounds__(256) void splitk_finish_x3_kernel(const GemmProblem P, const DropCfg drop, const float* __restrict__ part, int S, int ldp, size_t slice_stride,
                                                               int m_base, int rows) {
    const int nq = P.N >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * nq) return;
    const int ml = (int)(i / nq), n0 = (int)(i % nq) << 2;
    const float* src = part + (size_t)ml * ldp + n0;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        float t[4]; load4(src + (size_t)s * slice_stride, t);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    gemm_epilogue4_e<OutT, sf32>(P, P.epi, drop, m_base + ml, n0, v);
}
static int x3_splitk_slices(const GemmGroup& g, int out_f32) {
    if (!g_rgqa_nt_splitk || g.count != 1 || g.splitk_ws == nullptr || g.stamps != nullptr) return 0;
    const GemmProblem& p = g.p[0];
    if (p.K < 768 || (p.K % 32) != 0 || (p.N % 8) != 0) return 0;
    if (out_f32 && p.epi != EPI_BIAS) return 0;
    if (p.epi == EPI_ACCUM || (epi_needs_aux(p.epi) && p.aux == nullptr)) return 0;
    int S = p.K / 256;
    if (S > GEMM_NT_MAX_PROBLEMS) S = GEMM_NT_MAX_PROBLEMS;
    const int rows = p.M < X3_SPLITK_ROWS ? p.M : X3_SPLITK_ROWS;
    if (S < 2 || (size_t)S * rows * p.N > g.splitk_floats) return 0;
    return S;
}
static int launch_gemm_nt_x3_splitk(GemmGroup& g, int S, int out_f32, hipStream_t s) {
    const GemmProblem P = g.p[0];
    const int steps = P.K / 32, ldp = P.N;              // K-steps of 32 contraction elements (64 staged bf16 columns)
    for (int m0 = 0; m0 < P.M; m0 += X3_SPLITK_ROWS) {
        const int rows = P.M - m0 < X3_SPLITK_ROWS ? P.M - m0 : X3_SPLITK_ROWS;
        const size_t stride = (size_t)rows * ldp;
        GemmGroup g2; memset(&g2, 0, sizeof g2);
        g2.count = S; g2.drop = g.drop;
        for (int i = 0, k0 = 0; i < S; ++i) {
            const int ks = (steps / S + (i < steps % S ? 1 : 0)) * 32;
            GemmProblem& q = g2.p[i];
            q.A = reinterpret_cast<const sf32*>(P.A) + (size_t)m0 * P.lda + k0; q.lda = P.lda;      // 32-aligned columns: whole 128-byte lines
            q.B = reinterpret_cast<const sf32*>(P.B) + k0; q.ldb = P.ldb;
            q.C = g.splitk_ws + (size_t)i * stride; q.ldc = ldp;
            q.M = rows; q.N = P.N; q.K = ks; q.epi = EPI_BIAS;
            k0 += ks;
        }
        if (int r = launch256<float, EPI_BIAS, 2, true>(g2, s)) return r;
        const long n4 = (long)rows * (P.N >> 2);
        const int blocks = (int)((n4 + 255) / 256);
        if (out_f32) hipLaunchKernelGGL((splitk_finish_x3_kernel<float>), dim3(blocks), dim3(256), 0, s, P, g.drop, g.splitk_ws, S, ldp, stride, m0, rows);
        else hipLaunchKernelGGL((splitk_finish_x3_kernel<sf32>), dim3(blocks), dim3(256), 0, s, P, g.drop, g.splitk_ws, S, ldp, stride, m0, rows);
        RGQA_LAUNCH_CHECK("splitk_finish_x3_kernel");
    }
    return RGQA_OK;
}

int launch_gemm_nt_x3(GemmGroup& g, int out_f32, hipStream_t s) {
    RGQA_REQUIRE(g.count >= 1 && g.count <= GEMM_NT_MAX_PROBLEMS, "gemm_x3: bad problem count %d", g.count);
    const int epi = g.p[0].epi;
    for (int i = 0; i < g.count; ++i) {
        const GemmProblem& p = g.p[i];
        RGQA_REQUIRE(p.M > 0 && p.N > 0 && p.K >= 32 && p.A && p.B && p.C, "gemm_x3[%d]: empty problem %dx%dx%d", i, p.M, p.N, p.K);
        RGQA_REQUIRE(p.epi == epi, "gemm_x3: one epilogue per grouped launch");
        RGQA_REQUIRE((p.K % 32) == 0 && (p.N % 8) == 0, "gemm_x3[%d]: K %% 32 and N %% 8 required (N=%d K=%d)", i, p.N, p.K);
        // A / W rows may start at any 32-aligned column of a wider split-f32 matrix: the row pitch and the base keep whole lines
        RGQA_REQUIRE(x3_aligned(p.A, p.lda) && x3_aligned(p.B, p.ldb), "gemm_x3[%d]: operands must be 128-byte aligned with ld %% 32 == 0 (lda=%d ldb=%d)", i, p.lda, p.ldb);
        if (out_f32) RGQA_REQUIRE((p.ldc % 4) == 0 && (((uintptr_t)p.C) & 15) == 0, "gemm_x3[%d]: f32 result needs ldc %% 4 and 16-byte alignment", i);
        else {
            RGQA_REQUIRE(x3_aligned(p.C, p.ldc) && (p.c2_lp || x3_aligned(p.C2, p.ldc)), "gemm_x3[%d]: split-f32 result must be 128-byte aligned with ldc %% 32 == 0 (ldc=%d)", i, p.ldc);
            RGQA_REQUIRE((((uintptr_t)p.Cb) & 15) == 0 && (!p.c2_lp || (((uintptr_t)p.C2) & 15) == 0), "gemm_x3[%d]: the bf16 images must be 16-byte aligned", i);
        }
        if (epi_needs_aux(epi)) RGQA_REQUIRE(p.aux != nullptr && x3_aligned(p.aux, p.ldaux), "gemm_x3[%d]: epilogue %d needs a split-f32 aux operand (ldaux=%d)", i, epi, p.ldaux);
    }
    if (const int S = x3_splitk_slices(g, out_f32)) return launch_gemm_nt_x3_splitk(g, S, out_f32, s);
    long tiles = 0;
    int mt = pick_mt(g, tiles);
    if (g_rgqa_force_mt) mt = g_rgqa_force_mt;
    if (out_f32) {
        RGQA_REQUIRE(epi == EPI_BIAS, "gemm_x3: f32 results take the plain bias epilogue only");
        return launch256_mt<float, EPI_BIAS, true>(g, mt, s);
    }
    switch (epi) {
        case EPI_BIAS: return launch256_mt<sf32, EPI_BIAS, true>(g, mt, s);
        case EPI_GELU: return launch256_mt<sf32, EPI_GELU, true>(g, mt, s);
        case EPI_TANH: return launch256_mt<sf32, EPI_TANH, true>(g, mt, s);
        case EPI_RESID_DROP: return launch256_mt<sf32, EPI_RESID_DROP, true>(g, mt, s);
        case EPI_DGELU: return launch256_mt<sf32, EPI_DGELU, true>(g, mt, s);
        case EPI_ADD: return launch256_mt<sf32, EPI_ADD, true>(g, mt, s);
        case EPI_DTANH: return launch256_mt<sf32, EPI_DTANH, true>(g, mt, s);
        case EPI_RELU: return launch256_mt<sf32, EPI_RELU, true>(g, mt, s);                     // BUTD (butd.py:8-26, 170-178)
        case EPI_RELU_DROP: return launch256_mt<sf32, EPI_RELU_DROP, true>(g, mt, s);
        case EPI_DRELU_DROP: return launch256_mt<sf32, EPI_DRELU_DROP, true>(g, mt, s);
        default: rgqa_set_error("gemm_x3: no kernel for epilogue %d", epi); return RGQA_ERR_ARG;
    }
}

// ============================================================================ TN (wgrad), split-f32 operands
//   C[M,N] (f32) (+)= A[K,M]^T * B[K,N]:  dW[n,k] = sum_rows dY[row,n] X[row,k]
// Both operands are row-major over the CONTRACTION index (rows), their columns split [32 hi | 32 lo] per 128-byte line: read as
// bf16 matrices of 2M / 2N columns, a 16-column MFMA fragment is either the hi or the lo parts of 16 real columns.  Output tile 256 x 256
// real = 512 x 512 bf16 columns; a K-step is 32 rows (stage = 32 x (1 KiB A row + 1 KiB B row) = 64 KiB, two slots); 8 waves (2 x 4),
// 128 x 64 real outputs each = 32 accumulators; per K-step a wave reads 16 A and 8 B fragments (ds_read_b64_tr_b16, 32-byte granule
// swizzle as in the bf16 kernel) for 96 MFMAs - 1.5x the matrix work per staged byte of the bf16 wgrad loop, which is bound by operand
// delivery.  The bias gradient rides along as an all-ones MFMA column over the hi and the lo parts of A.
#define X3_TK 32
#define X3_TCOLS 512           // bf16 columns per stage row, each operand (= 256 real columns)
#define X3_PITCH (X3_TCOLS * 2)
template <int ACCUM>
__global__ __launch_bounds__(T256_THREADS) void gemm_tn_x3_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int A_BYTES = X3_TK * X3_PITCH, STAGE = 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // Block id -> tile.  The launcher sorts the problems by contraction length; problems of equal length form a class, and the ids of a class
    // are re-dealt so that the blocks of one XCD (equal id mod 8, dispatched in id order) walk ONE contiguous run of the class's tile list:
    // a run is a compact piece of one weight matrix - M-tiles of one N-tile first - whose tiles stream the same operand rows at the same time
    // through that XCD's L2.  (Per problem instead of per class, the runs were 3-5 tiles: L2 hit rate 0.39.)  Every XCD gets an eighth of
    // every class, so the XCDs stay balanced whatever the mix of lengths.
    const int tile0 = blockIdx.x;
    int p0 = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
        if (i < g.count && tile0 >= g.p[i].tile_start) p0 = i;
    const int K0 = g.p[p0].K;
    int c_lo = g.p[p0].tile_start, c_hi = g.total_tiles;
#pragma unroll
    for (int i = 0; i < GEMM_MAX_PROBLEMS; ++i)
        if (i < g.count) {
            if (g.p[i].K == K0) c_lo = min(c_lo, g.p[i].tile_start);
            else if (g.p[i].K < K0) c_hi = min(c_hi, g.p[i].tile_start);
        }
    const int tile = c_lo + xcd_remap256(tile0 - c_lo, c_hi - c_lo);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
        if (i < g.count && tile >= g.p[i].tile_start) pi = i;
    const GemmProblem& P = g.p[pi];
    const int tiles_m = cdiv(P.M, 256);
    const int tix = tile - P.tile_start;
    const int local = (tix % tiles_m) * P.tiles_n + (tix / tiles_m);
    const int m0 = (local / P.tiles_n) * 256, n0 = (local % P.tiles_n) * 256;      // real coordinates
    const int nkt = cdiv(P.K, X3_TK), ktail = P.K % X3_TK;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(P.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(P.B);
    const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g.zeros);
    const size_t lda2 = 2 * (size_t)P.lda, ldb2 = 2 * (size_t)P.ldb;

    // per-lane DMA sources: wave w fills rows 4w .. 4w+3 of both operands' stage images; lane l = physical 16-byte chunk l of the row =
    // logical chunk l ^ (tn_f(row) << 1); columns clamped in-bounds (what lies past M / N is computed and never stored)
    const bf16_t* asrc[4];
    const bf16_t* bsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 4 + i;
        const int lch = lane ^ (tn_f(row) << 1);
        long ca = 2L * m0 + lch * 8, cb = 2L * n0 + lch * 8;
        if (ca > (long)lda2 - 8) ca = (long)lda2 - 8;
        if (cb > (long)ldb2 - 8) cb = (long)ldb2 - 8;
        asrc[i] = A + (size_t)row * lda2 + ca;
        bsrc[i] = B + (size_t)row * ldb2 + cb;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    auto issue = [&](int stage, int kt) {
        const unsigned base = lds0 + stage * STAGE;
        const size_t ao = (size_t)kt * X3_TK * lda2, bo = (size_t)kt * X3_TK * ldb2;
        if (ktail != 0 && kt == nkt - 1) {       // block-uniform: A rows past K come from a zero line, B rows past K re-read row K-1
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wave * 4 + i;
                dma16(row < ktail ? asrc[i] + ao : zsrc, base + row * X3_PITCH);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wave * 4 + i;
                dma16(bsrc[i] + bo - (row < ktail ? (size_t)0 : (size_t)(row - (ktail - 1)) * ldb2), base + A_BYTES + row * X3_PITCH);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(asrc[i] + ao, base + (wave * 4 + i) * X3_PITCH);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(bsrc[i] + bo, base + A_BYTES + (wave * 4 + i) * X3_PITCH);
    };

    f32x4 acc[8][4], cs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_cs = P.colsum_out != nullptr && (local % P.tiles_n) == 0 && wn == 0;       // wave-uniform
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;

    // fragment columns (bf16 units) inside the stage row: real fragment f of a wave's strip -> hi parts at blk * 64 + (f & 1) * 16, lo + 32
    auto acol = [&](int i) { const int rm = i >> 1; return wm * 256 + (rm >> 1) * 64 + (rm & 1) * 16 + (i & 1) * 32; };
    auto bcol = [&](int rn, int part) { return wn * 128 + (rn >> 1) * 64 + (rn & 1) * 16 + part * 32; };

    issue(0, 0);
    auto kloop = [&](auto CS) {
        constexpr bool DO_CS = decltype(CS)::value;
        for (int kt = 0; kt < nkt; ++kt) {
            const int st = kt & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nkt) issue(st ^ 1, kt + 1);
            const unsigned char* a = lds + st * STAGE;
            const unsigned char* b = a + A_BYTES;
            constexpr int PD = 3, NF = 16;
            auto lda_ = [&](int i) { return tr_frag_dma<X3_PITCH>(a, 0, acol(i), lane); };
            bf16x8 bh[4], bl[4], ring[PD];
#pragma unroll
            for (int t = 0; t < 4; ++t) bh[t] = tr_frag_dma<X3_PITCH>(b, 0, bcol(t, 0), lane);
#pragma unroll
            for (int i = 0; i < PD; ++i) ring[i] = lda_(i);
#pragma unroll
            for (int t = 0; t < 4; ++t) bl[t] = tr_frag_dma<X3_PITCH>(b, 0, bcol(t, 1), lane);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int rm = i >> 1;
                const bf16x8 xa = ring[i % PD];
                if ((i & 1) == 0) {
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) acc[rm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[tn], xa, acc[rm][tn], 0, 0, 0);
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) acc[rm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[tn], xa, acc[rm][tn], 0, 0, 0);
                } else {
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) acc[rm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[tn], xa, acc[rm][tn], 0, 0, 0);
                }
                if (DO_CS) cs[rm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xa, cs[rm], 0, 0, 0);
                if (i + PD < NF) ring[i % PD] = lda_(i + PD);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (do_cs) kloop(std::true_type{}); else kloop(std::false_type{});

    float* Cc = reinterpret_cast<float*>(P.C);
    float* cs_out = P.colsum_out;
    const int ldc = P.ldc;
    constexpr bool accum = ACCUM != 0;
    if (do_cs && (lane >> 4) == 0) {
#pragma unroll
        for (int rm = 0; rm < 8; ++rm) {
            const int m = m0 + wm * 128 + rm * 16 + (lane & 15);
            if (m < P.M) cs_out[m] = accum ? cs_out[m] + cs[rm][0] : cs[rm][0];
        }
    }
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int rm = 0; rm < 8; ++rm) {
        const int m = m0 + wm * 128 + rm * 16 + fr;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int n = n0 + wn * 64 + tn * 16 + 4 * fq;
            if (m < P.M && n < P.N) {
                float* c = Cc + (size_t)m * ldc + n;
                float v[4] = {acc[rm][tn][0], acc[rm][tn][1], acc[rm][tn][2], acc[rm][tn][3]};
                if (n + 3 < P.N) {
                    if (accum) { float o[4]; load4(c, o); v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
                    store4(c, v);
                } else {
                    for (int i = 0; i < P.N - n; ++i) c[i] = accum ? c[i] + v[i] : v[i];
                }
            }
        }
    }
}

int launch_gemm_tn_x3(GemmGroup& g, hipStream_t s) {
    RGQA_REQUIRE(g.count >= 1 && g.count <= GEMM_MAX_PROBLEMS, "gemm_tn_x3: bad problem count %d", g.count);
    const int epi = g.p[0].epi;
    RGQA_REQUIRE(epi == EPI_BIAS || epi == EPI_ACCUM, "gemm_tn_x3: epilogue %d unsupported", epi);
    for (int i = 0; i < g.count; ++i) {
        const GemmProblem& p = g.p[i];
        RGQA_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.A && p.B && p.C && p.epi == epi && p.bias == nullptr, "gemm_tn_x3[%d]: bad problem", i);
        RGQA_REQUIRE(x3_aligned(p.A, p.lda) && x3_aligned(p.B, p.ldb) && p.lda >= 32 && p.ldb >= 32, "gemm_tn_x3[%d]: operands must be 128-byte aligned with ld %% 32 == 0 (lda=%d ldb=%d)", i, p.lda, p.ldb);
        RGQA_REQUIRE((p.ldc % 4) == 0 && (((uintptr_t)p.C) & 15) == 0, "gemm_tn_x3[%d]: ldc %% 4 and 16-byte aligned C required", i);
    }
    // longest contraction first: block ids are dispatched in order, so the dispatcher does LPT balancing
    for (int i = 1; i < g.count; ++i)
        for (int j = i; j > 0 && g.p[j].K > g.p[j - 1].K; --j) { GemmProblem t = g.p[j]; g.p[j] = g.p[j - 1]; g.p[j - 1] = t; }
    static void* zero_line = nullptr;
    if (zero_line == nullptr) {
        RGQA_HIP(hipMalloc(&zero_line, 256));
        RGQA_HIP(hipMemset(zero_line, 0, 256));
    }
    g.zeros = zero_line;
    constexpr int LDS_BYTES = 2 * 2 * X3_TK * X3_PITCH;
    static bool attr_set = false;
    if (!attr_set) {
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_x3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_x3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    gemm_group_finalize(g, 256, 256);
    if (epi == EPI_ACCUM) hipLaunchKernelGGL((gemm_tn_x3_kernel<1>), dim3(g.total_tiles), dim3(T256_THREADS), LDS_BYTES, s, g);
    else hipLaunchKernelGGL((gemm_tn_x3_kernel<0>), dim3(g.total_tiles), dim3(T256_THREADS), LDS_BYTES, s, g);
    RGQA_LAUNCH_CHECK("gemm_tn_x3_kernel");
    return RGQA_OK;
}
