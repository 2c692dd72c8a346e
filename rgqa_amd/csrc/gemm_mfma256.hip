// 256x256x64 bf16 MFMA grouped GEMM (NT) for gfx950: the throughput kernel for the encoder's big projections.
//
//   C[M,N] = A[M,K] * W[N,K]^T (+ epilogue), K % 64 == 0.
//
// One workgroup of 8 waves (2 along M x 4 along N, 128x64 outputs per wave = 128 accumulator registers) per CU,
// 128 KiB of LDS = 2 stages x (A 256x64 + W 256x64) bf16.  Operand tiles go HBM/L2 -> LDS directly
// (global_load_lds_dwordx4, no staging registers); each wave-instruction fills 1 KiB = 8 rows x 128 B, and the
// XOR swizzle that keeps ds_read_b128 conflict-free is applied on the per-lane SOURCE address (LDS destination
// of an LDS-DMA is lane-linear).  The next K-step's 64 KiB are in flight while the current one is consumed
// (64 MFMAs per wave = ~2k cycles per K-step per SIMD), which is what hides the L2/HBM latency that starved
// the 128x128 register-staged kernel.  Arithmetic intensity 128 FLOP/B of LDS fill vs 64 for the 128x128 tile.
#include "gemm.h"

#define TN 256
#define TK 64
#define T256_THREADS 512
// MT = 16-row m-tiles per wave (2 waves along M): tile height TM = 32*MT in {64,128,192,256}; stage = A then W

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

__device__ __forceinline__ int xcd_remap256(int b, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}
__device__ __forceinline__ int off256(int row, int ch) { return row * 128 + ((ch ^ (row & 7)) << 4); }

// EPI is a compile-time constant: the generic (run-time switched) epilogue inlined 32x stops the compiler from
// unrolling the accumulator loops and pushes the 128 accumulators into scratch.
template <typename OutT, int EPI, int MT>
__global__ __launch_bounds__(T256_THREADS) void gemm_nt256_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int tile = xcd_remap256(blockIdx.x, g.total_tiles), pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
        if (i < g.count && tile >= g.p[i].tile_start) pi = i;
    const GemmProblem& P = g.p[pi];
    const int local = tile - P.tile_start;
    constexpr int TM = 32 * MT, A_BYTES = TM * TK * 2, STAGE_BYTES = A_BYTES + TN * TK * 2, AG = MT / 2;
    const int m0 = (local / P.tiles_n) * TM, n0 = (local % P.tiles_n) * TN;
    const int M = P.M, N = P.N, nkt = P.K / TK;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(P.A);
    const bf16_t* W = reinterpret_cast<const bf16_t*>(P.B);

    // LDS-DMA: per stage wave w fills A row groups w*AG .. (8 rows each) and W row groups w*4 .. w*4+3
    const int lrow = lane >> 3, lch = (lane & 7) ^ lrow;   // source chunk for this lane's linear LDS slot
    const bf16_t* asrc[AG];
    const bf16_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        int am = m0 + (wave * AG + i) * 8 + lrow; if (am > M - 1) am = M - 1;      // clamp: rows past the edge are never stored
        asrc[i] = A + (size_t)am * P.lda + lch * 8;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int wn_ = n0 + (wave * 4 + i) * 8 + lrow; if (wn_ > N - 1) wn_ = N - 1;
        wsrc[i] = W + (size_t)wn_ * P.ldb + lch * 8;
    }
    auto issue = [&](int stage, int kt) {
        unsigned char* base = lds + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < AG; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(asrc[i] + kt * TK), (lds_void*)(base + (wave * AG + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(wsrc[i] + kt * TK), (lds_void*)(base + A_BYTES + (wave * 4 + i) * 1024), 16, 0, 0);
    };

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nkt) issue(st ^ 1, kt + 1);
        const unsigned char* a = lds + st * STAGE_BYTES;
        const unsigned char* w = a + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 xw[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) xw[t] = *reinterpret_cast<const bf16x8*>(w + off256(wn * 64 + t * 16 + fr, s * 4 + fq));
#pragma unroll
            for (int tm = 0; tm < MT; ++tm) {
                const bf16x8 xa = *reinterpret_cast<const bf16x8*>(a + off256(wm * (16 * MT) + tm * 16 + fr, s * 4 + fq));
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xw[tn], xa, acc[tm][tn], 0, 0, 0);
            }
        }
        __syncthreads();   // vmcnt(0): this wave's LDS-DMA for stage st^1 has landed; barrier: everyone is done with stage st
    }
    // epilogue: every bias / aux operand of this lane is fetched before the first store
    float bias_r[4][4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) epi_fetch_bias(P, n0 + wn * 64 + tn * 16 + 4 * fq, bias_r[tn]);
    AuxRaw<bf16_t> aux_r[MT][4];
#pragma unroll
    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
            epi_fetch_aux(P, EPI, m0 + wm * (16 * MT) + tm * 16 + fr, n0 + wn * 64 + tn * 16 + 4 * fq, aux_r[tm][tn]);
#pragma unroll
    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
            float a4[4];
            aux_unpack(aux_r[tm][tn], a4);
            epi_finish<OutT>(P, EPI, g.drop, m0 + wm * (16 * MT) + tm * 16 + fr, n0 + wn * 64 + tn * 16 + 4 * fq, bias_r[tn], a4, v);
        }
}

// Tile height per launch: the MT in {8,6,4,2} (TM = 256/192/128/64) that minimises rounds-over-256-CUs x per-tile cost.
static int pick_mt(const GemmGroup& g, long& tiles_out) {
    int best = 8; double best_cost = 1e30; long best_tiles = 0;
    const int cand[4] = {8, 6, 4, 2};
    for (int c = 0; c < 4; ++c) {
        const int mt = cand[c];
        long tiles = 0;
        for (int i = 0; i < g.count; ++i) tiles += (long)cdiv(g.p[i].M, 32 * mt) * cdiv(g.p[i].N, TN);
        const long rounds = (tiles + 255) / 256;
        const double cost = (double)rounds * (mt + 1.5);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = mt; best_tiles = tiles; }
    }
    tiles_out = best_tiles;
    return best;
}

// true when every problem of the group can run on the LDS-DMA kernel and the launch fills enough of the chip
bool gemm_nt256_eligible(const GemmGroup& g, int out_f32) {
    if (g.a_f32 || out_f32) return false;
    const int epi = g.p[0].epi;
    if (!(epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESID_DROP || epi == EPI_DGELU || epi == EPI_ADD)) return false;
    for (int i = 0; i < g.count; ++i) {
        const GemmProblem& p = g.p[i];
        if (p.epi != epi || p.K % TK != 0 || p.K < TK) return false;
    }
    long tiles = 0;
    pick_mt(g, tiles);
    return tiles >= 96;
}

template <int EPI, int MT>
static int launch256(GemmGroup& g, hipStream_t s) {
    constexpr int LDS_BYTES = 2 * (32 * MT * TK * 2 + TN * TK * 2);
    static bool attr_set = false;
    if (!attr_set) {
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256_kernel<bf16_t, EPI, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    gemm_group_finalize(g, 32 * MT, TN);
    hipLaunchKernelGGL((gemm_nt256_kernel<bf16_t, EPI, MT>), dim3(g.total_tiles), dim3(T256_THREADS), LDS_BYTES, s, g);
    RGQA_LAUNCH_CHECK("gemm_nt256_kernel");
    return RGQA_OK;
}

template <int EPI>
static int launch256_mt(GemmGroup& g, int mt, hipStream_t s) {
    switch (mt) {
        case 8: return launch256<EPI, 8>(g, s);
        case 6: return launch256<EPI, 6>(g, s);
        case 4: return launch256<EPI, 4>(g, s);
        default: return launch256<EPI, 2>(g, s);
    }
}

int g_rgqa_force_mt = 0;
int launch_gemm_nt256_bf16(GemmGroup& g, hipStream_t s) {
    long tiles = 0;
    int mt = pick_mt(g, tiles);
    if (g_rgqa_force_mt) mt = g_rgqa_force_mt;
    switch (g.p[0].epi) {
        case EPI_BIAS: return launch256_mt<EPI_BIAS>(g, mt, s);
        case EPI_GELU: return launch256_mt<EPI_GELU>(g, mt, s);
        case EPI_RESID_DROP: return launch256_mt<EPI_RESID_DROP>(g, mt, s);
        case EPI_DGELU: return launch256_mt<EPI_DGELU>(g, mt, s);
        default: return launch256_mt<EPI_ADD>(g, mt, s);
    }
}
