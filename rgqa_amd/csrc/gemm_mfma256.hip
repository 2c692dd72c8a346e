// bf16 launchers of the 256-wide LDS-DMA GEMM kernels (gemm_nt256.h) and the bf16 weight-gradient (TN) kernel.
#include <string.h>
#include <type_traits>
#include "gemm.h"
#include <stdio.h>
#include <stdlib.h>
#include "gemm_nt256.h"

int g_rgqa_force_mt = 0;      // rgqa_debug_set key 1
int g_rgqa_nt_stagger = 8;    // rgqa_debug_set key 22 (gemm_nt256.h): blocks of a persistent NT launch that walk one tile fewer start ~0.5 x K-steps x 2 us late; 0 = off.
                              // A/B on one box (profiles/r06_ab_records.txt): bf16x3_fwd 15.346 -> 15.288 ms per step (FFN1-class launches -6 %), bf16 11.023 -> 10.985; 20: worse
int g_rgqa_nt_panel = -1;     // rgqa_debug_set key 9

// true when every problem of the group can run on the LDS-DMA kernel
bool gemm_nt256_eligible(const GemmGroup& g, int out_f32) {
    if (g.count > GEMM_NT_MAX_PROBLEMS) return false;
    if (g.a_f32) return false;
    const int epi = g.p[0].epi;
    if (out_f32) {      // f32 result: only the plain-bias epilogue on 64-row tiles (deep-ring kernel), e.g. the logits GEMM
        if (epi != EPI_BIAS) return false;
        for (int i = 0; i < g.count; ++i) {
            const GemmProblem& p = g.p[i];
            if (p.epi != epi || p.K % TK != 0 || p.K < TK || (p.ldc % 4) != 0 || (p.N % 8) != 0 || p.M > 512) return false;
        }
        return true;
    }
    if (!(epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESID_DROP || epi == EPI_DGELU || epi == EPI_ADD || epi == EPI_TANH || epi == EPI_DTANH ||
          epi == EPI_RELU || epi == EPI_RELU_DROP || epi == EPI_DRELU_DROP)) return false;
    for (int i = 0; i < g.count; ++i) {
        const GemmProblem& p = g.p[i];
        if (p.epi != epi || p.K % TK != 0 || p.K < TK || (p.ldc % 8) != 0 || (p.N % 8) != 0) return false;
        if (epi_needs_aux(epi) && (p.ldaux % 8) != 0) return false;
    }
    return true;
}

// Probe launches (bench.py roofline.peak_sustained, tools/nt_stamps.py): ONE launch of a stamped instantiation of the bf16 NT kernels - the
// persistent loop at 256- / 224-row tiles, the deep ring at 160- / 64-row tiles, plain-bias or GELU epilogue.  8 words per block.
template <int EPI, int MT>
static int launch_probe(GemmGroup& g, hipStream_t s) {
    gemm_group_finalize(g, 32 * MT, TN);
    nt_set_panels<false>(g);
    static bool attr_set = false;
    if constexpr (MT == 2 || MT == 5) {
        constexpr int NSD = MT == 2 ? 4 : 3;
        constexpr int LDS_D = NSD * (32 * MT * TK * 2 + TN * TK * 2);
        if (!attr_set) {
            RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256d_kernel<bf16_t, EPI, MT, NSD, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_D));
            attr_set = true;
        }
        hipLaunchKernelGGL((gemm_nt256d_kernel<bf16_t, EPI, MT, NSD, false, true>), dim3(g.total_tiles), dim3(T256_THREADS), LDS_D, s, nt_prefix(g));
    } else {
        constexpr int LDS_BYTES = NT256_LDS(MT);
        if (!attr_set) {
            RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256_kernel<bf16_t, EPI, MT, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
            attr_set = true;
        }
        int grid = g.total_tiles;
        if (grid > rgqa_num_cus()) grid = rgqa_num_cus();
        hipLaunchKernelGGL((gemm_nt256_kernel<bf16_t, EPI, MT, false, true>), dim3(grid), dim3(T256_THREADS), LDS_BYTES, s, nt_prefix(g));
    }
    RGQA_LAUNCH_CHECK("gemm_nt256 probe");
    return RGQA_OK;
}
int launch_gemm_nt256_probe(GemmGroup& g, int mt, hipStream_t s) {
    RGQA_REQUIRE(g.stamps != nullptr && gemm_nt256_eligible(g, 0), "gemm probe: bad problem");
    const int epi = g.p[0].epi;
    RGQA_REQUIRE(epi == EPI_BIAS || epi == EPI_GELU, "gemm probe: bias or GELU epilogue");
    if (mt == 0) mt = 8;
    if (epi == EPI_GELU) {
        switch (mt) { case 8: return launch_probe<EPI_GELU, 8>(g, s); case 7: return launch_probe<EPI_GELU, 7>(g, s);
                      case 5: return launch_probe<EPI_GELU, 5>(g, s); case 2: return launch_probe<EPI_GELU, 2>(g, s); }
    } else {
        switch (mt) { case 8: return launch_probe<EPI_BIAS, 8>(g, s); case 7: return launch_probe<EPI_BIAS, 7>(g, s);
                      case 5: return launch_probe<EPI_BIAS, 5>(g, s); case 2: return launch_probe<EPI_BIAS, 2>(g, s); }
    }
    rgqa_set_error("gemm probe: tile height 32 * %d has no stamped instantiation (8, 7, 5, 2)", mt);
    return RGQA_ERR_ARG;
}

int launch_gemm_nt256_f32out(GemmGroup& g, hipStream_t s) { return g.b_kn ? launch256<float, EPI_BIAS, 2, false, true>(g, s) : launch256<float, EPI_BIAS, 2, false>(g, s); }

int launch_gemm_nt256_bf16(GemmGroup& g, hipStream_t s) {
    long tiles = 0;
    int mt = pick_mt(g, tiles);
    if (g_rgqa_force_mt) mt = g_rgqa_force_mt;
    if (g.b_kn) {       // B operands stored [K, N] (dgrad on the weight as it lies): the epilogues a dgrad uses
        switch (g.p[0].epi) {
            case EPI_BIAS: return launch256_mt<bf16_t, EPI_BIAS, false, true>(g, mt, s);
            case EPI_DGELU: return launch256_mt<bf16_t, EPI_DGELU, false, true>(g, mt, s);
            case EPI_ADD: return launch256_mt<bf16_t, EPI_ADD, false, true>(g, mt, s);
            case EPI_DTANH: return launch256_mt<bf16_t, EPI_DTANH, false, true>(g, mt, s);
            default: rgqa_set_error("gemm: no [K, N]-operand kernel for epilogue %d", g.p[0].epi); return RGQA_ERR_ARG;
        }
    }
    switch (g.p[0].epi) {
        case EPI_BIAS: return launch256_mt<bf16_t, EPI_BIAS, false>(g, mt, s);
        case EPI_GELU: return launch256_mt<bf16_t, EPI_GELU, false>(g, mt, s);
        case EPI_RESID_DROP: {
            bool lnf = g.p[0].ln_tk != nullptr;            // the LayerNorm behind the dense layer rides in the launch: every problem or none
            for (int i = 1; i < g.count; ++i) RGQA_REQUIRE((g.p[i].ln_tk != nullptr) == lnf, "gemm: the fused LayerNorm must be set on every problem of a group or on none");
            if (lnf) {
                for (int i = 0; i < g.count; ++i)
                    RGQA_REQUIRE(g.p[i].N == 768 && g.p[i].ln_g && g.p[i].ln_b && g.p[i].ln_y && g.p[i].ln_mean && g.p[i].ln_rstd, "gemm: the fused LayerNorm needs N == 768 and all of its operands");
                return launch256_mt<bf16_t, EPI_RESID_DROP, false, false, true>(g, mt, s);
            }
            return launch256_mt<bf16_t, EPI_RESID_DROP, false>(g, mt, s);
        }
        case EPI_DGELU: return launch256_mt<bf16_t, EPI_DGELU, false>(g, mt, s);
        case EPI_TANH: return launch256_mt<bf16_t, EPI_TANH, false>(g, mt, s);
        case EPI_DTANH: return launch256_mt<bf16_t, EPI_DTANH, false>(g, mt, s);
        case EPI_RELU: return launch256_mt<bf16_t, EPI_RELU, false>(g, mt, s);
        case EPI_RELU_DROP: return launch256_mt<bf16_t, EPI_RELU_DROP, false>(g, mt, s);
        case EPI_DRELU_DROP: return launch256_mt<bf16_t, EPI_DRELU_DROP, false>(g, mt, s);
        default: return launch256_mt<bf16_t, EPI_ADD, false>(g, mt, s);
    }
}

// ============================================================================ split-K for skinny NT problems
// The [CLS]-row tail of the last layer, the pooler and the answer head are GEMMs of M = B rows (256 in training): 3-12 tiles on a 256-CU chip,
// each walking a contraction of 1,536-3,072 alone (42 us for 1.2 GFLOP).  Their contraction is cut into S <= 12 slices that run as the S problems
// of ONE grouped launch of the f32-result kernel (64-row tiles: 4 x N/256 x S blocks, 4-5 K-steps each) into f32 partial tiles, and one pass
// sums the slices in slice order - a fixed order: bit-reproducible - and applies the epilogue (bias, activation, dropout, residual, second
// output) with the shared element-wise epilogue of gemm.h.  Two launches of ~6 us for one of 25-42.
// S depends on K alone and the path is chosen by CALL SITE (the caller hands scratch over), never by M: a batch of more than 256 rows runs as
// row groups of <= 256 rows, each with the same S, so a sample's arithmetic does not depend on the batch it is in (eval at B = 1024 and
// training at B = 256 fold the same slices in the same order; tests/test_gpu_engine.py::test_full_size_batch_independence...).
#define SPLITK_ROWS 256
int g_rgqa_nt_splitk = 1;     // rgqa_debug_set key 7: 0 = skinny problems stay whole
template <typename OutT>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const GemmProblem P, const DropCfg drop, const float* __restrict__ part, int S, int ldp, size_t slice_stride,
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
    gemm_epilogue4_e<OutT, bf16_t>(P, P.epi, drop, m_base + ml, n0, v);     // global row: addresses and the dropout stream are those of the whole problem
}
static int splitk_slices(const GemmGroup& g, int out_f32) {
    if (!g_rgqa_nt_splitk || g.count != 1 || g.a_f32 || g.splitk_ws == nullptr || g.stamps != nullptr) return 0;
    const GemmProblem& p = g.p[0];
    if (p.K < 1536 || (p.K % TK) != 0 || (p.N % 8) != 0 || (p.ldc % 4) != 0 || p.Cb != nullptr) return 0;
    if (out_f32 && p.epi != EPI_BIAS) return 0;
    if (p.epi == EPI_ACCUM || (epi_needs_aux(p.epi) && (p.aux == nullptr || (p.ldaux % 4) != 0))) return 0;
    int S = (p.K / TK) / 4;
    if (S > GEMM_NT_MAX_PROBLEMS) S = GEMM_NT_MAX_PROBLEMS;
    const int rows = p.M < SPLITK_ROWS ? p.M : SPLITK_ROWS;
    if (S < 2 || (size_t)S * rows * p.N > g.splitk_floats) return 0;
    return S;
}
static int launch_gemm_nt_splitk(GemmGroup& g, int S, int out_f32, hipStream_t s) {
    const GemmProblem P = g.p[0];
    const int steps = P.K / TK, ldp = P.N;
    for (int m0 = 0; m0 < P.M; m0 += SPLITK_ROWS) {         // row groups: the scratch is reused, the launches are ordered by the stream
        const int rows = P.M - m0 < SPLITK_ROWS ? P.M - m0 : SPLITK_ROWS;
        const size_t stride = (size_t)rows * ldp;
        GemmGroup g2; memset(&g2, 0, sizeof g2);
        g2.count = S; g2.drop = g.drop; g2.b_kn = g.b_kn;
        for (int i = 0, k0 = 0; i < S; ++i) {
            const int ks = (steps / S + (i < steps % S ? 1 : 0)) * TK;
            GemmProblem& q = g2.p[i];
            q.A = reinterpret_cast<const bf16_t*>(P.A) + (size_t)m0 * P.lda + k0; q.lda = P.lda;
            q.B = reinterpret_cast<const bf16_t*>(P.B) + (g.b_kn ? (size_t)k0 * P.ldb : (size_t)k0); q.ldb = P.ldb;
            q.C = g.splitk_ws + (size_t)i * stride; q.ldc = ldp;
            q.M = rows; q.N = P.N; q.K = ks; q.epi = EPI_BIAS;
            k0 += ks;
        }
        RGQA_REQUIRE(gemm_nt256_eligible(g2, 1), "gemm split-K: internal eligibility mismatch");
        if (int r = launch_gemm_nt256_f32out(g2, s)) return r;
        const long n4 = (long)rows * (P.N >> 2);
        const int blocks = (int)((n4 + 255) / 256);
        if (out_f32) hipLaunchKernelGGL((splitk_finish_kernel<float>), dim3(blocks), dim3(256), 0, s, P, g.drop, g.splitk_ws, S, ldp, stride, m0, rows);
        else hipLaunchKernelGGL((splitk_finish_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, P, g.drop, g.splitk_ws, S, ldp, stride, m0, rows);
        RGQA_LAUNCH_CHECK("splitk_finish_kernel");
    }
    return RGQA_OK;
}
// the bf16 NT entry of the LDS-DMA kernels: skinny single problems take the split-K path when the caller provides scratch
int launch_gemm_nt256_any(GemmGroup& g, int out_f32, hipStream_t s) {
    if (const int S = splitk_slices(g, out_f32)) {
        RGQA_REQUIRE(g.p[0].ln_tk == nullptr, "gemm: the split-K path has no fused LayerNorm");
        return launch_gemm_nt_splitk(g, S, out_f32, s);
    }
    return out_f32 ? launch_gemm_nt256_f32out(g, s) : launch_gemm_nt256_bf16(g, s);
}

// ============================================================================ TN (wgrad) with LDS-DMA
//   C[M,N] (f32) (+)= A[K,M]^T * B[K,N],  K % 64 == 0:  dW[n,k] = sum_rows dY[row,n] X[row,k]
// 128 x 256 output tile, 8 waves (2 x 4, 64x64 each), 3-stage ring x (A 64x128 + B 64x256) bf16 = 144 KiB.
// Both operands are row-major over the CONTRACTION index, so the LDS images are natural row-major copies filled
// by LDS-DMA and the MFMA fragments are fetched with ds_read_b64_tr_b16.  32-byte granules of a row are XOR-
// swizzled with f(row) = (row&3) | ((row>>3)&1)<<2 so that the 8 (row, 32 B) pieces a half-wave touches per
// transposed read land on 8 distinct bank groups.  Tiles are launched longest-contraction-first so the hardware
// dispatcher balances the unequal (lang / visn / shared) problems of one launch over the 256 CUs.
#define RGQA_TN_PIPE 3     // A fragments through a register ring 3 deep (in situ -0.26 ms per step against the plain loop; depth 4 spills)
#define WM 128
#define WN 256
// MTW = 16-row m-tiles per wave (2 waves along M): output tile WMV = 32*MTW rows x 256 columns.
//   MTW = 4: 128 x 256, 3-slot ring (48 KiB per K-step per CU);  MTW = 8: 256 x 256, 2 slots of 64 KiB: twice the MFMAs per
//   DMA byte.  The loop is bound by what the LDS-DMA path delivers per CU (~50 GB/s), not by the MFMAs, so the taller tile
//   costs ~1.5x less CU time per FLOP; it halves the tile count, which only pays when something else fills the idle CUs (the
//   launch runs on the side stream beside the next layer's chain).
// (Round 1-2: an XCD-local tile placement with the long contractions cut into chunks - "TnPlan" - took the serialised weight-gradient
// time from 3.65 to 3.0 ms per step but did not move the step with the launches on the side stream, where the main stream's kernels
// already fill the CUs an unbalanced launch leaves idle; removed in round 3, the measurements are in DESIGN.md.)
template <int ACCUM, int MTW>
__global__ __launch_bounds__(T256_THREADS) void gemm_tn_dma_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int WMV = 32 * MTW, NSLOT = MTW == 4 ? 3 : 2;
    constexpr int APW = MTW / 2;                       // A pieces (1 KiB) per wave per slot
    constexpr int ARPP = 1024 / (WMV * 2), ALPR = 64 / ARPP;   // rows per A piece, lanes per A row
    constexpr int A_BYTES = TK * WMV * 2, B_BYTES = TK * WN * 2, STAGE = A_BYTES + B_BYTES;
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
    const int tiles_m = cdiv(P.M, WMV);
    const int tix = tile - P.tile_start;
    const int local = (tix % tiles_m) * P.tiles_n + (tix / tiles_m);     // back to the m-major id used below
    const int m0 = (local / P.tiles_n) * WMV, n0 = (local % P.tiles_n) * WN;
    // contraction length need not be a multiple of the K-step (packed language rows): in the last, partial step the A rows
    // past K come from a zero line (they also feed the bias column sums) and the B rows past K re-read row K-1 (finite data
    // times zero), so no lane predicates its DMA.
    const int nkt = cdiv(P.K, TK), ktail = P.K % TK;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(P.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(P.B);
    const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g.zeros);

    // per-lane DMA sources (row within the K-step, column chunk after un-swizzling); columns clamped in-bounds
    const bf16_t* asrc[APW];
    const bf16_t* bsrc[4];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int row = (wave * APW + i) * ARPP + lane / ALPR;
        int col = m0 + (((lane % ALPR) ^ (tn_f(row) << 1)) << 3);
        if (col > P.lda - 8) col = P.lda - 8;
        asrc[i] = A + (size_t)row * P.lda + col;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 2 + (lane >> 5);
        int col = n0 + (((lane & 31) ^ (tn_f(row) << 1)) << 3);
        if (col > P.ldb - 8) col = P.ldb - 8;
        bsrc[i] = B + (size_t)row * P.ldb + col;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    auto issue = [&](int stage, int kt) {
        const unsigned base = lds0 + stage * STAGE;
        const size_t ao = (size_t)kt * TK * P.lda, bo = (size_t)kt * TK * P.ldb;
        if (ktail != 0 && kt == nkt - 1) {       // block-uniform
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const int row = (wave * APW + i) * ARPP + lane / ALPR;
                dma16(row < ktail ? asrc[i] + ao : zsrc, base + (wave * APW + i) * 1024);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (wave * 4 + i) * 2 + (lane >> 5);
                dma16(bsrc[i] + bo - (row < ktail ? (size_t)0 : (size_t)(row - (ktail - 1)) * P.ldb), base + A_BYTES + (wave * 4 + i) * 1024);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < APW; ++i) dma16(asrc[i] + ao, base + (wave * APW + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(bsrc[i] + bo, base + A_BYTES + (wave * 4 + i) * 1024);
    };

    f32x4 acc[MTW][4], cs[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // bias gradient = column sums of the A operand: one extra MFMA column against an all-ones fragment, done by the
    // wn == 0 waves of the first N-tile of every M-tile (wave-uniform condition)
    const bool do_cs = P.colsum_out != nullptr && (local % P.tiles_n) == 0 && wn == 0;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;

    // MTW = 4: 3-slot LDS ring, two K-steps of LDS-DMA in flight: every wave issues 6 DMA instructions per slot, so
    // "s_waitcnt vmcnt(6)" = this wave's slot kt has landed while slot kt+1 stays in flight; the barrier then makes
    // every wave's slot-kt data visible AND proves everyone is done reading slot kt-1, whose buffer the next issue
    // overwrites.  __syncthreads() would drain vmcnt to 0 (hipcc) and serialise each K-step behind a full DMA latency.
    // MTW = 8: two 64-KiB slots, one K-step in flight (vmcnt(0) + barrier per step, as in the NT kernel).
    issue(0, 0);
    if (NSLOT == 3 && nkt > 1) issue(1, 1);
    // the bias-gradient MFMA is wave-uniform: two copies of the loop instead of a branch after every fragment's MFMAs
    auto kloop = [&](auto CS) {
    constexpr bool DO_CS = decltype(CS)::value;
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt % NSLOT;
        if (NSLOT == 3 && kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NSLOT - 1 < nkt) issue((kt + NSLOT - 1) % NSLOT, kt + NSLOT - 1);
        const unsigned char* a = lds + st * STAGE;
        const unsigned char* b = a + A_BYTES;
        {   // A fragments through a register ring, as in nt256_kstep
            constexpr int PD = RGQA_TN_PIPE, NF = 2 * MTW;
            auto lda = [&](int i) { return tr_frag_dma<WMV * 2>(a, (i / MTW) * 32, wm * (16 * MTW) + (i % MTW) * 16, lane); };
            bf16x8 xb[2][4], ring[PD];
#pragma unroll
            for (int t = 0; t < 4; ++t) xb[0][t] = tr_frag_dma<WN * 2>(b, 0, wn * 64 + t * 16, lane);
#pragma unroll
            for (int i = 0; i < PD; ++i) ring[i] = lda(i);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int sh = i / MTW, tm = i % MTW;
                const bf16x8 xa = ring[i % PD];
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb[sh][tn], xa, acc[tm][tn], 0, 0, 0);
                if (DO_CS) cs[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xa, cs[tm], 0, 0, 0);
                if (i + PD < NF) ring[i % PD] = lda(i + PD);
                if (sh == 0 && tm == MTW - 3) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) xb[1][t] = tr_frag_dma<WN * 2>(b, 32, wn * 64 + t * 16, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
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
        for (int tm = 0; tm < MTW; ++tm) {
            const int m = m0 + wm * (16 * MTW) + tm * 16 + (lane & 15);
            if (m < P.M) cs_out[m] = accum ? cs_out[m] + cs[tm][0] : cs[tm][0];
        }
    }
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int tm = 0; tm < MTW; ++tm) {
        const int m = m0 + wm * (16 * MTW) + tm * 16 + fr;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int n = n0 + wn * 64 + tn * 16 + 4 * fq;
            if (m < P.M && n < P.N) {
                float* c = Cc + (size_t)m * ldc + n;
                float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
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

bool gemm_tn_dma_eligible(const GemmGroup& g) {
    if (g.a_f32) return false;
    const int epi = g.p[0].epi;
    if (epi != EPI_BIAS && epi != EPI_ACCUM) return false;
    long tiles = 0;
    for (int i = 0; i < g.count; ++i) {
        const GemmProblem& p = g.p[i];
        if (p.epi != epi || p.bias != nullptr || p.K < 1 || p.lda < WM || p.ldb < WN) return false;
        tiles += (long)cdiv(p.M, WM) * cdiv(p.N, WN);
    }
    return tiles >= 1;
}

int g_rgqa_tn_mtw = 0;     // rgqa_debug_set key 4: force the wgrad tile height (4 = 128 rows, 8 = 256 rows); 0 = default
template <int ACCUM, int MTW>
static int launch_tn(GemmGroup& g, hipStream_t s) {
    constexpr int WMV = 32 * MTW, LDS_BYTES = (MTW == 4 ? 3 : 2) * (TK * WMV * 2 + TK * WN * 2);
    static bool attr_set = false;
    if (!attr_set) {
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_dma_kernel<ACCUM, MTW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    gemm_group_finalize(g, WMV, WN);
    hipLaunchKernelGGL((gemm_tn_dma_kernel<ACCUM, MTW>), dim3(g.total_tiles), dim3(T256_THREADS), LDS_BYTES, s, g);
    RGQA_LAUNCH_CHECK("gemm_tn_dma_kernel");
    return RGQA_OK;
}

int launch_gemm_tn_dma_bf16(GemmGroup& g, hipStream_t s) {
    // longest contraction first: block ids are dispatched in order, so the dispatcher does LPT balancing
    for (int i = 1; i < g.count; ++i)
        for (int j = i; j > 0 && g.p[j].K > g.p[j - 1].K; --j) { GemmProblem t = g.p[j]; g.p[j] = g.p[j - 1]; g.p[j - 1] = t; }
    static void* zero_line = nullptr;
    if (zero_line == nullptr) {
        RGQA_HIP(hipMalloc(&zero_line, 256));
        RGQA_HIP(hipMemset(zero_line, 0, 256));
    }
    g.zeros = zero_line;
    // 256-row tiles when every problem has at least 256 output rows and the launch still spreads over >= half the CUs
    int mtw = g_rgqa_tn_mtw;
    if (mtw != 4 && mtw != 8) {
        long tiles8 = 0; bool tall = true;
        for (int i = 0; i < g.count; ++i) { tiles8 += (long)cdiv(g.p[i].M, 256) * cdiv(g.p[i].N, WN); if (g.p[i].M < 256 || g.p[i].lda < 256) tall = false; }
        mtw = (tall && tiles8 >= 128) ? 8 : 4;
    }
    const bool acc = g.p[0].epi == EPI_ACCUM;
    if (mtw == 8) return acc ? launch_tn<1, 8>(g, s) : launch_tn<0, 8>(g, s);
    return acc ? launch_tn<1, 4>(g, s) : launch_tn<0, 4>(g, s);
}
