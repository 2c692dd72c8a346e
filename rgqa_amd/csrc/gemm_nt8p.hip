// 256x256x64 bf16 NT GEMM, phase-interleaved ("8 phase") persistent kernel for gfx950.
//
//   C[M,N] = A[M,K] * W[N,K]^T (+ epilogue), K % 64 == 0, K >= 128.  Same operands, tiles (32*MT rows x 256 columns, MT = 5..8),
//   accumulator layout, K order (bit-identical results) and epilogue as gemm_nt256_kernel<.., MT>; what differs is how a K-tile
//   is fed and scheduled (described for MT = 8; for MT < 8 the two row halves of a wave are 16*ceil(MT/2) and 16*floor(MT/2) rows):
//
//   * A K-tile (64 deep) is four 16-KiB LDS UNITS cut along the order in which a wave consumes them, not along the operands:
//       U0 = A rows {wr*128 + 0..63},  U1 = W rows {wc*64 + 0..31},  U2 = W rows {wc*64 + 32..63},  U3 = A rows {wr*128 + 64..127}
//     (wr = wave >> 2, wc = wave & 3; a wave owns a 128 x 64 output = 2 x 2 quadrants of 64 x 32).  A K-tile is four PHASES, one
//     quadrant (16 MFMAs) each: phase 1 reads U0 + U1, phase 2 reads U2, phase 3 reads U3, phase 4 re-uses registers.
//   * Every phase issues ONE unit of LDS-DMA (2 x 1 KiB per wave), seven units ahead of the unit the phase consumes: a unit is
//     re-filled as soon as its last reader is done (U0 of K-tile t+2 goes out in phase 2 of K-tile t), so 5 units = 80 KiB stay in
//     flight under counted "s_waitcnt vmcnt(10)" waits - against one K-tile (64 KiB, issued and drained once per K-step) in the
//     two-slot kernel, whose loop ran at the pace of the operand delivery.  The DMA stream does not stop at a
//     tile boundary: it runs on into the block's next tile while the epilogue converts and stores.
//   * The two wave groups (waves 0-3, waves 4-7: one wave of each per SIMD) run half a phase apart: while one group issues its
//     16 MFMAs the other reads its fragments and issues its DMA, so each SIMD's matrix pipe always has a wave on it.  Both
//     groups re-align around the epilogue (a wave-private LDS transpose, no barriers inside).
//
// Hazards (LDS-DMA is ordered for a reader only by the issuing wave's vmcnt wait followed by a barrier the reader passes):
//   RAW  a unit read in phase p+1 is waited for (vmcnt) before the barrier that closes the read interval of phase p, by every wave;
//   WAR  every wave retires its fragment reads (lgkmcnt(0)) before the barrier that closes its read interval; a unit is re-issued
//        at the earliest two intervals later, i.e. after the trailing group has passed that barrier too.
#include "gemm256_dev.h"
#include <stdlib.h>

#define P8_UNIT 16384
#define P8_BUF 65536
#define P8_EPI_OFF 131072
#define P8_LDS (P8_EPI_OFF + 8 * 4096)

#define P8_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define P8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

int g_rgqa_nt8p = -1;    // rgqa_debug_set key 7: 1 = use this kernel for every eligible 256-row launch, 0 = never, -1 = env RGQA_NT8P / default

// one wave's share of an A unit of 8*RW rows: rows RW*w .. RW*w+RW-1, moved by two LDS-DMA instructions of R1 and RW-R1 rows
// (8 + 8, 8 + 4 or 4 + 4; the short ones run with the upper lanes masked off), so every unit costs every wave two vmcnt slots
template <int RW> struct P8Split { static constexpr int R1 = RW >= 12 ? 8 : 4; static constexpr int R2 = RW - R1; };

template <typename OutT, int EPI, int MT>
__global__ __launch_bounds__(T256_THREADS) void gemm_nt8p_kernel(const GemmGroup g) {
    constexpr int MA0 = (MT + 1) / 2, MA1 = MT / 2, TM = 32 * MT;
    constexpr int RW0 = 4 * MA0, RW1 = 4 * MA1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3;
    // DMA: wave w fills pieces w and w + 8 (8 rows x 128 B each) of every unit; source chunk = position ^ swizzle(unit row)
    const int lch = (lane & 7) ^ (((wave & 1) << 2) + (lrow >> 1));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    const int nblk = (int)gridDim.x;

    // ---- issue cursor: the (tile, K-tile) whose units are being requested
    int i_vt = blockIdx.x, i_kt = 0, i_nkt = 0;
    const bf16_t* asrc[2][2];     // [U0 | U3][instruction]
    const bf16_t* wsrc[2][2];     // [U1 | U2][piece]
    auto a_ptr = [&](const GemmProblem& P, int m0, int ur, int rows_per_wr, int sub_off) {      // unit row -> source row, swizzled chunk
        int am = m0 + (ur / rows_per_wr) * (16 * MT) + sub_off + (ur % rows_per_wr); if (am > P.M - 1) am = P.M - 1;   // rows past the edge are never stored
        return reinterpret_cast<const bf16_t*>(P.A) + (size_t)am * P.lda + (((lane & 7) ^ ((ur >> 1) & 7)) << 3);
    };
    auto locate_issue = [&](int vt) {
        const int tile = xcd_remap256(vt, g.total_tiles);
        int p = 0;
#pragma unroll
        for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
            if (i < g.count && tile >= g.p[i].tile_start) p = i;
        const GemmProblem& P = g.p[p];
        const int local = tile - P.tile_start;
        const int m0 = (local / P.tiles_n) * TM, n0 = (local % P.tiles_n) * 256;
        i_nkt = P.K / TK;
        const bf16_t* W = reinterpret_cast<const bf16_t*>(P.B);
        asrc[0][0] = a_ptr(P, m0, RW0 * wave + lrow, 16 * MA0, 0);
        asrc[0][1] = a_ptr(P, m0, RW0 * wave + P8Split<RW0>::R1 + lrow, 16 * MA0, 0);
        asrc[1][0] = a_ptr(P, m0, RW1 * wave + lrow, 16 * MA1, 16 * MA0);
        asrc[1][1] = a_ptr(P, m0, RW1 * wave + P8Split<RW1>::R1 + lrow, 16 * MA1, 16 * MA0);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                int wn_ = n0 + (pc * 2 + (wave >> 2)) * 64 + u * 32 + (wave & 3) * 8 + lrow; if (wn_ > P.N - 1) wn_ = P.N - 1;
                wsrc[u][pc] = W + (size_t)wn_ * P.ldb + lch * 8;
            }
    };
    auto advance = [&]() {      // next K-tile of this block's stream; past the last tile the last K-tile is re-requested (into dead units)
        if (i_kt + 1 < i_nkt) { ++i_kt; return; }
        const int nvt = i_vt + nblk;
        if (nvt < g.total_tiles) { i_vt = nvt; i_kt = 0; locate_issue(nvt); }
    };
    auto issue_a = [&](int u, unsigned dst) {
        if (u == 0) {
            if (P8Split<RW0>::R1 == 8 || lrow < P8Split<RW0>::R1) dma16(asrc[0][0] + i_kt * TK, dst + (RW0 * wave) * 128);
            if (P8Split<RW0>::R2 == 8 || lrow < P8Split<RW0>::R2) dma16(asrc[0][1] + i_kt * TK, dst + (RW0 * wave + P8Split<RW0>::R1) * 128);
        } else {
            if (P8Split<RW1>::R1 == 8 || lrow < P8Split<RW1>::R1) dma16(asrc[1][0] + i_kt * TK, dst + (RW1 * wave) * 128);
            if (P8Split<RW1>::R2 == 8 || lrow < P8Split<RW1>::R2) dma16(asrc[1][1] + i_kt * TK, dst + (RW1 * wave + P8Split<RW1>::R1) * 128);
        }
    };
    auto issue_w = [&](int u, unsigned dst) {
        dma16(wsrc[u][0] + i_kt * TK, dst + wave * 1024);
        dma16(wsrc[u][1] + i_kt * TK, dst + (wave + 8) * 1024);
    };

    // ---- compute cursor
    int c_vt = blockIdx.x, c_pi = 0, c_m0 = 0, c_n0 = 0, c_nkt = 0;
    auto locate_compute = [&](int vt) {
        const int tile = xcd_remap256(vt, g.total_tiles);
        int p = 0;
#pragma unroll
        for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
            if (i < g.count && tile >= g.p[i].tile_start) p = i;
        const int local = tile - g.p[p].tile_start;
        c_pi = p; c_m0 = (local / g.p[p].tiles_n) * TM; c_n0 = (local % g.p[p].tiles_n) * 256; c_nkt = g.p[p].K / TK;
    };

    // fragment addresses inside a K-tile buffer (unit order in LDS: U0, U1, U2, U3)
    const int a_row0 = wr * (16 * MA0) + fr, a_row1 = wr * (16 * MA1) + fr, w_row = wc * 32 + fr;
    auto lda = [&](const unsigned char* buf, int sub, int tm, int ks) {
        return *reinterpret_cast<const bf16x8*>(buf + (sub ? 3 * P8_UNIT : 0) + off256((sub ? a_row1 : a_row0) + tm * 16, ks * 4 + fq));
    };
    auto ldw = [&](const unsigned char* buf, int sub, int tn, int ks) {
        return *reinterpret_cast<const bf16x8*>(buf + (1 + sub) * P8_UNIT + off256(w_row + tn * 16, ks * 4 + fq));
    };

    // ---- prologue: units 0..6 of the stream
    locate_issue(i_vt);
    locate_compute(c_vt);
    issue_a(0, lds0 + 0 * P8_UNIT); issue_w(0, lds0 + 1 * P8_UNIT); issue_w(1, lds0 + 2 * P8_UNIT); issue_a(1, lds0 + 3 * P8_UNIT);
    advance();
    issue_a(0, lds0 + P8_BUF + 0 * P8_UNIT); issue_w(0, lds0 + P8_BUF + 1 * P8_UNIT); issue_w(1, lds0 + P8_BUF + 2 * P8_UNIT);
    P8_WAIT(10);                                    // U0, U1 of K-tile 0 have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // stagger: waves 4-7 run one interval behind waves 0-3

    int vk = 0;                                     // K-tiles consumed so far (buffer = vk & 1)
    for (;;) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < c_nkt; ++kt, ++vk) {
            const int b = vk & 1;
            const unsigned char* buf = lds + b * P8_BUF;
            const unsigned cur = lds0 + b * P8_BUF, oth = lds0 + (b ^ 1) * P8_BUF;
            bf16x8 a0[MA0][2], a1[MA1][2], w0[2][2], w1[2][2];
            // ---------------- phase 1: quadrant (0,0)
            issue_a(1, oth + 3 * P8_UNIT);          // U3 of K-tile vk+1
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) w0[tn][ks] = ldw(buf, 0, tn, ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < MA0; ++tm)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a0[tm][ks] = lda(buf, 0, tm, ks);
            P8_WAIT(10);                            // U2 of this K-tile
            P8_LGKM0();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tm = 0; tm < MA0; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[tn][ks], a0[tm][ks], acc[tm][tn], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- phase 2: quadrant (0,1)
            advance();
            issue_a(0, cur + 0 * P8_UNIT);          // U0 of K-tile vk+2
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) w1[tn][ks] = ldw(buf, 1, tn, ks);
            P8_WAIT(10);                            // U3 of this K-tile
            P8_LGKM0();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tm = 0; tm < MA0; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) acc[tm][2 + tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[tn][ks], a0[tm][ks], acc[tm][2 + tn], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- phase 3: quadrant (1,1)
            issue_w(0, cur + 1 * P8_UNIT);          // U1 of K-tile vk+2
#pragma unroll
            for (int tm = 0; tm < MA1; ++tm)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a1[tm][ks] = lda(buf, 1, tm, ks);
            P8_LGKM0();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tm = 0; tm < MA1; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) acc[MA0 + tm][2 + tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[tn][ks], a1[tm][ks], acc[MA0 + tm][2 + tn], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- phase 4: quadrant (1,0)
            issue_w(1, cur + 2 * P8_UNIT);          // U2 of K-tile vk+2
            P8_WAIT(10);                            // U0, U1 of K-tile vk+1
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tm = 0; tm < MA1; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) acc[MA0 + tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[tn][ks], a1[tm][ks], acc[MA0 + tm][tn], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        // ---- tile done: re-align the two groups, convert + store (no barriers inside), stagger again
        if (wr == 0) __builtin_amdgcn_s_barrier();
        const int nvt = c_vt + nblk;
        const bool more = nvt < g.total_tiles;
        nt256_epilogue<OutT, EPI, MT>(g, g.p[c_pi], lds + P8_EPI_OFF, wave, lane, c_m0, c_n0, wr, wc, acc, []() {});
        if (!more) break;
        c_vt = nvt;
        locate_compute(nvt);
        if (wr == 1) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing may land in LDS after this block has released it
}

bool gemm_nt8p_eligible(const GemmGroup& g) {
    static const int env = []() { const char* e = getenv("RGQA_NT8P"); return e ? atoi(e) : 0; }();
    const int mode = g_rgqa_nt8p >= 0 ? g_rgqa_nt8p : env;
    if (mode <= 0) return false;
    for (int i = 0; i < g.count; ++i)
        if (g.p[i].K < 2 * TK || (g.p[i].K % TK) != 0) return false;
    return true;
}

template <int EPI, int MT>
static int launch8p(GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<bf16_t, EPI, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS));
        attr_set = true;
    }
    gemm_group_finalize(g, 32 * MT, TN);
    g.ablate = 0;
    int ncu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    ncu &= ~7; if (ncu < 8) ncu = 8;
    const int grid = g.total_tiles < ncu ? g.total_tiles : ncu;
    hipLaunchKernelGGL((gemm_nt8p_kernel<bf16_t, EPI, MT>), dim3(grid), dim3(T256_THREADS), P8_LDS, s, g);
    RGQA_LAUNCH_CHECK("gemm_nt8p_kernel");
    return RGQA_OK;
}

template <int EPI>
static int launch8p_mt(GemmGroup& g, int mt, hipStream_t s) {
    switch (mt) {
        case 8: return launch8p<EPI, 8>(g, s);
        case 7: return launch8p<EPI, 7>(g, s);
        case 6: return launch8p<EPI, 6>(g, s);
        default: return launch8p<EPI, 5>(g, s);
    }
}

// same epilogue set as launch256_epi (gemm_mfma256.hip); the caller has checked gemm_nt256_eligible + gemm_nt8p_eligible; mt in 5..8
int launch_gemm_nt8p_bf16(GemmGroup& g, int mt, hipStream_t s) {
    switch (g.p[0].epi) {
        case EPI_BIAS: return launch8p_mt<EPI_BIAS>(g, mt, s);
        case EPI_GELU: return launch8p_mt<EPI_GELU>(g, mt, s);
        case EPI_RESID_DROP: return launch8p_mt<EPI_RESID_DROP>(g, mt, s);
        case EPI_DGELU: return launch8p_mt<EPI_DGELU>(g, mt, s);
        case EPI_TANH: return launch8p_mt<EPI_TANH>(g, mt, s);
        case EPI_DTANH: return launch8p_mt<EPI_DTANH>(g, mt, s);
        case EPI_RELU: return launch8p_mt<EPI_RELU>(g, mt, s);
        case EPI_RELU_DROP: return launch8p_mt<EPI_RELU_DROP>(g, mt, s);
        case EPI_DRELU_DROP: return launch8p_mt<EPI_DRELU_DROP>(g, mt, s);
        default: return launch8p_mt<EPI_ADD>(g, mt, s);
    }
}
