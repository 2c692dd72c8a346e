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

#define P8_WAIT_(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define P8_WAIT(n) P8_WAIT_(n)
#define P8_WAITV() do { if (P8_SPLIT) P8_WAIT_(9); else P8_WAIT(P8_INFLIGHT); } while (0)      /* split issue: the current unit has one instruction out */
// lab-only ablations of the K loop (results are garbage): P8_ABLATE bit 0 = no LDS-DMA in the loop, bit 1 = no fragment reads,
// bit 2 = no MFMAs, bit 3 = the W units fetched by plain global_load_dwordx4 into registers instead of LDS-DMA.  tools/lab/stamp_lab builds one binary per value.
#ifndef P8_ABLATE
#define P8_ABLATE 0
#endif
#if P8_ABLATE & 4
__device__ __forceinline__ f32x4 p8_no_mfma(bf16x8 a, bf16x8 b, f32x4 c) { asm volatile("" :: "v"(a), "v"(b)); return c; }
#define P8_MFMA(a, b, c, x, y, z) p8_no_mfma(a, b, c)
#else
#define P8_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#endif
#ifndef P8_DEFAULT_VARIANT
#define P8_DEFAULT_VARIANT 0
#endif
#ifndef P8_INFLIGHT      // LDS-DMA instructions a wave may leave in flight at a wait: 10 = 5 units (the schedule's maximum); lab builds lower it
#define P8_INFLIGHT 10
#endif
#define P8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// diagnostic build only (tools/lab/stamp_lab.cpp defines P8_STAMPS and g_p8_stamps): wave 0 of every block stamps the shader
// clock (s_memtime) and the 100 MHz wall clock (s_memrealtime) at the phase boundaries of its tiles; nothing reads the stamps
#ifdef P8_STAMPS
__device__ unsigned long long* g_p8_stamps;     // [blocks][32][2]
#define P8_STAMP() do { if (wave == 0 && n_stamp < 32) { unsigned long long* q_ = g_p8_stamps + ((size_t)blockIdx.x * 32 + n_stamp) * 2; \
        q_[0] = __builtin_amdgcn_s_memtime(); q_[1] = __builtin_amdgcn_s_memrealtime(); } ++n_stamp; } while (0)
// interval accounting (waves 0 and 4): cycles from interval start to "my part is done" (work) and from there to the barrier's release (wait)
#define P8_T() __builtin_amdgcn_s_memtime()
#ifdef P8_STAMPS_FINE   // 16 s_memtime per K-tile: slows the loop by ~30 %, read the RATIOS only
#define P8_ACC(slot) do { const unsigned long long n_ = P8_T(); iv[slot] += n_ - t_prev; t_prev = n_; } while (0)
#else
#define P8_ACC(slot) do { } while (0)
#endif
#else
#define P8_STAMP() do { } while (0)
#define P8_ACC(slot) do { } while (0)
#endif

int g_rgqa_nt8p = -1;    // rgqa_debug_set key 7: 2 = every eligible launch of 160..256-row tiles, 1 = 192-row tiles only, 0 = never, -1 = env RGQA_NT8P (default 1); 3 = as 2 with the balanced read schedule (A/B builds)

// one wave's share of an A unit of 8*RW rows: rows RW*w .. RW*w+RW-1, moved by two LDS-DMA instructions of R1 and RW-R1 rows
// (8 + 8, 8 + 4 or 4 + 4; the short ones run with the upper lanes masked off), so every unit costs every wave two vmcnt slots
template <int RW> struct P8Split { static constexpr int R1 = RW >= 12 ? 8 : 4; static constexpr int R2 = RW - R1; };

template <typename OutT, int EPI, int MT, int VARIANT>
__global__ __launch_bounds__(T256_THREADS) void gemm_nt8p_kernel(const GemmGroup g) {
    // VARIANT 0: both LDS-DMA instructions of a phase's unit go out in its read interval; 1: as 0 with the fragment reads balanced
    // over the phases; 2: one in the read interval, one between the two halves of the phase's MFMAs
    constexpr bool P8_BALANCED = VARIANT == 1, P8_SPLIT = VARIANT == 2;
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
#ifdef P8_STAMPS
    int n_stamp = 0;
    unsigned long long iv[4] = {0, 0, 0, 0}, t_prev = 0;      // read work, read-barrier wait, MFMA issue, MFMA-barrier wait
#endif
    P8_STAMP();                                     // 0: entry

    // ---- issue cursor: the (tile, K-tile) whose units are being requested.  Sources = wave-uniform base of the tile's operand
    // panel at the cursor's K-tile (SGPRs, advanced by 128 B per K-tile) + per-lane byte offsets (row clamped to the matrix, swizzled chunk)
    int i_vt = blockIdx.x, i_kt = 0, i_nkt = 0;
    unsigned aoff[2][2];          // [U0 | U3][instruction]
    unsigned woff[2][2];          // [U1 | U2][piece]
    const unsigned char* sA = nullptr;
    const unsigned char* sW = nullptr;
    auto a_off = [&](const GemmProblem& P, int m0, int ur, int rows_per_wr, int sub_off) {      // unit row -> tile row (clamped), swizzled chunk
        int r = (ur / rows_per_wr) * (16 * MT) + sub_off + (ur % rows_per_wr); if (r > P.M - 1 - m0) r = P.M - 1 - m0;   // rows past the edge are never stored
        return (unsigned)(r * P.lda * 2 + (((lane & 7) ^ ((ur >> 1) & 7)) << 4));
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
        sA = reinterpret_cast<const unsigned char*>(P.A) + (size_t)m0 * P.lda * 2;
        sW = reinterpret_cast<const unsigned char*>(P.B) + (size_t)n0 * P.ldb * 2;
        aoff[0][0] = a_off(P, m0, RW0 * wave + lrow, 16 * MA0, 0);
        aoff[0][1] = a_off(P, m0, RW0 * wave + P8Split<RW0>::R1 + lrow, 16 * MA0, 0);
        aoff[1][0] = a_off(P, m0, RW1 * wave + lrow, 16 * MA1, 16 * MA0);
        aoff[1][1] = a_off(P, m0, RW1 * wave + P8Split<RW1>::R1 + lrow, 16 * MA1, 16 * MA0);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                int r = (pc * 2 + (wave >> 2)) * 64 + u * 32 + (wave & 3) * 8 + lrow; if (r > P.N - 1 - n0) r = P.N - 1 - n0;
                woff[u][pc] = (unsigned)(r * P.ldb * 2 + lch * 16);
            }
    };
    auto advance = [&]() {      // next K-tile of this block's stream; past the last tile the last K-tile is re-requested (into dead units)
        if (i_kt + 1 < i_nkt) { ++i_kt; sA += TK * 2; sW += TK * 2; return; }
        const int nvt = i_vt + nblk;
        if (nvt < g.total_tiles) { i_vt = nvt; i_kt = 0; locate_issue(nvt); }
    };
    // which: 0 = the unit's first instruction, 1 = its second, 2 = both
    bool in_loop = false;
    auto issue_a = [&](int u, unsigned dst, int which) {
        if ((P8_ABLATE & 1) && in_loop) return;
        if (u == 0) {
            if (which != 1 && (P8Split<RW0>::R1 == 8 || lrow < P8Split<RW0>::R1)) dma16o(aoff[0][0], sA, dst + (RW0 * wave) * 128);
            if (which != 0 && (P8Split<RW0>::R2 == 8 || lrow < P8Split<RW0>::R2)) dma16o(aoff[0][1], sA, dst + (RW0 * wave + P8Split<RW0>::R1) * 128);
        } else {
            if (which != 1 && (P8Split<RW1>::R1 == 8 || lrow < P8Split<RW1>::R1)) dma16o(aoff[1][0], sA, dst + (RW1 * wave) * 128);
            if (which != 0 && (P8Split<RW1>::R2 == 8 || lrow < P8Split<RW1>::R2)) dma16o(aoff[1][1], sA, dst + (RW1 * wave + P8Split<RW1>::R1) * 128);
        }
    };
    auto issue_w = [&](int u, unsigned dst, int which) {
        if ((P8_ABLATE & 1) && in_loop) return;
        if ((P8_ABLATE & 8) && in_loop) {      // lab: the W half of the operand traffic as plain vector loads into (discarded) registers
            uint4 t0, t1;
            if (which != 1) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t0) : "v"(woff[u][0]), "s"(sW) : "memory");
            if (which != 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t1) : "v"(woff[u][1]), "s"(sW) : "memory");
            return;
        }
        if (which != 1) dma16o(woff[u][0], sW, dst + wave * 1024);
        if (which != 0) dma16o(woff[u][1], sW, dst + (wave + 8) * 1024);
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
        if (P8_ABLATE & 2) { bf16x8 z; for (int j = 0; j < 8; ++j) z[j] = (bf16_t)(float)(lane + tm + ks + sub); asm volatile("" : "+v"(z)); return z; }
        return *reinterpret_cast<const bf16x8*>(buf + (sub ? 3 * P8_UNIT : 0) + off256((sub ? a_row1 : a_row0) + tm * 16, ks * 4 + fq));
    };
    auto ldw = [&](const unsigned char* buf, int sub, int tn, int ks) {
        if (P8_ABLATE & 2) { bf16x8 z; for (int j = 0; j < 8; ++j) z[j] = (bf16_t)(float)(lane + tn + ks + sub); asm volatile("" : "+v"(z)); return z; }
        return *reinterpret_cast<const bf16x8*>(buf + (1 + sub) * P8_UNIT + off256(w_row + tn * 16, ks * 4 + fq));
    };

    // ---- prologue: units 0..6 of the stream
    locate_issue(i_vt);
    locate_compute(c_vt);
    issue_a(0, lds0 + 0 * P8_UNIT, 2); issue_w(0, lds0 + 1 * P8_UNIT, 2); issue_w(1, lds0 + 2 * P8_UNIT, 2); issue_a(1, lds0 + 3 * P8_UNIT, 2);
    advance();
    issue_a(0, lds0 + P8_BUF + 0 * P8_UNIT, 2); issue_w(0, lds0 + P8_BUF + 1 * P8_UNIT, 2); issue_w(1, lds0 + P8_BUF + 2 * P8_UNIT, 2);
    P8_WAITV();                                    // U0, U1 of K-tile 0 have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    P8_STAMP();                                     // 1: first units landed
    if (wr == 1) __builtin_amdgcn_s_barrier();      // stagger: waves 4-7 run one interval behind waves 0-3

    in_loop = true;
    int vk = 0;                                     // K-tiles consumed so far (buffer = vk & 1)
    for (;;) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        bf16x8 a0[MA0][2];                         // P8_BALANCED: carried across K-tiles (read one phase early)
#ifdef P8_STAMPS
        t_prev = P8_T();
#endif
        for (int kt = 0; kt < c_nkt; ++kt, ++vk) {
            const int b = vk & 1;
            const unsigned char* buf = lds + b * P8_BUF;
            const unsigned cur = lds0 + b * P8_BUF, oth = lds0 + (b ^ 1) * P8_BUF;
            bf16x8 a1[MA1][2], w0[2][2], w1[2][2];
            // ---------------- phase 1: quadrant (0,0)
            issue_a(1, oth + 3 * P8_UNIT, P8_SPLIT ? 0 : 2);          // U3 of K-tile vk+1
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) w0[tn][ks] = ldw(buf, 0, tn, ks);
            __builtin_amdgcn_sched_barrier(0);
            if (!P8_BALANCED || kt == 0) {          // balanced: only a tile's first K-tile reads its U0 here, the others did in phase 4
#pragma unroll
                for (int tm = 0; tm < MA0; ++tm)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) a0[tm][ks] = lda(buf, 0, tm, ks);
            }
            P8_WAITV();                            // U2 of this K-tile
            P8_LGKM0();
            __builtin_amdgcn_sched_barrier(0);
            P8_ACC(0);
            __builtin_amdgcn_s_barrier();
            P8_ACC(1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (P8_SPLIT && ks == 1) { __builtin_amdgcn_sched_barrier(0); issue_a(1, oth + 3 * P8_UNIT, 1); __builtin_amdgcn_sched_barrier(0); }      // the unit's second half, behind 8 MFMAs
#pragma unroll
                for (int tm = 0; tm < MA0; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = P8_MFMA(w0[tn][ks], a0[tm][ks], acc[tm][tn], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            P8_ACC(2);
            __builtin_amdgcn_s_barrier();
            P8_ACC(3);
            // ---------------- phase 2: quadrant (0,1)
            advance();
            issue_a(0, cur + 0 * P8_UNIT, P8_SPLIT ? 0 : 2);          // U0 of K-tile vk+2
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) w1[tn][ks] = ldw(buf, 1, tn, ks);
            P8_WAITV();                            // U3 of this K-tile
            P8_LGKM0();
            __builtin_amdgcn_sched_barrier(0);
            P8_ACC(0);
            __builtin_amdgcn_s_barrier();
            P8_ACC(1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (P8_SPLIT && ks == 1) { __builtin_amdgcn_sched_barrier(0); issue_a(0, cur + 0 * P8_UNIT, 1); __builtin_amdgcn_sched_barrier(0); }      // the unit's second half, behind 8 MFMAs
#pragma unroll
                for (int tm = 0; tm < MA0; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) acc[tm][2 + tn] = P8_MFMA(w1[tn][ks], a0[tm][ks], acc[tm][2 + tn], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            P8_ACC(2);
            __builtin_amdgcn_s_barrier();
            P8_ACC(3);
            // ---------------- phase 3: quadrant (1,1)
            issue_w(0, cur + 1 * P8_UNIT, P8_SPLIT ? 0 : 2);          // U1 of K-tile vk+2
#pragma unroll
            for (int tm = 0; tm < MA1; ++tm)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a1[tm][ks] = lda(buf, 1, tm, ks);
            if (P8_BALANCED) P8_WAITV();           // U0 of K-tile vk+1 (read in phase 4)
            P8_LGKM0();
            __builtin_amdgcn_sched_barrier(0);
            P8_ACC(0);
            __builtin_amdgcn_s_barrier();
            P8_ACC(1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (P8_SPLIT && ks == 1) { __builtin_amdgcn_sched_barrier(0); issue_w(0, cur + 1 * P8_UNIT, 1); __builtin_amdgcn_sched_barrier(0); }      // the unit's second half, behind 8 MFMAs
#pragma unroll
                for (int tm = 0; tm < MA1; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) acc[MA0 + tm][2 + tn] = P8_MFMA(w1[tn][ks], a1[tm][ks], acc[MA0 + tm][2 + tn], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            P8_ACC(2);
            __builtin_amdgcn_s_barrier();
            P8_ACC(3);
            // ---------------- phase 4: quadrant (1,0)
            issue_w(1, cur + 2 * P8_UNIT, P8_SPLIT ? 0 : 2);          // U2 of K-tile vk+2
            P8_WAITV();                            // U0, U1 of K-tile vk+1 (balanced: U1; U0 was waited for in phase 3)
            // P8_BALANCED spreads the fragment reads 4 / 4 / 8 / 8 over the phases instead of 12 / 4 / 8 / 0.  Measured (tools/lab, A/B in one
            // process): no gain, -1..+7 % - the read intervals are long because of the two LDS-DMA issues (~100-180 cycles each
            // beside fragment reads), not because of phase 1's read burst; kept for A/B builds only (-DP8_AB_BUILD).
            if (P8_BALANCED) {
                if (kt + 1 < c_nkt) {
                    const unsigned char* nbuf = lds + (b ^ 1) * P8_BUF;
#pragma unroll
                    for (int tm = 0; tm < MA0; ++tm)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) a0[tm][ks] = lda(nbuf, 0, tm, ks);
                }
                P8_LGKM0();
            }
            __builtin_amdgcn_sched_barrier(0);
            P8_ACC(0);
            __builtin_amdgcn_s_barrier();
            P8_ACC(1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (P8_SPLIT && ks == 1) { __builtin_amdgcn_sched_barrier(0); issue_w(1, cur + 2 * P8_UNIT, 1); __builtin_amdgcn_sched_barrier(0); }      // the unit's second half, behind 8 MFMAs
#pragma unroll
                for (int tm = 0; tm < MA1; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) acc[MA0 + tm][tn] = P8_MFMA(w0[tn][ks], a1[tm][ks], acc[MA0 + tm][tn], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            P8_ACC(2);
            __builtin_amdgcn_s_barrier();
            P8_ACC(3);
        }
        // ---- tile done: re-align the two groups, convert + store (no barriers inside), stagger again
        P8_STAMP();                                 // 2 + 3i: K loop of tile i done (wave 0)
        if (wr == 0) __builtin_amdgcn_s_barrier();
        P8_STAMP();                                 // 3 + 3i: groups re-aligned
        const int nvt = c_vt + nblk;
        const bool more = nvt < g.total_tiles;
        nt256_epilogue<OutT, EPI, MT>(g, g.p[c_pi], lds + P8_EPI_OFF, wave, lane, c_m0, c_n0, wr, wc, acc, []() {});
        P8_STAMP();                                 // 4 + 3i: epilogue issued (stores may still be in flight)
        if (!more) break;
        c_vt = nvt;
        locate_compute(nvt);
        if (wr == 1) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing may land in LDS after this block has released it
    P8_STAMP();                                     // last: everything drained
#ifdef P8_STAMPS
    if ((wave == 0 || wave == 4) && lane == 0) {
        unsigned long long* q = g_p8_stamps + 256 * 32 * 2 + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 4;
        q[0] = iv[0]; q[1] = iv[1]; q[2] = iv[2]; q[3] = iv[3];
    }
#endif
}

bool gemm_nt8p_eligible(const GemmGroup& g) {
    static const int env = []() { const char* e = getenv("RGQA_NT8P"); return e ? atoi(e) : 1; }();     // default: mode 1
    if (g_rgqa_nt8p < 0) g_rgqa_nt8p = env;
    if (g_rgqa_nt8p <= 0) return false;
    for (int i = 0; i < g.count; ++i)
        if (g.p[i].K < 2 * TK || (g.p[i].K % TK) != 0) return false;
    return true;
}

template <int EPI, int MT, int BAL>
static int launch8p(GemmGroup& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt8p_kernel<bf16_t, EPI, MT, BAL>), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS));
        attr_set = true;
    }
    gemm_group_finalize(g, 32 * MT, TN);
    g.ablate = 0;
    int ncu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    ncu &= ~7; if (ncu < 8) ncu = 8;
    const int grid = g.total_tiles < ncu ? g.total_tiles : ncu;
    hipLaunchKernelGGL((gemm_nt8p_kernel<bf16_t, EPI, MT, BAL>), dim3(grid), dim3(T256_THREADS), P8_LDS, s, g);
    RGQA_LAUNCH_CHECK("gemm_nt8p_kernel");
    return RGQA_OK;
}

template <int EPI>
static int launch8p_mt(GemmGroup& g, int mt, hipStream_t s) {
#ifdef P8_AB_BUILD      // lab builds carry every schedule variant: g_rgqa_nt8p == 3 balanced reads, 4 = split LDS-DMA issue
    if (g_rgqa_nt8p == 3 || g_rgqa_nt8p == 4) {
        const bool bal = g_rgqa_nt8p == 3;
        switch (mt) {
            case 8: return bal ? launch8p<EPI, 8, 1>(g, s) : launch8p<EPI, 8, 2>(g, s);
            case 7: return bal ? launch8p<EPI, 7, 1>(g, s) : launch8p<EPI, 7, 2>(g, s);
            case 6: return bal ? launch8p<EPI, 6, 1>(g, s) : launch8p<EPI, 6, 2>(g, s);
            default: return bal ? launch8p<EPI, 5, 1>(g, s) : launch8p<EPI, 5, 2>(g, s);
        }
    }
#endif
    switch (mt) {
        case 8: return launch8p<EPI, 8, P8_DEFAULT_VARIANT>(g, s);
        case 7: return launch8p<EPI, 7, P8_DEFAULT_VARIANT>(g, s);
        case 6: return launch8p<EPI, 6, P8_DEFAULT_VARIANT>(g, s);
        default: return launch8p<EPI, 5, P8_DEFAULT_VARIANT>(g, s);
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
