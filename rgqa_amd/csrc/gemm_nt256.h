// 256-wide LDS-DMA MFMA grouped GEMM kernels (NT) for gfx950: the throughput kernels of the encoder's projections, in two
// operand precisions (template parameter X3):
//
//   X3 = false   C[M,N] = A[M,K] * W[N,K]^T (+ epilogue), A / W bf16, K % 64 == 0.
//   X3 = true    the same product on split-f32 operands (common.h `sf32`: every element a bf16 pair hi + lo, 32 hi | 32 lo per
//                128-byte line), K % 32 == 0: the operand tiles are staged EXACTLY as bf16 tiles of 2K columns - a 64-column K-step
//                then holds the hi parts (columns 0..31) and the lo parts (32..63) of 32 contraction elements - and each pair of
//                fragments feeds three MFMAs, a_hi*w_lo + a_lo*w_hi + a_hi*w_hi, into one f32 accumulator.  Per byte staged that
//                is 1.5x the matrix work of the bf16 loop, which is bound by operand delivery (~46 GB/s per CU).
//
// One workgroup of 8 waves (2 along M x 4 along N, 16*MT x 64 outputs per wave) per CU, 128 KiB of LDS = 2 stages x
// (A 32*MT x 64 + W 256 x 64) bf16.  Operand tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no staging registers);
// each wave-instruction fills 1 KiB = 8 rows x 128 B, and the XOR swizzle that keeps ds_read_b128 conflict-free is applied on
// the per-lane SOURCE address (the LDS destination of an LDS-DMA is lane-linear).  The next K-step's 64 KiB are in flight while
// the current one is consumed.
#pragma once
#include "gemm256_dev.h"

// One K-step of MFMAs for a wave's (16*MT) x 64 slice.  The A fragments come through a register ring PD deep - the LDS read for
// fragment i+PD is issued right after the MFMAs of fragment i, pinned by sched_barrier - instead of being requested two at a time just
// before their use, which is what the compiler makes of the plain loop (every group of 8 MFMAs then starts behind a full LDS round trip
// that only the SIMD's other wave can cover).  Measured in situ (B=256 train step): ring 3 -0.18 ms; depth 4 = depth 3; depth 2 within
// noise of 3; iglp_opt on the plain loop +1..3 % against the ring.
// tile id inside a problem -> tile coordinates.  P.tiles_n carries the number of N-tiles in its low 16 bits and the PANEL width in its high 16
// (set by the NT launchers, nt_set_panels): tiles are numbered panel by panel - a panel = `pw` adjacent N-tiles over all M-tiles, N fastest
// inside it - so that the contiguous run of tile ids an XCD walks (xcd_remap256) stays inside one panel over its rounds: the panel's weight
// rows (pw x 256 x K elements, sized to fit the XCD's 4-MB L2 beside the streamed A rows) are fetched once instead of once per round, and
// a round's 32 tiles still share their A rows pw ways.  pw == tiles_n is the plain row-major numbering.
__device__ __forceinline__ void nt_tile_coords(const GemmProblem& P, int local, int tm_rows, int& m0, int& n0) {
    const int tn_ = P.tiles_n & 0xFFFF, pw = P.tiles_n >> 16;
    if (pw <= 0 || pw >= tn_) { m0 = (local / tn_) * tm_rows; n0 = (local % tn_) * TN; return; }
    const int tiles_m = cdiv(P.M, tm_rows), full = tn_ / pw;          // full panels
    int pnl = local / (tiles_m * pw);
    if (pnl > full) pnl = full;                                          // (cannot exceed: the last, narrower panel is index `full`)
    const int r = local - pnl * tiles_m * pw;
    const int w = pnl < full ? pw : tn_ - full * pw;                    // width of this panel
    m0 = (r / w) * tm_rows; n0 = (pnl * pw + r % w) * TN;
}

#define RGQA_NT_PIPE 3
// NN (bf16 only): the W stage is a [64 contraction rows][256 columns] image - the weight as it is stored, [out, in], serves the dgrad GEMMs without a
// transposed copy (round 5: the copy cost 0.16 ms per train step to re-make and 410 MB) - and its fragments come through ds_read_b64_tr_b16 like the
// weight-gradient kernel's operands (tr_frag_dma, 32-byte granule swizzle)
template <int MT, bool X3, bool NN = false>
__device__ __forceinline__ void nt256_kstep(const unsigned char* a, const unsigned char* w, int wm, int wn, int fr, int fq, f32x4 (&acc)[MT][4]) {
    static_assert(!(X3 && NN), "the [K, N] operand form exists for bf16 only");
    auto ldw = [&](int half, int t) -> bf16x8 {
        if constexpr (NN) return tr_frag_dma<TN * 2>(w, half * 32, wn * 64 + t * 16, fq * 16 + fr);
        else return *reinterpret_cast<const bf16x8*>(w + off256(wn * 64 + t * 16 + fr, half * 4 + fq));
    };
    constexpr int PD = RGQA_NT_PIPE, NF = 2 * MT;
    if constexpr (X3) {
        // fragment order hi(0), lo(0), hi(1), lo(1), ...: half 0 of the stage = hi parts, half 1 = lo parts of the same 32 contraction elements
        auto lda = [&](int i) { return *reinterpret_cast<const bf16x8*>(a + off256(wm * (16 * MT) + (i >> 1) * 16 + fr, (i & 1) * 4 + fq)); };
        bf16x8 wh[4], wl[4], ring[PD];
#pragma unroll
        for (int t = 0; t < 4; ++t) wh[t] = ldw(0, t);
#pragma unroll
        for (int i = 0; i < PD; ++i) ring[i] = lda(i);
#pragma unroll
        for (int t = 0; t < 4; ++t) wl[t] = ldw(1, t);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int tm = i >> 1;
            const bf16x8 xa = ring[i % PD];
            if ((i & 1) == 0) {         // a_hi: against w_lo (small term first), then w_hi
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[tn], xa, acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tn], xa, acc[tm][tn], 0, 0, 0);
            } else {                    // a_lo: against w_hi only (lo * lo is below 2^-16 of the product)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tn], xa, acc[tm][tn], 0, 0, 0);
            }
            if (i + PD < NF) ring[i % PD] = lda(i + PD);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        auto lda = [&](int i) { return *reinterpret_cast<const bf16x8*>(a + off256(wm * (16 * MT) + (i % MT) * 16 + fr, (i / MT) * 4 + fq)); };
        bf16x8 xw[2][4], ring[PD];
#pragma unroll
        for (int t = 0; t < 4; ++t) xw[0][t] = ldw(0, t);
#pragma unroll
        for (int i = 0; i < PD; ++i) ring[i] = lda(i);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int s = i / MT, tm = i % MT;
            const bf16x8 xa = ring[i % PD];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xw[s][tn], xa, acc[tm][tn], 0, 0, 0);
            if (i + PD < NF) ring[i % PD] = lda(i + PD);
            if (s == 0 && tm == MT - 1 - (MT > 2 ? 2 : 0)) {          // second half's W fragments, two fragments of lead
#pragma unroll
                for (int t = 0; t < 4; ++t) xw[1][t] = ldw(1, t);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// EPI is a compile-time constant: the generic (run-time switched) epilogue inlined 32x stops the compiler from
// unrolling the accumulator loops and pushes the 128 accumulators into scratch.
// Persistent for MT >= 4: one block per CU walks tiles blockIdx, + grid, ... (XCD-aware order); the next tile's first two K-steps are
// DMA'd behind the epilogue.
// STAMP (the probe's own instantiations, never a product launch): thread 0 of every block records into g.stamps[8 * block + ..]
// 0/1 s_memtime / s_memrealtime at entry, 2/3 at exit (shader cycles per 100-MHz tick over the block's life = the clock the chip holds
// under this loop), 4 s_memrealtime when the first tile's first operands have landed, 5 at the end of its K loop, 6 after its epilogue
// (stores issued, not drained), 7 the number of tiles the block walked.
// LNF (bf16, EPI_RESID_DROP): result tiles stored write-through and the row block's LayerNorm done by the last arriver (gemm256_dev.h nt256_ln_after_tile)
template <typename OutT, int EPI, int MT, bool X3, bool STAMP = false, bool NN = false, bool LNF = false>
__global__ __launch_bounds__(T256_THREADS) void gemm_nt256_kernel(const GemmGroupNT g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned long long st_c0 = 0, st_r0 = 0, st_r1 = 0, st_r2 = 0, st_r3 = 0, st_nt = 0;
    if (STAMP && threadIdx.x == 0) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr bool PERSIST = NT256_PERSIST(MT);
    constexpr int TM = 32 * MT, A_BYTES = TM * TK * 2, STAGE_BYTES = A_BYTES + TN * TK * 2, AG = (MT + 1) / 2, NAG = 4 * MT;
    constexpr int EPI_OFF = PERSIST ? 2 * STAGE_BYTES : 0;
    constexpr int KV = X3 ? 2 : 1;                       // bf16 columns per contraction element as the tiles are staged
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // LDS-DMA: per stage wave w fills the A row groups (8 rows each) w, w+8, .. below NAG = 4*MT (odd MT: waves 4-7 fill one
    // group fewer) and W row groups w*4 .. w*4+3
    const int lrow = lane >> 3;
    // source chunk for this lane's linear LDS slot in a piece of 8 rows: (lane&7) ^ ((row>>1)&7), row = piece*8 + lrow
    const int lch_a = (lane & 7) ^ (((wave & 1) << 2) + (lrow >> 1));
    const int lch_w[2] = {(lane & 7) ^ (lrow >> 1), (lane & 7) ^ (4 + (lrow >> 1))};
    const bf16_t* asrc[AG];
    const bf16_t* wsrc[4];
    size_t wstep = TK;      // W source advance per K-step: TK elements along a row (NT) or TK rows (NN)
    int pi = 0, m0 = 0, n0 = 0, nkt = 0;
    // tile id -> problem, tile origin and this lane's DMA source rows
    auto locate = [&](int vt) {
        const int tile = xcd_remap256(vt, g.total_tiles);
        int p = 0;
#pragma unroll
        for (int i = 1; i < GEMM_NT_MAX_PROBLEMS; ++i)
            if (i < g.count && tile >= g.p[i].tile_start) p = i;
        const GemmProblem& P = g.p[p];
        const int local = tile - P.tile_start;
        pi = p; nt_tile_coords(P, local, TM, m0, n0); nkt = P.K * KV / TK;
        const bf16_t* A = reinterpret_cast<const bf16_t*>(P.A);
        const bf16_t* W = reinterpret_cast<const bf16_t*>(P.B);
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            int am = m0 + (i * 8 + wave) * 8 + lrow; if (am > P.M - 1) am = P.M - 1;   // clamp: rows past the edge are never stored
            asrc[i] = A + (size_t)am * P.lda * KV + lch_a * 8;
        }
        if constexpr (NN) {
            wstep = (size_t)TK * P.ldb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {      // piece = 2 contraction rows x 256 columns; source columns un-swizzled per lane, clamped in-bounds (columns past N are never stored)
                const int row = (wave * 4 + i) * 2 + (lane >> 5);
                int col = n0 + (((lane & 31) ^ (tn_f(row) << 1)) << 3);
                if (col > P.ldb - 8) col = P.ldb - 8;
                wsrc[i] = W + (size_t)row * P.ldb + col;
            }
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int wn_ = n0 + (wave * 4 + i) * 8 + lrow; if (wn_ > P.N - 1) wn_ = P.N - 1;
            wsrc[i] = W + (size_t)wn_ * P.ldb * KV + lch_w[i & 1] * 8;
        }
        }
    };
    // LDS ring depth 2. Measured alternatives on these shapes (round 1): 3 stages for MT <= 4 lost 10..20 % (MT2 loses its
    // 2 blocks/CU, K is only 12 steps); a K-step-32 / 4-slot ring lost 15..33 % (64-B DMA rows halve the useful bytes per L2 line
    // request and the loop is close to delivery-bound); a ping-pong schedule (waves 4-7 one slot behind waves 0-3) WON the L2-hot
    // micro-benchmark by 1..9 % and LOST 4..8 % per launch inside the train step: there the operands come from MALL/HBM and a faster
    // compute slot only shortens the DMA's lead.  Round 2 (profiles/r02_nt8p_*): a phase-interleaved 80-KiB-in-flight variant tied
    // this loop in situ and was removed; the loop delivers a 64-KiB K-tile to a CU every ~1.5 us whatever the schedule.
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    auto issue = [&](int stage, int kt) {
        const unsigned base = lds0 + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < AG; ++i)
            if ((MT & 1) == 0 || i * 8 + wave < NAG) dma16(asrc[i] + kt * TK, base + (i * 8 + wave) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(wsrc[i] + kt * wstep, base + A_BYTES + (wave * 4 + i) * 1024);
    };

    const int fr = lane & 15, fq = lane >> 4;
    int vt = blockIdx.x;
    locate(vt);
    if (PERSIST && g.stagger > 0) {
        // All blocks of a launch move in lock-step: K loop (matrix pipe busy, HBM idle: the operands come out of the L2s), then the epilogue (every CU stores
        // its tile at once: HBM-bound, matrix pipe idle).  In a launch of 2.67 tile rounds a third of the blocks walk one tile fewer: they can afford to
        // start late, and then their K loops run beside the others' stores
        const int total = g.total_tiles, grid = (int)gridDim.x;
        const int mine = (total - (int)blockIdx.x + grid - 1) / grid, mx = (total + grid - 1) / grid;
        if (mine < mx) {
            const int loops = (nkt * g.stagger) >> 4;
            for (int i = 0; i < loops; ++i) __builtin_amdgcn_s_sleep(64);          // 64 x 64 clocks ~ 2 us
        }
    }
    issue(0, 0);
    bool pre1 = false;      // K-step 1 of the current tile was already issued (behind the previous tile's epilogue)
    for (;;) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nkt; ++kt) {
            const int st = kt & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (STAMP && threadIdx.x == 0 && kt == 0 && st_nt == 0) st_r1 = __builtin_amdgcn_s_memrealtime();
            if (kt + 1 < nkt && !(pre1 && kt == 0)) issue(st ^ 1, kt + 1);
            const unsigned char* a = lds + st * STAGE_BYTES;
            const unsigned char* w = a + A_BYTES;
            nt256_kstep<MT, X3, NN>(a, w, wm, wn, fr, fq, acc);
        }
        __syncthreads();   // every wave is done with the operand stages: they may be refilled (PERSIST) or reused as scratch
        if (STAMP && threadIdx.x == 0 && st_nt == 0) st_r2 = __builtin_amdgcn_s_memrealtime();
        const int cpi = pi, cm0 = m0, cn0 = n0;
        const int nvt = vt + (int)gridDim.x;
        const bool more = PERSIST && nvt < g.total_tiles;
        pre1 = false;
        nt256_epilogue<OutT, EPI, MT, LNF>(g, g.p[cpi], lds + EPI_OFF, wave, lane, cm0, cn0, wm, wn, acc, [&]() {
            if (more) {
                locate(nvt);
                issue(0, 0);
                pre1 = nkt > 1;
                if (pre1) issue(1, 1);
            }
        });
        if constexpr (LNF) nt256_ln_after_tile<TM>(g.p[cpi], cm0, reinterpret_cast<int*>(lds + EPI_OFF));
        if (STAMP && threadIdx.x == 0) { if (st_nt == 0) st_r3 = __builtin_amdgcn_s_memrealtime(); ++st_nt; }
        if (!more) break;
        vt = nvt;
    }
    if (STAMP && threadIdx.x == 0) {
        unsigned long long* o = g.stamps + (size_t)blockIdx.x * 8;
        o[0] = st_c0; o[1] = st_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = st_r1; o[5] = st_r2; o[6] = st_r3; o[7] = st_nt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// "Deep ring" variant for launches of 64 / 128 / 160-row tiles (the language-only stages, the N = 768 projections, the head, the
// BUTD GRU steps).  Such a launch cannot hide the operand-DMA latency behind other tiles: with one K-step in flight every step
// costs a full L2/MALL round trip (~1 us) whatever the MFMA work.  One tile per block, so the whole 160 KiB of LDS can hold
// the ring: NS = 4 slots for MT = 2 (40 KiB each) - three K-steps in flight under counted vmcnt waits; the epilogue scratch
// aliases the ring once the last step has been consumed.
template <typename OutT, int EPI, int MT, int NS, bool X3, bool STAMP = false, bool NN = false, bool LNF = false>
__global__ __launch_bounds__(T256_THREADS) void gemm_nt256d_kernel(const GemmGroupNT g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned long long st_c0 = 0, st_r0 = 0, st_r1 = 0, st_r2 = 0;
    if (STAMP && threadIdx.x == 0) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int TM = 32 * MT, A_BYTES = TM * TK * 2, STAGE_BYTES = A_BYTES + TN * TK * 2, AG = (MT + 1) / 2, NAG = 4 * MT;
    constexpr int KV = X3 ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lrow = lane >> 3;
    const int lch_a = (lane & 7) ^ (((wave & 1) << 2) + (lrow >> 1));
    const int lch_w[2] = {(lane & 7) ^ (lrow >> 1), (lane & 7) ^ (4 + (lrow >> 1))};
    const int tile = xcd_remap256(blockIdx.x, g.total_tiles);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_NT_MAX_PROBLEMS; ++i)
        if (i < g.count && tile >= g.p[i].tile_start) pi = i;
    const GemmProblem& P = g.p[pi];
    const int local = tile - P.tile_start;
    int m0, n0;
    nt_tile_coords(P, local, TM, m0, n0);
    const int nkt = P.K * KV / TK;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(P.A);
    const bf16_t* W = reinterpret_cast<const bf16_t*>(P.B);
    const bf16_t* asrc[AG];
    const bf16_t* wsrc[4];
    int my_a = 0;                                       // A pieces this wave issues per slot (wave-uniform)
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        if (i * 8 + wave < NAG) ++my_a;
        int am = m0 + (i * 8 + wave) * 8 + lrow; if (am > P.M - 1) am = P.M - 1;
        asrc[i] = A + (size_t)am * P.lda * KV + lch_a * 8;
    }
    size_t wstep = TK;
    if constexpr (NN) {
        wstep = (size_t)TK * P.ldb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 2 + (lane >> 5);
            int col = n0 + (((lane & 31) ^ (tn_f(row) << 1)) << 3);
            if (col > P.ldb - 8) col = P.ldb - 8;
            wsrc[i] = W + (size_t)row * P.ldb + col;
        }
    } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int wn_ = n0 + (wave * 4 + i) * 8 + lrow; if (wn_ > P.N - 1) wn_ = P.N - 1;
        wsrc[i] = W + (size_t)wn_ * P.ldb * KV + lch_w[i & 1] * 8;
    }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    auto issue = [&](int kt) {
        const unsigned base = lds0 + (kt % NS) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < AG; ++i)
            if (i * 8 + wave < NAG) dma16(asrc[i] + kt * TK, base + (i * 8 + wave) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(wsrc[i] + kt * wstep, base + A_BYTES + (wave * 4 + i) * 1024);
    };
    // wait until at most `slots` of my slots (my_a + 4 DMA instructions each) are still in flight
    auto wait_keep = [&](int slots) {
        if (slots <= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
        if (slots == 1) {
            if (my_a == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (my_a == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            return;
        }
        if (my_a == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (my_a == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    };
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (i < nkt) issue(i);
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int ahead = nkt - 1 - kt;                 // slots after this one that have been issued at most NS - 2
        wait_keep(ahead < NS - 2 ? ahead : NS - 2);
        __builtin_amdgcn_s_barrier();                   // slot kt visible to all; everyone is done reading slot kt-1
        if (STAMP && threadIdx.x == 0 && kt == 0) st_r1 = __builtin_amdgcn_s_memrealtime();
        if (kt + NS - 1 < nkt) issue(kt + NS - 1);      // refills slot (kt-1) % NS
        const unsigned char* a = lds + (kt % NS) * STAGE_BYTES;
        const unsigned char* w = a + A_BYTES;
        nt256_kstep<MT, X3, NN>(a, w, wm, wn, fr, fq, acc);
    }
    __syncthreads();   // the ring is dead: reuse it as the epilogue's transpose scratch
    if (STAMP && threadIdx.x == 0) st_r2 = __builtin_amdgcn_s_memrealtime();
    nt256_epilogue<OutT, EPI, MT, LNF>(g, P, lds, wave, lane, m0, n0, wm, wn, acc, []() {});
    if constexpr (LNF) { __syncthreads(); nt256_ln_after_tile<TM>(P, m0, reinterpret_cast<int*>(lds)); }
    if (STAMP && threadIdx.x == 0) {
        unsigned long long* o = g.stamps + (size_t)blockIdx.x * 8;
        const unsigned long long r3 = __builtin_amdgcn_s_memrealtime();
        o[0] = st_c0; o[1] = st_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = r3; o[4] = st_r1; o[5] = st_r2; o[6] = r3; o[7] = 1;
    }
}

static inline int rgqa_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n &= ~7;                                   // whole XCDs: the tile -> XCD map needs grid % 8 == 0
        if (n < 8) n = 8;
    }
    return n;
}

// Tile height per launch: the MT in {8,7,6,5,4,2} (TM = 32 * MT rows) that minimises rounds-over-256-CUs x per-tile cost.
// (the per-tile constant 1.5 - prologue + epilogue in units of a 32-row K loop - was swept 0.3 .. 3.0 in round 3 for both precisions: 1.5
// is within 0.3 % of the best for either; the split-f32 kernels have the longer K loop but also the heavier epilogue)
static inline int pick_mt(const GemmGroup& g, long& tiles_out) {
    int best = 8; double best_cost = 1e30; long best_tiles = 0;
    const int cand[6] = {8, 7, 6, 5, 4, 2};
    const int ncu = rgqa_num_cus();
    for (int c = 0; c < 6; ++c) {
        const int mt = cand[c];
        long tiles = 0;
        for (int i = 0; i < g.count; ++i) tiles += (long)cdiv(g.p[i].M, 32 * mt) * cdiv(g.p[i].N, TN);
        const long rounds = (tiles + ncu - 1) / ncu;
        const double cost = (double)rounds * (mt + 1.5);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = mt; best_tiles = tiles; }
    }
    tiles_out = best_tiles;
    return best;
}

extern int g_rgqa_force_mt;      // rgqa_debug_set key 1 (kernel parity tests: every tile height)
extern int g_rgqa_nt_stagger;    // rgqa_debug_set key 22: late start of the persistent NT blocks that walk one tile fewer (GemmGroupT::stagger); 0 = off
extern int g_rgqa_nt_panel;      // rgqa_debug_set key 9: 0 = row-major tile numbering everywhere, -1 = default, n > 0 = panel width n

// panel width per problem (see nt_tile_coords): the widest panel whose weight rows fit ~1.6 MB (split f32: 2.4 MB - three tiles at K = 768,
// the measured best), preferring one that divides the N-tile count; narrower than three tiles a panel shares too little inside a round -
// such problems keep the row-major numbering.  In the train step (tools/ab_debug.py 9): bf16 11.43 -> 11.39 ms, bf16x3 22.88 -> 22.81.
template <bool X3>
static inline void nt_set_panels(GemmGroup& g) {
    for (int i = 0; i < g.count; ++i) {
        GemmProblem& p = g.p[i];
        const int tn_ = p.tiles_n & 0xFFFF;
        int pw = tn_;
        if (g_rgqa_nt_panel > 0) pw = g_rgqa_nt_panel < tn_ ? g_rgqa_nt_panel : tn_;
        else if (g_rgqa_nt_panel < 0) {
            const size_t wblock = (size_t)TN * p.K * (X3 ? 4 : 2);
            const int fit = (int)(((X3 ? 2400u : 1600u) << 10) / wblock);
            if (fit >= 3 && tn_ > fit) {
                pw = fit;
                for (int c = fit; c >= 3; --c) if (tn_ % c == 0) { pw = c; break; }
            }
        }
        p.tiles_n = tn_ | (pw << 16);
    }
}

// Launch of one grouped problem set at tile height MT.  64-, 128- and 160-row tiles take the deep-ring kernel (4 / 3 LDS slots,
// one tile per block): measured IN SITU (operands arriving from MALL/HBM) -4..-25 % on those launches against the two-slot loop;
// forcing 160- or 128-row tiles on the big launches to get them onto the deep ring loses 5..30 %.
template <typename OutT, int EPI, int MT, bool X3, bool NN = false, bool LNF = false>
static int launch256(GemmGroup& g, hipStream_t s) {
    constexpr int LDS_BYTES = NT256_LDS(MT);
    gemm_group_finalize(g, 32 * MT, TN);
    nt_set_panels<X3>(g);
    if constexpr (MT == 2 || MT == 4 || MT == 5) {
        constexpr int NSD = MT == 2 ? 4 : 3;
        constexpr int LDS_D = NSD * (32 * MT * TK * 2 + TN * TK * 2);
        static bool attr_set_d = false;
        if (!attr_set_d) {
            RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256d_kernel<OutT, EPI, MT, NSD, X3, false, NN, LNF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_D));
            attr_set_d = true;
        }
        hipLaunchKernelGGL((gemm_nt256d_kernel<OutT, EPI, MT, NSD, X3, false, NN, LNF>), dim3(g.total_tiles), dim3(T256_THREADS), LDS_D, s, nt_prefix(g));
        RGQA_LAUNCH_CHECK("gemm_nt256d_kernel");
        return RGQA_OK;
    } else {
        static bool attr_set = false;
        if (!attr_set) {
            RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256_kernel<OutT, EPI, MT, X3, false, NN, LNF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
            attr_set = true;
        }
        int grid = g.total_tiles;
        if (grid > rgqa_num_cus()) grid = rgqa_num_cus();     // one persistent block per CU
        g.stagger = g_rgqa_nt_stagger;
        hipLaunchKernelGGL((gemm_nt256_kernel<OutT, EPI, MT, X3, false, NN, LNF>), dim3(grid), dim3(T256_THREADS), LDS_BYTES, s, nt_prefix(g));
        RGQA_LAUNCH_CHECK("gemm_nt256_kernel");
        return RGQA_OK;
    }
}

template <typename OutT, int EPI, bool X3, bool NN = false, bool LNF = false>
static int launch256_mt(GemmGroup& g, int mt, hipStream_t s) {
    switch (mt) {
        case 8: return launch256<OutT, EPI, 8, X3, NN, LNF>(g, s);
        case 7: return launch256<OutT, EPI, 7, X3, NN, LNF>(g, s);
        case 6: return launch256<OutT, EPI, 6, X3, NN, LNF>(g, s);
        case 5: return launch256<OutT, EPI, 5, X3, NN, LNF>(g, s);
        case 4: return launch256<OutT, EPI, 4, X3, NN, LNF>(g, s);
        default: return launch256<OutT, EPI, 2, X3, NN, LNF>(g, s);
    }
}
