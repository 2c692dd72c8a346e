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
#include <string.h>
#include <type_traits>
#include "gemm.h"
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

int g_rgqa_ablate = 0;   // rgqa_debug_set key 3
int g_rgqa_no_deep = 0;  // rgqa_debug_set key 5: 1 = never use the deep-ring single-round variant (A/B)

#include "gemm256_dev.h"

// One K-step of MFMAs for a wave's (16*MT) x 64 slice.  RGQA_NT_PIPE / RGQA_TN_PIPE (build-time ring depth, 0 = plain loop): the
// A fragments come through a register ring PD deep - the LDS read for fragment i+PD is issued right after the MFMAs of fragment i,
// pinned by sched_barrier - instead of being requested two at a time just before their use, which is what the compiler makes of the
// plain loop (2 ds_read_b128, s_waitcnt, 8 MFMAs, ...: every group of 8 MFMAs starts behind a full LDS round trip that only the
// SIMD's other wave can cover).  Measured in situ (B=256 train step, libraries built both ways, tools/ab_bench.sh): NT ring 3
// -0.18 ms, wgrad ring 3 -0.26 ms, together 13.26 -> 12.81 ms; depth 4 = depth 3 (NT) / spills (wgrad, +9 %); depth 2 within noise of 3.
// Issuing the next K-step's DMA instructions one or two per fragment slot inside this stream instead of all up front: +8..14 % (step).
// __builtin_amdgcn_iglp_opt(0 / 1) on the plain loop instead of the ring: +1..3 % against the ring.
// The next K-step's DMA issued after this step's first fragment reads (behind sched_barriers) instead of before them: 3.3x slower
// (the compiler then drains vmcnt before the fragment waits).
#ifndef RGQA_NT_PIPE
#define RGQA_NT_PIPE 3
#endif
#ifndef RGQA_TN_PIPE
#define RGQA_TN_PIPE 3
#endif
// NN: the W stage is a [64 contraction rows][256 columns] image (the weight as stored, [out, in], serves dgrad without a transposed copy):
// its fragments come through ds_read_b64_tr_b16 like the wgrad kernel's
template <int MT, bool NN>
__device__ __forceinline__ void nt256_kstep(const unsigned char* a, const unsigned char* w, int wm, int wn, int fr, int fq, f32x4 (&acc)[MT][4]) {
    const int lane_ = fq * 16 + fr;
    auto ldw = [&](int half, int t) -> bf16x8 {
        if constexpr (NN) return tr_frag_dma<TN * 2>(w, half * 32, wn * 64 + t * 16, lane_);
        else return *reinterpret_cast<const bf16x8*>(w + off256(wn * 64 + t * 16 + fr, half * 4 + fq));
    };
#if RGQA_NT_PIPE
    constexpr int PD = RGQA_NT_PIPE, NF = 2 * MT;
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
#else
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bf16x8 xw[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) xw[t] = ldw(s, t);
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
            const bf16x8 xa = *reinterpret_cast<const bf16x8*>(a + off256(wm * (16 * MT) + tm * 16 + fr, s * 4 + fq));
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xw[tn], xa, acc[tm][tn], 0, 0, 0);
        }
    }
#endif
}


// EPI is a compile-time constant: the generic (run-time switched) epilogue inlined 32x stops the compiler from
// unrolling the accumulator loops and pushes the 128 accumulators into scratch.
int g_rgqa_nt_static_blocks = 0;   // with a ticket counter (GemmGroup::sched), blocks below this start on a fixed tile and the rest are spares (0 = all of them fixed); set by the engine (RGQA_NT_TICKETS / rgqa_debug_set key 12)
template <typename OutT, int EPI, int MT, bool NN = false>
__global__ __launch_bounds__(T256_THREADS) void gemm_nt256_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr bool PERSIST = NT256_PERSIST(MT);
    constexpr int TM = 32 * MT, A_BYTES = TM * TK * 2, STAGE_BYTES = A_BYTES + TN * TK * 2, AG = (MT + 1) / 2, NAG = 4 * MT;
    constexpr int EPI_OFF = PERSIST ? 2 * STAGE_BYTES : 0;
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
        for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
            if (i < g.count && tile >= g.p[i].tile_start) p = i;
        const GemmProblem& P = g.p[p];
        const int local = tile - P.tile_start;
        pi = p; m0 = (local / P.tiles_n) * TM; n0 = (local % P.tiles_n) * TN; nkt = P.K / TK;
        const bf16_t* A = reinterpret_cast<const bf16_t*>(P.A);
        const bf16_t* W = reinterpret_cast<const bf16_t*>(P.B);
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            int am = m0 + (i * 8 + wave) * 8 + lrow; if (am > P.M - 1) am = P.M - 1;   // clamp: rows past the edge are never stored
            asrc[i] = A + (size_t)am * P.lda + lch_a * 8;
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
                wsrc[i] = W + (size_t)wn_ * P.ldb + lch_w[i & 1] * 8;
            }
        }
    };
    // LDS ring depth 2. Measured alternatives on these shapes (round 1): 3 stages for MT <= 4 lost 10..20 % (MT2 loses its
    // 2 blocks/CU, K is only 12 steps); a K-step-32 / 4-slot ring lost 15..33 % with or without register-double-buffered
    // fragments (64-B DMA rows halve the useful bytes per L2 line request and the loop is close to delivery-bound: DMA alone
    // takes 0.85x of LDS+MFMA alone, rgqa_debug_set key 3); moving the barrier between the two K32 halves with the fragments
    // double-buffered across it (256 VGPRs) changed nothing (+-3 %); reading the W fragments of both halves at the top of the
    // step behind a second barrier, so that W(kt+2) is requested a step earlier (96 KiB in flight), lost 2..7 % hot and changed nothing in situ (+-0.5 %); s_setprio(1) around
    // each group of 4 MFMAs lost 5..14 % (it pays only inside a multi-phase schedule, as the CDNA guide notes).
    // A ping-pong schedule (four barrier slots per K-step - read half 0 / 32 MFMAs / read half 1 / 32 MFMAs - with waves 4-7 one slot
    // behind waves 0-3, so every SIMD always has one wave on the MFMA pipe) WON the L2-hot micro-benchmark by 1..9 % and LOST 4..8 %
    // per launch inside the train step: there the operands come from MALL/HBM, a K-step's 64 KiB take ~1.25 us to arrive with one
    // step of lead (DMA alone: 11.5 TB/s aggregate; 16.6 TB/s with two steps in flight, rgqa_debug_set key 3 = 5), and a faster
    // compute slot only shortens that lead.  The lever is more operand bytes in flight (a third slot does not fit 160 KiB of LDS
    // at this tile size), not a denser MFMA stream.  Lesson for A/B work on this loop: decide in situ (tools/ab_bench.sh +
    // RGQA_PROF_DUMP), not on a hot micro-benchmark.
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
    // Tile order.  Static: block b walks tiles b, b + grid, ...  With a ticket counter (g.sched, zeroed by the caller): blocks below
    // g.sched_static start on tile b, every further tile is the next ticket - a block that reaches its CU late (another stream's
    // kernel holds it) then takes fewer tiles instead of finishing its fixed share long after the others; with sched_static < grid the
    // blocks above it are spares that only run if a CU is free while tickets remain.  The ticket for the tile after this one is requested
    // at the top of the K loop (one lane, returned long before the loop ends) and handed to the eight waves through the first word of
    // each wave's PRIVATE epilogue scratch, written before the barrier that ends the loop: no extra barrier, no extra LDS.
    const bool dyn = PERSIST && g.sched != nullptr;
    constexpr int EPI_WAVE_BYTES = PERSIST ? NT256_TP(MT) * 4096 : 0;
    int vt = blockIdx.x;
    if (dyn && (int)blockIdx.x >= g.sched_static) {       // spare block: first tile by ticket, too (block-uniform)
        if (tid == 0) {
            int t0;        // (inline asm: hipcc's rewrite of a uniform atomic leaves the LDS-DMA statements below with a vector M0 operand)
            asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(t0) : "v"(g.sched), "v"(1) : "memory");
#pragma unroll
            for (int w = 0; w < 8; ++w) *reinterpret_cast<int*>(lds + EPI_OFF + w * EPI_WAVE_BYTES) = t0;
        }
        __syncthreads();
        vt = g.sched_static + __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(lds + EPI_OFF + wave * EPI_WAVE_BYTES));
        __syncthreads();
        if (vt >= g.total_tiles) return;
    }
    locate(vt);
    issue(0, 0);
    bool pre1 = false;      // K-step 1 of the current tile was already issued (behind the previous tile's epilogue)
    for (;;) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Issued from inline asm and NOT waited for here (hipcc would turn a builtin atomic on a uniform address into an atomic followed by
        // s_waitcnt vmcnt(0), a 2-us stall per tile): the value is read after the K loop, whose first step waits for vmcnt(0).
        // tools/check_ticket_isa.py verifies on the compiled code that nothing touches the register in between.
        int ticket = 0;
        if (dyn && tid == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "+v"(ticket) : "v"(g.sched), "v"(1) : "memory");

        for (int kt = 0; kt < nkt; ++kt) {
            const int st = kt & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nkt && !(pre1 && kt == 0) && g.ablate != 1) issue(st ^ 1, kt + 1);
            const unsigned char* a = lds + st * STAGE_BYTES;
            const unsigned char* w = a + A_BYTES;
            if (g.ablate == 5 && kt + 1 < nkt) issue(st, kt + 1);      // DMA-only with twice the bytes in flight: latency- or bandwidth-bound?
            if (g.ablate == 2 || g.ablate == 5) continue;
            nt256_kstep<MT, NN>(a, w, wm, wn, fr, fq, acc);
        }
        if (dyn && tid == 0) {
#pragma unroll
            for (int w = 0; w < 8; ++w) *reinterpret_cast<int*>(lds + EPI_OFF + w * EPI_WAVE_BYTES) = ticket;
        }
        __syncthreads();   // every wave is done with the operand stages: they may be refilled (PERSIST) or reused as scratch
        const int cpi = pi, cm0 = m0, cn0 = n0;
        const int nvt = dyn ? g.sched_static + __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(lds + EPI_OFF + wave * EPI_WAVE_BYTES)) : vt + (int)gridDim.x;
        const bool more = PERSIST && nvt < g.total_tiles;
        nt256_epilogue<OutT, EPI, MT>(g, g.p[cpi], lds + EPI_OFF, wave, lane, cm0, cn0, wm, wn, acc, [&]() {
            if (more) {
                locate(nvt);
                issue(0, 0);
                pre1 = nkt > 1;
                if (pre1) issue(1, 1);
            }
        });
        if (!more) break;
        vt = nvt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// "Deep ring" variant for launches that fit ONE round of tiles (tiles <= CUs: the language-only stages, the head, the BUTD
// GRU steps).  Such a launch cannot hide the operand-DMA latency behind other tiles: with one K-step in flight every step
// costs a full L2/MALL round trip (~1 us) whatever the MFMA work.  One tile per block, so the whole 160 KiB of LDS can hold
// the ring: NS = 4 slots for MT = 2 (40 KiB each) - three K-steps in flight under counted vmcnt waits; the epilogue scratch
// aliases the ring once the last step has been consumed.  (Written for any MT <= 5 / NS; only <MT 2, NS 4> is instantiated.)
template <typename OutT, int EPI, int MT, int NS, bool NN = false>
__global__ __launch_bounds__(T256_THREADS) void gemm_nt256d_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TM = 32 * MT, A_BYTES = TM * TK * 2, STAGE_BYTES = A_BYTES + TN * TK * 2, AG = (MT + 1) / 2, NAG = 4 * MT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lrow = lane >> 3;
    const int lch_a = (lane & 7) ^ (((wave & 1) << 2) + (lrow >> 1));
    const int lch_w[2] = {(lane & 7) ^ (lrow >> 1), (lane & 7) ^ (4 + (lrow >> 1))};
    const int tile = xcd_remap256(blockIdx.x, g.total_tiles);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
        if (i < g.count && tile >= g.p[i].tile_start) pi = i;
    const GemmProblem& P = g.p[pi];
    const int local = tile - P.tile_start;
    const int m0 = (local / P.tiles_n) * TM, n0 = (local % P.tiles_n) * TN, nkt = P.K / TK;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(P.A);
    const bf16_t* W = reinterpret_cast<const bf16_t*>(P.B);
    const bf16_t* asrc[AG];
    const bf16_t* wsrc[4];
    int my_a = 0;                                       // A pieces this wave issues per slot (wave-uniform)
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        if (i * 8 + wave < NAG) ++my_a;
        int am = m0 + (i * 8 + wave) * 8 + lrow; if (am > P.M - 1) am = P.M - 1;
        asrc[i] = A + (size_t)am * P.lda + lch_a * 8;
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
            wsrc[i] = W + (size_t)wn_ * P.ldb + lch_w[i & 1] * 8;
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
        if (kt + NS - 1 < nkt) issue(kt + NS - 1);      // refills slot (kt-1) % NS
        const unsigned char* a = lds + (kt % NS) * STAGE_BYTES;
        const unsigned char* w = a + A_BYTES;
        nt256_kstep<MT, NN>(a, w, wm, wn, fr, fq, acc);
    }
    __syncthreads();   // the ring is dead: reuse it as the epilogue's transpose scratch
    nt256_epilogue<OutT, EPI, MT>(g, P, lds, wave, lane, m0, n0, wm, wn, acc, []() {});
}

static int rgqa_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n &= ~7;                                   // whole XCDs: the tile -> XCD map needs grid % 8 == 0
        if (n < 8) n = 8;
    }
    return n;
}

// Tile height per launch: the MT in {8,6,4,2} (TM = 256/192/128/64) that minimises rounds-over-256-CUs x per-tile cost.
static int pick_mt(const GemmGroup& g, long& tiles_out) {
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

// true when every problem of the group can run on the LDS-DMA kernel and the launch fills enough of the chip
bool gemm_nt256_eligible(const GemmGroup& g, int out_f32) {
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
    static const bool basic_only = getenv("RGQA_NT256_EPI_BASIC") != nullptr;     // A/B: leave tanh / relu epilogues to the 128x128 kernel
    if (basic_only && !(epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESID_DROP || epi == EPI_DGELU || epi == EPI_ADD)) return false;
    for (int i = 0; i < g.count; ++i) {
        const GemmProblem& p = g.p[i];
        if (p.epi != epi || p.K % TK != 0 || p.K < TK || (p.ldc % 8) != 0 || (p.N % 8) != 0) return false;
        if (epi_needs_aux(epi) && (p.ldaux % 8) != 0) return false;
    }
    long tiles = 0;
    pick_mt(g, tiles);
    static const long min_tiles = []() { const char* e = getenv("RGQA_NT256_MIN_TILES"); return e ? atol(e) : 1L; }();
    return tiles >= min_tiles;
}

template <int EPI, int MT, bool NN>
static int launch256(GemmGroup& g, hipStream_t s) {
    constexpr int LDS_BYTES = NT256_LDS(MT);
    static bool attr_set = false;
    if (!attr_set) {
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256_kernel<bf16_t, EPI, MT, NN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    gemm_group_finalize(g, 32 * MT, TN);
    g.ablate = g_rgqa_ablate;
    // Launches whose tile height is 64, 128 or 160 rows (the language-only stages, the N = 768 projections, the head, BUTD's GRU
    // steps) take the deep-ring kernel: 4 / 3 LDS slots, one tile per block.  Measured IN SITU (RGQA_PROF_DUMP, operands arriving
    // from MALL/HBM): -4..-25 % on those launches, NT total -4 %, step -1.3 %; the L2-hot micro-benchmark had shown -3..7 % for most
    // of them (there the one-step lead of the 2-slot loop already covers the DMA latency) and +15..18 % only for 64-row tiles at
    // K >= 2304.  Forcing 160- or 128-row tiles on the big launches to get them onto the deep ring loses 5..30 %.
    // RGQA_NT_DEEP: 0 never, 1 single-round 64-row launches with K >= 2048 only, 2 every launch of 64-row tiles, 3 (default) also
    // 128- and 160-row tiles
    static const int deep_mode = []() { const char* e = getenv("RGQA_NT_DEEP"); return e ? atoi(e) : 3; }();
    bool deep = false;
    if (!g_rgqa_no_deep && deep_mode > 0) {
        if (deep_mode == 1) {
            deep = MT == 2 && g.total_tiles <= rgqa_num_cus() && g.total_tiles >= 64;
            for (int i = 0; i < g.count && deep; ++i) if (g.p[i].K < 32 * TK) deep = false;
        } else deep = MT == 2 || (deep_mode >= 3 && (MT == 4 || MT == 5));
    }
    if constexpr (MT == 2 || MT == 4 || MT == 5) {
        if (deep) {
            constexpr int NSD = MT == 2 ? 4 : 3;
            constexpr int LDS_D = NSD * (32 * MT * TK * 2 + TN * TK * 2);
            static bool attr_set_d = false;
            if (!attr_set_d) {
                RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256d_kernel<bf16_t, EPI, MT, NSD, NN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_D));
                attr_set_d = true;
            }
            hipLaunchKernelGGL((gemm_nt256d_kernel<bf16_t, EPI, MT, NSD, NN>), dim3(g.total_tiles), dim3(T256_THREADS), LDS_D, s, g);
            RGQA_LAUNCH_CHECK("gemm_nt256d_kernel");
            return RGQA_OK;
        }
    }
    int grid = g.total_tiles;
    static const bool nonpersist = getenv("RGQA_NT_NONPERSIST") != nullptr;   // experiment: one tile per block, hardware dispatch order
    if (NT256_PERSIST(MT) && !nonpersist && grid > rgqa_num_cus()) grid = rgqa_num_cus();     // one persistent block per CU
    if (grid == g.total_tiles) g.sched = nullptr;          // one tile per block anyway
    g.sched_static = grid;
    if (g.sched != nullptr && g_rgqa_nt_static_blocks > 0 && g_rgqa_nt_static_blocks < grid) g.sched_static = g_rgqa_nt_static_blocks;
    hipLaunchKernelGGL((gemm_nt256_kernel<bf16_t, EPI, MT, NN>), dim3(grid), dim3(T256_THREADS), LDS_BYTES, s, g);
    RGQA_LAUNCH_CHECK("gemm_nt256_kernel");
    return RGQA_OK;
}

template <int EPI, bool NN = false>
static int launch256_mt(GemmGroup& g, int mt, hipStream_t s) {
    switch (mt) {
        case 8: return launch256<EPI, 8, NN>(g, s);
        case 7: return launch256<EPI, 7, NN>(g, s);
        case 6: return launch256<EPI, 6, NN>(g, s);
        case 5: return launch256<EPI, 5, NN>(g, s);
        case 4: return launch256<EPI, 4, NN>(g, s);
        default: return launch256<EPI, 2, NN>(g, s);
    }
}

int g_rgqa_force_mt = 0;
static int launch256_epi(GemmGroup& g, int mt, hipStream_t s) {
    if (g.b_kn) {       // B operand stored [K, N] (dgrad on the weight as it is): the epilogues a dgrad uses
        switch (g.p[0].epi) {
            case EPI_BIAS: return launch256_mt<EPI_BIAS, true>(g, mt, s);
            case EPI_DGELU: return launch256_mt<EPI_DGELU, true>(g, mt, s);
            case EPI_ADD: return launch256_mt<EPI_ADD, true>(g, mt, s);
            case EPI_DTANH: return launch256_mt<EPI_DTANH, true>(g, mt, s);
            case EPI_DRELU_DROP: return launch256_mt<EPI_DRELU_DROP, true>(g, mt, s);
            default: rgqa_set_error("gemm: no [K,N]-operand kernel for epilogue %d", g.p[0].epi); return RGQA_ERR_ARG;
        }
    }
    switch (g.p[0].epi) {
        case EPI_BIAS: return launch256_mt<EPI_BIAS>(g, mt, s);
        case EPI_GELU: return launch256_mt<EPI_GELU>(g, mt, s);
        case EPI_RESID_DROP: return launch256_mt<EPI_RESID_DROP>(g, mt, s);
        case EPI_DGELU: return launch256_mt<EPI_DGELU>(g, mt, s);
        case EPI_TANH: return launch256_mt<EPI_TANH>(g, mt, s);
        case EPI_DTANH: return launch256_mt<EPI_DTANH>(g, mt, s);
        case EPI_RELU: return launch256_mt<EPI_RELU>(g, mt, s);
        case EPI_RELU_DROP: return launch256_mt<EPI_RELU_DROP>(g, mt, s);
        case EPI_DRELU_DROP: return launch256_mt<EPI_DRELU_DROP>(g, mt, s);
        default: return launch256_mt<EPI_ADD>(g, mt, s);
    }
}

// Tile height by measurement: the first time a launch signature (epilogue + every problem's M,N,K) is seen, each candidate
// is timed on the caller's stream (the NT epilogues are pure functions of their inputs, so re-running a launch is harmless;
// the result is bit-identical for every MT - same K order per output) and the fastest is cached for the process.
// Opt-in (RGQA_GEMM_AUTOTUNE=1): on the config-2 shapes the pick_mt() cost model already lands on the measured best or
// within ~2 % of it for 20 of 22 signatures (16.05 vs 16.02 ms/step), so the default stays model-driven and sync-free.
static int tuned_mt(GemmGroup& g, int model_mt, hipStream_t s) {
    static const bool enabled = []() { const char* e = getenv("RGQA_GEMM_AUTOTUNE"); return e && e[0] == '1'; }();
    if (!enabled) return model_mt;
    static std::mutex mu;
    static std::unordered_map<uint64_t, int> cache;
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
    mix((uint64_t)g.p[0].epi); mix((uint64_t)g.count);
    for (int i = 0; i < g.count; ++i) { mix((uint64_t)g.p[i].M); mix((uint64_t)g.p[i].N); mix((uint64_t)g.p[i].K); }
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return model_mt;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return model_mt;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return model_mt; }
    const int cand[6] = {8, 7, 6, 5, 4, 2};
    float best[6] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
    bool ok = true;
    int* const sched_saved = g.sched;
    g.sched = nullptr;                 // the timing launches must not draw this launch's tile tickets
    for (int round = 0; round < 4 && ok; ++round)
        for (int c = 0; c < 6 && ok; ++c) {
            ok = hipEventRecord(e0, s) == hipSuccess && launch256_epi(g, cand[c], s) == RGQA_OK && hipEventRecord(e1, s) == hipSuccess &&
                 hipEventSynchronize(e1) == hipSuccess;
            float ms = 0.f;
            if (ok && round > 0 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best[c]) best[c] = ms;   // round 0 warms up
        }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    g.sched = sched_saved;
    int mt = model_mt;
    if (ok) {
        int bi = 0;
        for (int c = 1; c < 6; ++c) if (best[c] < best[bi]) bi = c;
        mt = cand[bi];
        if (getenv("RGQA_GEMM_AUTOTUNE_LOG"))
            fprintf(stderr, "[rgqa] nt256 tune epi=%d n=%d M0=%d N0=%d K0=%d: MT8 %.1f MT7 %.1f MT6 %.1f MT5 %.1f MT4 %.1f MT2 %.1f us -> MT%d (model MT%d)\n", g.p[0].epi, g.count,
                    g.p[0].M, g.p[0].N, g.p[0].K, best[0] * 1e3f, best[1] * 1e3f, best[2] * 1e3f, best[3] * 1e3f, best[4] * 1e3f, best[5] * 1e3f, mt, model_mt);
    }
    cache[key] = mt;
    return mt;
}

int launch_gemm_nt256_f32out(GemmGroup& g, hipStream_t s) {
    constexpr int LDS_D = 4 * (32 * 2 * TK * 2 + TN * TK * 2);
    static bool attr_set = false;
    if (!attr_set) {
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256d_kernel<float, EPI_BIAS, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_D));
        attr_set = true;
    }
    gemm_group_finalize(g, 64, TN);
    g.ablate = 0;
    hipLaunchKernelGGL((gemm_nt256d_kernel<float, EPI_BIAS, 2, 4>), dim3(g.total_tiles), dim3(T256_THREADS), LDS_D, s, g);
    RGQA_LAUNCH_CHECK("gemm_nt256d_kernel<float>");
    return RGQA_OK;
}

bool gemm_nt8p_eligible(const GemmGroup& g);
int launch_gemm_nt8p_bf16(GemmGroup& g, int mt, hipStream_t s);
extern int g_rgqa_nt8p;

int launch_gemm_nt256_bf16(GemmGroup& g, hipStream_t s) {
    long tiles = 0;
    int mt = pick_mt(g, tiles);
    static const int env_mt = []() { const char* e = getenv("RGQA_NT_FORCE_MT"); return e ? atoi(e) : 0; }();     // experiments
    if (g_rgqa_force_mt) mt = g_rgqa_force_mt;
    else if (env_mt) mt = env_mt;
    else mt = tuned_mt(g, mt, s);
    // The phase-interleaved kernel (gemm_nt8p.hip) keeps 80 KiB of operands in flight instead of <= 64: on cold operands it is 7-13 %
    // faster than the two-slot kernel at every tile height (tools/lab/gemm_lab: 8192^3 1.39 vs 1.20 PFLOP/s; the encoder's
    // N = 2304 / 3072 launches -7..-10 %).  IN SITU (activations just written by the previous kernel; rocprofv3 kernel trace of
    // bench.py, profiles/r02_*): 256- and 224-row tiles tie (+-1 %, +4 % for the DGELU dgrad), 192-row tiles gain 2-6 %, and the
    // single-round 160-row launches lose 12 % to the deep-ring kernel, which has 104 KiB in flight.  Default (RGQA_NT8P=1): 192-row
    // tiles only; 2 = every launch of 160..256-row tiles (what a cold / large-K caller wants); 0 = never.
    if (mt >= 5 && !g.b_kn && gemm_nt8p_eligible(g) && (g_rgqa_nt8p >= 2 || mt == 6)) return launch_gemm_nt8p_bf16(g, mt, s);
    return launch256_epi(g, mt, s);
}

// ============================================================================ TN (wgrad) with LDS-DMA
//   C[M,N] (f32) (+)= A[K,M]^T * B[K,N],  K % 64 == 0:  dW[n,k] = sum_rows dY[row,n] X[row,k]
// 128 x 256 output tile, 8 waves (2 x 4, 64x64 each), 3-stage ring x (A 64x128 + B 64x256) bf16 = 144 KiB.
// Both operands are row-major over the CONTRACTION index, so the LDS images are natural row-major copies filled
// by LDS-DMA and the MFMA fragments are fetched with ds_read_b64_tr_b16.  32-byte granules of a row are XOR-
// swizzled with f(row) = (row&3) | ((row>>3)&1)<<2 so that the 8 (row, 32 B) pieces a half-wave touches per
// transposed read land on 8 distinct bank groups.  Tiles are launched longest-contraction-first so the hardware
// dispatcher balances the unequal (lang / visn / shared) problems of one launch over the 256 CUs.
#define WM 128
#define WN 256
// MTW = 16-row m-tiles per wave (2 waves along M): output tile WMV = 32*MTW rows x 256 columns.
//   MTW = 4: 128 x 256, 3-slot ring (48 KiB per K-step per CU);  MTW = 8: 256 x 256, 2 slots of 64 KiB: twice the MFMAs per
//   DMA byte.  The loop is bound by what the LDS-DMA path delivers per CU (~50 GB/s), not by the MFMAs, so the taller tile
//   costs ~1.5x less CU time per FLOP; it halves the tile count, which only pays when something else fills the idle CUs (the
//   launch runs on the side stream beside the next layer's chain).
// ---- XCD-local placement + split contraction (TnPlan, built by plan_tn below).  A grouped weight-gradient launch has few output
// tiles (36 for a [768,3072] weight) and long, UNEQUAL contractions (3.1 k packed language rows vs 9.2 k vision rows).  Spread
// round-robin over the 8 XCDs, the tiles of one problem share their operand panels through 8 different L2s: rocprofv3 FETCH_SIZE
// showed 2.9x the unique operand bytes coming from beyond L2 and the loop ran at the fabric's pace (2.4 us per K-step, 4.2 TB/s of
// LDS-DMA chip-wide), with the language tiles finishing 3x earlier than the vision tiles beside them.  The plan cuts every
// problem's contraction into chunks of similar length, keeps the tiles of one (problem, chunk) group on ONE XCD (workgroup b runs
// on XCD b % 8), balances the XCDs by total K-steps and orders each XCD's list longest-first.  Chunk 0 writes the gradient itself,
// chunks >= 1 write f32 partials that tn_fold_kernel adds in a fixed order (deterministic, unlike atomics).
#define TN_MAX_GROUPS 48
struct TnGroupDesc { int prob, chunk, xcd, start, tiles, tile0, kt0, nk; };      // plain ints: hipcc 7.2 mis-reads 16-bit kernarg fields under a dynamic index
struct TnPlan { int use, ngroups; TnGroupDesc grp[TN_MAX_GROUPS]; };

template <int ACCUM, int MTW>
__global__ __launch_bounds__(T256_THREADS) void gemm_tn_dma_kernel(const GemmGroup g, const TnPlan plan) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int WMV = 32 * MTW, NSLOT = MTW == 4 ? 3 : 2;
    constexpr int APW = MTW / 2;                       // A pieces (1 KiB) per wave per slot
    constexpr int ARPP = 1024 / (WMV * 2), ALPR = 64 / ARPP;   // rows per A piece, lanes per A row
    constexpr int A_BYTES = TK * WMV * 2, B_BYTES = TK * WN * 2, STAGE = A_BYTES + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int tile = blockIdx.x, pi = 0, chunk = 0, kt0 = 0, nk = -1, tix = -1;
    if (plan.use) {
        const int x = blockIdx.x & 7, pos = blockIdx.x >> 3;
        int gi = -1;
        for (int i = 0; i < plan.ngroups; ++i)
            if (plan.grp[i].xcd == x && pos >= plan.grp[i].start && pos < plan.grp[i].start + plan.grp[i].tiles) gi = i;
        if (gi < 0) return;                            // padding block of a shorter XCD list (block-uniform, before any barrier)
        const TnGroupDesc& gd = plan.grp[gi];
        pi = gd.prob; chunk = gd.chunk; kt0 = gd.kt0; nk = gd.nk; tix = gd.tile0 + pos - gd.start;
    } else {
#pragma unroll
        for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
            if (i < g.count && tile >= g.p[i].tile_start) pi = i;
    }
    const GemmProblem& P = g.p[pi];
    // tile order inside a problem: consecutive ids run over the M-tiles of one N-tile first, so neighbours stream the same B
    // operand (the wider one).  Without a plan, blocks that share an XCD (equal id mod 8) get consecutive ids.
    const int tiles_m = cdiv(P.M, WMV);
    if (tix < 0) tix = xcd_remap256(tile - P.tile_start, tiles_m * P.tiles_n);
    const int local = (tix % tiles_m) * P.tiles_n + (tix / tiles_m);     // back to the m-major id used below
    const int m0 = (local / P.tiles_n) * WMV, n0 = (local % P.tiles_n) * WN;
    // contraction length need not be a multiple of the K-step (packed language rows): in the last, partial step the A rows
    // past K come from a zero line (they also feed the bias column sums) and the B rows past K re-read row K-1 (finite data
    // times zero), so no lane predicates its DMA.
    const int nkt_all = cdiv(P.K, TK), ktail = P.K % TK;
    const int nkt = nk < 0 ? nkt_all : nk;             // this block's K-steps: kt0 .. kt0 + nkt of the problem's nkt_all
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
        const size_t ao = (size_t)(kt0 + kt) * TK * P.lda, bo = (size_t)(kt0 + kt) * TK * P.ldb;
        if (ktail != 0 && kt0 + kt == nkt_all - 1) {       // block-uniform
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
#if RGQA_TN_PIPE
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
#else
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 xb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) xb[t] = tr_frag_dma<WN * 2>(b, s * 32, wn * 64 + t * 16, lane);
#pragma unroll
            for (int tm = 0; tm < MTW; ++tm) {
                const bf16x8 xa = tr_frag_dma<WMV * 2>(a, s * 32, wm * (16 * MTW) + tm * 16, lane);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb[tn], xa, acc[tm][tn], 0, 0, 0);
                if (DO_CS) cs[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xa, cs[tm], 0, 0, 0);
            }
        }
#endif
    }
    };
    if (do_cs) kloop(std::true_type{}); else kloop(std::false_type{});
    // chunk 0 -> the gradient (and bias gradient) itself; chunk c >= 1 -> dense f32 partial c-1: [M, N] then the M column sums
    float* Cc = reinterpret_cast<float*>(P.C);
    float* cs_out = P.colsum_out;
    int ldc = P.ldc;
    const bool accum = ACCUM && chunk == 0;
    if (chunk > 0) {
        Cc = reinterpret_cast<float*>(P.C2) + (size_t)(chunk - 1) * ((size_t)P.M * P.N + P.M);
        cs_out = Cc + (size_t)P.M * P.N;
        ldc = P.N;
    }
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
    static const long min_tiles = []() { const char* e = getenv("RGQA_TN_DMA_MIN_TILES"); return e ? atol(e) : 1L; }();
    return tiles >= min_tiles;
}

int g_rgqa_tn_mtw = 0;     // rgqa_debug_set key 4: force the wgrad tile height (4 = 128 rows, 8 = 256 rows); 0 = default
int g_rgqa_tn_plan = -1;   // rgqa_debug_set key 6: 0 = round-robin tiles, no split; 1 = XCD-local placement only; 2 = placement + split contraction; -1 = env RGQA_TN_PLAN / default (0)

// dst (+ its bias gradient) += partial 0 + partial 1 + ... in that order; one entry per split problem
#define TN_FOLD_MAX 12
struct TnFoldEntry { float* dst; float* cs_dst; const float* src; int M, N, ldc, nparts, blk0; };
struct TnFoldArgs { int n, total_blocks; TnFoldEntry e[TN_FOLD_MAX]; };
#define TN_FOLD_PER_BLOCK 2048          // floats per block: 256 threads x 2 float4
__global__ __launch_bounds__(256) void tn_fold_kernel(const TnFoldArgs a) {
    int ei = 0;
#pragma unroll
    for (int i = 1; i < TN_FOLD_MAX; ++i)
        if (i < a.n && (int)blockIdx.x >= a.e[i].blk0) ei = i;
    const TnFoldEntry& e = a.e[ei];
    const size_t mn = (size_t)e.M * e.N, pstride = mn + e.M;
    const size_t base = (size_t)((int)blockIdx.x - e.blk0) * TN_FOLD_PER_BLOCK;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const size_t i = base + (size_t)(r * 256 + threadIdx.x) * 4;
        if (i < mn) {                                   // N % 4 == 0: a float4 never straddles rows
            float* d = e.dst + (i / e.N) * e.ldc + (i % e.N);
            float v[4]; load4(d, v);
            for (int j = 0; j < e.nparts; ++j) { float t[4]; load4(e.src + j * pstride + i, t); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
            store4(d, v);
        } else if (e.cs_dst != nullptr) {               // the M column sums behind the matrix, element-wise
            for (int k = 0; k < 4; ++k) {
                const size_t m = i - mn + k;
                if (m < (size_t)e.M) { float v = e.cs_dst[m]; for (int j = 0; j < e.nparts; ++j) v += e.src[j * pstride + mn + m]; e.cs_dst[m] = v; }
            }
        }
    }
}

// Builds the placement for one grouped launch (problems already sorted longest contraction first). Returns the grid size.
static int plan_tn(GemmGroup& g, int wmv, int split, TnPlan& pl, TnFoldArgs& fa) {
    memset(&pl, 0, sizeof pl); memset(&fa, 0, sizeof fa);
    struct Grp { int prob, chunk, tile0, tiles, kt0, nk; long w; int xcd, start; };
    Grp gr[TN_MAX_GROUPS]; int ng = 0;
    int ks[GEMM_MAX_PROBLEMS], nch[GEMM_MAX_PROBLEMS], tiles[GEMM_MAX_PROBLEMS];
    int kmin = 1 << 30;
    for (int i = 0; i < g.count; ++i) {
        ks[i] = cdiv(g.p[i].K, TK); tiles[i] = cdiv(g.p[i].M, wmv) * cdiv(g.p[i].N, WN);
        if (ks[i] < kmin) kmin = ks[i];
        if (tiles[i] > 60000 || ks[i] > 60000) return -1;
    }
    const int lref = kmin < 32 ? 32 : kmin;
    // chunk counts: contraction / reference length, rounded, at most 4; partials must fit the scratch and stay 16-B aligned
    size_t need = 0; int nfold = 0;
    for (int i = 0; i < g.count; ++i) {
        int c = split ? (ks[i] + lref / 2) / lref : 1;
        if (c < 1) c = 1;
        if (c > 4) c = 4;
        if ((g.p[i].N % 4) != 0 || (g.p[i].M % 4) != 0 || g.p[i].C2 != nullptr) c = 1;
        nch[i] = c;
        if (c > 1) { need += (size_t)(c - 1) * ((size_t)g.p[i].M * g.p[i].N + g.p[i].M) * sizeof(float); ++nfold; }
    }
    if (need > g.tn_scratch_bytes || g.tn_scratch == nullptr || nfold > TN_FOLD_MAX)
        for (int i = 0; i < g.count; ++i) nch[i] = 1;
    int ngroups0 = 0;
    for (int i = 0; i < g.count; ++i) ngroups0 += nch[i];
    if (ngroups0 > TN_MAX_GROUPS) return -1;
    float* sc = reinterpret_cast<float*>(g.tn_scratch);
    long W = 0; int fold_blocks = 0;
    for (int i = 0; i < g.count; ++i) {
        for (int j = 0; j < nch[i]; ++j) {
            Grp& q = gr[ng++];
            q.prob = i; q.chunk = j; q.tile0 = 0; q.tiles = tiles[i];
            q.kt0 = (int)((long)ks[i] * j / nch[i]); q.nk = (int)((long)ks[i] * (j + 1) / nch[i]) - q.kt0;
            q.w = (long)q.tiles * q.nk; W += q.w;
        }
        if (nch[i] > 1) {
            GemmProblem& p = g.p[i];
            p.C2 = sc;
            TnFoldEntry& e = fa.e[fa.n++];
            e.dst = reinterpret_cast<float*>(p.C); e.cs_dst = p.colsum_out; e.src = sc; e.M = p.M; e.N = p.N; e.ldc = p.ldc; e.nparts = nch[i] - 1; e.blk0 = fold_blocks;
            const size_t per = (size_t)p.M * p.N + p.M;
            fold_blocks += (int)((per + TN_FOLD_PER_BLOCK - 1) / TN_FOLD_PER_BLOCK);
            sc += (size_t)(nch[i] - 1) * per;
        }
    }
    fa.total_blocks = fold_blocks;
    // no group heavier than half an XCD's fair share (so the LPT assignment below can balance): halve the heaviest by tiles
    const long cap = W / 16 > 0 ? W / 16 : 1;
    while (ng < TN_MAX_GROUPS) {
        int h = 0;
        for (int i = 1; i < ng; ++i) if (gr[i].w > gr[h].w) h = i;
        if (gr[h].w <= cap || gr[h].tiles < 2) break;
        Grp& a = gr[h]; Grp& b = gr[ng++];
        b = a;
        a.tiles = a.tiles / 2; b.tile0 = a.tile0 + a.tiles; b.tiles -= a.tiles;
        a.w = (long)a.tiles * a.nk; b.w = (long)b.tiles * b.nk;
    }
    // LPT: heaviest group first onto the lightest XCD
    int order[TN_MAX_GROUPS];
    for (int i = 0; i < ng; ++i) order[i] = i;
    for (int i = 1; i < ng; ++i)
        for (int j = i; j > 0 && gr[order[j]].w > gr[order[j - 1]].w; --j) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < ng; ++k) {
        int x = 0;
        for (int i = 1; i < 8; ++i) if (load[i] < load[x]) x = i;
        gr[order[k]].xcd = x; load[x] += gr[order[k]].w;
    }
    // per XCD: longest chunks first (the XCD's 32 CUs take the list in order as they free up)
    for (int i = 1; i < ng; ++i)
        for (int j = i; j > 0 && gr[order[j]].nk > gr[order[j - 1]].nk; --j) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    int len[8] = {0, 0, 0, 0, 0, 0, 0, 0}, maxlen = 0;
    for (int k = 0; k < ng; ++k) { Grp& q = gr[order[k]]; q.start = len[q.xcd]; len[q.xcd] += q.tiles; if (len[q.xcd] > maxlen) maxlen = len[q.xcd]; }
    if (maxlen > 60000) return -1;
    pl.use = 1; pl.ngroups = ng;
    for (int i = 0; i < ng; ++i) {
        TnGroupDesc& d = pl.grp[i];
        d.prob = gr[i].prob; d.chunk = gr[i].chunk; d.xcd = gr[i].xcd;
        d.start = gr[i].start; d.tiles = gr[i].tiles; d.tile0 = gr[i].tile0; d.kt0 = gr[i].kt0; d.nk = gr[i].nk;
    }
    return 8 * maxlen;
}

template <int ACCUM, int MTW>
static int launch_tn(GemmGroup& g, int plan_mode, hipStream_t s) {
    constexpr int WMV = 32 * MTW, LDS_BYTES = (MTW == 4 ? 3 : 2) * (TK * WMV * 2 + TK * WN * 2);
    static bool attr_set = false;
    if (!attr_set) {
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_dma_kernel<ACCUM, MTW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    gemm_group_finalize(g, WMV, WN);
    TnPlan pl; TnFoldArgs fa;
    int grid = plan_mode > 0 ? plan_tn(g, WMV, plan_mode > 1, pl, fa) : -1;
    if (grid <= 0) { memset(&pl, 0, sizeof pl); fa.n = 0; grid = g.total_tiles; }
    hipLaunchKernelGGL((gemm_tn_dma_kernel<ACCUM, MTW>), dim3(grid), dim3(T256_THREADS), LDS_BYTES, s, g, pl);
    RGQA_LAUNCH_CHECK("gemm_tn_dma_kernel");
    if (fa.n > 0) {
        hipLaunchKernelGGL(tn_fold_kernel, dim3(fa.total_blocks), dim3(256), 0, s, fa);
        RGQA_LAUNCH_CHECK("tn_fold_kernel");
    }
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
    static const int env_mtw = []() { const char* e = getenv("RGQA_TN_MTW"); return e ? atoi(e) : 0; }();
    int mtw = g_rgqa_tn_mtw ? g_rgqa_tn_mtw : env_mtw;
    if (mtw != 4 && mtw != 8) {
        long tiles8 = 0; bool tall = true;
        for (int i = 0; i < g.count; ++i) { tiles8 += (long)cdiv(g.p[i].M, 256) * cdiv(g.p[i].N, WN); if (g.p[i].M < 256 || g.p[i].lda < 256) tall = false; }
        static const long tall_min = []() { const char* e = getenv("RGQA_TN_TALL_MIN"); return e ? atol(e) : 128L; }();
        mtw = (tall && tiles8 >= tall_min) ? 8 : 4;
    }
    // Measured (B=256, 1 x MI355X): with the launches serialised on one stream, placement + split contraction cut the weight-gradient
    // time from 3.65 to 3.0 ms per step (step 14.33 -> 13.69 ms; placement alone 14.29: the loop is not L2-miss bound, one tile
    // alone on an idle chip still needs 1.7 us per K-step against 0.86 us of MFMAs - the per-CU LDS-DMA rate, ~38 GB/s here).  On
    // the side stream, where the main stream's kernels already fill the CUs the unbalanced launch leaves idle, the step did not
    // move (13.04-13.12 without, 13.13-13.32 ms with).  Hence opt-in: RGQA_TN_PLAN=1 (placement) / 2 (placement + split).
    static const int env_plan = []() { const char* e = getenv("RGQA_TN_PLAN"); return e ? atoi(e) : 0; }();
    const int pm = g_rgqa_tn_plan >= 0 ? g_rgqa_tn_plan : env_plan;
    const bool acc = g.p[0].epi == EPI_ACCUM;
    if (mtw == 8) return acc ? launch_tn<1, 8>(g, pm, s) : launch_tn<0, 8>(g, pm, s);
    return acc ? launch_tn<1, 4>(g, pm, s) : launch_tn<0, 4>(g, pm, s);
}
