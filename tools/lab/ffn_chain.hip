// LAB (VERDICT r4 #1a; never linked into the product): ONE persistent launch that owns the tile lists of TWO dependent grouped GEMMs -
//   phase 1   H = gelu(X W1^T + b1), H' = gelu'(..)      N = 3072, K = 768    (BertIntermediate, lxrt/modeling.py:389-401; EPI1)
//   phase 2   Z = dropout(H W2^T + b2) + X               N = 768,  K = 3072   (BertOutput before its LayerNorm, :404-415; EPI2)
// (and, with EPI1 = EPI_DGELU / EPI2 = EPI_ADD, backward's dFFN2 -> dFFN1 pair) - against the same two problems as two launches of the product
// kernels.  What the chain can win: CUs that have no tile in phase 1's short last round fall straight into phase 2 (a phase-2 tile of rows that
// phase 1 has finished starts while other CUs are still in phase 1), one kernel boundary and one prologue burst disappear.
//
// Schedule: static.  Both tile lists are numbered row block by row block (N fastest; no panels) and cut into eight contiguous runs, one per XCD
// (xcd_remap256), so that an XCD walks the same rows in both phases.  Block b = (XCD b & 7, slot b >> 3) takes phase-1 tiles slot, slot + 32,
// ... of its XCD's run; the slots that get no tile in the run's last round are "early": in phase 2 the early slots take the first tiles of the
// XCD's run (rows whose phase-1 tiles were all in full rounds), the others the rest.
// Hand-off (MI355X_MICROARCH.md, Valid forms): phase-1 result rows are stored write-through (sc1); a block signals a finished tile only after
// EVERY wave's `s_waitcnt vmcnt(0)` + the workgroup barrier - the wait at the top of its NEXT tile's K loop (or an explicit one after its last
// tile) - by ONE agent-scope atomic add on the counter of (problem, phase-1 row block).  A phase-2 tile polls (sc1 loads, one lane, bounded) the
// counters of the phase-1 row blocks its rows lie in until each has seen all N1/256 tiles, then ONE agent acquire (buffer_inv sc1), that
// lane's vmcnt(0), the barrier, and only then its first LDS-DMA.  Results do not depend on placement or timing; a spin that gives up sets a
// timeout word and the tile proceeds (the harness checks the word and the outputs).
#include <string.h>
#include "gemm_nt256.h"

bool gemm_nt256_eligible(const GemmGroup& g, int out_f32);      // csrc/gemm_mfma256.hip

struct ChainArgs {
    GemmGroupNT g1, g2;          // finalized at tile heights 32 * MT1 / 32 * MT2; row-major tile numbering (tiles_n high half = 0)
    int* counters;               // zeroed before the launch: counters[cnt_base[p] + row block]
    int cnt_base[4];
    unsigned* timeout;           // zeroed before the launch
    int rows1;                   // 32 * MT1
};

__device__ __forceinline__ int xcd_run_begin(int xcd, int nwg) { const int q = nwg >> 3, r = nwg & 7; return xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q; }
__device__ __forceinline__ int xcd_run_len(int xcd, int nwg) { return (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0); }

template <int EPI1, int EPI2, int MT1, int MT2>
__global__ __launch_bounds__(T256_THREADS) void ffn_chain_kernel(const ChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lrow = lane >> 3;
    const int lch_a = (lane & 7) ^ (((wave & 1) << 2) + (lrow >> 1));
    const int lch_w[2] = {(lane & 7) ^ (lrow >> 1), (lane & 7) ^ (4 + (lrow >> 1))};
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    const int fr = lane & 15, fq = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = (int)gridDim.x >> 3;     // grid % 8 == 0

    // ================================================================= phase 1: the persistent two-slot loop of gemm_nt256_kernel
    {
        constexpr int MT = MT1, TM = 32 * MT, A_BYTES = TM * TK * 2, STAGE_BYTES = A_BYTES + TN * TK * 2, AG = (MT + 1) / 2, NAG = 4 * MT;
        constexpr int EPI_OFF = 2 * STAGE_BYTES;
        const GemmGroupNT& g = c.g1;
        const int run0 = xcd_run_begin(xcd, g.total_tiles), runl = xcd_run_len(xcd, g.total_tiles);
        const bf16_t* asrc[AG];
        const bf16_t* wsrc[4];
        int pi = 0, m0 = 0, n0 = 0, nkt = 0;
        auto locate = [&](int k) {          // k-th tile of this XCD's run
            const int tile = run0 + k;
            int p = 0;
#pragma unroll
            for (int i = 1; i < 4; ++i)
                if (i < g.count && tile >= g.p[i].tile_start) p = i;
            const GemmProblem& P = g.p[p];
            pi = p; nt_tile_coords(P, tile - P.tile_start, TM, m0, n0); nkt = P.K / TK;
            const bf16_t* A = reinterpret_cast<const bf16_t*>(P.A);
            const bf16_t* W = reinterpret_cast<const bf16_t*>(P.B);
#pragma unroll
            for (int i = 0; i < AG; ++i) {
                int am = m0 + (i * 8 + wave) * 8 + lrow; if (am > P.M - 1) am = P.M - 1;
                asrc[i] = A + (size_t)am * P.lda + lch_a * 8;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int wn_ = n0 + (wave * 4 + i) * 8 + lrow; if (wn_ > P.N - 1) wn_ = P.N - 1;
                wsrc[i] = W + (size_t)wn_ * P.ldb + lch_w[i & 1] * 8;
            }
        };
        auto issue = [&](int stage, int kt) {
            const unsigned base = lds0 + stage * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < AG; ++i)
                if ((MT & 1) == 0 || i * 8 + wave < NAG) dma16(asrc[i] + kt * TK, base + (i * 8 + wave) * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16(wsrc[i] + kt * TK, base + A_BYTES + (wave * 4 + i) * 1024);
        };
        int sig_p = -1, sig_rb = 0;          // the finished tile whose stores are not yet known to be drained
        auto signal = [&]() {                // call behind a point where EVERY wave has passed `s_waitcnt vmcnt(0)` and the workgroup barrier
            if (sig_p >= 0 && tid == 0) __hip_atomic_fetch_add(c.counters + c.cnt_base[sig_p] + sig_rb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sig_p = -1;
        };
        int k = slot;
        if (k < runl) {
            locate(k);
            issue(0, 0);
            bool pre1 = false;
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
                    if (kt == 0) signal();       // the previous tile's write-through stores were issued before this wait by every wave
                    if (kt + 1 < nkt && !(pre1 && kt == 0)) issue(st ^ 1, kt + 1);
                    const unsigned char* a = lds + st * STAGE_BYTES;
                    nt256_kstep<MT, false>(a, a + A_BYTES, wm, wn, fr, fq, acc);
                }
                __syncthreads();
                const int cpi = pi, cm0 = m0, cn0 = n0;
                const int nk = k + slots;
                const bool more = nk < runl;
                pre1 = false;
                nt256_epilogue<bf16_t, EPI1, MT, true>(g, g.p[cpi], lds + EPI_OFF, wave, lane, cm0, cn0, wm, wn, acc, [&]() {
                    if (more) { locate(nk); issue(0, 0); pre1 = nkt > 1; if (pre1) issue(1, 1); }
                });
                sig_p = cpi; sig_rb = cm0 / TM;
                if (!more) break;
                k = nk;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last tile's stores, every wave
            __syncthreads();
            signal();
        }
    }
    __syncthreads();          // phase 1's LDS (operand stages, epilogue scratch) is dead

    // ================================================================= phase 2: the deep ring of gemm_nt256d_kernel, one tile after another
    {
        constexpr int MT = MT2, NS = 3, TM = 32 * MT, A_BYTES = TM * TK * 2, STAGE_BYTES = A_BYTES + TN * TK * 2, AG = (MT + 1) / 2, NAG = 4 * MT;
        const GemmGroupNT& g = c.g2;
        const int run0 = xcd_run_begin(xcd, g.total_tiles), runl = xcd_run_len(xcd, g.total_tiles);
        // early slots (no tile in phase 1's last round of this XCD's run) first
        const int runl1 = xcd_run_len(xcd, c.g1.total_tiles);
        const int rem = runl1 % slots;                      // slots [0, rem) work in phase 1's last round (rem == 0: nobody is early)
        int k = slot >= rem ? slot - rem : (slots - rem) + slot;
        for (; k < runl; k += slots) {
            const int tile = run0 + k;
            int pi = 0;
#pragma unroll
            for (int i = 1; i < 4; ++i)
                if (i < g.count && tile >= g.p[i].tile_start) pi = i;
            const GemmProblem& P = g.p[pi];
            int m0, n0;
            nt_tile_coords(P, tile - P.tile_start, TM, m0, n0);
            const int nkt = P.K / TK;
            // ---- wait for the phase-1 row blocks that hold rows [m0, m0 + TM) of problem pi
            if (tid == 0) {
                const int mlast = (m0 + TM - 1 < P.M - 1) ? m0 + TM - 1 : P.M - 1;
                const int need = c.g1.p[pi].tiles_n & 0xFFFF;
                int ok = 1;
                for (int rb = m0 / c.rows1; rb <= mlast / c.rows1; ++rb) {
                    const int* cp = c.counters + c.cnt_base[pi] + rb;
                    unsigned spins = 0;
                    while (__hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1u << 22)) { ok = 0; break; }
                    }
                }
                if (!ok) __hip_atomic_fetch_add(c.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // ONE buffer_inv sc1 after the match: this CU's stale L1 lines go
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            const bf16_t* A = reinterpret_cast<const bf16_t*>(P.A);
            const bf16_t* W = reinterpret_cast<const bf16_t*>(P.B);
            const bf16_t* asrc[AG];
            const bf16_t* wsrc[4];
            int my_a = 0;
#pragma unroll
            for (int i = 0; i < AG; ++i) {
                if (i * 8 + wave < NAG) ++my_a;
                int am = m0 + (i * 8 + wave) * 8 + lrow; if (am > P.M - 1) am = P.M - 1;
                asrc[i] = A + (size_t)am * P.lda + lch_a * 8;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int wn_ = n0 + (wave * 4 + i) * 8 + lrow; if (wn_ > P.N - 1) wn_ = P.N - 1;
                wsrc[i] = W + (size_t)wn_ * P.ldb + lch_w[i & 1] * 8;
            }
            auto issue = [&](int kt) {
                const unsigned base = lds0 + (kt % NS) * STAGE_BYTES;
#pragma unroll
                for (int i = 0; i < AG; ++i)
                    if (i * 8 + wave < NAG) dma16(asrc[i] + kt * TK, base + (i * 8 + wave) * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) dma16(wsrc[i] + kt * TK, base + A_BYTES + (wave * 4 + i) * 1024);
            };
            auto wait_keep = [&](int n) {
                if (n <= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
                if (my_a == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else if (my_a == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            };
            f32x4 acc[MT][4];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NS - 1; ++i)
                if (i < nkt) issue(i);
            for (int kt = 0; kt < nkt; ++kt) {
                const int ahead = nkt - 1 - kt;
                wait_keep(ahead < NS - 2 ? ahead : NS - 2);
                __builtin_amdgcn_s_barrier();
                if (kt + NS - 1 < nkt) issue(kt + NS - 1);
                const unsigned char* a = lds + (kt % NS) * STAGE_BYTES;
                nt256_kstep<MT, false>(a, a + A_BYTES, wm, wn, fr, fq, acc);
            }
            __syncthreads();
            nt256_epilogue<bf16_t, EPI2, MT>(g, P, lds, wave, lane, m0, n0, wm, wn, acc, []() {});
            __syncthreads();          // the scratch aliases the ring the next tile fills
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- host side (C entry points)
static void fill_group(GemmGroup& g, int np, const int* rows, const void* const* A, const void* const* W, const float* const* bias, void* const* C,
                       void* const* C2, const void* const* aux, int N, int K, int epi, float drop_p) {
    memset(&g, 0, sizeof g);
    g.count = np; g.drop = make_drop(drop_p, 0x1234567ull, 0);
    for (int i = 0; i < np; ++i) {
        GemmProblem& p = g.p[i];
        p.A = A[i]; p.lda = K; p.B = W[i]; p.ldb = K; p.C = C[i]; p.ldc = N; p.C2 = C2 ? C2[i] : nullptr; p.bias = bias ? bias[i] : nullptr;
        p.aux = aux ? aux[i] : nullptr; p.ldaux = N; p.M = rows[i]; p.N = N; p.K = K; p.epi = epi; p.drop_site = 17 + 4 * i;
    }
}

// the chained launch; counters: >= 256 ints + 1 (timeout word at counters[255]) of device memory, zeroed here on the stream
template <int EPI1, int EPI2, int MT1, int MT2>
static int launch_chain(ChainArgs& c, hipStream_t s) {
    constexpr int LDS1 = NT256_LDS(MT1), LDS2 = 3 * (32 * MT2 * TK * 2 + TN * TK * 2);
    constexpr int LDS_BYTES = LDS1 > LDS2 ? LDS1 : LDS2;
    static bool attr = false;
    if (!attr) { RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_chain_kernel<EPI1, EPI2, MT1, MT2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); attr = true; }
    hipLaunchKernelGGL((ffn_chain_kernel<EPI1, EPI2, MT1, MT2>), dim3(rgqa_num_cus()), dim3(T256_THREADS), LDS_BYTES, s, c);
    RGQA_LAUNCH_CHECK("ffn_chain_kernel");
    return RGQA_OK;
}
extern "C" {
// the two product launches (what the engine issues today): grouped FFN1 then grouped FFN2 on stream `stream`
int lab_ffn_two_launches(int np, const int* rows, const void* const* X, const void* const* W1, const float* const* b1, void* const* H, void* const* Hp,
                         const void* const* W2, const float* const* b2, void* const* Z, int H_, int I_, int epi1, int epi2, const void* const* aux1,
                         const void* const* aux2, float drop_p, void* stream) {
    GemmGroup g;
    fill_group(g, np, rows, X, W1, b1, H, Hp, aux1, I_, H_, epi1, 0.f);
    if (int r = launch_gemm_nt_bf16(g, 0, (hipStream_t)stream)) return r;
    fill_group(g, np, rows, (const void* const*)H, W2, b2, Z, nullptr, aux2, H_, I_, epi2, drop_p);
    return launch_gemm_nt_bf16(g, 0, (hipStream_t)stream);
}

int lab_ffn_chain(int np, const int* rows, const void* const* X, const void* const* W1, const float* const* b1, void* const* H, void* const* Hp,
                  const void* const* W2, const float* const* b2, void* const* Z, int H_, int I_, int epi1, int epi2, const void* const* aux1,
                  const void* const* aux2, float drop_p, int mt1, int mt2, int* counters, void* stream) {
    RGQA_REQUIRE(np >= 1 && np <= 4, "lab_ffn_chain: 1..4 problems");
    GemmGroup g1, g2;
    fill_group(g1, np, rows, X, W1, b1, H, Hp, aux1, I_, H_, epi1, 0.f);
    fill_group(g2, np, rows, (const void* const*)H, W2, b2, Z, nullptr, aux2, H_, I_, epi2, drop_p);
    RGQA_REQUIRE(gemm_nt256_eligible(g1, 0) && gemm_nt256_eligible(g2, 0), "lab_ffn_chain: shapes not eligible for the LDS-DMA kernels");
    gemm_group_finalize(g1, 32 * mt1, TN);
    gemm_group_finalize(g2, 32 * mt2, TN);
    ChainArgs c; memset(&c, 0, sizeof c);
    c.g1 = nt_prefix(g1); c.g2 = nt_prefix(g2);
    int nb = 0;
    for (int i = 0; i < np; ++i) { c.cnt_base[i] = nb; nb += cdiv(rows[i], 32 * mt1); }
    RGQA_REQUIRE(nb <= 255, "lab_ffn_chain: too many row blocks (%d)", nb);
    c.counters = counters; c.timeout = reinterpret_cast<unsigned*>(counters + 255); c.rows1 = 32 * mt1;
    RGQA_HIP(hipMemsetAsync(counters, 0, 256 * sizeof(int), (hipStream_t)stream));
    hipStream_t s = (hipStream_t)stream;
    if (epi1 == EPI_GELU && epi2 == EPI_RESID_DROP) {
        if (mt1 == 7 && mt2 == 5) return launch_chain<EPI_GELU, EPI_RESID_DROP, 7, 5>(c, s);
        if (mt1 == 8 && mt2 == 5) return launch_chain<EPI_GELU, EPI_RESID_DROP, 8, 5>(c, s);
        if (mt1 == 6 && mt2 == 5) return launch_chain<EPI_GELU, EPI_RESID_DROP, 6, 5>(c, s);
        if (mt1 == 7 && mt2 == 4) return launch_chain<EPI_GELU, EPI_RESID_DROP, 7, 4>(c, s);
    }
    if (epi1 == EPI_DGELU && epi2 == EPI_ADD) {
        if (mt1 == 7 && mt2 == 5) return launch_chain<EPI_DGELU, EPI_ADD, 7, 5>(c, s);
    }
    rgqa_set_error("lab_ffn_chain: no instantiation for epilogues %d -> %d at tile heights %d / %d", epi1, epi2, 32 * mt1, 32 * mt2);
    return RGQA_ERR_ARG;
}
int lab_force_mt(int mt) { g_rgqa_force_mt = mt; return 0; }
}
