// Device helpers shared by the LDS-DMA GEMM kernels (gemm_nt256.h, gemm_mfma256.hip, gemm_x3.hip): LDS-DMA issue, the swizzled
// operand image, the XCD-aware tile map and the common epilogue.
#pragma once
#include <type_traits>
#include "gemm.h"

#define TN 256
#define TK 64
#define T256_THREADS 512
// MT >= 4: persistent tile loop, the epilogue's transpose scratch lives BEHIND the two operand stages (so the next tile's
// first two K-steps are already being DMA'd while this tile's outputs are converted and stored); 160 KiB of LDS in all.
// MT == 2 keeps two co-resident 80-KiB blocks per CU (scratch aliases the dead stages, one tile per block).
#define NT256_PERSIST(MT) ((MT) >= 4)
#define NT256_TP(MT) ((MT) == 4 ? 2 : ((MT) >= 5 ? 1 : (MT)))
#define NT256_LDS(MT) (2 * (32 * (MT) * TK * 2 + TN * TK * 2) + (NT256_PERSIST(MT) ? 8 * NT256_TP(MT) * 4096 : 0))
// MT = 16-row m-tiles per wave (2 waves along M): tile height TM = 32*MT in {64,128,160,192,224,256}; stage = A then W

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// LDS-DMA issued from inline asm: invisible to hipcc's s_waitcnt bookkeeping, so the compiler does not drain vmcnt(0)
// in front of every ds_read of a K-step (it cannot prove the reads do not alias the in-flight DMA). Ordering is then
// entirely ours: counted "s_waitcnt vmcnt(N)" + s_barrier before a stage is read.  lds_dst: wave-uniform LDS byte
// address of the 1-KiB piece; gsrc: this lane's 16 source bytes. M0 is saved and restored inside the statement.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(p);
}

__device__ __forceinline__ int xcd_remap256(int b, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}
// LDS image of a [rows][64] bf16 operand tile: 128-B rows, 16-B chunk c of row r at r*128 + ((c ^ ((r>>1)&7))<<4).
// A ds_read_b128 is served per 16-lane group = 16 consecutive rows at one logical chunk: bank slot (mod 256 B) is
// (r&1)*8 + (c ^ ((r>>1)&7)) - 16 distinct slots, conflict-free.  (c ^ (r&7), used first, repeats every 8 rows at equal
// parity: a 2-way conflict on every fragment read.)
__device__ __forceinline__ int off256(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// ---- shared epilogue of the LDS-DMA NT kernels. The accumulators are transposed through a wave-private LDS region
// (the operand stages are dead after the last barrier) so that every global access is a full 128-B line: 8 lanes x
// 16 B per output row, instead of 16 rows x 32 B per store straight out of the MFMA layout (which ran HBM writes
// at ~1.5-2.4 TB/s).
// OutT = bf16_t / float / sf32 (split f32: the aux operand is then sf32 as well).  With an sf32 aux operand the tile's aux rows are
// NOT all requested up front (twice the registers): they are loaded pass by pass, and `after_loads` - the persistent kernel's DMA of
// the next tile's first K-steps - runs after the last of them (compiler-tracked loads issued behind the untracked LDS-DMA would make
// every aux wait drain the DMA as well).
// WT (tools/lab/ffn_chain.hip only; the product launches pass false): the bf16 result rows are stored WRITE-THROUGH (sc1) from inline asm, so that
// another workgroup of the same launch may read them after a counter hand-off without an L2 write-back (MI355X_MICROARCH.md, visibility: R1).
// Such stores are invisible to hipcc's s_waitcnt bookkeeping: the caller drains them (s_waitcnt vmcnt(0)) before it signals.
template <bool WT>
__device__ __forceinline__ void nt256_store16(bf16_t* p, const bf16x8& v) {
    if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    else *reinterpret_cast<bf16x8*>(p) = v;
}
template <typename OutT, int EPI, int MT, bool WT = false, typename F>
__device__ __forceinline__ void nt256_epilogue(const GemmGroupNT& g, const GemmProblem& P, unsigned char* lds, int wave, int lane,
                                               int m0, int n0, int wm, int wn, f32x4 (&acc)[MT][4], F&& after_loads) {
    const int M = P.M, N = P.N;
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int TP = NT256_TP(MT);                    // m-tiles (16 rows) per pass; TP*4 KiB of f32 per wave
    constexpr bool AUX = (EPI == EPI_RESID_DROP || EPI == EPI_DGELU || EPI == EPI_ADD || EPI == EPI_DTANH || EPI == EPI_DRELU_DROP);
    constexpr bool SF = std::is_same<OutT, sf32>::value;       // split-f32 result and aux operand
    constexpr bool AUX_UP = AUX && !SF;                        // bf16 aux: the whole tile's rows in flight before anything else
    unsigned char* wl = lds + wave * (TP * 4096);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    const int nb = n0 + wn * 64 + ecol;                 // first of this lane's 8 output columns
    // N % 8 == 0 (eligibility): a lane's 8 columns are all inside or all outside; loads use clamped (always valid) addresses
    // so that they are branch-free and stay in flight together, only the stores are guarded.
    const int nbc = nb < N ? nb : N - 8;
    float bias8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias8[j] = 0.f;
    if (P.bias != nullptr) { load4(P.bias + nbc, bias8); load4(P.bias + nbc + 4, bias8 + 4); }
    DropCfg dcf = g.drop; dcf.seed_hi ^= P.drop_site;
    // the whole tile's aux rows (residual / gelu'), coalesced 16 B per lane, in flight before anything else happens
    uint4 auxv[AUX_UP ? 2 * MT : 1];
    if (AUX_UP) {
#pragma unroll
        for (int it = 0; it < 2 * MT; ++it) {
            int m = m0 + wm * (16 * MT) + it * 8 + erow; if (m > M - 1) m = M - 1;
            auxv[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(P.aux) + (size_t)m * P.ldaux + nbc);
        }
    }
    if (!(AUX && SF)) after_loads();      // persistent kernel: the next tile's first K-steps are DMA'd from here on
#pragma unroll
    for (int pass = 0; pass < MT / TP; ++pass) {
#pragma unroll
        for (int t = 0; t < TP; ++t)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                const int r = t * 16 + fr, c = tn * 4 + fq;
                *reinterpret_cast<f32x4*>(wl + r * 256 + ((c ^ (r & 15)) << 4)) = acc[pass * TP + t][tn];
            }
        // Pin the waits for the (lane-conditional) bias / aux loads HERE, on every path: left to their first use inside the
        // row guard below, they stay "possibly pending" on the path that skips it, and in the persistent kernel the
        // compiler then drains vmcnt(0) - our in-flight operand DMA included - before the first ds_read of every K-step.
        if (pass == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(bias8[j]));
            if (AUX_UP) {
#pragma unroll
                for (int it = 0; it < 2 * MT; ++it) asm volatile("" :: "v"(auxv[it].x), "v"(auxv[it].y), "v"(auxv[it].z), "v"(auxv[it].w));
            }
        }
        // split-f32 aux rows of this pass: 2 x 16 B (hi, lo) per lane and row
        uint4 auxh[AUX && SF ? 2 * TP : 1], auxl[AUX && SF ? 2 * TP : 1];
        if (AUX && SF) {
#pragma unroll
            for (int it = 0; it < 2 * TP; ++it) {
                int m = m0 + wm * (16 * MT) + pass * TP * 16 + it * 8 + erow; if (m > M - 1) m = M - 1;
                const unsigned char* h = sf_hi(reinterpret_cast<const sf32*>(P.aux) + (size_t)m * P.ldaux + nbc);
                auxh[it] = *reinterpret_cast<const uint4*>(h); auxl[it] = *reinterpret_cast<const uint4*>(h + 64);
            }
        }
#pragma unroll
        for (int it = 0; it < 2 * TP; ++it) {
            const int r = it * 8 + erow;
            const int m = m0 + wm * (16 * MT) + pass * TP * 16 + r;
            const int c0 = (lane & 7) * 2;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(wl + r * 256 + ((c0 ^ (r & 15)) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(wl + r * 256 + (((c0 + 1) ^ (r & 15)) << 4));
            if (m >= M || nb >= N) continue;
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            float pre[8];
            float ax[8];
            if (AUX && SF) {
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(&auxh[AUX && SF ? it : 0]), xl = *reinterpret_cast<const bf16x8*>(&auxl[AUX && SF ? it : 0]);
#pragma unroll
                for (int j = 0; j < 8; ++j) ax[j] = (float)xh[j] + (float)xl[j];
            } else {
                const bf16x8 xb = *reinterpret_cast<const bf16x8*>(&auxv[AUX_UP ? pass * 2 * TP + it : 0]);
#pragma unroll
                for (int j = 0; j < 8; ++j) ax[j] = (float)xb[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += bias8[j];
            if (EPI == EPI_RELU || EPI == EPI_RELU_DROP) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (EPI == EPI_RESID_DROP || EPI == EPI_RELU_DROP) drop_apply_vec<8>(dcf, (uint32_t)m * (uint32_t)N + (uint32_t)nb, v);   // N % 8 == 0: even index
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = v[j];
                pre[j] = x;
                if (EPI == EPI_GELU) gelu_and_grad_fast(pre[j], x, pre[j]);      // C = gelu, C2 = gelu' (consumed by EPI_DGELU)
                else if (EPI == EPI_TANH) x = tanhf(x);
                else if (EPI == EPI_RESID_DROP) x = x + ax[j];
                else if (EPI == EPI_DGELU) x = x * ax[j];
                else if (EPI == EPI_ADD) x = x + ax[j];
                else if (EPI == EPI_DTANH) x = x * (1.0f - ax[j] * ax[j]);
                else if (EPI == EPI_DRELU_DROP) x = ax[j] > 0.f ? x * g.drop.scale : 0.f;
                v[j] = x;
            }
            if (std::is_same<OutT, float>::value) {        // f32 result (the logits GEMM): two 16-B stores per lane, a full 256-B run per 8 lanes
                float* cf = reinterpret_cast<float*>(P.C) + (size_t)m * P.ldc + nb;
                store4(cf, v); store4(cf + 4, v + 4);
                continue;
            }
            if (SF) {       // split f32: 16 B of hi parts + 16 B of lo parts per lane; 4 lanes write one whole 128-B line
                sf_store8(reinterpret_cast<sf32*>(P.C) + (size_t)m * P.ldc + nb, v);
                if (P.Cb != nullptr) {          // bf16x3_fwd precision: the bf16 image the backward pass reads
                    bf16x8 ob;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ob[j] = (bf16_t)v[j];
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(P.Cb) + (size_t)m * P.ldc + nb) = ob;
                }
                if (EPI == EPI_GELU && P.C2 != nullptr) {
                    if (P.c2_lp) {
                        bf16x8 op;
#pragma unroll
                        for (int j = 0; j < 8; ++j) op[j] = (bf16_t)pre[j];
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(P.C2) + (size_t)m * P.ldc + nb) = op;
                    } else sf_store8(reinterpret_cast<sf32*>(P.C2) + (size_t)m * P.ldc + nb, pre);
                }
                continue;
            }
            bf16x8 o, op;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o[j] = (bf16_t)v[j]; op[j] = (bf16_t)pre[j]; }
            bf16_t* cp = reinterpret_cast<bf16_t*>(P.C) + (size_t)m * P.ldc + nb;
            nt256_store16<WT>(cp, o);
            if (EPI == EPI_GELU && P.C2 != nullptr) nt256_store16<WT>(reinterpret_cast<bf16_t*>(P.C2) + (size_t)m * P.ldc + nb, op);
        }
    }
    if (AUX && SF) after_loads();
}

// ---- LayerNorm of a finished row block, inside the GEMM launch that produced it (round 5).
// The N = 768 projections that end an attention / FFN sub-block are followed by a LayerNorm over whole rows - three 256-column tiles that three
// different workgroups compute.  Until round 5 a separate launch read the pre-LN sum back (34 launches of 8.5 us at the HBM floor + a kernel
// boundary each).  Now every workgroup stores its tile write-through (sc1), every wave drains, the workgroup barrier, ONE lane draws a ticket of the
// row block (agent-scope add) - and the workgroup whose ticket is the last (told by the value its add returned: MI355X_MICROARCH.md, Valid forms,
// first table row) normalises the row block: sc1 loads of the rows, half a wave per row, 16 bytes per lane and access - the arithmetic of
// ln_fwd16_kernel (csrc/norm.hip) in the same order: bit-identical to the separate launch.  Nobody ever waits: no residency assumption.
__device__ __forceinline__ float nt256_half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// rows [m0, m0 + TM) of problem P (N == 768), by the 512 threads of the workgroup: 16 half-waves, one row each per trip, U trips' loads in flight
// at once (the rows come from the memory side of the L2s - sc1 - at its latency: one trip at a time cost ~8 us per launch, measured)
template <int TM>
__device__ __forceinline__ void nt256_ln_rows(const GemmProblem& P, int m0) {
    constexpr int NC = 3, N = 768, TRIPS = TM / 16;
    constexpr int U = (TRIPS % 5 == 0) ? 5 : (TRIPS % 7 == 0) ? 7 : (TRIPS % 6 == 0) ? 6 : 4;
    static_assert(TRIPS % U == 0, "row trips per batch");
    const int hl = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(P.C, 0, (int)((size_t)P.M * P.ldc * 2), 0x00020000);
    bf16_t* y = reinterpret_cast<bf16_t*>(P.ln_y);
    float g[NC][8], b[NC][8];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = (hl + 32 * i) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 gv = *reinterpret_cast<const float4*>(P.ln_g + c + 4 * h), bv = *reinterpret_cast<const float4*>(P.ln_b + c + 4 * h);
            g[i][4 * h] = gv.x; g[i][4 * h + 1] = gv.y; g[i][4 * h + 2] = gv.z; g[i][4 * h + 3] = gv.w;
            b[i][4 * h] = bv.x; b[i][4 * h + 1] = bv.y; b[i][4 * h + 2] = bv.z; b[i][4 * h + 3] = bv.w;
        }
    }
    using raw_t = decltype(__builtin_amdgcn_raw_buffer_load_b128(rs_c, 0u, 0, 16));
    for (int t0 = 0; t0 < TRIPS; t0 += U) {
        if (m0 + t0 * 16 >= P.M) break;                     // block-uniform
        raw_t raw[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int row = m0 + (t0 + u) * 16 + hw;
            row = row < P.M ? row : P.M - 1;                // a valid address; the result is dropped below
#pragma unroll
            for (int i = 0; i < NC; ++i)
                raw[u][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, (unsigned)(((size_t)row * P.ldc + (hl + 32 * i) * 8) * 2), 0, 16);      // aux 16 = sc1
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = m0 + (t0 + u) * 16 + hw;        // uniform over the half-wave: the shuffles below stay inside a half
            float v[NC][8];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const bf16x8 t = *reinterpret_cast<const bf16x8*>(&raw[u][i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = (float)t[j];
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[i][j];
            const float mu = nt256_half_sum(s) * (1.0f / (float)N);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mu; q += d * d; }
            const float rs = rsqrtf(nt256_half_sum(q) * (1.0f / (float)N) + P.ln_eps);
            if (row < P.M) {
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((v[i][j] - mu) * rs * g[i][j] + b[i][j]);
                    *reinterpret_cast<bf16x8*>(y + (size_t)row * P.ldc + (hl + 32 * i) * 8) = o;
                }
                if (hl == 0) { P.ln_mean[row] = mu; P.ln_rstd[row] = rs; }
            }
        }
    }
}
// called by EVERY thread of the workgroup after the tile's epilogue (its stores were issued write-through); `flag`: 4 bytes of LDS nobody else uses now
template <int TM>
__device__ __forceinline__ void nt256_ln_after_tile(const GemmProblem& P, int m0, int* flag) {
    if (P.ln_tk == nullptr) return;                           // block-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave
    __syncthreads();
    if (threadIdx.x == 0) {
        int* tk = P.ln_tk + m0 / TM;
        const int t = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (P.tiles_n & 0xFFFF) - 1);
        if (last) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch that uses this ticket
        *flag = last;
    }
    __syncthreads();
    if (*flag) nt256_ln_rows<TM>(P, m0);
    __syncthreads();                                          // `flag` may be overwritten (it lives in scratch the next tile reuses)
}

// ---- row-major-over-the-contraction LDS images (wgrad operands; the weight operand of the NN dgrad): 32-byte granule swizzle + transposed fragment reads
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

__device__ __forceinline__ int tn_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <int PITCH>
__device__ __forceinline__ bf16x8 tr_frag_dma(const unsigned char* tile, int r0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = r0 + 8 * g + q;
    const int ch = (c0 >> 3) + (p >> 1);
    const int sw = tn_f(row) << 1;      // identical for row and row + 4
    const unsigned char* a1 = tile + row * PITCH + ((ch ^ sw) << 4) + 8 * (p & 1);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(a1));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(a1 + 4 * PITCH));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
