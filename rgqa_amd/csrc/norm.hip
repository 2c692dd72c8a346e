// Wavefront-reduced LayerNorm (eps inside the sqrt, biased variance: torch.nn.LayerNorm semantics,
// reference lxrt/modeling.py:261) forward / backward, one 64-lane wave per row, f32 statistics.
// HBM-bound: algorithmic bytes per row = N * (sizeof(T) in + sizeof(T) out) forward.
#include "kernels.h"
#include <type_traits>

#define LN_MAXV 8  // up to 8 vec4 chunks per lane -> N <= 2048 (NV = chunks per lane, compile-time)

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int ldx, const float* gamma,
                                                     const float* beta, T* __restrict__ y, int ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int N, float eps,
                                                     int split, const float* __restrict__ gamma2, const float* __restrict__ beta2, bf16_t* __restrict__ yb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    if (row >= split) { gamma = gamma2; beta = beta2; }        // second module's parameters (language | vision rows of one launch)
    const int nv = N >> 2;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            load4(x + (size_t)row * ldx + c * 4, v[i]);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mu = wave_sum(s) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[i][j] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)N + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float g[4], b[4], o[4];
            load4(gamma + c * 4, g);
            load4(beta + c * 4, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
            store4(y + (size_t)row * ldy + c * 4, o);
            if (yb) store4(yb + (size_t)row * ldy + c * 4, o);
        }
    }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// bf16 rows with N = 256 * NC columns (768, 1536): HALF a wave per row, every access a full 16 bytes per lane (the 8-byte bf16x4 accesses
// of the generic kernel run at 0.54-0.70x the 16-byte rate, MI355X_MICROARCH.md).  Lane l of a half owns the 8-column chunks l, l + 32, ...
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// 8 consecutive elements (first index a multiple of 8) of a bf16 / split-f32 row: one / two 16-byte accesses per lane
__device__ __forceinline__ void ld8(const bf16_t* p, float v[8]) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
}
__device__ __forceinline__ void ld8(const sf32* p, float v[8]) { sf_load8(p, v); }
__device__ __forceinline__ void st8(bf16_t* p, const float v[8]) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (bf16_t)v[j];
    *reinterpret_cast<bf16x8*>(p) = t;
}
__device__ __forceinline__ void st8(sf32* p, const float v[8]) { sf_store8(p, v); }
// T = bf16_t or sf32 (round 4: the split-f32 rows of the bf16x3 precisions ran on the generic kernel's 8-byte accesses); yb: optional bf16
// image of a split-f32 result (bf16x3_fwd precision)
template <typename T, int NC>
__global__ __launch_bounds__(256) void ln_fwd16_kernel(const T* __restrict__ x, int ldx, const float* gamma, const float* beta, T* __restrict__ y, int ldy,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int M, float eps, int split,
                                                       const float* __restrict__ gamma2, const float* __restrict__ beta2, bf16_t* __restrict__ yb) {
    constexpr int N = 256 * NC;
    const int hl = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= M) return;                                   // a whole half-wave leaves: the shuffles below stay inside a half
    if (row >= split) { gamma = gamma2; beta = beta2; }
    float v[NC][8];
#pragma unroll
    for (int i = 0; i < NC; ++i) ld8(x + (size_t)row * ldx + (hl + 32 * i) * 8, v[i]);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][j];
    const float mu = half_sum(s) * (1.0f / (float)N);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mu; q += d * d; }
    const float rs = rsqrtf(half_sum(q) * (1.0f / (float)N) + eps);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = (hl + 32 * i) * 8;
        float g[8], b[8], o[8];
        load4(gamma + c, g); load4(gamma + c + 4, g + 4);
        load4(beta + c, b); load4(beta + c + 4, b + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
        st8(y + (size_t)row * ldy + c, o);
        if (yb) st8(yb + (size_t)row * ldy + c, o);
    }
    if (hl == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// Backward. dz = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma, xhat = (z-mean)*rstd.
// Also emits (optionally) dzd = dropout-masked dz (the gradient entering the preceding dense layer,
// whose forward epilogue applied that mask) and per-block partial column sums:
//   part[blk][0][n] = sum dy*xhat (dgamma), part[blk][1][n] = sum dy (dbeta), part[blk][2][n] = sum dzd (dense bias grad)
// One or two row segments per launch (adjacent modules: language | vision): blocks [0, nblk0) work on segment 0, the rest on
// segment 1; partial column sums are indexed by the global block id, so each segment's partials are a contiguous block range.
template <typename T>
struct LnBwdSeg {
    const T* dy; const T* z; const float* gamma; const float* mean; const float* rstd; T* dz; T* dzd;
    int M; DropCfg drop, drop_in;
    int z_split = 0;      // (T = bf16_t, ln_bwd16 only) z points at a SPLIT-f32 tensor (common.h sf32, ldz in its elements): its hi parts are read in place
};
template <typename T, int NV, int LN_BWD_THREADS>
__global__ __launch_bounds__(LN_BWD_THREADS) void ln_bwd_kernel(const LnBwdSeg<T> sg0, const LnBwdSeg<T> sg1, int nblk0, int lddy, int ldz, int lddz,
                                                     float* __restrict__ part, int N, float dy_scale) {
    const bool second = (int)blockIdx.x >= nblk0;            // block-uniform
    const LnBwdSeg<T>& sg = second ? sg1 : sg0;
    const T* __restrict__ dy = sg.dy; const T* __restrict__ z = sg.z;
    const float* __restrict__ gamma = sg.gamma; const float* __restrict__ mean = sg.mean; const float* __restrict__ rstd = sg.rstd;
    T* __restrict__ dz = sg.dz; T* __restrict__ dzd = sg.dzd;
    const int M = sg.M;
    const DropCfg drop = sg.drop, drop_in = sg.drop_in;
    const int lblk = second ? (int)blockIdx.x - nblk0 : (int)blockIdx.x, lgrid = second ? (int)gridDim.x - nblk0 : nblk0;
    // N <= 768: 16 waves per block (4 per SIMD: the loop is one dependent HBM round trip per row, so it lives on
    // occupancy; fits the 128-VGPR budget); wider rows: 4 waves per block.  One row per wave per trip, next row's loads
    // issued before the current row's reductions.
    constexpr int NW = LN_BWD_THREADS / 64;
    __shared__ float red[NW / 2][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = N >> 2;
    float ag[NV][4], ab[NV][4], ad[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; ad[i][j] = 0.f; }
    }
    const int stride = lgrid * NW;
    int row = lblk * NW + wave;
    raw4<T> rd[NV], rz[NV];
    float mu = 0.f, rs = 0.f;
    if (row < M) {
        mu = mean[row]; rs = rstd[row];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) { load_raw4(dy + (size_t)row * lddy + c * 4, rd[i]); load_raw4(z + (size_t)row * ldz + c * 4, rz[i]); }
        }
    }
    for (; row < M; row += stride) {
        float g[NV][4], xh[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float d[4], zz[4], gm[4];
                load4(gamma + c * 4, gm);   // L1/L2-resident; not kept in registers (the 128-VGPR budget of a 16-wave block)
                cvt_raw4(rd[i], d);
                cvt_raw4(rz[i], zz);
                drop_apply_vec<4>(drop_in, (uint32_t)row * (uint32_t)N + (uint32_t)(c * 4), d);     // N % 4 == 0: even index
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    d[j] *= dy_scale;
                    xh[i][j] = (zz[j] - mu) * rs;
                    g[i][j] = d[j] * gm[j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                    ag[i][j] += d[j] * xh[i][j];
                    ab[i][j] += d[j];
                }
            }
        }
        const float rs_cur = rs;
        const int nrow = row + stride;
        if (nrow < M) {
            mu = mean[nrow]; rs = rstd[nrow];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < nv) { load_raw4(dy + (size_t)nrow * lddy + c * 4, rd[i]); load_raw4(z + (size_t)nrow * ldz + c * 4, rz[i]); }
            }
        }
        s1 = wave_sum(s1) / (float)N;
        s2 = wave_sum(s2) / (float)N;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float o[4], od[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = rs_cur * (g[i][j] - s1 - xh[i][j] * s2); od[j] = o[j]; }
                drop_apply_vec<4>(drop, (uint32_t)row * (uint32_t)N + (uint32_t)(c * 4), od);
#pragma unroll
                for (int j = 0; j < 4; ++j) ad[i][j] += od[j];
                store4(dz + (size_t)row * lddz + c * 4, o);
                if (dzd) store4(dzd + (size_t)row * lddz + c * 4, od);
            }
        }
    }
    if (part == nullptr) return;
    // cross-wave reduction of the three column accumulators through LDS, fixed order (bit-reproducible):
    // waves NW/2.. park theirs, waves 0..NW/2-1 add and park the pair sums, then every thread folds NW/2 values per column.
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        __syncthreads();
        if (wave >= NW / 2) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < nv) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        red[wave - NW / 2][c * 4 + j] = which == 0 ? ag[i][j] : (which == 1 ? ab[i][j] : ad[i][j]);
                }
            }
        }
        __syncthreads();
        if (wave < NW / 2) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < nv) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        red[wave][c * 4 + j] += which == 0 ? ag[i][j] : (which == 1 ? ab[i][j] : ad[i][j]);
                }
            }
        }
        __syncthreads();
        for (int n = threadIdx.x; n < N; n += LN_BWD_THREADS) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW / 2; ++w) t += red[w][n];
            part[((size_t)blockIdx.x * 3 + which) * N + n] = t;
        }
    }
}

// bf16 rows with N = 256 * NC columns (768, 1536): HALF a wave per row, every access 16 bytes per lane, as in ln_fwd16_kernel.  Lane l of a
// half owns the 8-column chunks l, l + 32, ...: 3 x 8 x NC column accumulators per lane, which is why the block is 8 waves (256-VGPR budget;
// a 16-wave block at 128 VGPRs spilled: 52 us against 24 us per launch, round 1).  Two rows per wave per trip, the next pair's loads issued
// before the current pair's reductions; the two halves of a wave own the same columns, so their column sums meet in registers (one
// shuffle) before the 8 waves fold through LDS in a fixed order.  Same partials layout as ln_bwd_kernel.
// ZSF (round 6, bf16x3_fwd precision): the pre-LayerNorm sums are read straight out of the forward pass's split-f32 tensor - the hi parts ARE the bf16
// image the forward pass used to store beside it (16 bytes of a 64-byte hi run per lane: the same accesses, at twice the row pitch) - so the projections'
// epilogues no longer write that image: 2 of their 10 bytes per element.
template <int NC, bool ZSF>
__global__ __launch_bounds__(512) void ln_bwd16_kernel(const LnBwdSeg<bf16_t> sg0, const LnBwdSeg<bf16_t> sg1, int nblk0, int lddy, int ldz, int lddz,
                                                       float* __restrict__ part, float dy_scale) {
    constexpr int N = 256 * NC, NHW = 16, NW = 8;
    const bool second = (int)blockIdx.x >= nblk0;            // block-uniform
    const LnBwdSeg<bf16_t>& sg = second ? sg1 : sg0;
    const bf16_t* __restrict__ dy = sg.dy; const bf16_t* __restrict__ z = sg.z;
    const float* __restrict__ mean = sg.mean; const float* __restrict__ rstd = sg.rstd;
    bf16_t* __restrict__ dz = sg.dz; bf16_t* __restrict__ dzd = sg.dzd;
    const int M = sg.M;
    const DropCfg drop = sg.drop, drop_in = sg.drop_in;
    const int lblk = second ? (int)blockIdx.x - nblk0 : (int)blockIdx.x, lgrid = second ? (int)gridDim.x - nblk0 : nblk0;
    __shared__ float red[NW / 2][N];
    const int hl = threadIdx.x & 31, hw = threadIdx.x >> 5, wave = threadIdx.x >> 6;
    float ag[NC][8], ab[NC][8], ad[NC][8], gm[NC][8];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        load4(sg.gamma + (hl + 32 * i) * 8, gm[i]); load4(sg.gamma + (hl + 32 * i) * 8 + 4, gm[i] + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; ad[i][j] = 0.f; }
    }
    const int stride = lgrid * NHW;
    int row = lblk * NHW + hw;
    auto ldz8 = [&](size_t r, int e0) -> bf16x8 {
        if constexpr (ZSF) return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(z) + r * (size_t)ldz * 4 + (size_t)(e0 >> 5) * 128 + (e0 & 31) * 2);
        else return *reinterpret_cast<const bf16x8*>(z + r * (size_t)ldz + e0);
    };
    bf16x8 rd[NC], rz[NC];
    float mu = 0.f, rs = 0.f;
    if (row < M) {
        mu = mean[row]; rs = rstd[row];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            rd[i] = *reinterpret_cast<const bf16x8*>(dy + (size_t)row * lddy + (hl + 32 * i) * 8);
            rz[i] = ldz8((size_t)row, (hl + 32 * i) * 8);
        }
    }
    // (a half-wave whose rows have run out keeps taking part in the shuffles below with zeros: the loop is uniform over the WAVE)
    for (int base = lblk * NHW + (wave << 1); base < M; base += stride) {
        const bool live = row < M;
        float g[NC][8], xh[NC][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            float d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = live ? (float)rd[i][j] : 0.f;
            if (live) drop_apply_vec<8>(drop_in, (uint32_t)row * (uint32_t)N + (uint32_t)((hl + 32 * i) * 8), d);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                d[j] *= dy_scale;
                xh[i][j] = live ? ((float)rz[i][j] - mu) * rs : 0.f;
                g[i][j] = d[j] * gm[i][j];
                s1 += g[i][j];
                s2 += g[i][j] * xh[i][j];
                ag[i][j] += d[j] * xh[i][j];
                ab[i][j] += d[j];
            }
        }
        const float rs_cur = rs;
        const int crow = row;
        row += stride;
        if (row < M) {
            mu = mean[row]; rs = rstd[row];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                rd[i] = *reinterpret_cast<const bf16x8*>(dy + (size_t)row * lddy + (hl + 32 * i) * 8);
                rz[i] = ldz8((size_t)row, (hl + 32 * i) * 8);
            }
        }
        s1 = half_sum(s1) * (1.0f / (float)N);
        s2 = half_sum(s2) * (1.0f / (float)N);
        if (live) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                float o[8], od[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { o[j] = rs_cur * (g[i][j] - s1 - xh[i][j] * s2); od[j] = o[j]; }
                drop_apply_vec<8>(drop, (uint32_t)crow * (uint32_t)N + (uint32_t)((hl + 32 * i) * 8), od);
                bf16x8 ob, odb;
#pragma unroll
                for (int j = 0; j < 8; ++j) { ad[i][j] += od[j]; ob[j] = (bf16_t)o[j]; odb[j] = (bf16_t)od[j]; }
                *reinterpret_cast<bf16x8*>(dz + (size_t)crow * lddz + (hl + 32 * i) * 8) = ob;
                if (dzd) *reinterpret_cast<bf16x8*>(dzd + (size_t)crow * lddz + (hl + 32 * i) * 8) = odb;
            }
        }
    }
    if (part == nullptr) return;
    // the two halves of a wave own the same columns: lower half + upper half (fixed order), then the 8 waves through LDS as in ln_bwd_kernel
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        float v[NC][8];
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float mine = which == 0 ? ag[i][j] : (which == 1 ? ab[i][j] : ad[i][j]);
                const float other = __shfl_xor(mine, 32, 64);
                v[i][j] = (threadIdx.x & 32) ? other + mine : mine + other;       // lower-half value first in both halves
            }
        __syncthreads();
        if (wave >= NW / 2 && (threadIdx.x & 32) == 0) {
#pragma unroll
            for (int i = 0; i < NC; ++i) { store4(&red[wave - NW / 2][(hl + 32 * i) * 8], v[i]); store4(&red[wave - NW / 2][(hl + 32 * i) * 8 + 4], v[i] + 4); }
        }
        __syncthreads();
        if (wave < NW / 2 && (threadIdx.x & 32) == 0) {
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[wave][(hl + 32 * i) * 8 + j] += v[i][j];
        }
        __syncthreads();
        for (int n = threadIdx.x; n < N; n += 512) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW / 2; ++w) t += red[w][n];
            part[((size_t)blockIdx.x * 3 + which) * N + n] = t;
        }
    }
}

// out[q][n*stride] (+)= sum_blk part[blk][q][n] for q < nq (null output pointers are skipped).
// Block = 16 columns (4 lanes x float4) x 64 partial groups: every thread sums nblk/64 float4 partials (fixed order ->
// bit-reproducible), then the 64 group sums are folded in a fixed order through LDS.  N % 4 == 0.
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nq, int N, FinOut fo, int accumulate) {
    __shared__ float red[64][17];
    const int cq = threadIdx.x & 3, grp = threadIdx.x >> 2;
    const int n4 = blockIdx.x * 16 + cq * 4;
    const int q = blockIdx.y;
    float* o = fo.p[q];
    if (o == nullptr) return;
    const int qs = fo.qsrc[q], b0 = fo.b0[q], b1 = fo.b1[q];     // quantity index inside a block's partials; block range to fold
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (n4 < N)
        for (int b = b0 + grp; b < b1; b += 64) {
            float v[4];
            load4(part + ((size_t)b * nq + qs) * N + n4, v);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[grp][cq * 4 + j] = s[j];
    __syncthreads();
    const int c = threadIdx.x, n = blockIdx.x * 16 + c;
    if (c < 16 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 64; ++g) t += red[g][c];
        o += (size_t)n * fo.stride[q];
        *o = accumulate ? *o + t : t;
    }
}

// nq quantities per block in `part`; output q folds quantity q of blocks [0, nblk)
int k_colsum_finalize(const float* part, int nblk, int nq, int N, const FinOut& fo_in, int accumulate, hipStream_t s) {
    RGQA_REQUIRE(N % 4 == 0, "colsum_finalize: N=%d must be a multiple of 4", N);
    FinOut fo = fo_in;
    for (int q = 0; q < nq; ++q) { fo.qsrc[q] = q; fo.b0[q] = 0; fo.b1[q] = nblk; }
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(N, 16), nq), dim3(256), 0, s, part, nq, N, fo, accumulate);
    RGQA_LAUNCH_CHECK("colsum_finalize_kernel");
    return RGQA_OK;
}
// general form: nout outputs, each with its own source quantity (fo.qsrc) and block range (fo.b0 .. fo.b1)
int k_colsum_finalize_ranges(const float* part, int nq_part, int nout, int N, const FinOut& fo, int accumulate, hipStream_t s) {
    RGQA_REQUIRE(N % 4 == 0 && nout <= FIN_MAXQ, "colsum_finalize: N=%d / %d outputs", N, nout);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(N, 16), nout), dim3(256), 0, s, part, nq_part, N, fo, accumulate);
    RGQA_LAUNCH_CHECK("colsum_finalize_kernel");
    return RGQA_OK;
}

// plain column sum of a [M,N] matrix into per-block partials part[blk][0][n] (nq = 1)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ldx, float* __restrict__ part, int M, int N) {
    // thread owns 4 consecutive columns; block strides over rows
    const int nv = N >> 2;
    for (int c = threadIdx.x; c < nv; c += 256) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int row = blockIdx.x; row < M; row += gridDim.x) {
            float v[4];
            load4(x + (size_t)row * ldx + c * 4, v);
            a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        }
        store4(part + (size_t)blockIdx.x * N + c * 4, a);
    }
}

template <typename T>
int k_ln_fwd2(const T* x, int ldx, const float* gamma, const float* beta, const float* gamma2, const float* beta2, int split, T* y, int ldy, float* mean, float* rstd,
              int M, int N, float eps, hipStream_t s, bf16_t* y_b) {
    RGQA_REQUIRE(N % 4 == 0 && N <= LN_MAXV * 256 && ldx % 4 == 0 && ldy % 4 == 0, "layernorm: N=%d must be a multiple of 4 and <= %d", N, LN_MAXV * 256);
    RGQA_REQUIRE(y_b == nullptr || (std::is_same<T, sf32>::value && ((uintptr_t)y_b % 16) == 0 && ldy % 8 == 0), "layernorm: the bf16 image goes with split-f32 rows (16-byte aligned, ldy %% 8)");
    if (M <= 0) return RGQA_OK;
    if constexpr (!std::is_same<T, float>::value) {
        constexpr int AL = std::is_same<T, sf32>::value ? 128 : 16, LDM = std::is_same<T, sf32>::value ? 32 : 8;
        if ((N == 768 || N == 1536) && ldx % LDM == 0 && ldy % LDM == 0 && ((uintptr_t)x % AL) == 0 && ((uintptr_t)y % AL) == 0) {
            if (N == 768) hipLaunchKernelGGL((ln_fwd16_kernel<T, 3>), dim3(cdiv(M, 8)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M, eps, split, gamma2, beta2, y_b);
            else hipLaunchKernelGGL((ln_fwd16_kernel<T, 6>), dim3(cdiv(M, 8)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M, eps, split, gamma2, beta2, y_b);
            RGQA_LAUNCH_CHECK("ln_fwd16_kernel");
            return RGQA_OK;
        }
    }
#define LN_FWD(NVV) hipLaunchKernelGGL((ln_fwd_kernel<T, NVV>), dim3(cdiv(M, 4)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M, N, eps, split, gamma2, beta2, y_b)
    const int nvl = cdiv(N / 4, 64);
    if (nvl <= 1) LN_FWD(1); else if (nvl == 2) LN_FWD(2); else if (nvl == 3) LN_FWD(3); else if (nvl == 4) LN_FWD(4);
    else if (nvl <= 6) LN_FWD(6); else LN_FWD(8);
#undef LN_FWD
    RGQA_LAUNCH_CHECK("ln_fwd_kernel");
    return RGQA_OK;
}
template <typename T>
int k_ln_fwd(const T* x, int ldx, const float* gamma, const float* beta, T* y, int ldy, float* mean, float* rstd, int M, int N, float eps, hipStream_t s, bf16_t* y_b) {
    return k_ln_fwd2<T>(x, ldx, gamma, beta, gamma, beta, M, y, ldy, mean, rstd, M, N, eps, s, y_b);
}

// waves per block: narrow rows live on occupancy (16 waves = 4 per SIMD at <= 128 VGPRs with 2-byte elements); the 4-byte element types (f32,
// split f32) carry twice the staged registers per row and spill at that budget (528 us instead of ~30 per launch, rocprofv3): 8 waves for them
template <typename T> static int ln_bwd_waves(int N) { return cdiv(N / 4, 64) <= 3 ? (sizeof(T) == 2 ? 16 : 8) : 4; }
template <typename T> static int ln_bwd_blocks_t(int M, int N) { const int nw = ln_bwd_waves<T>(N); const int b = cdiv(M, nw), cap = nw == 4 ? 512 : 256; return b > cap ? cap : b; }
int ln_bwd_blocks(int M, int N) { return ln_bwd_blocks_t<bf16_t>(M, N); }

template <typename T>
static int ln_bwd_launch(const LnBwdSeg<T>& a, const LnBwdSeg<T>& b, int nblk0, int nblk, int lddy, int ldz, int lddz, float* part, int N, float dy_scale, hipStream_t s) {
    if constexpr (std::is_same<T, bf16_t>::value) {
        const bool al = lddy % 8 == 0 && ldz % 8 == 0 && lddz % 8 == 0 && ((uintptr_t)a.dy % 16) == 0 && ((uintptr_t)a.z % 16) == 0 && ((uintptr_t)a.dz % 16) == 0 &&
                        (a.dzd == nullptr || ((uintptr_t)a.dzd % 16) == 0) && ((uintptr_t)b.dy % 16) == 0 && ((uintptr_t)b.z % 16) == 0 && ((uintptr_t)b.dz % 16) == 0 &&
                        (b.dzd == nullptr || ((uintptr_t)b.dzd % 16) == 0) && ((uintptr_t)a.gamma % 16) == 0 && ((uintptr_t)b.gamma % 16) == 0;
        RGQA_REQUIRE(a.z_split == b.z_split && (!a.z_split || (al && N == 768 && ldz % 32 == 0 && ((uintptr_t)a.z % 128) == 0 && ((uintptr_t)b.z % 128) == 0)),
                     "layernorm bwd: split-f32 z needs N = 768 and 128-byte aligned rows (N=%d ldz=%d)", N, ldz);
        if (al && N == 768) {            // (N = 1536, the answer head's LayerNorm over 256 rows, stays on the generic kernel: 144 column accumulators per lane would spill)
            if (a.z_split) hipLaunchKernelGGL((ln_bwd16_kernel<3, true>), dim3(nblk), dim3(512), 0, s, a, b, nblk0, lddy, ldz, lddz, part, dy_scale);
            else hipLaunchKernelGGL((ln_bwd16_kernel<3, false>), dim3(nblk), dim3(512), 0, s, a, b, nblk0, lddy, ldz, lddz, part, dy_scale);
            RGQA_LAUNCH_CHECK("ln_bwd16_kernel");
            return RGQA_OK;
        }
    } else RGQA_REQUIRE(!a.z_split, "layernorm bwd: split-f32 z is read by the bf16 kernel only");
#define LN_BWD(NVV) hipLaunchKernelGGL((ln_bwd_kernel<T, NVV, (NVV <= 3 ? (sizeof(T) == 2 ? 1024 : 512) : 256)>), dim3(nblk), dim3(NVV <= 3 ? (sizeof(T) == 2 ? 1024 : 512) : 256), 0, s, a, b, nblk0, lddy, ldz, lddz, part, N, dy_scale)
    const int nvl = cdiv(N / 4, 64);
    if (nvl <= 1) LN_BWD(1); else if (nvl == 2) LN_BWD(2); else if (nvl == 3) LN_BWD(3); else if (nvl == 4) LN_BWD(4);
    else if (nvl <= 6) LN_BWD(6); else LN_BWD(8);
#undef LN_BWD
    RGQA_LAUNCH_CHECK("ln_bwd_kernel");
    return RGQA_OK;
}

template <typename T>
int k_ln_bwd(const T* dy, int lddy, const T* z, int ldz, const float* gamma, const float* mean, const float* rstd, T* dz, T* dzd, int lddz,
             float* part, float* dgamma, float* dbeta, float* dbias, int accumulate, int M, int N, DropCfg drop, DropCfg drop_in, float dy_scale, hipStream_t s,
             FinDefer* defer, int z_split) {
    RGQA_REQUIRE(N % 4 == 0 && N <= LN_MAXV * 256, "layernorm bwd: N=%d unsupported", N);
    if (M <= 0) return RGQA_OK;
    const int nblk = ln_bwd_blocks_t<T>(M, N);
    if (part && defer && !defer->room(nblk, 3, N)) defer = nullptr;
    const int blk0 = defer ? defer->blk : 0;
    if (part && defer) part = defer->take(nblk);
    LnBwdSeg<T> a; a.dy = dy; a.z = z; a.gamma = gamma; a.mean = mean; a.rstd = rstd; a.dz = dz; a.dzd = dzd; a.M = M; a.drop = drop; a.drop_in = drop_in; a.z_split = z_split;
    int r = ln_bwd_launch<T>(a, a, nblk, nblk, lddy, ldz, lddz, part, N, dy_scale, s);
    if (r) return r;
    if (part && defer) {
        defer->add(dgamma, 0, blk0, blk0 + nblk); defer->add(dbeta, 1, blk0, blk0 + nblk); defer->add(dbias, 2, blk0, blk0 + nblk);
        return RGQA_OK;
    }
    if (part) {
        FinOut fo = {};
        fo.p[0] = dgamma; fo.p[1] = dbeta; fo.p[2] = dbias;
        fo.stride[0] = fo.stride[1] = fo.stride[2] = 1;
        return k_colsum_finalize(part, nblk, 3, N, fo, accumulate, s);
    }
    return RGQA_OK;
}
int fin_flush(FinDefer& d, int accumulate, hipStream_t s) {
    if (d.nout == 0) return RGQA_OK;
    int r = k_colsum_finalize_ranges(d.base, 3, d.nout, d.N, d.fo, accumulate, s);
    d.blk = 0; d.nout = 0; d.fo = FinOut{};
    return r;
}

// two adjacent row segments (rows [0,M0) and [M0, M0+M1) of the same buffers, different modules) in one launch + one finalize
template <typename T>
int k_ln_bwd2(const T* dy, int lddy, const T* z, int ldz, const float* mean, const float* rstd, T* dz, T* dzd, int lddz, float* part, int N, int accumulate,
              int M0, const float* gamma0, float* dgamma0, float* dbeta0, float* dbias0, DropCfg drop0,
              int M1, const float* gamma1, float* dgamma1, float* dbeta1, float* dbias1, DropCfg drop1, hipStream_t s, FinDefer* defer, int z_split) {
    RGQA_REQUIRE(N % 4 == 0 && N <= LN_MAXV * 256 && part != nullptr && M0 > 0 && M1 > 0, "layernorm bwd2: bad arguments (N=%d)", N);
    const DropCfg nodrop = make_drop(0.f, 0, 0);
    int nb0 = ln_bwd_blocks_t<T>(M0, N), nb1 = ln_bwd_blocks_t<T>(M1, N);
    if (nb0 > 256) nb0 = 256;                       // the partial-sum scratch holds 512 blocks
    if (nb1 > 256) nb1 = 256;
    if (nb0 + nb1 > 256) {
        // ONE block per CU over both segments, split in proportion to their rows: every block then walks the same number of rows and pays
        // its set-up and its column-sum fold once (two full-size grids back to back ran 453 blocks of 1-2 rows per half-wave)
        int a0 = (int)((256.0 * M0) / ((double)M0 + M1) + 0.5);
        if (a0 < 1) a0 = 1;
        if (a0 > 255) a0 = 255;
        if (a0 < nb0) nb0 = a0;
        if (256 - nb0 < nb1) nb1 = 256 - nb0;
    }
    LnBwdSeg<T> a, b;
    a.dy = dy; a.z = z; a.gamma = gamma0; a.mean = mean; a.rstd = rstd; a.dz = dz; a.dzd = drop0.thresh ? dzd : nullptr; a.M = M0; a.drop = drop0; a.drop_in = nodrop;
    a.z_split = b.z_split = z_split;
    b.dy = dy + (size_t)M0 * lddy; b.z = z + (size_t)M0 * ldz * (z_split ? 4 / sizeof(T) : 1); b.gamma = gamma1; b.mean = mean + M0; b.rstd = rstd + M0;
    b.dz = dz + (size_t)M0 * lddz; b.dzd = drop1.thresh ? dzd + (size_t)M0 * lddz : nullptr; b.M = M1; b.drop = drop1; b.drop_in = nodrop;
    if (defer && !defer->room(nb0 + nb1, 6, N)) defer = nullptr;
    const int blk0 = defer ? defer->blk : 0;
    if (defer) part = defer->take(nb0 + nb1);
    int r = ln_bwd_launch<T>(a, b, nb0, nb0 + nb1, lddy, ldz, lddz, part, N, 1.0f, s);
    if (r) return r;
    float* outs[6] = {dgamma0, dbeta0, dbias0, dgamma1, dbeta1, dbias1};
    if (defer) {
        for (int q = 0; q < 6; ++q) defer->add(outs[q], q % 3, blk0 + (q < 3 ? 0 : nb0), blk0 + (q < 3 ? nb0 : nb0 + nb1));
        return RGQA_OK;
    }
    FinOut fo = {};
    for (int q = 0; q < 6; ++q) { fo.p[q] = outs[q]; fo.stride[q] = 1; fo.qsrc[q] = q % 3; fo.b0[q] = q < 3 ? 0 : nb0; fo.b1[q] = q < 3 ? nb0 : nb0 + nb1; }
    return k_colsum_finalize_ranges(part, 3, 6, N, fo, accumulate, s);
}

template <typename T>
int k_colsum(const T* x, int ldx, float* part, float* out, int accumulate, int M, int N, hipStream_t s) {
    RGQA_REQUIRE(N % 4 == 0 && ldx % 4 == 0, "colsum: N %% 4");
    if (M <= 0) return RGQA_OK;
    int nblk = M < 256 ? M : 256;
    hipLaunchKernelGGL(colsum_kernel<T>, dim3(nblk), dim3(256), 0, s, x, ldx, part, M, N);
    RGQA_LAUNCH_CHECK("colsum_kernel");
    FinOut fo = {};
    fo.p[0] = out; fo.stride[0] = 1;
    return k_colsum_finalize(part, nblk, 1, N, fo, accumulate, s);
}

template int k_ln_fwd2<float>(const float*, int, const float*, const float*, const float*, const float*, int, float*, int, float*, float*, int, int, float, hipStream_t, bf16_t*);
template int k_ln_fwd2<bf16_t>(const bf16_t*, int, const float*, const float*, const float*, const float*, int, bf16_t*, int, float*, float*, int, int, float, hipStream_t, bf16_t*);
template int k_ln_fwd2<sf32>(const sf32*, int, const float*, const float*, const float*, const float*, int, sf32*, int, float*, float*, int, int, float, hipStream_t, bf16_t*);
template int k_ln_fwd<float>(const float*, int, const float*, const float*, float*, int, float*, float*, int, int, float, hipStream_t, bf16_t*);
template int k_ln_fwd<bf16_t>(const bf16_t*, int, const float*, const float*, bf16_t*, int, float*, float*, int, int, float, hipStream_t, bf16_t*);
template int k_ln_fwd<sf32>(const sf32*, int, const float*, const float*, sf32*, int, float*, float*, int, int, float, hipStream_t, bf16_t*);
template int k_ln_bwd2<float>(const float*, int, const float*, int, const float*, const float*, float*, float*, int, float*, int, int, int, const float*, float*, float*, float*, DropCfg, int, const float*, float*, float*, float*, DropCfg, hipStream_t, FinDefer*, int);
template int k_ln_bwd2<bf16_t>(const bf16_t*, int, const bf16_t*, int, const float*, const float*, bf16_t*, bf16_t*, int, float*, int, int, int, const float*, float*, float*, float*, DropCfg, int, const float*, float*, float*, float*, DropCfg, hipStream_t, FinDefer*, int);
template int k_ln_bwd2<sf32>(const sf32*, int, const sf32*, int, const float*, const float*, sf32*, sf32*, int, float*, int, int, int, const float*, float*, float*, float*, DropCfg, int, const float*, float*, float*, float*, DropCfg, hipStream_t, FinDefer*, int);
template int k_ln_bwd<float>(const float*, int, const float*, int, const float*, const float*, const float*, float*, float*, int, float*, float*, float*, float*, int, int, int, DropCfg, DropCfg, float, hipStream_t, FinDefer*, int);
template int k_ln_bwd<bf16_t>(const bf16_t*, int, const bf16_t*, int, const float*, const float*, const float*, bf16_t*, bf16_t*, int, float*, float*, float*, float*, int, int, int, DropCfg, DropCfg, float, hipStream_t, FinDefer*, int);
template int k_ln_bwd<sf32>(const sf32*, int, const sf32*, int, const float*, const float*, const float*, sf32*, sf32*, int, float*, float*, float*, float*, int, int, int, DropCfg, DropCfg, float, hipStream_t, FinDefer*, int);
template int k_colsum<float>(const float*, int, float*, float*, int, int, int, hipStream_t);
template int k_colsum<bf16_t>(const bf16_t*, int, float*, float*, int, int, int, hipStream_t);
template int k_colsum<sf32>(const sf32*, int, float*, float*, int, int, int, hipStream_t);
