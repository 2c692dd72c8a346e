// Wavefront-reduced LayerNorm (eps inside the sqrt, biased variance: torch.nn.LayerNorm semantics,
// reference lxrt/modeling.py:261) forward / backward, one 64-lane wave per row, f32 statistics.
// HBM-bound: algorithmic bytes per row = N * (sizeof(T) in + sizeof(T) out) forward.
#include "kernels.h"

#define LN_MAXV 8  // up to 8 vec4 chunks per lane -> N <= 2048 (NV = chunks per lane, compile-time)

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y, int ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int N, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
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
        }
    }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// Backward. dz = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma, xhat = (z-mean)*rstd.
// Also emits (optionally) dzd = dropout-masked dz (the gradient entering the preceding dense layer,
// whose forward epilogue applied that mask) and per-block partial column sums:
//   part[blk][0][n] = sum dy*xhat (dgamma), part[blk][1][n] = sum dy (dbeta), part[blk][2][n] = sum dzd (dense bias grad)
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ z, int ldz,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dz, T* __restrict__ dzd,
                                                     int lddz, float* __restrict__ part, int M, int N, DropCfg drop, DropCfg drop_in, float dy_scale) {
    __shared__ float red[4][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = N >> 2;
    float ag[NV][4], ab[NV][4], ad[NV][4], gm[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; ad[i][j] = 0.f; gm[i][j] = 0.f; }
        if (c < nv) load4(gamma + c * 4, gm[i]);
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float g[NV][4], xh[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float d[4], zz[4];
                load4(dy + (size_t)row * lddy + c * 4, d);
                load4(z + (size_t)row * ldz + c * 4, zz);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    d[j] = drop_apply(drop_in, (uint32_t)row * (uint32_t)N + (uint32_t)(c * 4 + j), d[j]) * dy_scale;
                    xh[i][j] = (zz[j] - mu) * rs;
                    g[i][j] = d[j] * gm[i][j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                    ag[i][j] += d[j] * xh[i][j];
                    ab[i][j] += d[j];
                }
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
                for (int j = 0; j < 4; ++j) {
                    o[j] = rs * (g[i][j] - s1 - xh[i][j] * s2);
                    od[j] = drop_apply(drop, (uint32_t)row * (uint32_t)N + (uint32_t)(c * 4 + j), o[j]);
                    ad[i][j] += od[j];
                }
                store4(dz + (size_t)row * lddz + c * 4, o);
                if (dzd) store4(dzd + (size_t)row * lddz + c * 4, od);
            }
        }
    }
    if (part == nullptr) return;
    // cross-wave reduction of the three column accumulators, one at a time through LDS
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    red[wave][c * 4 + j] = which == 0 ? ag[i][j] : (which == 1 ? ab[i][j] : ad[i][j]);
            }
        }
        __syncthreads();
        for (int n = threadIdx.x; n < N; n += 256)
            part[((size_t)blockIdx.x * 3 + which) * N + n] = red[0][n] + red[1][n] + red[2][n] + red[3][n];
    }
}

// out[q][n*stride] (+)= sum_blk part[blk][q][n] for q < nq (null output pointers are skipped).
// Block = 16 columns x 16 partial groups: every thread sums nblk/16 partials (fixed order -> bit-reproducible),
// then the 16 group sums are folded in a fixed order through LDS.
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nblk, int nq, int N, FinOut fo, int accumulate) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + c;
    const int q = blockIdx.y;
    float* o = fo.p[q];
    if (o == nullptr) return;
    float s = 0.f;
    if (n < N)
        for (int b = grp; b < nblk; b += 16) s += part[((size_t)b * nq + q) * N + n];
    red[grp][c] = s;
    __syncthreads();
    if (grp == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][c];
        o += (size_t)n * fo.stride[q];
        *o = accumulate ? *o + t : t;
    }
}

int k_colsum_finalize(const float* part, int nblk, int nq, int N, const FinOut& fo, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(N, 16), nq), dim3(256), 0, s, part, nblk, nq, N, fo, accumulate);
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
int k_ln_fwd(const T* x, int ldx, const float* gamma, const float* beta, T* y, int ldy, float* mean, float* rstd, int M, int N, float eps, hipStream_t s) {
    RGQA_REQUIRE(N % 4 == 0 && N <= LN_MAXV * 256 && ldx % 4 == 0 && ldy % 4 == 0, "layernorm: N=%d must be a multiple of 4 and <= %d", N, LN_MAXV * 256);
    if (M <= 0) return RGQA_OK;
#define LN_FWD(NVV) hipLaunchKernelGGL((ln_fwd_kernel<T, NVV>), dim3(cdiv(M, 4)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M, N, eps)
    const int nvl = cdiv(N / 4, 64);
    if (nvl <= 1) LN_FWD(1); else if (nvl == 2) LN_FWD(2); else if (nvl == 3) LN_FWD(3); else if (nvl == 4) LN_FWD(4);
    else if (nvl <= 6) LN_FWD(6); else LN_FWD(8);
#undef LN_FWD
    RGQA_LAUNCH_CHECK("ln_fwd_kernel");
    return RGQA_OK;
}

int ln_bwd_blocks(int M) { int b = cdiv(M, 4); return b > 512 ? 512 : b; }

template <typename T>
int k_ln_bwd(const T* dy, int lddy, const T* z, int ldz, const float* gamma, const float* mean, const float* rstd, T* dz, T* dzd, int lddz,
             float* part, float* dgamma, float* dbeta, float* dbias, int accumulate, int M, int N, DropCfg drop, DropCfg drop_in, float dy_scale, hipStream_t s) {
    RGQA_REQUIRE(N % 4 == 0 && N <= LN_MAXV * 256, "layernorm bwd: N=%d unsupported", N);
    if (M <= 0) return RGQA_OK;
    const int nblk = ln_bwd_blocks(M);
#define LN_BWD(NVV) hipLaunchKernelGGL((ln_bwd_kernel<T, NVV>), dim3(nblk), dim3(256), 0, s, dy, lddy, z, ldz, gamma, mean, rstd, dz, dzd, lddz, part, M, N, drop, drop_in, dy_scale)
    const int nvl = cdiv(N / 4, 64);
    if (nvl <= 1) LN_BWD(1); else if (nvl == 2) LN_BWD(2); else if (nvl == 3) LN_BWD(3); else if (nvl == 4) LN_BWD(4);
    else if (nvl <= 6) LN_BWD(6); else LN_BWD(8);
#undef LN_BWD
    RGQA_LAUNCH_CHECK("ln_bwd_kernel");
    if (part) {
        FinOut fo = {};
        fo.p[0] = dgamma; fo.p[1] = dbeta; fo.p[2] = dbias;
        fo.stride[0] = fo.stride[1] = fo.stride[2] = 1;
        return k_colsum_finalize(part, nblk, 3, N, fo, accumulate, s);
    }
    return RGQA_OK;
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

template int k_ln_fwd<float>(const float*, int, const float*, const float*, float*, int, float*, float*, int, int, float, hipStream_t);
template int k_ln_fwd<bf16_t>(const bf16_t*, int, const float*, const float*, bf16_t*, int, float*, float*, int, int, float, hipStream_t);
template int k_ln_bwd<float>(const float*, int, const float*, int, const float*, const float*, const float*, float*, float*, int, float*, float*, float*, float*, int, int, int, DropCfg, DropCfg, float, hipStream_t);
template int k_ln_bwd<bf16_t>(const bf16_t*, int, const bf16_t*, int, const float*, const float*, const float*, bf16_t*, bf16_t*, int, float*, float*, float*, float*, int, int, int, DropCfg, DropCfg, float, hipStream_t);
template int k_colsum<float>(const float*, int, float*, float*, int, int, int, hipStream_t);
template int k_colsum<bf16_t>(const bf16_t*, int, float*, float*, int, int, int, hipStream_t);
