// Fused global-norm clip + BertAdam (reference nn.utils.clip_grad_norm_(.., 5.) at tasks/gqa_conf.py:201 and
// BertAdam.step at lxrt/optimization.py:101-180) over a flat f32 parameter arena, plus the bf16 re-cast of
// the updated weights.  HBM-bound: 4 f32 reads + 3 f32 writes (+1 bf16 write) per parameter.
#include "kernels.h"

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    const size_t nv = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[(nv << 2) + threadIdx.x]; acc += v * v; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out, int accumulate) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { float v = (float)(red[0] + red[1] + red[2] + red[3]); *out = accumulate ? *out + v : v; }
}

int k_sumsq(const float* g, size_t n, float* partial, float* out_sumsq, int accumulate_into_out, hipStream_t s) {
    RGQA_REQUIRE(((uintptr_t)g % 16) == 0, "sumsq: 16-byte alignment required");
    int nblk = (int)((n / 4 + 255) / 256);
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, s, g, n, partial);
    RGQA_LAUNCH_CHECK("sumsq_kernel");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, partial, nblk, out_sumsq, accumulate_into_out);
    RGQA_LAUNCH_CHECK("sumsq_final_kernel");
    return RGQA_OK;
}

__device__ __forceinline__ void adam_update4(const AdamArgs& a, float coef, float p[4], const float g[4], float m[4], float v[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float gg = g[j] * coef;
        m[j] = m[j] * a.b1 + (1.f - a.b1) * gg;              // optimization.py:142
        v[j] = v[j] * a.b2 + (1.f - a.b2) * gg * gg;         // :143
        float u = m[j] / (sqrtf(v[j]) + a.eps);              // :144 (no bias correction, :175-178)
        u += a.wd * p[j];                                    // :153-154 (every parameter)
        p[j] -= a.lr_t * u;                                  // :170-171
    }
}
// the n % 4 trailing elements
__device__ __forceinline__ void adam_tail(const AdamArgs& a, float coef, size_t nv) {
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const size_t i = (nv << 2) + threadIdx.x;
        const float gg = a.g[i] * coef;
        float m = a.m[i] * a.b1 + (1.f - a.b1) * gg, v = a.v[i] * a.b2 + (1.f - a.b2) * gg * gg;
        float u = m / (sqrtf(v) + a.eps) + a.wd * a.p[i];
        a.p[i] -= a.lr_t * u; a.m[i] = m; a.v[i] = v;
        if (a.p_lp) reinterpret_cast<bf16_t*>(a.p_lp)[i] = (bf16_t)a.p[i];
    }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
// NT: non-temporal accesses (streamed once: keep the 6 GB of optimizer state out of the caches the forward pass beside it lives in)
template <bool NT> __device__ __forceinline__ void ld4(const float* p, float v[4]) {
    const f32x4v t = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p)) : *reinterpret_cast<const f32x4v*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <bool NT> __device__ __forceinline__ void st4(float* p, const float v[4]) {
    const f32x4v t = {v[0], v[1], v[2], v[3]};
    if (NT) __builtin_nontemporal_store(t, reinterpret_cast<f32x4v*>(p)); else *reinterpret_cast<f32x4v*>(p) = t;
}
// NT: non-temporal 16-byte accesses for the streamed-once f32 state (p, g, m, v in; p, m, v out): 6 GB per step that would otherwise wash
// through the 256 MB Infinity Cache and displace what the next forward pass re-reads; the bf16 weight copy, which that pass reads, stays a
// normal store.  In situ (tools/ab_bench.sh, three interleaved rounds): 11.64 vs 11.84 ms per step.
template <bool NT>
__global__ __launch_bounds__(256) void bertadam_kernel(const AdamArgs a) {
    // clip coefficient exactly as torch's clip_grad_norm_: coef = max_norm / (norm + 1e-6), applied when < 1
    float coef = a.grad_prescale;
    if (a.sumsq) {
        const float norm = sqrtf(*a.sumsq) * a.grad_prescale;
        const float c = a.max_norm / (norm + 1e-6f);
        if (c < 1.f) coef *= c;
    }
    const size_t nv = a.n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        float p[4], g[4], m[4], v[4];
        ld4<NT>(a.p + i * 4, p); ld4<NT>(a.g + i * 4, g); ld4<NT>(a.m + i * 4, m); ld4<NT>(a.v + i * 4, v);
        adam_update4(a, coef, p, g, m, v);
        st4<NT>(a.p + i * 4, p); st4<NT>(a.m + i * 4, m); st4<NT>(a.v + i * 4, v);
        if (a.p_lp) store4(reinterpret_cast<bf16_t*>(a.p_lp) + i * 4, p);
    }
    adam_tail(a, coef, nv);
}

// The same update confined to gridDim.x CUs: 1024-thread blocks that each claim more than half of a CU's LDS (so two never share a
// CU), two groups of four 16-byte loads in flight per lane.  Used when the update runs on a side stream BESIDE the next forward
// pass (Engine.adam_step(pipeline="background")): the persistent GEMM blocks of that pass need whole CUs (all registers, 128 KiB of
// LDS) - an update spread over every CU starves them until it has drained, one that owns a quarter of the chip leaves them the rest.
#define NARROW_THREADS 1024
#define NARROW_LDS (84 * 1024)
template <bool NT, int UNROLL>
__global__ __launch_bounds__(NARROW_THREADS) void bertadam_narrow_kernel(const AdamArgs a) {
    extern __shared__ unsigned char narrow_pad[];
    float coef = a.grad_prescale;
    if (a.sumsq) {
        const float norm = sqrtf(*a.sumsq) * a.grad_prescale;
        const float c = a.max_norm / (norm + 1e-6f);
        if (c < 1.f) coef *= c;
    }
    const size_t nv = a.n >> 2, stride = (size_t)gridDim.x * NARROW_THREADS;
    size_t i = (size_t)blockIdx.x * NARROW_THREADS + threadIdx.x;
    auto one = [&](size_t k) {
        float p[4], g[4], m[4], v[4];
        ld4<NT>(a.p + k * 4, p); ld4<NT>(a.g + k * 4, g); ld4<NT>(a.m + k * 4, m); ld4<NT>(a.v + k * 4, v);
        adam_update4(a, coef, p, g, m, v);
        st4<NT>(a.p + k * 4, p); st4<NT>(a.m + k * 4, m); st4<NT>(a.v + k * 4, v);
        if (a.p_lp) store4(reinterpret_cast<bf16_t*>(a.p_lp) + k * 4, p);      // the forward pass reads this copy next: cached
    };
    if (UNROLL == 2) {
        for (; i + stride < nv; i += 2 * stride) {
            const size_t i2 = i + stride;
            float p[4], g[4], m[4], v[4], p2[4], g2[4], m2[4], v2[4];
            ld4<NT>(a.p + i * 4, p); ld4<NT>(a.g + i * 4, g); ld4<NT>(a.m + i * 4, m); ld4<NT>(a.v + i * 4, v);
            ld4<NT>(a.p + i2 * 4, p2); ld4<NT>(a.g + i2 * 4, g2); ld4<NT>(a.m + i2 * 4, m2); ld4<NT>(a.v + i2 * 4, v2);
            adam_update4(a, coef, p, g, m, v);
            st4<NT>(a.p + i * 4, p); st4<NT>(a.m + i * 4, m); st4<NT>(a.v + i * 4, v);
            if (a.p_lp) store4(reinterpret_cast<bf16_t*>(a.p_lp) + i * 4, p);
            adam_update4(a, coef, p2, g2, m2, v2);
            st4<NT>(a.p + i2 * 4, p2); st4<NT>(a.m + i2 * 4, m2); st4<NT>(a.v + i2 * 4, v2);
            if (a.p_lp) store4(reinterpret_cast<bf16_t*>(a.p_lp) + i2 * 4, p2);
        }
    }
    for (; i < nv; i += stride) one(i);
    adam_tail(a, coef, nv);
}

int g_rgqa_narrow_cus = 0;     // rgqa_debug_set key 13: > 0 = the optimizer / weight-copy kernels launched next stay on that many CUs (one 1024-thread block each)
static int narrow_blocks() {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bertadam_narrow_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, NARROW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bertadam_narrow_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, NARROW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bertadam_narrow_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, NARROW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bertadam_narrow_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, NARROW_LDS);
        attr = true;
    }
    return g_rgqa_narrow_cus;
}

int k_bertadam(const AdamArgs& a, hipStream_t s) {
    if (a.n == 0) return RGQA_OK;
    RGQA_REQUIRE(((uintptr_t)a.p % 16) == 0 && ((uintptr_t)a.g % 16) == 0 && ((uintptr_t)a.m % 16) == 0 && ((uintptr_t)a.v % 16) == 0, "bertadam: 16-byte alignment required");
    size_t nb = (a.n / 4 + 255) / 256;
    if (const int ncu = narrow_blocks(); ncu > 0 && nb > (size_t)ncu * 4) {
        static const int variant = []() { const char* e = getenv("RGQA_ADAM_NARROW_VARIANT"); return e ? atoi(e) : 3; }();     // bit 0: non-temporal, bit 1: two groups of loads in flight
        switch (variant & 3) {
            case 0: hipLaunchKernelGGL((bertadam_narrow_kernel<false, 1>), dim3(ncu), dim3(NARROW_THREADS), NARROW_LDS, s, a); break;
            case 1: hipLaunchKernelGGL((bertadam_narrow_kernel<true, 1>), dim3(ncu), dim3(NARROW_THREADS), NARROW_LDS, s, a); break;
            case 2: hipLaunchKernelGGL((bertadam_narrow_kernel<false, 2>), dim3(ncu), dim3(NARROW_THREADS), NARROW_LDS, s, a); break;
            default: hipLaunchKernelGGL((bertadam_narrow_kernel<true, 2>), dim3(ncu), dim3(NARROW_THREADS), NARROW_LDS, s, a); break;
        }
        RGQA_LAUNCH_CHECK("bertadam_narrow_kernel");
        return RGQA_OK;
    }
    static const bool nt = []() { const char* e = getenv("RGQA_ADAM_NT"); return !(e != nullptr && e[0] == '0'); }();     // default on
    int nblk = nb > 2048 ? 2048 : (nb < 1 ? 1 : (int)nb);
    if (nt) hipLaunchKernelGGL(bertadam_kernel<true>, dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(bertadam_kernel<false>, dim3(nblk), dim3(256), 0, s, a);
    RGQA_LAUNCH_CHECK("bertadam_kernel");
    return RGQA_OK;
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
    const size_t nv = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        float v[4];
        load4(src + i * 4, v);
        store4(dst + i * 4, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(nv << 2) + threadIdx.x] = (bf16_t)src[(nv << 2) + threadIdx.x];
}
int k_cast_bf16(const float* src, void* dst, size_t n, hipStream_t s) {
    if (n == 0) return RGQA_OK;
    size_t nb = (n / 4 + 255) / 256;
    int nblk = nb > 2048 ? 2048 : (nb < 1 ? 1 : (int)nb);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(nblk), dim3(256), 0, s, src, reinterpret_cast<bf16_t*>(dst), n);
    RGQA_LAUNCH_CHECK("cast_bf16_kernel");
    return RGQA_OK;
}

// Data-parallel exchange (rgqa_amd/parallel.py): dst[i] = sum_r f32(parts[r * stride + i]), r ascending (f32 accumulation of the
// bf16 gradient shards every rank received for the range it owns - deterministic, unlike an in-network bf16 reduction)
__global__ __launch_bounds__(256) void sum_bf16_parts_kernel(const bf16_t* __restrict__ parts, size_t stride, int nparts, float* __restrict__ dst, size_t n) {
    const size_t nv = n >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < nparts; ++r) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(parts + (size_t)r * stride + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        }
        store4(dst + i * 8, acc); store4(dst + i * 8 + 4, acc + 4);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const size_t i = (nv << 3) + threadIdx.x;
        float a = 0.f;
        for (int r = 0; r < nparts; ++r) a += (float)parts[(size_t)r * stride + i];
        dst[i] = a;
    }
}
int k_sum_bf16_parts(const void* parts, size_t stride, int nparts, float* dst, size_t n, hipStream_t s) {
    if (n == 0) return RGQA_OK;
    RGQA_REQUIRE(((uintptr_t)parts % 16) == 0 && ((uintptr_t)dst % 16) == 0 && (stride % 8) == 0, "sum_bf16_parts: 16-byte alignment / stride %% 8 required");
    size_t nb = (n / 8 + 255) / 256;
    int nblk = nb > 2048 ? 2048 : (nb < 1 ? 1 : (int)nb);
    hipLaunchKernelGGL(sum_bf16_parts_kernel, dim3(nblk), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(parts), stride, nparts, dst, n);
    RGQA_LAUNCH_CHECK("sum_bf16_parts_kernel");
    return RGQA_OK;
}

// Batched cast + transpose of every linear weight: dst[k][n] = bf16(src[n][k]); TRANSPOSE_TILE^2 (64x64) tiles through LDS:
// float4 reads of 256-B row pieces, 8-B writes of full 128-B destination lines (the 32x32 / 2-B-store version ran at 58 % of
// the copy's byte floor).
// one 64 x 64 tile `t` by 256 threads (ltid), in two phases around a workgroup barrier of the caller
template <typename S>
__device__ __forceinline__ void transpose_tile_load(const S* __restrict__ src, const TransDesc* __restrict__ desc, int ndesc, int t, float (*tile)[TRANSPOSE_TILE + 1], int ltid, TransDesc& d, int& n0, int& k0) {
    constexpr int TT = TRANSPOSE_TILE;
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (desc[mid].tile_start <= t) lo = mid; else hi = mid - 1; }
    d = desc[lo];
    const int lt = t - d.tile_start, tk = cdiv(d.K, TT);
    n0 = (lt / tk) * TT; k0 = (lt % tk) * TT;
    const int tx = ltid & 15, ty = ltid >> 4;      // 16 threads x 4 elements per 64-wide row, 16 rows per pass
    const bool k4 = (d.K & 3) == 0;
#pragma unroll
    for (int r = 0; r < TT / 16; ++r) {
        const int n = n0 + ty + r * 16, k = k0 + tx * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (n < d.N) {
            const S* sp = src + d.src_off + (size_t)n * d.K + k;
            if (k4 && k + 4 <= d.K) load4(sp, v);
            else { for (int j = 0; j < 4; ++j) if (k + j < d.K) v[j] = to_f32(sp[j]); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[ty + r * 16][tx * 4 + j] = v[j];
    }
}
__device__ __forceinline__ void transpose_tile_store(bf16_t* __restrict__ dst, const TransDesc& d, int n0, int k0, float (*tile)[TRANSPOSE_TILE + 1], int ltid, bool nt) {
    constexpr int TT = TRANSPOSE_TILE;
    const int tx = ltid & 15, ty = ltid >> 4;
    const bool n4 = (d.ld_dst & 3) == 0;
#pragma unroll
    for (int r = 0; r < TT / 16; ++r) {
        const int k = k0 + ty + r * 16, n = n0 + tx * 4;
        if (k >= d.K || n >= d.ld_dst) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = tile[tx * 4 + j][ty + r * 16];
        bf16_t* dp = dst + d.dst_off + (size_t)k * d.ld_dst + n;
        if (n4 && n + 4 <= d.ld_dst) {
            if (nt) {       // the transposed copy is read by a backward pass milliseconds later: do not let it displace cached data now
                bf16x4 o; o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
                __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(dp));
            } else store4(dp, v);
        }
        else { for (int j = 0; j < 4; ++j) if (n + j < d.ld_dst) dp[j] = (bf16_t)v[j]; }
    }
}
template <typename S>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const S* __restrict__ src, bf16_t* __restrict__ dst, const TransDesc* __restrict__ desc, int ndesc, int nt) {
    __shared__ float tile[TRANSPOSE_TILE][TRANSPOSE_TILE + 1];
    TransDesc d; int n0, k0;
    transpose_tile_load<S>(src, desc, ndesc, blockIdx.x, tile, threadIdx.x, d, n0, k0);
    __syncthreads();
    transpose_tile_store(dst, d, n0, k0, tile, threadIdx.x, nt != 0);
}
// confined to gridDim.x CUs (see bertadam_narrow_kernel): four 256-thread groups per block, each walking its own tiles
template <typename S>
__global__ __launch_bounds__(NARROW_THREADS) void cast_transpose_narrow_kernel(const S* __restrict__ src, bf16_t* __restrict__ dst, const TransDesc* __restrict__ desc, int ndesc, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char narrow_lds[];
    const int sub = threadIdx.x >> 8, ltid = threadIdx.x & 255;
    float (*tile)[TRANSPOSE_TILE + 1] = reinterpret_cast<float (*)[TRANSPOSE_TILE + 1]>(narrow_lds) + sub * TRANSPOSE_TILE;
    const int step = (int)gridDim.x * 4;
    for (int base = 0; base < total_tiles; base += step) {        // block-uniform trip count: every thread reaches both barriers
        const int t = base + (int)blockIdx.x * 4 + sub;
        TransDesc d; int n0 = 0, k0 = 0;
        if (t < total_tiles) transpose_tile_load<S>(src, desc, ndesc, t, tile, ltid, d, n0, k0);
        __syncthreads();
        if (t < total_tiles) transpose_tile_store(dst, d, n0, k0, tile, ltid, true);
        __syncthreads();
    }
}
// src: the f32 master weights, or (src_is_bf16) their bf16 copy at the same element offsets - half the bytes to read, same result
int k_cast_transpose(const void* src, int src_is_bf16, void* dst_bf16, const TransDesc* desc_dev, int ndesc, int total_tiles, hipStream_t s) {
    if (ndesc <= 0 || total_tiles <= 0) return RGQA_OK;
    if (const int ncu = narrow_blocks(); ncu > 0 && total_tiles > ncu * 16) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cast_transpose_narrow_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, NARROW_LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cast_transpose_narrow_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, NARROW_LDS);
            attr = true;
        }
        if (src_is_bf16) hipLaunchKernelGGL(cast_transpose_narrow_kernel<bf16_t>, dim3(ncu), dim3(NARROW_THREADS), NARROW_LDS, s, reinterpret_cast<const bf16_t*>(src), reinterpret_cast<bf16_t*>(dst_bf16), desc_dev, ndesc, total_tiles);
        else hipLaunchKernelGGL(cast_transpose_narrow_kernel<float>, dim3(ncu), dim3(NARROW_THREADS), NARROW_LDS, s, reinterpret_cast<const float*>(src), reinterpret_cast<bf16_t*>(dst_bf16), desc_dev, ndesc, total_tiles);
        RGQA_LAUNCH_CHECK("cast_transpose_narrow_kernel");
        return RGQA_OK;
    }
    const int ntf = 0;       // (non-temporal stores of the transposed copy, of the wgrad output, of gelu' and non-temporal loads in the gradient norm: +-0 in situ, 11.73 vs 11.69 ms; only the optimizer's state streams pay, RGQA_ADAM_NT)
    if (src_is_bf16) hipLaunchKernelGGL(cast_transpose_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(src), reinterpret_cast<bf16_t*>(dst_bf16), desc_dev, ndesc, ntf);
    else hipLaunchKernelGGL(cast_transpose_kernel<float>, dim3(total_tiles), dim3(256), 0, s, reinterpret_cast<const float*>(src), reinterpret_cast<bf16_t*>(dst_bf16), desc_dev, ndesc, ntf);
    RGQA_LAUNCH_CHECK("cast_transpose_kernel");
    return RGQA_OK;
}
