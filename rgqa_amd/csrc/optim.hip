// Fused global-norm clip + BertAdam (reference nn.utils.clip_grad_norm_(.., 5.) at tasks/gqa_conf.py:201 and
// BertAdam.step at lxrt/optimization.py:101-180) over a flat f32 parameter arena, plus the bf16 re-cast of
// the updated weights.  HBM-bound: 4 f32 reads + 3 f32 writes (+1 bf16 write) per parameter.
#include "kernels.h"

// sum(g^2) in ONE launch: every block parks its partial, the block whose ticket is last folds all partials in index order (fixed order:
// bit-reproducible) and rewinds the ticket for the next launch.  partial: >= 1024 floats + 1 ticket word (partial[1024], as an int).
// Two entry points: k_sumsq zeroes the ticket word itself on the stream (any scratch will do, shared or recycled: the BUTD engine, the public
// rgqa_grad_sumsq); k_sumsq_owned is for a scratch that only these launches ever touch and that its owner zeroed once (the engine's
// per-segment slots: 21 launches per step without a memset node each).  (Round 2 ran this as two launches per gradient segment: 42 launches
// per step, the second one 5 us of pure launch latency.)
// The cross-block hand-off of sumsq_kernel / sum_parts_kernel: one lane parks its block's partial and draws a ticket; the block whose ticket
// is last reads every partial.  What makes it correct ON THIS TARGET (and cheap):
//   * every access to a partial - the publish AND the last block's reads - is a RETURNING agent-scope read-modify-write: on gfx950 such an
//     atomic is performed at the memory side (MI355X_MICROARCH.md, "atomic DROP it"), never served from a CU's L1 or another XCD's L2, and its
//     return value exists only once it has been performed;
//   * the publishing lane WAITS for that return (hardware: s_waitcnt vmcnt(0), forced by the workgroup-scope release fence below - which emits
//     the wait and NO L2 write-back, unlike an agent-scope release, round 3's 0.33 ms per train step) before it draws its ticket, so the ticket
//     order is a happens-after order of the publishes at the memory side;
//   * the last block's reads sit behind a workgroup-scope acquire fence + the workgroup barrier.
// The LLVM memory model alone gives no happens-before between relaxed agent atomics on different addresses: the argument above is a
// property of gfx942 / gfx950 with coarse-grained hipMalloc memory, so the file refuses to build for anything else, and
// tests/test_gpu_ops.py::test_sumsq_handoff_under_load checks the result against a two-pass sum beside GEMMs on all eight XCDs.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "optim.hip: the fence-free partial hand-off is verified on gfx942 / gfx950 only; use agent-scope release / acquire fences elsewhere"
#endif
__device__ __forceinline__ bool publish_partial_draw_ticket(unsigned* slot, unsigned bits, int* ticket, int nblocks) {
    const unsigned old = __hip_atomic_exchange(slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" :: "v"(old) : "memory");                          // the returned value is waited for here
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // s_waitcnt only (no buffer_wbl2): the exchange above has been performed
    const bool last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");            // the partial reads of the last block are not hoisted above its ticket
    return last;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial, float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    __shared__ int last;
    float acc = 0.f;
    const size_t nv = n >> 2;
    // four independent 16-byte loads in flight per lane (one per iteration left the loop waiting on one memory round trip per 16 bytes:
    // 17 us for a 28-MB segment, 1.6 TB/s); the order of the additions is fixed by (grid, n) alone, so the sum stays bit-reproducible
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (; i + 3 * stride < nv; i += 4 * stride) {
        const float4 v0 = reinterpret_cast<const float4*>(g)[i], v1 = reinterpret_cast<const float4*>(g)[i + stride];
        const float4 v2 = reinterpret_cast<const float4*>(g)[i + 2 * stride], v3 = reinterpret_cast<const float4*>(g)[i + 3 * stride];
        a0 += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
        a1 += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
        a2 += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
        a3 += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
    }
    for (; i < nv; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        a0 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    acc = (a0 + a1) + (a2 + a3);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[(nv << 2) + threadIdx.x]; acc += v * v; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    int* ticket = reinterpret_cast<int*>(partial + 1024);
    if (threadIdx.x == 0) {
        // hand-off without agent-scope fences: publish_partial_draw_ticket above
        last = publish_partial_draw_ticket(reinterpret_cast<unsigned*>(partial) + blockIdx.x, __float_as_uint(red[0] + red[1] + red[2] + red[3]), ticket, (int)gridDim.x);
    }
    __syncthreads();
    if (!last) return;                                          // block-uniform
    double t = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)
        t += (double)__uint_as_float(__hip_atomic_fetch_or(reinterpret_cast<unsigned*>(partial) + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    __shared__ double redd[4];
    if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(redd[0] + redd[1] + redd[2] + redd[3]);
        *out = accumulate ? *out + v : v;
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Up to SUMSQ_GROUP_MAX ranges in ONE launch (blockIdx.y = range), each with its own partials / ticket / output slot: the shares of the gradient
// segments one weight-gradient launch finalises (round 5: six launches of 14 us in a row at the end of backward, in front of the optimizer).  The
// arithmetic per range is sumsq_kernel's with the same grid: the same bits.
__global__ __launch_bounds__(256) void sumsq_group_kernel(const SumsqGroup gr) {
    const SumsqRange& r = gr.r[blockIdx.y];
    if ((int)blockIdx.x >= r.nblk) return;                      // block-uniform
    const float* __restrict__ g = r.g;
    const size_t n = r.n;
    float* partial = r.partial;
    __shared__ float red[4];
    __shared__ int last;
    const size_t nv = n >> 2;
    const size_t stride = (size_t)r.nblk * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (; i + 3 * stride < nv; i += 4 * stride) {
        const float4 v0 = reinterpret_cast<const float4*>(g)[i], v1 = reinterpret_cast<const float4*>(g)[i + stride];
        const float4 v2 = reinterpret_cast<const float4*>(g)[i + 2 * stride], v3 = reinterpret_cast<const float4*>(g)[i + 3 * stride];
        a0 += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
        a1 += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
        a2 += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
        a3 += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
    }
    for (; i < nv; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        a0 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    float acc = (a0 + a1) + (a2 + a3);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[(nv << 2) + threadIdx.x]; acc += v * v; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    int* ticket = reinterpret_cast<int*>(partial + 1024);
    if (threadIdx.x == 0)
        last = publish_partial_draw_ticket(reinterpret_cast<unsigned*>(partial) + blockIdx.x, __float_as_uint(red[0] + red[1] + red[2] + red[3]), ticket, r.nblk);
    __syncthreads();
    if (!last) return;                                          // block-uniform
    double t = 0.0;
    for (int j = threadIdx.x; j < r.nblk; j += 256)
        t += (double)__uint_as_float(__hip_atomic_fetch_or(reinterpret_cast<unsigned*>(partial) + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    __shared__ double redd[4];
    if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        *r.out = (float)(redd[0] + redd[1] + redd[2] + redd[3]);
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int k_sumsq_owned_group(SumsqGroup& gr, hipStream_t s) {
    if (gr.count <= 0) return RGQA_OK;
    RGQA_REQUIRE(gr.count <= SUMSQ_GROUP_MAX, "sumsq group: %d ranges", gr.count);
    int mx = 1;
    for (int k = 0; k < gr.count; ++k) {
        RGQA_REQUIRE(((uintptr_t)gr.r[k].g % 16) == 0, "sumsq: 16-byte alignment required");
        int nblk = (int)((gr.r[k].n / 4 + 255) / 256);
        if (nblk > 512) nblk = 512;
        if (nblk < 1) nblk = 1;
        gr.r[k].nblk = nblk;
        if (nblk > mx) mx = nblk;
    }
    hipLaunchKernelGGL(sumsq_group_kernel, dim3(mx, gr.count), dim3(256), 0, s, gr);
    RGQA_LAUNCH_CHECK("sumsq_group_kernel");
    gr.count = 0;
    return RGQA_OK;
}

int k_sumsq(const float* g, size_t n, float* partial, float* out_sumsq, int accumulate_into_out, hipStream_t s) {
    RGQA_HIP(hipMemsetAsync(partial + 1024, 0, sizeof(int), s));
    return k_sumsq_owned(g, n, partial, out_sumsq, accumulate_into_out, s);
}
int k_sumsq_owned(const float* g, size_t n, float* partial, float* out_sumsq, int accumulate_into_out, hipStream_t s) {
    RGQA_REQUIRE(((uintptr_t)g % 16) == 0, "sumsq: 16-byte alignment required");
    int nblk = (int)((n / 4 + 255) / 256);
    if (nblk > 512) nblk = 512;          // two blocks per CU; the last block then folds two partials per lane
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, s, g, n, partial, out_sumsq, accumulate_into_out);
    RGQA_LAUNCH_CHECK("sumsq_kernel");
    return RGQA_OK;
}

// In-place part of nn.utils.clip_grad_norm_ (tasks/gqa_conf.py:201) for callers that clip and step separately (the drop-in BertAdam):
// g *= max_norm / (sqrt(sumsq) + 1e-6) when that coefficient is below 1 - and nothing at all, not a byte of traffic, when it is not.
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, size_t n, const float* __restrict__ sumsq, float max_norm) {
    const float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
    if (!(c < 1.f)) return;                 // block-uniform
    const size_t nv = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<float4*>(g)[i];
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        reinterpret_cast<float4*>(g)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) g[(nv << 2) + threadIdx.x] *= c;
}
int k_clip_scale(float* g, size_t n, const float* sumsq, float max_norm, hipStream_t s) {
    if (n == 0) return RGQA_OK;
    RGQA_REQUIRE(((uintptr_t)g % 16) == 0, "clip_scale: 16-byte alignment required");
    size_t nb = (n / 4 + 255) / 256;
    int nblk = nb > 2048 ? 2048 : (nb < 1 ? 1 : (int)nb);
    hipLaunchKernelGGL(clip_scale_kernel, dim3(nblk), dim3(256), 0, s, g, n, sumsq, max_norm);
    RGQA_LAUNCH_CHECK("clip_scale_kernel");
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
        if (a.p_lp && a.lp_split) sf_store1(reinterpret_cast<sf32*>(a.p_lp) + i, a.p[i]);
        else if (a.p_lp) reinterpret_cast<bf16_t*>(a.p_lp)[i] = (bf16_t)a.p[i];
    }
}
// the low-precision operand copy of 4 updated parameters: bf16, or the split-f32 pair (bf16x3 precision) at the same element offsets
__device__ __forceinline__ void adam_store_lp(const AdamArgs& a, size_t i4, const float p[4]) {
    if (a.p_lp == nullptr) return;
    if (a.lp_split) store4(reinterpret_cast<sf32*>(a.p_lp) + i4, p);
    else store4(reinterpret_cast<bf16_t*>(a.p_lp) + i4, p);
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
// through the 256 MB Infinity Cache and displace what the next forward pass re-reads; the low-precision weight copy, which that pass
// reads, stays a normal store.  In situ (three interleaved A/B rounds, round 2): 11.64 vs 11.84 ms per step.
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
        adam_store_lp(a, i * 4, p);
    }
    adam_tail(a, coef, nv);
}

int g_rgqa_adam_blocks = 2048;      // rgqa_debug_set key 20: cap on the update kernel's (grid-stride) grid
int k_bertadam(const AdamArgs& a, hipStream_t s) {
    if (a.n == 0) return RGQA_OK;
    RGQA_REQUIRE(((uintptr_t)a.p % 16) == 0 && ((uintptr_t)a.g % 16) == 0 && ((uintptr_t)a.m % 16) == 0 && ((uintptr_t)a.v % 16) == 0, "bertadam: 16-byte alignment required");
    size_t nb = (a.n / 4 + 255) / 256;
    const size_t cap = g_rgqa_adam_blocks > 0 ? (size_t)g_rgqa_adam_blocks : 2048;
    int nblk = nb > cap ? (int)cap : (nb < 1 ? 1 : (int)nb);
    hipLaunchKernelGGL(bertadam_kernel<true>, dim3(nblk), dim3(256), 0, s, a);
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

// f32 -> split f32 (bf16x3 precision), element i of src -> slot i of dst: the weight arena's operand copy, the RoI features
// img (bf16x3_fwd precision, the RoI features): the bf16 image the backward pass reads, from the same registers (round 5: a second pass over the f32 source)
__global__ __launch_bounds__(256) void cast_split_kernel(const float* __restrict__ src, sf32* __restrict__ dst, size_t n, bf16_t* __restrict__ img) {
    const size_t nv = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        float v[4];
        load4(src + i * 4, v);
        store4(dst + i * 4, v);
        if (img) store4(img + i * 4, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        sf_store1(dst + (nv << 2) + threadIdx.x, src[(nv << 2) + threadIdx.x]);
        if (img) img[(nv << 2) + threadIdx.x] = (bf16_t)src[(nv << 2) + threadIdx.x];
    }
}
int k_cast_split(const float* src, void* dst, size_t n, hipStream_t s, bf16_t* img) {
    if (n == 0) return RGQA_OK;
    RGQA_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0 && ((uintptr_t)img % 8) == 0, "cast_split: 16-byte alignment required");
    size_t nb = (n / 4 + 255) / 256;
    int nblk = nb > 2048 ? 2048 : (nb < 1 ? 1 : (int)nb);
    hipLaunchKernelGGL(cast_split_kernel, dim3(nblk), dim3(256), 0, s, src, reinterpret_cast<sf32*>(dst), n, img);
    RGQA_LAUNCH_CHECK("cast_split_kernel");
    return RGQA_OK;
}

// Data-parallel exchange (rgqa_amd/parallel.py): dst[i] = sum_r f32(parts[r * stride + i]), r ascending (f32 accumulation of the
// gradient shards every rank received for the range it owns - deterministic, unlike an in-network reduction).  PT = bf16_t (bf16 /
// bf16x3_fwd engines: bf16 payload) or float (f32 / bf16x3 engines).  With sq_out the kernel also leaves sum(dst^2) behind - the owner's share
// of the clip norm, taken while every element is in registers instead of by a second pass over the range: per-block partials, the last
// block (agent-scope ticket at sq_ws[1024], zeroed on the stream by the launcher) folds them in index order and ADDS to *sq_out.
__device__ __forceinline__ void ld8p(const bf16_t* p, float v[8]) {         // one 16-byte access
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
}
__device__ __forceinline__ void ld8p(const float* p, float v[8]) { load4(p, v); load4(p + 4, v + 4); }
template <typename PT>
__global__ __launch_bounds__(256) void sum_parts_kernel(const PT* __restrict__ parts, size_t stride, int nparts, float* __restrict__ dst, size_t n,
                                                        float* __restrict__ sq_ws, float* __restrict__ sq_out) {
    const size_t nv = n >> 3;
    float sq = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < nparts; ++r) {
            float v[8];
            ld8p(parts + (size_t)r * stride + i * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sq += acc[j] * acc[j];
        store4(dst + i * 8, acc); store4(dst + i * 8 + 4, acc + 4);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const size_t i = (nv << 3) + threadIdx.x;
        float a = 0.f;
        for (int r = 0; r < nparts; ++r) a += to_f32(parts[(size_t)r * stride + i]);
        dst[i] = a;
        sq += a * a;
    }
    if (sq_out == nullptr) return;
    __shared__ float red[4];
    __shared__ int last;
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    int* ticket = reinterpret_cast<int*>(sq_ws + 1024);
    if (threadIdx.x == 0) {
        // as in sumsq_kernel
        last = publish_partial_draw_ticket(reinterpret_cast<unsigned*>(sq_ws) + blockIdx.x, __float_as_uint(red[0] + red[1] + red[2] + red[3]), ticket, (int)gridDim.x);
    }
    __syncthreads();
    if (!last) return;                                          // block-uniform
    double t = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)
        t += (double)__uint_as_float(__hip_atomic_fetch_or(reinterpret_cast<unsigned*>(sq_ws) + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    __shared__ double redd[4];
    if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) *sq_out += (float)(redd[0] + redd[1] + redd[2] + redd[3]);
}
int k_sum_parts(const void* parts, int parts_f32, size_t stride, int nparts, float* dst, size_t n, float* sq_ws, float* sq_out, hipStream_t s) {
    if (n == 0) return RGQA_OK;
    RGQA_REQUIRE(((uintptr_t)parts % 16) == 0 && ((uintptr_t)dst % 16) == 0 && (stride % 8) == 0, "sum_parts: 16-byte alignment / stride %% 8 required");
    RGQA_REQUIRE((sq_out == nullptr) == (sq_ws == nullptr), "sum_parts: the norm share needs both its scratch (>= 1025 floats) and its output");
    size_t nb = (n / 8 + 255) / 256;
    int nblk = nb > 1024 ? 1024 : (nb < 1 ? 1 : (int)nb);
    if (sq_ws) RGQA_HIP(hipMemsetAsync(sq_ws + 1024, 0, sizeof(int), s));
    if (parts_f32) hipLaunchKernelGGL(sum_parts_kernel<float>, dim3(nblk), dim3(256), 0, s, reinterpret_cast<const float*>(parts), stride, nparts, dst, n, sq_ws, sq_out);
    else hipLaunchKernelGGL(sum_parts_kernel<bf16_t>, dim3(nblk), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(parts), stride, nparts, dst, n, sq_ws, sq_out);
    RGQA_LAUNCH_CHECK("sum_parts_kernel");
    return RGQA_OK;
}
int k_sum_bf16_parts(const void* parts, size_t stride, int nparts, float* dst, size_t n, hipStream_t s) { return k_sum_parts(parts, 0, stride, nparts, dst, n, nullptr, nullptr, s); }

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
template <typename D>
__device__ __forceinline__ void transpose_tile_store(D* __restrict__ dst, const TransDesc& d, int n0, int k0, float (*tile)[TRANSPOSE_TILE + 1], int ltid) {
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
        D* dp = dst + d.dst_off + (size_t)k * d.ld_dst + n;
        if (n4 && n + 4 <= d.ld_dst) store4(dp, v);
        else { for (int j = 0; j < 4; ++j) if (n + j < d.ld_dst) st_elem(dp + j, v[j]); }
    }
}
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const S* __restrict__ src, D* __restrict__ dst, const TransDesc* __restrict__ desc, int ndesc) {
    __shared__ float tile[TRANSPOSE_TILE][TRANSPOSE_TILE + 1];
    TransDesc d; int n0, k0;
    transpose_tile_load<S>(src, desc, ndesc, blockIdx.x, tile, threadIdx.x, d, n0, k0);
    __syncthreads();
    transpose_tile_store<D>(dst, d, n0, k0, tile, threadIdx.x);
}
// src: the f32 master weights, or (src_is_bf16) their bf16 copy at the same element offsets - half the bytes to read, same result;
// dst_split: the transposed copy is written in the split-f32 layout (bf16x3 precision; the source must then be the f32 master)
int k_cast_transpose(const void* src, int src_is_bf16, void* dst, int dst_split, const TransDesc* desc_dev, int ndesc, int total_tiles, hipStream_t s) {
    if (ndesc <= 0 || total_tiles <= 0) return RGQA_OK;
    RGQA_REQUIRE(!(dst_split && src_is_bf16), "cast_transpose: the split-f32 copy is made from the f32 master weights");
    if (dst_split) hipLaunchKernelGGL((cast_transpose_kernel<float, sf32>), dim3(total_tiles), dim3(256), 0, s, reinterpret_cast<const float*>(src), reinterpret_cast<sf32*>(dst), desc_dev, ndesc);
    else if (src_is_bf16) hipLaunchKernelGGL((cast_transpose_kernel<bf16_t, bf16_t>), dim3(total_tiles), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(src), reinterpret_cast<bf16_t*>(dst), desc_dev, ndesc);
    else hipLaunchKernelGGL((cast_transpose_kernel<float, bf16_t>), dim3(total_tiles), dim3(256), 0, s, reinterpret_cast<const float*>(src), reinterpret_cast<bf16_t*>(dst), desc_dev, ndesc);
    RGQA_LAUNCH_CHECK("cast_transpose_kernel");
    return RGQA_OK;
}
