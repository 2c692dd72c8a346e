// RoI-mixup device gather (reference tasks/gqa_mixup_vis.py:134-181) and small utilities.
#include "kernels.h"

// rows [B,2B) of feats [2B,O,F] / boxes [2B,O,4]: RoI o of sample j comes from the positive sample j when
// take_pos[j][o], else from the partner sample (zeros for feats in mixup_v3). One block per (j, o): an
// 8 KB row copy for F=2048, HBM-bound.
__global__ __launch_bounds__(256) void mixup_gather_kernel(float* __restrict__ feats, float* __restrict__ boxes, const int32_t* __restrict__ partner,
                                                           const uint8_t* __restrict__ take_pos, int B, int O, int F, int v3) {
    const int j = blockIdx.x / O, o = blockIdx.x % O;
    const bool pos = take_pos[(size_t)j * O + o] != 0;
    const int srcb = pos ? j : partner[j];
    const float* sf = feats + ((size_t)srcb * O + o) * F;
    float* df = feats + ((size_t)(B + j) * O + o) * F;
    const bool zero = v3 && !pos;
    for (int c = threadIdx.x; c < (F >> 2); c += 256) {
        float4 v = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4*>(sf)[c];
        reinterpret_cast<float4*>(df)[c] = v;
    }
    if (threadIdx.x < 4) boxes[((size_t)(B + j) * O + o) * 4 + threadIdx.x] = boxes[((size_t)srcb * O + o) * 4 + threadIdx.x];
}
int k_mixup_gather(float* feats, float* boxes, const int32_t* partner, const uint8_t* take_pos, int B, int O, int F, int mode_v3, hipStream_t s) {
    RGQA_REQUIRE(F % 4 == 0 && B > 0 && O > 0, "mixup_gather: bad shape");
    hipLaunchKernelGGL(mixup_gather_kernel, dim3(B * O), dim3(256), 0, s, feats, boxes, partner, take_pos, B, O, F, mode_v3);
    RGQA_LAUNCH_CHECK("mixup_gather_kernel");
    return RGQA_OK;
}

// 'perturb' (gqa_mixup_vis.py:124-133): rows [B,2B) = the features again, boxes[B+j][o] = boxes[j][perm[o]] (one permutation for the batch)
__global__ __launch_bounds__(256) void mixup_perturb_kernel(float* __restrict__ feats, float* __restrict__ boxes, const int32_t* __restrict__ perm, int B, int O, int F) {
    const int j = blockIdx.x / O, o = blockIdx.x % O;
    const float* sf = feats + ((size_t)j * O + o) * F;
    float* df = feats + ((size_t)(B + j) * O + o) * F;
    for (int c = threadIdx.x; c < (F >> 2); c += 256) reinterpret_cast<float4*>(df)[c] = reinterpret_cast<const float4*>(sf)[c];
    if (threadIdx.x < 4) boxes[((size_t)(B + j) * O + o) * 4 + threadIdx.x] = boxes[((size_t)j * O + perm[o]) * 4 + threadIdx.x];
}
int k_mixup_perturb(float* feats, float* boxes, const int32_t* perm, int B, int O, int F, hipStream_t s) {
    RGQA_REQUIRE(F % 4 == 0 && B > 0 && O > 0, "mixup_perturb: bad shape");
    hipLaunchKernelGGL(mixup_perturb_kernel, dim3(B * O), dim3(256), 0, s, feats, boxes, perm, B, O, F);
    RGQA_LAUNCH_CHECK("mixup_perturb_kernel");
    return RGQA_OK;
}

// 'weighted_sum' (gqa_mixup_vis.py:217-244): feats[B+j] = feats[j] * p[j] + feats[partner[j]] * q[j], every product rounded to f32
// before the sum as torch does (fp contraction off: no fused multiply-add); q[j] = f32(1 - prop) is formed on the host in double
// like the reference's Python expression; boxes repeated
__global__ __launch_bounds__(256) void mixup_wsum_kernel(float* __restrict__ feats, float* __restrict__ boxes, const int32_t* __restrict__ partner,
                                                         const float* __restrict__ p, const float* __restrict__ q, int B, int O, int F) {
#pragma clang fp contract(off)      // without this the sum becomes an FMA (one rounding instead of torch's three)
    const int j = blockIdx.x / O, o = blockIdx.x % O;
    const float pj = p[j], qj = q[j];
    const float4* sp = reinterpret_cast<const float4*>(feats + ((size_t)j * O + o) * F);
    const float4* sn = reinterpret_cast<const float4*>(feats + ((size_t)partner[j] * O + o) * F);
    float4* df = reinterpret_cast<float4*>(feats + ((size_t)(B + j) * O + o) * F);
    for (int c = threadIdx.x; c < (F >> 2); c += 256) {
        const float4 a = sp[c], b = sn[c];
        float4 r;
        // plain operators under the pragma above (the __fmul_rn / __fadd_rn wrappers are inlined WITH the header's contraction setting)
        const float ax = a.x * pj, ay = a.y * pj, az = a.z * pj, aw = a.w * pj;
        const float bx = b.x * qj, by = b.y * qj, bz = b.z * qj, bw = b.w * qj;
        r.x = ax + bx; r.y = ay + by; r.z = az + bz; r.w = aw + bw;
        df[c] = r;
    }
    if (threadIdx.x < 4) boxes[((size_t)(B + j) * O + o) * 4 + threadIdx.x] = boxes[((size_t)j * O + o) * 4 + threadIdx.x];
}
int k_mixup_weighted_sum(float* feats, float* boxes, const int32_t* partner, const float* p, const float* q, int B, int O, int F, hipStream_t s) {
    RGQA_REQUIRE(F % 4 == 0 && B > 0 && O > 0, "mixup_weighted_sum: bad shape");
    hipLaunchKernelGGL(mixup_wsum_kernel, dim3(B * O), dim3(256), 0, s, feats, boxes, partner, p, q, B, O, F);
    RGQA_LAUNCH_CHECK("mixup_wsum_kernel");
    return RGQA_OK;
}

// target[row0 + j][:] = target[j][:] * prop[j]
__global__ void scale_rows_kernel(float* __restrict__ target, const float* __restrict__ prop, int NA, int ld, int row0) {
    const int j = blockIdx.x;
    const float p = prop[j];
    for (int n = threadIdx.x; n < NA; n += blockDim.x) target[(size_t)(row0 + j) * ld + n] = target[(size_t)j * ld + n] * p;
}
int k_scale_rows(float* target, const float* prop, int B, int NA, int ld, int row0, hipStream_t s) {
    hipLaunchKernelGGL(scale_rows_kernel, dim3(B), dim3(256), 0, s, target, prop, NA, ld, row0);
    RGQA_LAUNCH_CHECK("scale_rows_kernel");
    return RGQA_OK;
}

template <typename T>
__global__ void fill_rows_kernel(T* __restrict__ dst, int ld, const T* __restrict__ src, int lds, int rows, int cols) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[(size_t)r * ld + c] = src[(size_t)r * lds + c];
}
template <typename T>
int k_fill_rows(T* dst, int ld, const T* src, int lds, int rows, int cols, hipStream_t s) {
    if (rows <= 0) return RGQA_OK;
    hipLaunchKernelGGL(fill_rows_kernel<T>, dim3(rows), dim3(256), 0, s, dst, ld, src, lds, rows, cols);
    RGQA_LAUNCH_CHECK("fill_rows_kernel");
    return RGQA_OK;
}
template int k_fill_rows<float>(float*, int, const float*, int, int, int, hipStream_t);
template int k_fill_rows<bf16_t>(bf16_t*, int, const bf16_t*, int, int, int, hipStream_t);

// dst[r][c] = (T) src[r][c] for c < cols, 0 for cols <= c < ldd  (pads K-side buffers with exact zeros)
template <typename T>
__global__ void cast_pad_kernel(const float* __restrict__ src, int lds, T* __restrict__ dst, int ldd, int cols, float scale) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < ldd; c += blockDim.x) st_elem(dst + (size_t)r * ldd + c, c < cols ? src[(size_t)r * lds + c] * scale : 0.f);
}
template <typename T>
int k_cast_pad(const float* src, int lds, T* dst, int ldd, int rows, int cols, float scale, hipStream_t s) {
    if (rows <= 0) return RGQA_OK;
    hipLaunchKernelGGL(cast_pad_kernel<T>, dim3(rows), dim3(256), 0, s, src, lds, dst, ldd, cols, scale);
    RGQA_LAUNCH_CHECK("cast_pad_kernel");
    return RGQA_OK;
}
template int k_cast_pad<float>(const float*, int, float*, int, int, int, float, hipStream_t);
template int k_cast_pad<bf16_t>(const float*, int, bf16_t*, int, int, int, float, hipStream_t);
template int k_cast_pad<sf32>(const float*, int, sf32*, int, int, int, float, hipStream_t);

// dst f32 [rows, cols] = src T
template <typename T>
__global__ void to_f32_kernel(const T* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int cols) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[(size_t)r * ldd + c] = ld_elem(src + (size_t)r * lds + c);
}
template <typename T>
int k_to_f32(const T* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s) {
    if (rows <= 0) return RGQA_OK;
    hipLaunchKernelGGL(to_f32_kernel<T>, dim3(rows), dim3(256), 0, s, src, lds, dst, ldd, cols);
    RGQA_LAUNCH_CHECK("to_f32_kernel");
    return RGQA_OK;
}
template int k_to_f32<float>(const float*, int, float*, int, int, int, hipStream_t);
template int k_to_f32<bf16_t>(const bf16_t*, int, float*, int, int, int, hipStream_t);
template int k_to_f32<sf32>(const sf32*, int, float*, int, int, int, hipStream_t);

// y = dy * dg, dg = the gelu'(pre) saved by the forward EPI_GELU epilogue (head backward: no GEMM between LN-bwd and GeLU)
template <typename T>
__global__ void dgelu_mul_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        st_elem(out + i, ld_elem(dy + i) * ld_elem(pre + i));
}
template <typename T>
int k_dgelu_mul(const T* dy, const T* pre, T* out, size_t n, hipStream_t s) {
    if (n == 0) return RGQA_OK;
    int nb = (int)((n + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(dgelu_mul_kernel<T>, dim3(nb), dim3(256), 0, s, dy, pre, out, n);
    RGQA_LAUNCH_CHECK("dgelu_mul_kernel");
    return RGQA_OK;
}
template int k_dgelu_mul<float>(const float*, const float*, float*, size_t, hipStream_t);
template int k_dgelu_mul<bf16_t>(const bf16_t*, const bf16_t*, bf16_t*, size_t, hipStream_t);
template int k_dgelu_mul<sf32>(const sf32*, const sf32*, sf32*, size_t, hipStream_t);

// out = dy * (1 - y^2)   (backward through tanh given its output)
template <typename T>
__global__ void dtanh_mul_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float t = ld_elem(y + i);
        st_elem(out + i, ld_elem(dy + i) * (1.0f - t * t));
    }
}
template <typename T>
int k_dtanh_mul(const T* dy, const T* y, T* out, size_t n, hipStream_t s) {
    if (n == 0) return RGQA_OK;
    int nb = (int)((n + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(dtanh_mul_kernel<T>, dim3(nb), dim3(256), 0, s, dy, y, out, n);
    RGQA_LAUNCH_CHECK("dtanh_mul_kernel");
    return RGQA_OK;
}
template int k_dtanh_mul<float>(const float*, const float*, float*, size_t, hipStream_t);
template int k_dtanh_mul<bf16_t>(const bf16_t*, const bf16_t*, bf16_t*, size_t, hipStream_t);
template int k_dtanh_mul<sf32>(const sf32*, const sf32*, sf32*, size_t, hipStream_t);

// ---- unpadded ("varlen") language rows: per-sample lengths -> cu (exclusive prefix sums) and the packed-row -> token map.
// The lengths travel as kernel ARGUMENTS (copied at launch time): no host buffer has to outlive the call and nothing
// synchronises, unlike a pageable hipMemcpyAsync.
struct LenPack { int v[512]; };
__global__ void store_lens_kernel(const LenPack p, int n, int off, int* __restrict__ lens) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lens[off + i] = p.v[i];
}
__global__ __launch_bounds__(1024) void build_cu_kernel(const int* __restrict__ lens, int B, int Tn, int* __restrict__ cu, int* __restrict__ row_src) {
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < B; ++b) { cu[b] = acc; acc += lens[b]; }
        cu[B] = acc;
    }
    __threadfence_block();
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int c0 = cu[b], n = lens[b];
        for (int t = 0; t < n; ++t) row_src[c0 + t] = b * Tn + t;
    }
}
int k_set_lengths(const int* lens_host, int B, int Tn, int* lens_dev, int* cu_dev, int* row_src_dev, hipStream_t s) {
    for (int off = 0; off < B; off += 512) {
        LenPack p;
        const int n = B - off < 512 ? B - off : 512;
        for (int i = 0; i < n; ++i) p.v[i] = lens_host[off + i];
        for (int i = n; i < 512; ++i) p.v[i] = 0;
        hipLaunchKernelGGL(store_lens_kernel, dim3(2), dim3(256), 0, s, p, n, off, lens_dev);
        RGQA_LAUNCH_CHECK("store_lens_kernel");
    }
    hipLaunchKernelGGL(build_cu_kernel, dim3(1), dim3(1024), 0, s, lens_dev, B, Tn, cu_dev, row_src_dev);
    RGQA_LAUNCH_CHECK("build_cu_kernel");
    return RGQA_OK;
}

// dst[b][:] = src[row(b)][:] (gather) / dst[row(b)][:] = src[b][:] (scatter); row(b) = cu ? cu[b] : b * stride_rows.
// The [CLS] rows feeding BertPooler (modeling.py:575-581) in either layout.
template <typename T, bool SCATTER>
__global__ void pick_rows_kernel(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, const int* __restrict__ cu, int stride_rows, int cols) {
    const int b = blockIdx.x;
    const size_t r = cu ? (size_t)cu[b] : (size_t)b * stride_rows;
    const T* sp = src + (SCATTER ? (size_t)b : r) * lds;
    T* dp = dst + (SCATTER ? r : (size_t)b) * ldd;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dp[c] = sp[c];
}
template <typename T>
int k_gather_rows(const T* src, int lds, const int* cu, int stride_rows, T* dst, int ldd, int rows, int cols, hipStream_t s) {
    if (rows <= 0) return RGQA_OK;
    hipLaunchKernelGGL((pick_rows_kernel<T, false>), dim3(rows), dim3(256), 0, s, src, lds, dst, ldd, cu, stride_rows, cols);
    RGQA_LAUNCH_CHECK("pick_rows_kernel(gather)");
    return RGQA_OK;
}
template <typename T>
int k_scatter_rows(const T* src, int lds, T* dst, int ldd, const int* cu, int stride_rows, int rows, int cols, hipStream_t s) {
    if (rows <= 0) return RGQA_OK;
    hipLaunchKernelGGL((pick_rows_kernel<T, true>), dim3(rows), dim3(256), 0, s, src, lds, dst, ldd, cu, stride_rows, cols);
    RGQA_LAUNCH_CHECK("pick_rows_kernel(scatter)");
    return RGQA_OK;
}
template int k_gather_rows<float>(const float*, int, const int*, int, float*, int, int, int, hipStream_t);
template int k_gather_rows<bf16_t>(const bf16_t*, int, const int*, int, bf16_t*, int, int, int, hipStream_t);
template int k_scatter_rows<float>(const float*, int, float*, int, const int*, int, int, int, hipStream_t);
template int k_scatter_rows<bf16_t>(const bf16_t*, int, bf16_t*, int, const int*, int, int, int, hipStream_t);
// split-f32 rows are copied slot by slot (whole 128-byte lines: cols % 32 == 0), which moves the hi and lo parts unchanged
template int k_gather_rows<sf32>(const sf32*, int, const int*, int, sf32*, int, int, int, hipStream_t);
template int k_scatter_rows<sf32>(const sf32*, int, sf32*, int, const int*, int, int, int, hipStream_t);

// bf16 image of a split-f32 matrix (bf16x3_fwd precision): the hi part of every element IS bf16(x); 8 elements per thread, two 16-byte accesses
__global__ __launch_bounds__(256) void sf_image_kernel(const sf32* __restrict__ src, int lds, bf16_t* __restrict__ dst, int ldd, int rows, int cols8) {
    const size_t n = (size_t)rows * cols8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cols8), c = (int)(i % cols8) * 8;
        *reinterpret_cast<bf16x8*>(dst + (size_t)r * ldd + c) = *reinterpret_cast<const bf16x8*>(sf_hi(src + (size_t)r * lds + c));
    }
}
int k_sf_image(const sf32* src, int lds, bf16_t* dst, int ldd, int rows, int cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return RGQA_OK;
    RGQA_REQUIRE(cols % 8 == 0 && lds % 32 == 0 && ldd % 8 == 0 && ((uintptr_t)src % 128) == 0 && ((uintptr_t)dst % 16) == 0, "sf_image: cols %% 8, whole split-f32 lines and a 16-byte aligned image required");
    const size_t nb = ((size_t)rows * (cols / 8) + 255) / 256;
    hipLaunchKernelGGL(sf_image_kernel, dim3(nb > 2048 ? 2048 : (int)nb), dim3(256), 0, s, src, lds, dst, ldd, rows, cols / 8);
    RGQA_LAUNCH_CHECK("sf_image_kernel");
    return RGQA_OK;
}

// out[i] (+)= sum_s part[s][i]  (fixed order: deterministic split-K reduction of a weight gradient)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int S, size_t n, float* __restrict__ out, int accumulate) {
    const size_t nv = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (accumulate) load4(out + i * 4, a);
        for (int s = 0; s < S; ++s) {
            float v[4];
            load4(part + (size_t)s * n + i * 4, v);
            a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        }
        store4(out + i * 4, a);
    }
}
int k_sum_partials(const float* part, int S, size_t n, float* out, int accumulate, hipStream_t s) {
    RGQA_REQUIRE(n % 4 == 0 && S >= 1, "sum_partials: n=%zu must be a multiple of 4", n);
    size_t nb = (n / 4 + 255) / 256;
    const int nblk = nb > 2048 ? 2048 : (int)nb;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(nblk), dim3(256), 0, s, part, S, n, out, accumulate);
    RGQA_LAUNCH_CHECK("sum_partials_kernel");
    return RGQA_OK;
}
