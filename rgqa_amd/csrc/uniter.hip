// UNITER single-stream backbone (reference uniter/modeling.py:560-635): the kernels its embedding front-end
// needs beyond the shared ones.  The encoder itself is 12 BertLayers (uniter/modeling.py:418-557, the same module code as LXMERT's
// language layers) over ONE sequence per sample: [text tokens ; 36 image regions], which the engine lays out as consecutive rows
// (sample b owns rows cu[b] .. cu[b+1]-1 = its real text tokens followed by its regions), so the existing GEMM / LayerNorm /
// attention kernels run unchanged with 56-row windows.
//   text:   UniterTextEmbeddings (:560-591)  = word + position + token-type -> LayerNorm -> dropout   (embed.hip, with a row map)
//   image:  UniterImageEmbeddings (:594-612) = LN(img_linear(feat)) + LN(pos_linear(pos7)) + type_emb[1] -> LayerNorm -> dropout
#include "kernels.h"

// destination rows in the joint layout: text_dst[tcu[b] + t] = jcu[b] + t (t < text length), img_dst[b*O + o] = jcu[b] + len + o
__global__ void uniter_dst_kernel(const int* __restrict__ tcu, const int* __restrict__ jcu, int O, int* __restrict__ text_dst, int* __restrict__ img_dst) {
    const int b = blockIdx.x;
    const int t0 = tcu[b], len = tcu[b + 1] - t0, j0 = jcu[b];
    for (int t = threadIdx.x; t < len; t += blockDim.x) text_dst[t0 + t] = j0 + t;
    for (int o = threadIdx.x; o < O; o += blockDim.x) img_dst[b * O + o] = j0 + len + o;
}
int k_uniter_dst(const int* tcu, const int* jcu, int B, int O, int* text_dst, int* img_dst, hipStream_t s) {
    hipLaunchKernelGGL(uniter_dst_kernel, dim3(B), dim3(64), 0, s, tcu, jcu, O, text_dst, img_dst);
    RGQA_LAUNCH_CHECK("uniter_dst_kernel");
    return RGQA_OK;
}

// additive key mask of the joint sequence, padded layout: [B, T+O] = [(1 - input_mask) * -10000 ; 0]   (uniter/modeling.py:624-627)
__global__ void uniter_mask_kernel(const int64_t* __restrict__ m, float* __restrict__ out, int B, int T, int O) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (T + O)) return;
    const int b = i / (T + O), j = i % (T + O);
    out[i] = j < T ? (1.0f - (float)m[b * T + j]) * -10000.0f : 0.f;
}
int k_uniter_mask(const int64_t* input_mask, float* out, int B, int T, int O, hipStream_t s) {
    hipLaunchKernelGGL(uniter_mask_kernel, dim3(cdiv(B * (T + O), 256)), dim3(256), 0, s, input_mask, out, B, T, O);
    RGQA_LAUNCH_CHECK("uniter_mask_kernel");
    return RGQA_OK;
}

#define UNITER_MAXPOS 8
// zp[row, n] = bp[n] + sum_k pos[row, k] Wp[n, k]      (pos_linear, K = 7: no GEMM)
template <typename T>
__global__ __launch_bounds__(256) void pos_proj_kernel(const float* __restrict__ pos, int pd, const float* __restrict__ Wp, const float* __restrict__ bp,
                                                       T* __restrict__ out, int ldo, int M, int H) {
    const int row = blockIdx.x;
    float px[UNITER_MAXPOS];
#pragma unroll
    for (int k = 0; k < UNITER_MAXPOS; ++k) px[k] = k < pd ? pos[(size_t)row * pd + k] : 0.f;
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        float a = bp[n];
        for (int k = 0; k < pd; ++k) a = fmaf(px[k], Wp[(size_t)n * pd + k], a);
        st_elem(out + (size_t)row * ldo + n, a);
    }
}
template <typename T>
int k_pos_proj(const float* pos, int pd, const float* Wp, const float* bp, T* out, int ldo, int M, int H, hipStream_t s) {
    RGQA_REQUIRE(pd >= 1 && pd <= UNITER_MAXPOS, "pos_proj: pos_dim %d unsupported", pd);
    if (M <= 0) return RGQA_OK;
    hipLaunchKernelGGL(pos_proj_kernel<T>, dim3(M), dim3(256), 0, s, pos, pd, Wp, bp, out, ldo, M, H);
    RGQA_LAUNCH_CHECK("pos_proj_kernel");
    return RGQA_OK;
}

// dWp[n, k] (+)= sum_row dzp[row, n] pos[row, k]: per-block partials part[blk][k][H], folded in a fixed order with output stride pd
template <typename T>
__global__ __launch_bounds__(256) void pos_wgrad_kernel(const T* __restrict__ dzp, int ld, const float* __restrict__ pos, int pd, float* __restrict__ part, int M, int H) {
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        float acc[UNITER_MAXPOS];
#pragma unroll
        for (int k = 0; k < UNITER_MAXPOS; ++k) acc[k] = 0.f;
        for (int row = blockIdx.x; row < M; row += gridDim.x) {
            const float g = ld_elem(dzp + (size_t)row * ld + n);
#pragma unroll
            for (int k = 0; k < UNITER_MAXPOS; ++k) if (k < pd) acc[k] = fmaf(g, pos[(size_t)row * pd + k], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < UNITER_MAXPOS; ++k) if (k < pd) part[((size_t)blockIdx.x * pd + k) * H + n] = acc[k];
    }
}
template <typename T>
int k_pos_wgrad(const T* dzp, int ld, const float* pos, int pd, float* part, float* dWp, int accumulate, int M, int H, hipStream_t s) {
    RGQA_REQUIRE(pd >= 1 && pd <= UNITER_MAXPOS && pd <= FIN_MAXQ && H % 4 == 0, "pos_wgrad: pos_dim %d / H %d unsupported", pd, H);
    if (M <= 0) return RGQA_OK;
    const int nblk = M < 128 ? M : 128;
    hipLaunchKernelGGL(pos_wgrad_kernel<T>, dim3(nblk), dim3(256), 0, s, dzp, ld, pos, pd, part, M, H);
    RGQA_LAUNCH_CHECK("pos_wgrad_kernel");
    FinOut fo = {};
    for (int k = 0; k < pd; ++k) { fo.p[k] = dWp + k; fo.stride[k] = pd; }
    return k_colsum_finalize(part, nblk, pd, H, fo, accumulate, s);
}

// x3[row] = a[row] + b[row] + trow;  out[dst[row]] = dropout(LN(x3[row]));  x3 and the statistics are kept per LOCAL row (the
// LayerNorm backward runs on the contiguous image rows); the dropout stream is indexed by the local row as well
template <typename T, int NV>
__global__ __launch_bounds__(256) void sum3_ln_kernel(const T* __restrict__ a, const T* __restrict__ b, int ld, const float* __restrict__ trow, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const int* __restrict__ dst, T* __restrict__ out, int ldo, T* __restrict__ xsave,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int M, int H, float eps, DropCfg drop) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    const int nv = H >> 2;
    float v[NV][4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float x[4], y[4], t[4];
            load4(a + (size_t)row * ld + c * 4, x);
            load4(b + (size_t)row * ld + c * 4, y);
            load4(trow + c * 4, t);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = x[j] + y[j] + t[j];
            store4(xsave + (size_t)row * ld + c * 4, v[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += v[i][j];
        }
    }
    const float mu = wave_sum(sum) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)H + eps);
    const size_t orow = (size_t)dst[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float g[4], bb[4], o[4];
            load4(gamma + c * 4, g);
            load4(beta + c * 4, bb);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = drop_apply(drop, (uint32_t)row * (uint32_t)H + (uint32_t)(c * 4 + j), (v[i][j] - mu) * rs * g[j] + bb[j]);
            store4(out + orow * ldo + c * 4, o);
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
template <typename T>
int k_sum3_ln_fwd(const T* a, const T* b, int ld, const float* trow, const float* gamma, const float* beta, const int* dst, T* out, int ldo, T* xsave, float* mean, float* rstd,
                  int M, int H, float eps, DropCfg drop, hipStream_t s) {
    RGQA_REQUIRE(H % 4 == 0 && H <= 1024 && ld % 4 == 0 && ldo % 4 == 0, "sum3_ln: H=%d unsupported", H);
    if (M <= 0) return RGQA_OK;
#define S3(NVV) hipLaunchKernelGGL((sum3_ln_kernel<T, NVV>), dim3(cdiv(M, 4)), dim3(256), 0, s, a, b, ld, trow, gamma, beta, dst, out, ldo, xsave, mean, rstd, M, H, eps, drop)
    const int nvl = cdiv(H / 4, 64);
    if (nvl <= 1) S3(1); else if (nvl == 2) S3(2); else if (nvl == 3) S3(3); else S3(4);
#undef S3
    RGQA_LAUNCH_CHECK("sum3_ln_kernel");
    return RGQA_OK;
}

template int k_pos_proj<float>(const float*, int, const float*, const float*, float*, int, int, int, hipStream_t);
template int k_pos_proj<bf16_t>(const float*, int, const float*, const float*, bf16_t*, int, int, int, hipStream_t);
template int k_pos_proj<sf32>(const float*, int, const float*, const float*, sf32*, int, int, int, hipStream_t);
template int k_pos_wgrad<float>(const float*, int, const float*, int, float*, float*, int, int, int, hipStream_t);
template int k_pos_wgrad<bf16_t>(const bf16_t*, int, const float*, int, float*, float*, int, int, int, hipStream_t);
template int k_pos_wgrad<sf32>(const sf32*, int, const float*, int, float*, float*, int, int, int, hipStream_t);
template int k_sum3_ln_fwd<float>(const float*, const float*, int, const float*, const float*, const float*, const int*, float*, int, float*, float*, float*, int, int, float, DropCfg, hipStream_t);
template int k_sum3_ln_fwd<bf16_t>(const bf16_t*, const bf16_t*, int, const float*, const float*, const float*, const int*, bf16_t*, int, bf16_t*, float*, float*, int, int, float, DropCfg, hipStream_t);
template int k_sum3_ln_fwd<sf32>(const sf32*, const sf32*, int, const float*, const float*, const float*, const int*, sf32*, int, sf32*, float*, float*, int, int, float, DropCfg, hipStream_t);
