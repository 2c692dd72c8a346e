// BertEmbeddings (reference lxrt/modeling.py:264-292): word + position + token-type gather, LayerNorm, dropout.
// One wave per token; HBM-bound gather (3 table rows in, 1 activation row out).
#include "kernels.h"

// row_src (varlen mode): packed row -> b*Tn + t of the token it holds; null = identity (padded layout)
// row_dst (UNITER joint layout): output row of `out` for this token; the saved pre-LN sum, the statistics and the dropout stream stay
// indexed by the token's own (contiguous) row
template <typename T, int NV>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ seg, const int* __restrict__ row_src, const int* __restrict__ row_dst, const float* __restrict__ word,
                                                        const float* __restrict__ pos, const float* __restrict__ type, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, int ldo, T* __restrict__ zsave,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int rows, int Tn, int H, float eps, DropCfg drop) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int src = row_src ? row_src[row] : row;
    const int t = src % Tn;
    const int64_t id = ids[src], sg = seg ? seg[src] : 0;
    const int nv = H >> 2;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float a[4], b[4], d[4];
            load4(word + (size_t)id * H + c * 4, a);
            load4(pos + (size_t)t * H + c * 4, b);
            load4(type + (size_t)sg * H + c * 4, d);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[i][j] = a[j] + b[j] + d[j]; s += v[i][j]; }
            if (zsave) store4(zsave + (size_t)row * ldo + c * 4, v[i]);
        }
    }
    const float mu = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[i][j] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float g[4], b[4], o[4];
            load4(gamma + c * 4, g);
            load4(beta + c * 4, b);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = drop_apply(drop, (uint32_t)row * (uint32_t)H + (uint32_t)(c * 4 + j), (v[i][j] - mu) * rs * g[j] + b[j]);
            store4(out + (size_t)(row_dst ? row_dst[row] : row) * ldo + c * 4, o);
        }
    }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// de [rows, H] f32 -> atomic scatter-add into the tables (dense f32 gradients, as the reference's
// nn.Embedding produces; rows with index 0 are skipped: padding_idx=0 on all three tables in LXMERT (pad0_all), on the word
// table only in UNITER (uniter/modeling.py:563-568)).
template <typename T>
__global__ __launch_bounds__(256) void embed_scatter_kernel(const T* __restrict__ de, const int64_t* __restrict__ ids, const int64_t* __restrict__ seg, const int* __restrict__ row_src,
                                                            float* __restrict__ dword, float* __restrict__ dpos, float* __restrict__ dtype, int rows, int Tn, int H, int pad0_all) {
    const int row = blockIdx.x;
    if (row >= rows) return;
    const int src = row_src ? row_src[row] : row;
    const int t = src % Tn;
    const int64_t id = ids[src], sg = seg ? seg[src] : 0;
    for (int n = threadIdx.x; n < H; n += 256) {
        const float g = ld_elem(de + (size_t)row * H + n);
        if (id != 0) atomicAdd(dword + (size_t)id * H + n, g);
        if (t != 0 || !pad0_all) atomicAdd(dpos + (size_t)t * H + n, g);
        if (sg != 0 || !pad0_all) atomicAdd(dtype + (size_t)sg * H + n, g);
    }
}

__global__ void make_mask_kernel(const int64_t* __restrict__ m, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (1.0f - (float)m[i]) * -10000.0f;
}

template <typename T>
int k_embed_fwd(const int64_t* ids, const int64_t* seg, const int* row_src, const int* row_dst, int rows, const float* word, const float* pos, const float* type, const float* gamma, const float* beta,
                T* out, int ldo, T* zsave, float* mean, float* rstd, int B, int Tn, int H, int vocab, int type_vocab, float eps, DropCfg drop, hipStream_t s) {
    RGQA_REQUIRE(H % 4 == 0 && H <= 2048 && ldo % 4 == 0, "embed: hidden %d unsupported", H);
    RGQA_REQUIRE(rows <= B * Tn, "embed: %d rows exceed B*T = %d", rows, B * Tn);
    if (rows <= 0) return RGQA_OK;
#define EMB(NVV) hipLaunchKernelGGL((embed_fwd_kernel<T, NVV>), dim3(cdiv(rows, 4)), dim3(256), 0, s, ids, seg, row_src, row_dst, word, pos, type, gamma, beta, out, ldo, zsave, mean, rstd, rows, Tn, H, eps, drop)
    const int nvl = cdiv(H / 4, 64);
    if (nvl <= 1) EMB(1); else if (nvl == 2) EMB(2); else if (nvl == 3) EMB(3); else if (nvl == 4) EMB(4); else EMB(8);
#undef EMB
    RGQA_LAUNCH_CHECK("embed_fwd_kernel");
    return RGQA_OK;
}

template <typename T>
int k_embed_scatter(const T* de, const int64_t* ids, const int64_t* seg, const int* row_src, int rows, float* dword, float* dpos, float* dtype, int B, int Tn, int H, int pad0_all, hipStream_t s) {
    RGQA_REQUIRE(rows <= B * Tn, "embed scatter: %d rows exceed B*T = %d", rows, B * Tn);
    if (rows <= 0) return RGQA_OK;
    hipLaunchKernelGGL(embed_scatter_kernel<T>, dim3(rows), dim3(256), 0, s, de, ids, seg, row_src, dword, dpos, dtype, rows, Tn, H, pad0_all);
    RGQA_LAUNCH_CHECK("embed_scatter_kernel");
    return RGQA_OK;
}

int k_make_mask(const int64_t* input_mask, float* out, int n, hipStream_t s) {
    if (n <= 0) return RGQA_OK;
    hipLaunchKernelGGL(make_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, input_mask, out, n);
    RGQA_LAUNCH_CHECK("make_mask_kernel");
    return RGQA_OK;
}

template int k_embed_fwd<float>(const int64_t*, const int64_t*, const int*, const int*, int, const float*, const float*, const float*, const float*, const float*, float*, int, float*, float*, float*, int, int, int, int, int, float, DropCfg, hipStream_t);
template int k_embed_fwd<bf16_t>(const int64_t*, const int64_t*, const int*, const int*, int, const float*, const float*, const float*, const float*, const float*, bf16_t*, int, bf16_t*, float*, float*, int, int, int, int, int, float, DropCfg, hipStream_t);
template int k_embed_fwd<sf32>(const int64_t*, const int64_t*, const int*, const int*, int, const float*, const float*, const float*, const float*, const float*, sf32*, int, sf32*, float*, float*, int, int, int, int, int, float, DropCfg, hipStream_t);
template int k_embed_scatter<sf32>(const sf32*, const int64_t*, const int64_t*, const int*, int, float*, float*, float*, int, int, int, int, hipStream_t);
template int k_embed_scatter<float>(const float*, const int64_t*, const int64_t*, const int*, int, float*, float*, float*, int, int, int, int, hipStream_t);
template int k_embed_scatter<bf16_t>(const bf16_t*, const int64_t*, const int64_t*, const int*, int, float*, float*, float*, int, int, int, int, hipStream_t);
