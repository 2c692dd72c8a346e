// Kernels of the BUTD GQA path (reference src/butd/butd.py; BASELINE config 5): GloVe embedding
// gather, GRU gates, product-fusion attention over the 36 regions (softmax over regions, weighted feature sum),
// weight-norm (scalar g) weight preparation / gradient, and small elementwise helpers.  The projections themselves run
// on the GEMM kernels shared with the LXMERT path.  All HBM-bound; T = float (parity) or bf16.
#include <type_traits>
#include "kernels.h"
#include "butd.h"

// ---------------------------------------------------------------- embedding
template <typename T>
__global__ void butd_embed_fwd_kernel(const int64_t* __restrict__ toks, const float* __restrict__ table, T* __restrict__ out, int E, int Ep) {
    const int row = blockIdx.x;
    const int64_t id = toks[row];
    for (int c = threadIdx.x; c < Ep; c += blockDim.x) st_elem(out + ((size_t)row * Ep + c), c < E ? table[(size_t)id * E + c] : 0.f);
}
template <typename T>
__global__ void butd_embed_bwd_kernel(const int64_t* __restrict__ toks, const T* __restrict__ dx, float* __restrict__ dtable, int E, int Ep, int pad_idx) {
    const int row = blockIdx.x;
    const int64_t id = toks[row];
    if (id == pad_idx) return;           // nn.Embedding(padding_idx=ntoken): no gradient (butd.py:36)
    for (int c = threadIdx.x; c < E; c += blockDim.x) atomicAdd(dtable + (size_t)id * E + c, ld_elem(dx + ((size_t)row * Ep + c)));
}

// ---------------------------------------------------------------- GRU gates (gate order r, z, n as torch.nn.GRU)
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T>
__global__ void gru_gate_fwd_kernel(const T* __restrict__ gi, long ldgi, const T* __restrict__ gh, const T* __restrict__ hprev,
                                    T* __restrict__ hnew, T* __restrict__ rs, T* __restrict__ zs, T* __restrict__ ns, T* __restrict__ ghn, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, h = i % H;
    const T* gib = gi + (size_t)b * ldgi;
    const T* ghb = gh + (size_t)b * 3 * H;
    const float r = sigm(ld_elem(gib + (h)) + ld_elem(ghb + (h)));
    const float z = sigm(ld_elem(gib + (H + h)) + ld_elem(ghb + (H + h)));
    const float gn = ld_elem(ghb + (2 * H + h));
    const float n = tanhf(ld_elem(gib + (2 * H + h)) + r * gn);
    const float hp = ld_elem(hprev + (i));
    st_elem(hnew + (i), (1.f - z) * n + z * hp);
    st_elem(rs + (i), r); st_elem(zs + (i), z); st_elem(ns + (i), n); st_elem(ghn + (i), gn);
}
// given dh (gradient w.r.t. h_t): dgi / dgh for this step and the direct part of dh_{t-1} (= dh * z)
template <typename T>
__global__ void gru_gate_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ hprev, const T* __restrict__ rs, const T* __restrict__ zs,
                                    const T* __restrict__ ns, const T* __restrict__ ghn, T* __restrict__ dgi, long lddgi, T* __restrict__ dgh,
                                    T* __restrict__ dhprev, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, h = i % H;
    const float d = ld_elem(dh + (i)), r = ld_elem(rs + (i)), z = ld_elem(zs + (i)), n = ld_elem(ns + (i)), gn = ld_elem(ghn + (i)), hp = ld_elem(hprev + (i));
    const float dn = d * (1.f - z) * (1.f - n * n);
    const float dz = d * (hp - n) * z * (1.f - z);
    const float dr = dn * gn * r * (1.f - r);
    T* gi = dgi + (size_t)b * lddgi;
    T* gh = dgh + (size_t)b * 3 * H;
    st_elem(gi + (h), dr); st_elem(gi + (H + h), dz); st_elem(gi + (2 * H + h), dn);
    st_elem(gh + (h), dr); st_elem(gh + (H + h), dz); st_elem(gh + (2 * H + h), dn * r);
    st_elem(dhprev + (i), d * z);
}

// ---------------------------------------------------------------- image features = cat(feat, pos), padded to a multiple of 8 columns
template <typename T>
__global__ void concat_cast_kernel(const float* __restrict__ feat, const float* __restrict__ pos, T* __restrict__ out, int F, int Pd, int Dp) {
    const int row = blockIdx.x;
    for (int c = threadIdx.x; c < Dp; c += blockDim.x) {
        float v = 0.f;
        if (c < F) v = feat[(size_t)row * F + c];
        else if (c < F + Pd) v = pos[(size_t)row * Pd + (c - F)];
        st_elem(out + ((size_t)row * Dp + c), v);
    }
}

// ---------------------------------------------------------------- attention over regions (butd.py:87-104, 207-208)
// one workgroup (256 threads) per sample; O <= 64 regions
template <typename T>
__global__ __launch_bounds__(256) void butd_attend_fwd_kernel(const T* __restrict__ ip, const T* __restrict__ qp, const float* __restrict__ wlin, const float* __restrict__ blin,
                                                              const T* __restrict__ imgf, float* __restrict__ att, T* __restrict__ img_enc, int O, int H, int Dp, DropCfg drop) {
    __shared__ float lg[64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* ipb = ip + (size_t)b * O * H;
    for (int k = wave; k < O; k += 4) {
        float s = 0.f;
        for (int h = lane; h < H; h += 64) {
            const float j = ld_elem(ipb + ((size_t)k * H + h)) * ld_elem(qp + ((size_t)b * H + h));
            s += drop_apply(drop, (uint32_t)((b * O + k) * H + h), j) * wlin[h];
        }
        s = wave_sum(s);
        if (lane == 0) lg[k] = s + blin[0];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const float v = lane < O ? lg[lane] : -INFINITY;
        const float m = wave_max(v);
        const float e = lane < O ? __expf(v - m) : 0.f;
        const float sum = wave_sum(e);
        if (lane < O) { lg[lane] = e / sum; att[(size_t)b * O + lane] = e / sum; }
    }
    __syncthreads();
    const T* fb = imgf + (size_t)b * O * Dp;
    for (int f = threadIdx.x; f < Dp; f += 256) {
        float s = 0.f;
        for (int k = 0; k < O; ++k) s = fmaf(lg[k], ld_elem(fb + ((size_t)k * Dp + f)), s);
        st_elem(img_enc + ((size_t)b * Dp + f), s);
    }
}

// backward: d_img_enc -> d(image_proj pre-ReLU) [B,O,H], d(question_proj pre-ReLU) [B,H], per-sample partials of dw_lin [B,H] and db_lin [B]
template <typename T>
__global__ __launch_bounds__(256) void butd_attend_bwd_kernel(const T* __restrict__ dimg, const T* __restrict__ imgf, const float* __restrict__ att, const T* __restrict__ ip,
                                                              const T* __restrict__ qp, const float* __restrict__ wlin, T* __restrict__ dip, T* __restrict__ dqp,
                                                              float* __restrict__ dw_part, float* __restrict__ db_part, int O, int H, int Dp, DropCfg drop) {
    __shared__ float dl[64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* fb = imgf + (size_t)b * O * Dp;
    for (int k = wave; k < O; k += 4) {
        float s = 0.f;
        for (int f = lane; f < Dp; f += 64) s = fmaf(ld_elem(dimg + ((size_t)b * Dp + f)), ld_elem(fb + ((size_t)k * Dp + f)), s);
        s = wave_sum(s);
        if (lane == 0) dl[k] = s;                 // d att_k
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const float a = lane < O ? att[(size_t)b * O + lane] : 0.f;
        const float da = lane < O ? dl[lane] : 0.f;
        const float dot = wave_sum(a * da);
        const float d = a * (da - dot);            // d logit_k
        const float dbs = wave_sum(d);
        if (lane < O) dl[lane] = d;
        if (lane == 0) db_part[b] = dbs;
    }
    __syncthreads();
    const T* ipb = ip + (size_t)b * O * H;
    T* dipb = dip + (size_t)b * O * H;
    for (int h = threadIdx.x; h < H; h += 256) {
        const float q = ld_elem(qp + ((size_t)b * H + h)), w = wlin[h];
        float dq = 0.f, dw = 0.f;
        for (int k = 0; k < O; ++k) {
            const float x = ld_elem(ipb + ((size_t)k * H + h));
            const float keep = drop_apply(drop, (uint32_t)((b * O + k) * H + h), 1.0f);
            const float dj = dl[k] * w * keep;      // d joint[k][h]
            dq = fmaf(dj, x, dq);
            dw = fmaf(dl[k] * keep, x * q, dw);
            st_elem(dipb + ((size_t)k * H + h), x > 0.f ? dj * q : 0.f);     // through the ReLU of image_proj
        }
        st_elem(dqp + ((size_t)b * H + h), q > 0.f ? dq : 0.f);               // through the ReLU of question_proj
        dw_part[(size_t)b * H + h] = dw;
    }
}

// ---- bf16 fast paths of the two kernels above (round 5): 16-byte / 8-byte accesses instead of one 2-byte element per lane and load (100 / 120 us per
// launch for 57 / 76 MB: 0.6 TB/s), the per-sample vectors (question projection, w_lin, d img_enc) held in registers.  Same arithmetic per element,
// the sums in another order.  H % 512 == 0, Dp % 8 == 0, H <= 2048.
template <int HQ>      // HQ = H / 512: bf16x8 chunks per lane of an H-long row
__global__ __launch_bounds__(256) void butd_attend_fwd_bf16_kernel(const bf16_t* __restrict__ ip, const bf16_t* __restrict__ qp, const float* __restrict__ wlin,
                                                                   const float* __restrict__ blin, const bf16_t* __restrict__ imgf, float* __restrict__ att,
                                                                   bf16_t* __restrict__ img_enc, int O, int Dp, DropCfg drop) {
    constexpr int H = HQ * 512;
    __shared__ float lg[64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16_t* ipb = ip + (size_t)b * O * H;
    float q[HQ][8], w[HQ][8];
#pragma unroll
    for (int c = 0; c < HQ; ++c) {
        const int h0 = (c * 64 + lane) * 8;
        const bf16x8 qv = *reinterpret_cast<const bf16x8*>(qp + (size_t)b * H + h0);
        load4(wlin + h0, w[c]); load4(wlin + h0 + 4, w[c] + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) q[c][j] = (float)qv[j];
    }
    for (int k = wave; k < O; k += 4) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HQ; ++c) {
            const int h0 = (c * 64 + lane) * 8;
            const bf16x8 xv = *reinterpret_cast<const bf16x8*>(ipb + (size_t)k * H + h0);
            float j8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) j8[j] = (float)xv[j] * q[c][j];
            drop_apply_vec<8>(drop, (uint32_t)((b * O + k) * H + h0), j8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s = fmaf(j8[j], w[c][j], s);
        }
        s = wave_sum(s);
        if (lane == 0) lg[k] = s + blin[0];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const float v = lane < O ? lg[lane] : -INFINITY;
        const float m = wave_max(v);
        const float e = lane < O ? __expf(v - m) : 0.f;
        const float sum = wave_sum(e);
        if (lane < O) { lg[lane] = e / sum; att[(size_t)b * O + lane] = e / sum; }
    }
    __syncthreads();
    const bf16_t* fb = imgf + (size_t)b * O * Dp;
    for (int c = threadIdx.x; c < (Dp >> 3); c += 256) {
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < O; ++k) {
            const bf16x8 fv = *reinterpret_cast<const bf16x8*>(fb + (size_t)k * Dp + c * 8);
            const float a = lg[k];
#pragma unroll
            for (int j = 0; j < 8; ++j) s8[j] = fmaf(a, (float)fv[j], s8[j]);
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)s8[j];
        *reinterpret_cast<bf16x8*>(img_enc + (size_t)b * Dp + c * 8) = o;
    }
}
__global__ __launch_bounds__(256) void butd_attend_bwd_bf16_kernel(const bf16_t* __restrict__ dimg, const bf16_t* __restrict__ imgf, const float* __restrict__ att,
                                                                   const bf16_t* __restrict__ ip, const bf16_t* __restrict__ qp, const float* __restrict__ wlin,
                                                                   bf16_t* __restrict__ dip, bf16_t* __restrict__ dqp, float* __restrict__ dw_part, float* __restrict__ db_part,
                                                                   int O, int H, int Dp, DropCfg drop) {
    __shared__ float dl[64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16_t* fb = imgf + (size_t)b * O * Dp;
    const int nch = Dp >> 3;
    for (int k = wave; k < O; k += 4) {
        float s = 0.f;
        for (int c = lane; c < nch; c += 64) {
            const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dimg + (size_t)b * Dp + c * 8), fv = *reinterpret_cast<const bf16x8*>(fb + (size_t)k * Dp + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s = fmaf((float)dv[j], (float)fv[j], s);
        }
        s = wave_sum(s);
        if (lane == 0) dl[k] = s;                 // d att_k
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const float a = lane < O ? att[(size_t)b * O + lane] : 0.f;
        const float da = lane < O ? dl[lane] : 0.f;
        const float dot = wave_sum(a * da);
        const float d = a * (da - dot);            // d logit_k
        const float dbs = wave_sum(d);
        if (lane < O) dl[lane] = d;
        if (lane == 0) db_part[b] = dbs;
    }
    __syncthreads();
    const bf16_t* ipb = ip + (size_t)b * O * H;
    bf16_t* dipb = dip + (size_t)b * O * H;
    for (int h0 = threadIdx.x * 4; h0 < H; h0 += 1024) {
        const bf16x4 qv = *reinterpret_cast<const bf16x4*>(qp + (size_t)b * H + h0);
        float w[4]; load4(wlin + h0, w);
        float q[4], dq[4] = {0.f, 0.f, 0.f, 0.f}, dw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = (float)qv[j];
        for (int k = 0; k < O; ++k) {
            const bf16x4 xv = *reinterpret_cast<const bf16x4*>(ipb + (size_t)k * H + h0);
            float keep[4] = {1.f, 1.f, 1.f, 1.f};
            drop_apply_vec<4>(drop, (uint32_t)((b * O + k) * H + h0), keep);
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = (float)xv[j];
                const float dj = dl[k] * w[j] * keep[j];      // d joint[k][h]
                dq[j] = fmaf(dj, x, dq[j]);
                dw[j] = fmaf(dl[k] * keep[j], x * q[j], dw[j]);
                o[j] = (bf16_t)(x > 0.f ? dj * q[j] : 0.f);       // through the ReLU of image_proj
            }
            *reinterpret_cast<bf16x4*>(dipb + (size_t)k * H + h0) = o;
        }
        bf16x4 oq;
#pragma unroll
        for (int j = 0; j < 4; ++j) oq[j] = (bf16_t)(q[j] > 0.f ? dq[j] : 0.f);      // through the ReLU of question_proj
        *reinterpret_cast<bf16x4*>(dqp + (size_t)b * H + h0) = oq;
        store4(dw_part + (size_t)b * H + h0, dw);
    }
}

// ---------------------------------------------------------------- joint = q_repr * img_repr (both post-ReLU)
template <typename T>
__global__ void mul_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_elem(out + (i), ld_elem(a + (i)) * ld_elem(b + (i)));
}
template <typename T>
__global__ void mul_relu_bwd_kernel(const T* __restrict__ dj, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ da, T* __restrict__ db, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = ld_elem(dj + (i)), x = ld_elem(a + (i)), y = ld_elem(b + (i));
        st_elem(da + (i), x > 0.f ? d * y : 0.f);
        st_elem(db + (i), y > 0.f ? d * x : 0.f);
    }
}

// ---------------------------------------------------------------- weight norm (scalar g): W = g * V / ||V||_F
// writes the effective weight [N, ldo] (zero padded past K) and, when wt != null, its transpose [K.., ldt]
template <typename T>
__global__ __launch_bounds__(256) void wn_eff_kernel(const float* __restrict__ v, const float* __restrict__ gptr, const float* __restrict__ sumsq,
                                                     T* __restrict__ w, int ldo, T* __restrict__ wt, int ldt, int N, int K) {
    __shared__ float tile[32][33];
    const float scale = gptr ? gptr[0] * rsqrtf(sumsq[0]) : 1.0f;
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + ty + r * 8, k = k0 + tx;
        const float x = (n < N && k < K) ? v[(size_t)n * K + k] * scale : 0.f;
        tile[ty + r * 8][tx] = x;
        if (n < N && k < ldo) st_elem(w + ((size_t)n * ldo + k), x);
    }
    if (wt == nullptr) return;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + ty + r * 8, n = n0 + tx;
        if (k < K && n < ldt) st_elem(wt + ((size_t)k * ldt + n), tile[tx][ty + r * 8]);
    }
}
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int N, int K, float* __restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)N * K; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i % K);
        acc = fmaf(a[(size_t)n * lda + k], b[i], acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// dV = s (dW - (dot / ||V||^2) V), dg = dot / ||V||, s = g / ||V||
__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ dw, int lddw, const float* __restrict__ v, const float* __restrict__ gptr,
                                                     const float* __restrict__ sumsq, const float* __restrict__ partial, int nblk, float* __restrict__ dv,
                                                     float* __restrict__ dg, int N, int K, int accumulate) {
    __shared__ float dots;
    if (threadIdx.x == 0) { float d = 0.f; for (int i = 0; i < nblk; ++i) d += partial[i]; dots = d; }
    __syncthreads();
    const float dot = dots, ss = sumsq[0], nrm = sqrtf(ss), s = gptr[0] / nrm;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)N * K; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i % K);
        const float g = s * (dw[(size_t)n * lddw + k] - dot / ss * v[i]);
        dv[i] = accumulate ? dv[i] + g : g;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) dg[0] = accumulate ? dg[0] + dot / nrm : dot / nrm;
}
// plain copy of a [N, K] block out of a padded gradient buffer (non-weight-normed weights)
__global__ void copy_block_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int N, int K, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)N * K; i += (size_t)gridDim.x * blockDim.x) {
        const float g = src[(size_t)(i / K) * lds + (i % K)];
        dst[i] = accumulate ? dst[i] + g : g;
    }
}

// ---------------------------------------------------------------- weight norm, every linear layer of the model in ONE launch per phase (round 5)
// Until round 5 each of the nine layers had its own sum-of-squares launch (+ the memset of its ticket), effective-weight launch, gradient dot
// product, gradient launch, bias column sum and bias copy: ~70 launches of 2-30 us per train step, 0.8 ms of a 4.5-ms step
// (profiles/r05_butd_kernel_stats_before.md).  Now: forward = wn_partials_group_kernel (the partial sums of squares of every weight-normed
// layer) + wn_eff_group_kernel (every layer's effective weight and its transpose; each block first folds its layer's partials in index order:
// deterministic, identical in every block); backward = the same pair for <dW, V> and dV / dg.
__device__ __forceinline__ int wn_find(const WnDesc* d, int n, int b, int WnDesc::*first) {
    int l = 0;
    for (int i = 1; i < n; ++i) if (b >= d[i].*first) l = i;
    return l;
}
__device__ __forceinline__ float4 wn_dw4(const WnDesc& d, size_t off) {
    float4 w = *reinterpret_cast<const float4*>(d.dw + off);
    for (int s = 1; s < d.nsplit; ++s) {
        const float4 x = *reinterpret_cast<const float4*>(d.dw + (size_t)s * d.split_stride + off);
        w.x += x.x; w.y += x.y; w.z += x.z; w.w += x.w;
    }
    return w;
}
__device__ __forceinline__ float wn_dw1(const WnDesc& d, size_t off) {
    float w = d.dw[off];
    for (int s = 1; s < d.nsplit; ++s) w += d.dw[(size_t)s * d.split_stride + off];
    return w;
}
// MODE 0: partial[b] = sum over the block's stripe of V^2;  MODE 1: of dW * V
template <int MODE>
__global__ __launch_bounds__(256) void wn_partials_group_kernel(const WnDesc* __restrict__ descs, int nd, float* __restrict__ partial) {
    __shared__ float red[4];
    const int l = wn_find(descs, nd, blockIdx.x, &WnDesc::blk0);
    const WnDesc d = descs[l];
    float acc = 0.f;
    if (d.g != nullptr) {
        const size_t n = (size_t)d.out * d.in;
        if ((d.in & 3) == 0 && (d.lddw & 3) == 0) {          // 16-byte accesses: a row holds a whole number of quads, so a quad never straddles rows
            const size_t nq = n >> 2;
            const int inq = d.in >> 2;
            for (size_t q = (size_t)(blockIdx.x - d.blk0) * 256 + threadIdx.x; q < nq; q += (size_t)d.nblk * 256) {
                const float4 v = reinterpret_cast<const float4*>(d.v)[q];
                if (MODE == 0) acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                else {
                    const float4 w = wn_dw4(d, (size_t)(q / inq) * d.lddw + (q % inq) * 4);
                    acc += w.x * v.x + w.y * v.y + w.z * v.z + w.w * v.w;
                }
            }
        } else
        for (size_t i = (size_t)(blockIdx.x - d.blk0) * 256 + threadIdx.x; i < n; i += (size_t)d.nblk * 256) {
            const float v = d.v[i];
            if (MODE == 0) acc = fmaf(v, v, acc);
            else acc = fmaf(wn_dw1(d, (size_t)(i / d.in) * d.lddw + (i % d.in)), v, acc);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float wn_fold(const float* partial, const WnDesc& d) {
    float s = 0.f;
    for (int i = 0; i < d.nblk; ++i) s += partial[d.blk0 + i];
    return s;
}
// effective weight [op, kp] (zero padded past `in`; rows past `out` are the caller's zeros) and its transpose [kp, op]; 32 x 32 tiles
template <typename T>
__global__ __launch_bounds__(256) void wn_eff_group_kernel(const WnDesc* __restrict__ descs, int nd, const float* __restrict__ partial) {
    __shared__ float tile[32][33];
    const int l = wn_find(descs, nd, blockIdx.x, &WnDesc::tile0);
    const WnDesc d = descs[l];
    float scale = 1.0f;
    if (d.g != nullptr) {
        const float ss = wn_fold(partial, d);
        scale = d.g[0] * rsqrtf(ss);
        if (blockIdx.x == d.tile0 && threadIdx.x == 0) d.sumsq[0] = ss;          // the backward pass needs ||V||^2 again
    }
    const int lt = blockIdx.x - d.tile0;
    const int n0 = (lt / d.tiles_x) * 32, k0 = (lt % d.tiles_x) * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    T* w = reinterpret_cast<T*>(d.eff);
    T* wt = reinterpret_cast<T*>(d.efft);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + ty + r * 8, k = k0 + tx;
        const float x = (n < d.out && k < d.in) ? d.v[(size_t)n * d.in + k] * scale : 0.f;
        tile[ty + r * 8][tx] = x;
        if (n < d.out && k < d.kp) st_elem(w + ((size_t)n * d.kp + k), x);
        if (d.eff_f32 != nullptr && n < d.out && k < d.in) d.eff_f32[(size_t)n * d.in + k] = x;
    }
    if (wt == nullptr) return;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + ty + r * 8, n = n0 + tx;
        if (k < d.in && n < d.op) st_elem(wt + ((size_t)k * d.op + n), tile[tx][ty + r * 8]);
    }
}
// dV = s (dW - (dot / ||V||^2) V), dg = dot / ||V||, s = g / ||V||; plain layers (g == null): dV = dW out of the padded scratch
__global__ __launch_bounds__(256) void wn_bwd_group_kernel(const WnDesc* __restrict__ descs, int nd, const float* __restrict__ partial) {
    const int l = wn_find(descs, nd, blockIdx.x, &WnDesc::blk0);
    const WnDesc d = descs[l];
    const size_t n = (size_t)d.out * d.in;
    if (d.bpart != nullptr && blockIdx.x == d.blk0)          // bias gradient: the partial column sums of the row slices, in slice order
        for (int m = threadIdx.x; m < d.out; m += 256) {
            float b = 0.f;
            for (int s = 0; s < d.nsplit; ++s) b += d.bpart[(size_t)s * d.out + m];
            d.db[m] = b;
        }
    if (d.g == nullptr) {
        if (d.dv == d.dw) return;                // the wgrad GEMM wrote straight into the gradient arena
        if ((d.in & 3) == 0 && (d.lddw & 3) == 0) {
            const size_t nq = n >> 2;
            const int inq = d.in >> 2;
            for (size_t q = (size_t)(blockIdx.x - d.blk0) * 256 + threadIdx.x; q < nq; q += (size_t)d.nblk * 256)
                reinterpret_cast<float4*>(d.dv)[q] = wn_dw4(d, (size_t)(q / inq) * d.lddw + (q % inq) * 4);
            return;
        }
        for (size_t i = (size_t)(blockIdx.x - d.blk0) * 256 + threadIdx.x; i < n; i += (size_t)d.nblk * 256)
            d.dv[i] = wn_dw1(d, (size_t)(i / d.in) * d.lddw + (i % d.in));
        return;
    }
    const float dot = wn_fold(partial, d), ss = d.sumsq[0], nrm = sqrtf(ss), s = d.g[0] / nrm, c = dot / ss;
    if ((d.in & 3) == 0 && (d.lddw & 3) == 0) {
        const size_t nq = n >> 2;
        const int inq = d.in >> 2;
        for (size_t q = (size_t)(blockIdx.x - d.blk0) * 256 + threadIdx.x; q < nq; q += (size_t)d.nblk * 256) {
            const float4 v = reinterpret_cast<const float4*>(d.v)[q];
            const float4 w = wn_dw4(d, (size_t)(q / inq) * d.lddw + (q % inq) * 4);
            reinterpret_cast<float4*>(d.dv)[q] = make_float4(s * (w.x - c * v.x), s * (w.y - c * v.y), s * (w.z - c * v.z), s * (w.w - c * v.w));
        }
    } else
    for (size_t i = (size_t)(blockIdx.x - d.blk0) * 256 + threadIdx.x; i < n; i += (size_t)d.nblk * 256)
        d.dv[i] = s * (wn_dw1(d, (size_t)(i / d.in) * d.lddw + (i % d.in)) - c * d.v[i]);
    if (blockIdx.x == d.blk0 && threadIdx.x == 0) d.dgp[0] = dot / nrm;
}

int kb_wn_forward_group(const WnDesc* descs_dev, int nd, int total_blocks, int total_tiles, float* partial, int elem /* 0 f32, 1 bf16, 2 split f32 */, hipStream_t s) {
    hipLaunchKernelGGL(wn_partials_group_kernel<0>, dim3(total_blocks), dim3(256), 0, s, descs_dev, nd, partial);
    RGQA_LAUNCH_CHECK("wn_partials_group_kernel");
    if (elem == 1) hipLaunchKernelGGL(wn_eff_group_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, s, descs_dev, nd, partial);
    else if (elem == 2) hipLaunchKernelGGL(wn_eff_group_kernel<sf32>, dim3(total_tiles), dim3(256), 0, s, descs_dev, nd, partial);
    else hipLaunchKernelGGL(wn_eff_group_kernel<float>, dim3(total_tiles), dim3(256), 0, s, descs_dev, nd, partial);
    RGQA_LAUNCH_CHECK("wn_eff_group_kernel");
    return RGQA_OK;
}
int kb_wn_backward_group(const WnDesc* descs_dev, int nd, int total_blocks, float* partial, hipStream_t s) {
    hipLaunchKernelGGL(wn_partials_group_kernel<1>, dim3(total_blocks), dim3(256), 0, s, descs_dev, nd, partial);
    RGQA_LAUNCH_CHECK("wn_partials_group_kernel");
    hipLaunchKernelGGL(wn_bwd_group_kernel, dim3(total_blocks), dim3(256), 0, s, descs_dev, nd, partial);
    RGQA_LAUNCH_CHECK("wn_bwd_group_kernel");
    return RGQA_OK;
}

// ---------------------------------------------------------------- host wrappers
#define GRID1(n) dim3((unsigned)(((n) + 255) / 256 > 4096 ? 4096 : ((n) + 255) / 256))

template <typename T> int kb_embed_fwd(const int64_t* toks, const float* table, T* out, int rows, int E, int Ep, hipStream_t s) {
    hipLaunchKernelGGL(butd_embed_fwd_kernel<T>, dim3(rows), dim3(128), 0, s, toks, table, out, E, Ep);
    RGQA_LAUNCH_CHECK("butd_embed_fwd_kernel"); return RGQA_OK;
}
template <typename T> int kb_embed_bwd(const int64_t* toks, const T* dx, float* dtable, int rows, int E, int Ep, int pad_idx, hipStream_t s) {
    hipLaunchKernelGGL(butd_embed_bwd_kernel<T>, dim3(rows), dim3(128), 0, s, toks, dx, dtable, E, Ep, pad_idx);
    RGQA_LAUNCH_CHECK("butd_embed_bwd_kernel"); return RGQA_OK;
}
template <typename T> int kb_gru_fwd(const T* gi, long ldgi, const T* gh, const T* hprev, T* hnew, T* rs, T* zs, T* ns, T* ghn, int B, int H, hipStream_t s) {
    hipLaunchKernelGGL(gru_gate_fwd_kernel<T>, dim3(cdiv((long)B * H, 256)), dim3(256), 0, s, gi, ldgi, gh, hprev, hnew, rs, zs, ns, ghn, B, H);
    RGQA_LAUNCH_CHECK("gru_gate_fwd_kernel"); return RGQA_OK;
}
template <typename T> int kb_gru_bwd(const T* dh, const T* hprev, const T* rs, const T* zs, const T* ns, const T* ghn, T* dgi, long lddgi, T* dgh, T* dhprev, int B, int H, hipStream_t s) {
    hipLaunchKernelGGL(gru_gate_bwd_kernel<T>, dim3(cdiv((long)B * H, 256)), dim3(256), 0, s, dh, hprev, rs, zs, ns, ghn, dgi, lddgi, dgh, dhprev, B, H);
    RGQA_LAUNCH_CHECK("gru_gate_bwd_kernel"); return RGQA_OK;
}
template <typename T> int kb_concat(const float* feat, const float* pos, T* out, int rows, int F, int Pd, int Dp, hipStream_t s) {
    hipLaunchKernelGGL(concat_cast_kernel<T>, dim3(rows), dim3(256), 0, s, feat, pos, out, F, Pd, Dp);
    RGQA_LAUNCH_CHECK("concat_cast_kernel"); return RGQA_OK;
}
template <typename T> int kb_attend_fwd(const T* ip, const T* qp, const float* wlin, const float* blin, const T* imgf, float* att, T* img_enc, int B, int O, int H, int Dp, DropCfg drop, hipStream_t s) {
    RGQA_REQUIRE(O <= 64, "butd attention: at most 64 regions (got %d)", O);
    if constexpr (std::is_same<T, bf16_t>::value) {
        if ((Dp & 7) == 0 && (H == 1024 || H == 512 || H == 2048)) {
            if (H == 1024) hipLaunchKernelGGL(butd_attend_fwd_bf16_kernel<2>, dim3(B), dim3(256), 0, s, ip, qp, wlin, blin, imgf, att, img_enc, O, Dp, drop);
            else if (H == 512) hipLaunchKernelGGL(butd_attend_fwd_bf16_kernel<1>, dim3(B), dim3(256), 0, s, ip, qp, wlin, blin, imgf, att, img_enc, O, Dp, drop);
            else hipLaunchKernelGGL(butd_attend_fwd_bf16_kernel<4>, dim3(B), dim3(256), 0, s, ip, qp, wlin, blin, imgf, att, img_enc, O, Dp, drop);
            RGQA_LAUNCH_CHECK("butd_attend_fwd_bf16_kernel"); return RGQA_OK;
        }
    }
    hipLaunchKernelGGL(butd_attend_fwd_kernel<T>, dim3(B), dim3(256), 0, s, ip, qp, wlin, blin, imgf, att, img_enc, O, H, Dp, drop);
    RGQA_LAUNCH_CHECK("butd_attend_fwd_kernel"); return RGQA_OK;
}
template <typename T> int kb_attend_bwd(const T* dimg, const T* imgf, const float* att, const T* ip, const T* qp, const float* wlin, T* dip, T* dqp, float* dw_part, float* db_part,
                                        int B, int O, int H, int Dp, DropCfg drop, hipStream_t s) {
    if constexpr (std::is_same<T, bf16_t>::value) {
        if ((Dp & 7) == 0 && (H & 3) == 0) {
            hipLaunchKernelGGL(butd_attend_bwd_bf16_kernel, dim3(B), dim3(256), 0, s, dimg, imgf, att, ip, qp, wlin, dip, dqp, dw_part, db_part, O, H, Dp, drop);
            RGQA_LAUNCH_CHECK("butd_attend_bwd_bf16_kernel"); return RGQA_OK;
        }
    }
    hipLaunchKernelGGL(butd_attend_bwd_kernel<T>, dim3(B), dim3(256), 0, s, dimg, imgf, att, ip, qp, wlin, dip, dqp, dw_part, db_part, O, H, Dp, drop);
    RGQA_LAUNCH_CHECK("butd_attend_bwd_kernel"); return RGQA_OK;
}
template <typename T> int kb_mul_fwd(const T* a, const T* b, T* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(mul_fwd_kernel<T>, GRID1(n), dim3(256), 0, s, a, b, out, n);
    RGQA_LAUNCH_CHECK("mul_fwd_kernel"); return RGQA_OK;
}
template <typename T> int kb_mul_relu_bwd(const T* dj, const T* a, const T* b, T* da, T* db, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(mul_relu_bwd_kernel<T>, GRID1(n), dim3(256), 0, s, dj, a, b, da, db, n);
    RGQA_LAUNCH_CHECK("mul_relu_bwd_kernel"); return RGQA_OK;
}
template <typename T> int kb_wn_eff(const float* v, const float* g, const float* sumsq, T* w, int ldo, T* wt, int ldt, int N, int K, hipStream_t s) {
    const int kk = ldo > K ? ldo : K;
    hipLaunchKernelGGL(wn_eff_kernel<T>, dim3(cdiv(kk, 32), cdiv(wt ? (ldt > N ? ldt : N) : N, 32)), dim3(256), 0, s, v, g, sumsq, w, ldo, wt, ldt, N, K);
    RGQA_LAUNCH_CHECK("wn_eff_kernel"); return RGQA_OK;
}
int kb_wn_bwd(const float* dw, int lddw, const float* v, const float* g, const float* sumsq, float* partial, float* dv, float* dg, int N, int K, int accumulate, hipStream_t s) {
    const int nblk = cdiv((long)N * K, 256 * 8) > 256 ? 256 : cdiv((long)N * K, 256 * 8);
    if (g == nullptr) {
        hipLaunchKernelGGL(copy_block_kernel, GRID1((size_t)N * K), dim3(256), 0, s, dw, lddw, dv, N, K, accumulate);
        RGQA_LAUNCH_CHECK("copy_block_kernel"); return RGQA_OK;
    }
    hipLaunchKernelGGL(dot_kernel, dim3(nblk), dim3(256), 0, s, dw, lddw, v, N, K, partial);
    RGQA_LAUNCH_CHECK("dot_kernel");
    hipLaunchKernelGGL(wn_bwd_kernel, GRID1((size_t)N * K), dim3(256), 0, s, dw, lddw, v, g, sumsq, partial, nblk, dv, dg, N, K, accumulate);
    RGQA_LAUNCH_CHECK("wn_bwd_kernel"); return RGQA_OK;
}

#define INST(T)                                                                                                                         \
    template int kb_embed_fwd<T>(const int64_t*, const float*, T*, int, int, int, hipStream_t);                                          \
    template int kb_embed_bwd<T>(const int64_t*, const T*, float*, int, int, int, int, hipStream_t);                                     \
    template int kb_gru_fwd<T>(const T*, long, const T*, const T*, T*, T*, T*, T*, T*, int, int, hipStream_t);                           \
    template int kb_gru_bwd<T>(const T*, const T*, const T*, const T*, const T*, const T*, T*, long, T*, T*, int, int, hipStream_t);     \
    template int kb_concat<T>(const float*, const float*, T*, int, int, int, int, hipStream_t);                                          \
    template int kb_attend_fwd<T>(const T*, const T*, const float*, const float*, const T*, float*, T*, int, int, int, int, DropCfg, hipStream_t); \
    template int kb_attend_bwd<T>(const T*, const T*, const float*, const T*, const T*, const float*, T*, T*, float*, float*, int, int, int, int, DropCfg, hipStream_t); \
    template int kb_mul_fwd<T>(const T*, const T*, T*, size_t, hipStream_t);                                                             \
    template int kb_mul_relu_bwd<T>(const T*, const T*, const T*, T*, T*, size_t, hipStream_t);                                          \
    template int kb_wn_eff<T>(const float*, const float*, const float*, T*, int, T*, int, int, int, hipStream_t);
INST(float)
INST(bf16_t)
INST(sf32)       // bf16x3 precision: element access through ld_elem / st_elem (rows are whole split-f32 lines: every ld a multiple of 32)
