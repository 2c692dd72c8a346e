// VisualFeatEncoder tail (reference lxrt/modeling.py:507-517): given zf = feats Wf^T + bf (from the GEMM),
//   out = dropout( (LN1(zf) + LN2(boxes Wb^T + bb)) / 2 )
// The K=4 box projection is computed in-register (no GEMM); one wave per RoI row; HBM-bound.
#include "kernels.h"

#define VISN_MAXPOS 8

// PDT = compile-time pos_dim (4 on the GQA path: the projection weights sit in registers as float4) or 0 = run-time pos_dim.
// Rows are grid-strided so that the per-lane constants (box weights, both LayerNorm affine pairs) are loaded once per wave: with one
// row per wave and a run-time K loop the kernel issued ~60 scalar-sized global loads per row and ran at 0.4 TB/s.
template <typename T, int NV, int PDT>
__global__ __launch_bounds__(256) void visn_fwd_kernel(const T* __restrict__ zf, int ldz, const float* __restrict__ boxes, const float* __restrict__ Wb,
                                                       const float* __restrict__ bb, const float* __restrict__ g1, const float* __restrict__ b1,
                                                       const float* __restrict__ g2, const float* __restrict__ b2, T* __restrict__ out, int ldo,
                                                       float* __restrict__ stats, int M, int H, int pd_rt, float eps, DropCfg drop) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = H >> 2;
    const int pd = PDT ? PDT : pd_rt;
    constexpr int PDM = PDT ? PDT : VISN_MAXPOS;
    float W[NV][4][PDM], Bb[NV][4], G1[NV][4], B1[NV][4], G2[NV][4], B2[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Bb[i][j] = G1[i][j] = B1[i][j] = G2[i][j] = B2[i][j] = 0.f;
#pragma unroll
            for (int k = 0; k < PDM; ++k) W[i][j][k] = 0.f;
        }
        if (c < nv) {
            load4(bb + c * 4, Bb[i]); load4(g1 + c * 4, G1[i]); load4(b1 + c * 4, B1[i]); load4(g2 + c * 4, G2[i]); load4(b2 + c * 4, B2[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = c * 4 + j;
                if (PDT == 4) load4(Wb + (size_t)n * 4, W[i][j]);
                else {
#pragma unroll
                    for (int k = 0; k < PDM; ++k) if (k < pd) W[i][j][k] = Wb[(size_t)n * pd + k];
                }
            }
        }
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        float bx[PDM];
#pragma unroll
        for (int c = 0; c < PDM; ++c) bx[c] = c < pd ? boxes[(size_t)row * pd + c] : 0.f;
        float x[NV][4], y[NV][4];
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                load4(zf + (size_t)row * ldz + c * 4, x[i]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = Bb[i][j];
#pragma unroll
                    for (int k = 0; k < PDM; ++k) a = fmaf(bx[k], W[i][j][k], a);
                    y[i][j] = a;
                    sx += x[i][j];
                    sy += a;
                }
            }
        }
        const float mx = wave_sum(sx) / (float)H, my = wave_sum(sy) / (float)H;
        float qx = 0.f, qy = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { float d = x[i][j] - mx; qx += d * d; float e = y[i][j] - my; qy += e * e; }
            }
        }
        const float rx = rsqrtf(wave_sum(qx) / (float)H + eps), ry = rsqrtf(wave_sum(qy) / (float)H + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = 0.5f * (((x[i][j] - mx) * rx * G1[i][j] + B1[i][j]) + ((y[i][j] - my) * ry * G2[i][j] + B2[i][j]));
                    o[j] = drop_apply(drop, (uint32_t)row * (uint32_t)H + (uint32_t)(c * 4 + j), v);
                }
                store4(out + (size_t)row * ldo + c * 4, o);
            }
        }
        if (lane == 0 && stats) {
            stats[(size_t)row * 4 + 0] = mx; stats[(size_t)row * 4 + 1] = rx;
            stats[(size_t)row * 4 + 2] = my; stats[(size_t)row * 4 + 3] = ry;
        }
    }
}

// partial layout: part[blk][q][n], q: 0 dg1, 1 db1, 2 dbias_fc, 3 dg2, 4 db2, 5 dbb, 6.. dWb[:,k]
template <typename T, int NV, int PDT>
__global__ __launch_bounds__(256) void visn_bwd_kernel(const T* __restrict__ dout, int lddo, const T* __restrict__ zf, int ldz, const float* __restrict__ boxes,
                                                       const float* __restrict__ Wb, const float* __restrict__ bb, const float* __restrict__ g1,
                                                       const float* __restrict__ g2, const float* __restrict__ stats, T* __restrict__ dzf, int lddz,
                                                       float* __restrict__ part, int M, int H, int pd_rt, DropCfg drop, float* __restrict__ dboxes) {
    __shared__ float red[4][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pd = PDT ? PDT : pd_rt;
    const int nv = H >> 2, nq = 6 + pd;
    float acc[6 + 4][NV][4];   // pos_dim <= 4 accumulated in registers
    float G1[NV][4], G2[NV][4];
    float W[NV][4][4], Bb[NV][4];   // box projection weights / bias of this lane's columns, loaded once
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Bb[i][j] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) W[i][j][k] = 0.f;
            if (c < nv) {
                const int n = c * 4 + j;
                Bb[i][j] = bb[n];
                if (PDT == 4) load4(Wb + (size_t)n * 4, W[i][j]);
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < pd) W[i][j][k] = Wb[(size_t)n * pd + k];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int q = 0; q < 10; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[q][i][j] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { G1[i][j] = 0.f; G2[i][j] = 0.f; }
        if (c < nv) { load4(g1 + c * 4, G1[i]); load4(g2 + c * 4, G2[i]); }
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float mx = stats[(size_t)row * 4], rx = stats[(size_t)row * 4 + 1], my = stats[(size_t)row * 4 + 2], ry = stats[(size_t)row * 4 + 3];
        float bx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) bx[k] = k < pd ? boxes[(size_t)row * pd + k] : 0.f;
        float d[NV][4], xh[NV][4], yh[NV][4];
        float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;
        float dbx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float dd[4], zz[4];
                load4(dout + (size_t)row * lddo + c * 4, dd);
                load4(zf + (size_t)row * ldz + c * 4, zz);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = c * 4 + j;
                    d[i][j] = 0.5f * drop_apply(drop, (uint32_t)row * (uint32_t)H + (uint32_t)n, dd[j]);
                    xh[i][j] = (zz[j] - mx) * rx;
                    float a = Bb[i][j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) a = fmaf(bx[k], W[i][j][k], a);
                    yh[i][j] = (a - my) * ry;
                    const float gx = d[i][j] * G1[i][j], gy = d[i][j] * G2[i][j];
                    s1 += gx; s2 += gx * xh[i][j]; t1 += gy; t2 += gy * yh[i][j];
                    acc[0][i][j] += d[i][j] * xh[i][j];
                    acc[1][i][j] += d[i][j];
                    acc[3][i][j] += d[i][j] * yh[i][j];
                    acc[4][i][j] += d[i][j];
                }
            }
        }
        s1 = wave_sum(s1) / (float)H; s2 = wave_sum(s2) / (float)H;
        t1 = wave_sum(t1) / (float)H; t2 = wave_sum(t2) / (float)H;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = rx * (d[i][j] * G1[i][j] - s1 - xh[i][j] * s2);
                    const float dzb = ry * (d[i][j] * G2[i][j] - t1 - yh[i][j] * t2);
                    acc[2][i][j] += o[j];
                    acc[5][i][j] += dzb;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[6 + k][i][j] += dzb * bx[k];
                    if (dboxes) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) dbx[k] = fmaf(dzb, W[i][j][k], dbx[k]);
                    }
                }
                store4(dzf + (size_t)row * lddz + c * 4, o);
            }
        }
        if (dboxes) {       // gradient w.r.t. the box coordinates (input gradients: ODIN, tasks/gqa_odin.py:97-121)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = wave_sum(dbx[k]);
                if (lane == 0 && k < pd) dboxes[(size_t)row * pd + k] = v;
            }
        }
    }
    for (int q = 0; q < nq; ++q) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = 0.f;
#pragma unroll
                    for (int qq = 0; qq < 10; ++qq) v = (qq == q) ? acc[qq][i][j] : v;
                    red[wave][c * 4 + j] = v;
                }
            }
        }
        __syncthreads();
        for (int n = threadIdx.x; n < H; n += 256)
            part[((size_t)blockIdx.x * nq + q) * H + n] = red[0][n] + red[1][n] + red[2][n] + red[3][n];
    }
}

template <typename T>
int k_visn_combine_fwd(const T* zf, int ldz, const float* boxes, const float* Wb, const float* bb, const float* g1, const float* b1, const float* g2,
                       const float* b2, T* out, int ldo, float* stats, int M, int H, int pos_dim, float eps, DropCfg drop, hipStream_t s) {
    RGQA_REQUIRE(H % 4 == 0 && H <= 1024 && pos_dim >= 1 && pos_dim <= 4, "visn_combine: H=%d pos_dim=%d unsupported", H, pos_dim);
    if (M <= 0) return RGQA_OK;
    const int nblk = cdiv(M, 4) > 1024 ? 1024 : cdiv(M, 4);      // 4 rows (waves) per block, grid-stride
#define VF(NVV) do { if (pos_dim == 4) hipLaunchKernelGGL((visn_fwd_kernel<T, NVV, 4>), dim3(nblk), dim3(256), 0, s, zf, ldz, boxes, Wb, bb, g1, b1, g2, b2, out, ldo, stats, M, H, pos_dim, eps, drop); \
                     else hipLaunchKernelGGL((visn_fwd_kernel<T, NVV, 0>), dim3(nblk), dim3(256), 0, s, zf, ldz, boxes, Wb, bb, g1, b1, g2, b2, out, ldo, stats, M, H, pos_dim, eps, drop); } while (0)
    const int nvl = cdiv(H / 4, 64);
    if (nvl <= 1) VF(1); else if (nvl == 2) VF(2); else if (nvl == 3) VF(3); else VF(4);
#undef VF
    RGQA_LAUNCH_CHECK("visn_fwd_kernel");
    return RGQA_OK;
}

template <typename T>
int k_visn_combine_bwd(const T* dout, int lddo, const T* zf, int ldz, const float* boxes, const float* Wb, const float* bb, const float* g1, const float* g2,
                       const float* stats, T* dzf, int lddz, float* part, float* dg1, float* db1, float* dg2, float* db2, float* dbias_fc,
                       float* dWb, float* dbb, int accumulate, int M, int H, int pos_dim, DropCfg drop, float* dboxes, hipStream_t s) {
    RGQA_REQUIRE(H % 4 == 0 && H <= 1024 && pos_dim >= 1 && pos_dim <= 4, "visn_combine bwd: H=%d pos_dim=%d unsupported", H, pos_dim);
    if (M <= 0) return RGQA_OK;
    const int nblk = cdiv(M, 4) > 512 ? 512 : cdiv(M, 4);   // 4 rows (waves) per block, grid-stride
#define VB(NVV) do { if (pos_dim == 4) hipLaunchKernelGGL((visn_bwd_kernel<T, NVV, 4>), dim3(nblk), dim3(256), 0, s, dout, lddo, zf, ldz, boxes, Wb, bb, g1, g2, stats, dzf, lddz, part, M, H, pos_dim, drop, dboxes); \
                     else hipLaunchKernelGGL((visn_bwd_kernel<T, NVV, 0>), dim3(nblk), dim3(256), 0, s, dout, lddo, zf, ldz, boxes, Wb, bb, g1, g2, stats, dzf, lddz, part, M, H, pos_dim, drop, dboxes); } while (0)
    const int nvl = cdiv(H / 4, 64);
    if (nvl <= 1) VB(1); else if (nvl == 2) VB(2); else if (nvl == 3) VB(3); else VB(4);
#undef VB
    RGQA_LAUNCH_CHECK("visn_bwd_kernel");
    FinOut fo = {};
    fo.p[0] = dg1; fo.p[1] = db1; fo.p[2] = dbias_fc; fo.p[3] = dg2; fo.p[4] = db2; fo.p[5] = dbb;
    for (int q = 0; q < 6; ++q) fo.stride[q] = 1;
    for (int k = 0; k < pos_dim; ++k) { fo.p[6 + k] = dWb + k; fo.stride[6 + k] = pos_dim; }
    return k_colsum_finalize(part, nblk, 6 + pos_dim, H, fo, accumulate, s);
}

template int k_visn_combine_fwd<float>(const float*, int, const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, int, float*, int, int, int, float, DropCfg, hipStream_t);
template int k_visn_combine_fwd<bf16_t>(const bf16_t*, int, const float*, const float*, const float*, const float*, const float*, const float*, const float*, bf16_t*, int, float*, int, int, int, float, DropCfg, hipStream_t);
template int k_visn_combine_fwd<sf32>(const sf32*, int, const float*, const float*, const float*, const float*, const float*, const float*, const float*, sf32*, int, float*, int, int, int, float, DropCfg, hipStream_t);
template int k_visn_combine_bwd<float>(const float*, int, const float*, int, const float*, const float*, const float*, const float*, const float*, const float*, float*, int, float*, float*, float*, float*, float*, float*, float*, float*, int, int, int, int, DropCfg, float*, hipStream_t);
template int k_visn_combine_bwd<bf16_t>(const bf16_t*, int, const bf16_t*, int, const float*, const float*, const float*, const float*, const float*, const float*, bf16_t*, int, float*, float*, float*, float*, float*, float*, float*, float*, int, int, int, int, DropCfg, float*, hipStream_t);
template int k_visn_combine_bwd<sf32>(const sf32*, int, const sf32*, int, const float*, const float*, const float*, const float*, const float*, const float*, sf32*, int, float*, float*, float*, float*, float*, float*, float*, float*, int, int, int, int, DropCfg, float*, hipStream_t);
