// MFMA attention core of the bf16x3 precision (split-f32 operands, common.h `sf32`), head size 64, Lq, Lk <= 64.
// Reference semantics: BertAttention.forward, lxrt/modeling.py:326-346 (f32 arithmetic).
//
// Same structure as attn_mfma.hip - one 64-lane wave per (sample, head), scores produced transposed so that the softmax is lane-local
// plus two shuffles, every gradient a chain of MFMAs fed from accumulators - with every product formed as hi*hi + hi*lo + lo*hi:
// the Q / K / V / dO fragments come as (hi, lo) pairs straight from the split layout (a head's 64 dims are two 128-byte lines
// [32 hi | 32 lo]), the probabilities / dS / dropout(P) are split in registers after the f32 softmax arithmetic.  The bf16 kernels
// keep their matrix pipe 3-5 % busy (they are instruction-issue bound), so tripling the MFMAs is cheap next to the doubled operand bytes.
#include "kernels.h"

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#define ROWB 144  // LDS row pitch in bytes for a [rows][64] bf16 image (128 + 16: spreads 8 consecutive rows over banks)
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

struct pair8 { bf16x8 h, l; };
// c += A * B on split operands: the two cross terms first, the leading term last
__device__ __forceinline__ f32x4 mfma3(const pair8& A, const pair8& B, f32x4 c) {
    c = MFMA(A.l, B.h, c);
    c = MFMA(A.h, B.l, c);
    return MFMA(A.h, B.h, c);
}
// Operand fragment of contraction step s (dims 32s .. 32s+31) of a head slice: `base` = the bf16 view of element (row 0, dim 0) of the
// head, ld2 = bf16 columns per row (2 x the element pitch).  Rows past the sample's length are read from its last valid row: whatever
// such a row holds only ever multiplies a probability that is exactly 0.
__device__ __forceinline__ pair8 ldfrag_x3(const bf16_t* base, int row, int nrows, size_t ld2, int s, int g) {
    const int r = row < nrows ? row : nrows - 1;
    const bf16_t* p = base + (size_t)r * ld2 + s * 64 + g * 8;
    pair8 f;
    f.h = *reinterpret_cast<const bf16x8*>(p);
    f.l = *reinterpret_cast<const bf16x8*>(p + 32);
    return f;
}
// transposed fragment: element jj = tile[ (jj<4 ? r0a : r0b) + 4*g + (jj&3) ][ c0 + (lane&15) ]
template <int PITCH = ROWB>
__device__ __forceinline__ bf16x8 trfrag(const unsigned char* tile, int r0a, int r0b, bool has_b, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const unsigned char* a1 = tile + (r0a + 4 * g + q) * PITCH + (c0 + 4 * p) * 2;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a1);
    bf16x4 hi = {0, 0, 0, 0};
    if (has_b) {
        const unsigned char* a2 = tile + (r0b + 4 * g + q) * PITCH + (c0 + 4 * p) * 2;
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a2);
    }
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
template <int PITCH = ROWB>
__device__ __forceinline__ pair8 trfrag2(const unsigned char* th, const unsigned char* tl, int r0a, int r0b, bool has_b, int c0, int lane) {
    pair8 f;
    f.h = trfrag<PITCH>(th, r0a, r0b, has_b, c0, lane);
    f.l = trfrag<PITCH>(tl, r0a, r0b, has_b, c0, lane);
    return f;
}
// 8 f32 values (two accumulator quads) -> (hi, lo) bf16 fragments
__device__ __forceinline__ pair8 split8(const f32x4& a, const f32x4& b) {
    pair8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16_t h, l;
        sf_split(a[i], h, l); r.h[i] = h; r.l[i] = l;
        sf_split(b[i], h, l); r.h[4 + i] = h; r.l[4 + i] = l;
    }
    return r;
}
__device__ __forceinline__ const bf16_t* head_base(const void* p, size_t row0, int ld, int h) {
    return reinterpret_cast<const bf16_t*>(reinterpret_cast<const sf32*>(p) + row0 * ld + h * 64);
}

// ============================================================================ forward
// (body + one-problem and two-problem kernels, as in attn_mfma.hip)
template <int NQT, int NKT>
__device__ __forceinline__ void attn_fwd_x3_body(const AttnArgs& a, const int blk, unsigned char* smem /* 2 * NKT * 16 * ROWB bytes */) {
    unsigned char* vsh = smem;
    unsigned char* vsl = smem + NKT * 16 * ROWB;
    const int lane = threadIdx.x, fr = lane & 15, g = lane >> 4;
    const int b = blk / a.nh, h = blk % a.nh;
    ATTN_SAMPLE_ROWS(a, b)
    const bf16_t* Q = head_base(a.q, q0, a.ldq, h);
    const bf16_t* K = head_base(a.k, k0, a.ldk, h);
    const bf16_t* V = head_base(a.v, k0, a.ldv, h);
    const size_t ldq2 = 2 * (size_t)a.ldq, ldk2 = 2 * (size_t)a.ldk, ldv2 = 2 * (size_t)a.ldv;
    pair8 qf[NQT][2], kf[NKT][2], vf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) { kf[kt][s] = ldfrag_x3(K, kt * 16 + fr, Lk, ldk2, s, g); vf[kt][s] = ldfrag_x3(V, kt * 16 + fr, Lk, ldv2, s, g); }
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[qt][s] = ldfrag_x3(Q, qt * 16 + fr, Lq, ldq2, s, g);
    f32x4 acc[NKT][NQT];
    float mk[NKT][4];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            mk[kt][r] = key < Lk ? (a.mask ? a.mask[(size_t)b * a.Lk + key] : 0.f) : -INFINITY;
        }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            *reinterpret_cast<bf16x8*>(vsh + (kt * 16 + fr) * ROWB + s * 64 + g * 16) = vf[kt][s].h;
            *reinterpret_cast<bf16x8*>(vsl + (kt * 16 + fr) * ROWB + s * 64 + g * 16) = vf[kt][s].l;
        }
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            c = mfma3(kf[kt][0], qf[qt][0], c);
            c = mfma3(kf[kt][1], qf[qt][1], c);
            acc[kt][qt] = c;
        }
    DropCfg dc = a.drop; dc.seed_hi ^= a.drop_site;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const int q = qt * 16 + fr;
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { float s = acc[kt][qt][r] * a.scale + mk[kt][r]; acc[kt][qt][r] = s; m = fmaxf(m, s); }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { float p = __expf(acc[kt][qt][r] - m); acc[kt][qt][r] = p; sum += p; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        if (g == 0 && q < Lq && a.lse) a.lse[((size_t)b * a.nh + h) * a.Lq + q] = m + __logf(sum);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const uint32_t idx = (uint32_t)(((b * a.nh + h) * a.Lq + q) * a.Lk + kt * 16 + 4 * g);
            float v[4] = {acc[kt][qt][0] * inv, acc[kt][qt][1] * inv, acc[kt][qt][2] * inv, acc[kt][qt][3] * inv};
            if ((a.Lk & 1) == 0) drop_apply_vec<4>(dc, idx, v);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = drop_apply(dc, idx + (uint32_t)r, v[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[kt][qt][r] = v[r];
        }
    }
    // ctx^T[d][q] = sum_key V^T[d][key] P^T[key][q]
    sf32* O = reinterpret_cast<sf32*>(a.out) + q0 * a.ldo + h * 64;
    bf16_t* Ob = a.out_b ? reinterpret_cast<bf16_t*>(a.out_b) + q0 * a.ldo + h * 64 : nullptr;      // bf16x3_fwd precision: the image the bf16 backward reads
    constexpr int NKS = (NKT + 1) / 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    pair8 pb[NKS][NQT];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
            pb[ks][qt] = split8(acc[2 * ks][qt], (2 * ks + 1 < NKT) ? acc[(2 * ks + 1 < NKT) ? 2 * ks + 1 : 0][qt] : zero4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        pair8 va[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) va[ks] = trfrag2(vsh, vsl, 2 * ks * 16, (2 * ks + 1) * 16, 2 * ks + 1 < NKT, dt * 16, lane);
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) o = mfma3(va[ks], pb[ks][qt], o);
            const int q = qt * 16 + fr;
            if (q < Lq) {
                float v[4] = {o[0], o[1], o[2], o[3]};
                store4(O + (size_t)q * a.ldo + dt * 16 + 4 * g, v);
                if (Ob) store4(Ob + (size_t)q * a.ldo + dt * 16 + 4 * g, v);
            }
        }
    }
}

// ============================================================================ backward
// Query-major pass: P recomputed from the saved log-sum-exp, dP = dO V^T, dS = P (dP - delta) scale, dQ = dS K; dS and dropout(P) are
// parked (hi and lo images) for the key-major pass, which fetches them transposed: dK = dS^T Q, dV = dropout(P)^T dO.
template <int NQT, int NKT>
__host__ __device__ constexpr int attn_bwd_x3_smem() {
    return 2 * ((NKT * 16 * ROWB > NQT * 16 * (NKT * 32 + 16) ? NKT * 16 * ROWB : NQT * 16 * (NKT * 32 + 16)) + 2 * NQT * 16 * ROWB + NQT * 16 * (NKT * 32 + 16));
}
template <int NQT, int NKT>
__device__ __forceinline__ void attn_bwd_x3_body(const AttnArgs& a, const int blk, unsigned char* smem /* attn_bwd_x3_smem<NQT, NKT>() bytes */) {
    constexpr int P2 = NKT * 32 + 16;                  // pitch of the [query][key] images
    constexpr int KS_BYTES = NKT * 16 * ROWB > NQT * 16 * P2 ? NKT * 16 * ROWB : NQT * 16 * P2;
    unsigned char* ksh = smem; unsigned char* ksl = ksh + KS_BYTES;               // K images; after the query-major pass: dS[q][key]
    unsigned char* qsh = ksl + KS_BYTES; unsigned char* qsl = qsh + NQT * 16 * ROWB;
    unsigned char* osh = qsl + NQT * 16 * ROWB; unsigned char* osl = osh + NQT * 16 * ROWB;
    unsigned char* pdh = osl + NQT * 16 * ROWB; unsigned char* pdl = pdh + NQT * 16 * P2;     // Pd[q][key] = dropout(P)
    const int lane = threadIdx.x, fr = lane & 15, g = lane >> 4;
    const int b = blk / a.nh, h = blk % a.nh;
    ATTN_SAMPLE_ROWS(a, b)
    const bf16_t* Q = head_base(a.q, q0, a.ldq, h);
    const bf16_t* K = head_base(a.k, k0, a.ldk, h);
    const bf16_t* V = head_base(a.v, k0, a.ldv, h);
    const bf16_t* dO = head_base(a.dout, q0, a.lddo, h);
    const size_t ldq2 = 2 * (size_t)a.ldq, ldk2 = 2 * (size_t)a.ldk, ldv2 = 2 * (size_t)a.ldv, ldo2 = 2 * (size_t)a.lddo;
    const float* lse = a.lse + ((size_t)b * a.nh + h) * a.Lq;
    pair8 kf[NKT][2], vf[NKT][2], qf[NQT][2], of[NQT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) { kf[kt][s] = ldfrag_x3(K, kt * 16 + fr, Lk, ldk2, s, g); vf[kt][s] = ldfrag_x3(V, kt * 16 + fr, Lk, ldv2, s, g); }
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int s = 0; s < 2; ++s) { qf[qt][s] = ldfrag_x3(Q, qt * 16 + fr, Lq, ldq2, s, g); of[qt][s] = ldfrag_x3(dO, qt * 16 + fr, Lq, ldo2, s, g); }
    float lse_t[NQT], mk[NKT][4];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) lse_t[qt] = qt * 16 + fr < Lq ? lse[qt * 16 + fr] : 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            mk[kt][r] = key < Lk ? (a.mask ? a.mask[(size_t)b * a.Lk + key] : 0.f) : -INFINITY;
        }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            *reinterpret_cast<bf16x8*>(ksh + (kt * 16 + fr) * ROWB + s * 64 + g * 16) = kf[kt][s].h;
            *reinterpret_cast<bf16x8*>(ksl + (kt * 16 + fr) * ROWB + s * 64 + g * 16) = kf[kt][s].l;
        }
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            *reinterpret_cast<bf16x8*>(qsh + (qt * 16 + fr) * ROWB + s * 64 + g * 16) = qf[qt][s].h;
            *reinterpret_cast<bf16x8*>(qsl + (qt * 16 + fr) * ROWB + s * 64 + g * 16) = qf[qt][s].l;
            *reinterpret_cast<bf16x8*>(osh + (qt * 16 + fr) * ROWB + s * 64 + g * 16) = of[qt][s].h;
            *reinterpret_cast<bf16x8*>(osl + (qt * 16 + fr) * ROWB + s * 64 + g * 16) = of[qt][s].l;
        }
    __syncthreads();
    DropCfg dc = a.drop; dc.seed_hi ^= a.drop_site;
    const uint32_t idx0 = (uint32_t)((b * a.nh + h) * a.Lq) * (uint32_t)a.Lk;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int NKS = (NKT + 1) / 2, NQS = (NQT + 1) / 2;

    // ---------------- pass T: lane = query, registers = keys  ->  dQ; Pd -> pd images, dS held split in registers
    bf16x4 dsh[NQT][NKT], dsl[NQT][NKT];
    {
        sf32* dQ = reinterpret_cast<sf32*>(a.dq) + q0 * a.lddq + h * 64;
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            const int q = qt * 16 + fr;
            const float lq = lse_t[qt];
            f32x4 pp[NKT], dpp[NKT];
            float delta = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp4 = {0.f, 0.f, 0.f, 0.f};
                s4 = mfma3(kf[kt][0], qf[qt][0], s4); s4 = mfma3(kf[kt][1], qf[qt][1], s4);
                dp4 = mfma3(vf[kt][0], of[qt][0], dp4); dp4 = mfma3(vf[kt][1], of[qt][1], dp4);
                float keep4[4] = {1.f, 1.f, 1.f, 1.f};
                {
                    const uint32_t idx = idx0 + (uint32_t)(q * a.Lk + kt * 16 + 4 * g);
                    if ((a.Lk & 1) == 0) drop_apply_vec<4>(dc, idx, keep4);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) keep4[r] = drop_apply(dc, idx + (uint32_t)r, 1.0f);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = q < Lq ? __expf(s4[r] * a.scale + mk[kt][r] - lq) : 0.f;
                    const float keep = keep4[r];
                    const float dp = dp4[r] * keep;
                    delta += p * dp;
                    s4[r] = p; dp4[r] = dp;
                    keep4[r] = p * keep;               // Pd
                }
                pp[kt] = s4; dpp[kt] = dp4;
                {
                    bf16x4 ph, pl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { bf16_t x, y; sf_split(keep4[r], x, y); ph[r] = x; pl[r] = y; }
                    *reinterpret_cast<bf16x4*>(pdh + q * P2 + 8 * g + kt * 32) = ph;      // keys kt*16 + 4g .. +3 of query row q
                    *reinterpret_cast<bf16x4*>(pdl + q * P2 + 8 * g + kt * 32) = pl;
                }
            }
            delta += __shfl_xor(delta, 16, 64);
            delta += __shfl_xor(delta, 32, 64);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pp[kt][r] = pp[kt][r] * (dpp[kt][r] - delta) * a.scale;   // dS^T
                    bf16_t x, y; sf_split(pp[kt][r], x, y); dsh[qt][kt][r] = x; dsl[qt][kt][r] = y;
                }
            }
            // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
            pair8 db[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) db[ks] = split8(pp[2 * ks], (2 * ks + 1 < NKT) ? pp[(2 * ks + 1 < NKT) ? 2 * ks + 1 : 0] : zero4);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const pair8 ka = trfrag2(ksh, ksl, 2 * ks * 16, (2 * ks + 1) * 16, 2 * ks + 1 < NKT, dt * 16, lane);
                    o = mfma3(ka, db[ks], o);
                }
                if (q < Lq) {
                    float v[4] = {o[0], o[1], o[2], o[3]};
                    store4(dQ + (size_t)q * a.lddq + dt * 16 + 4 * g, v);
                }
            }
        }
    }
    __syncthreads();   // every dQ product has read the K images
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            *reinterpret_cast<bf16x4*>(ksh + (qt * 16 + fr) * P2 + kt * 32 + 8 * g) = dsh[qt][kt];
            *reinterpret_cast<bf16x4*>(ksl + (qt * 16 + fr) * P2 + kt * 32 + 8 * g) = dsl[qt][kt];
        }
    __syncthreads();
    // ---------------- pass N: lane = key  ->  dK, dV from the parked dS / Pd
    {
        sf32* dK = reinterpret_cast<sf32*>(a.dk) + k0 * a.lddk + h * 64;
        sf32* dV = reinterpret_cast<sf32*>(a.dv) + k0 * a.lddv + h * 64;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int key = kt * 16 + fr;
            pair8 dsb[NQS], pdb[NQS];
#pragma unroll
            for (int qs = 0; qs < NQS; ++qs) {
                const bool two = 2 * qs + 1 < NQT;
                dsb[qs] = trfrag2<P2>(ksh, ksl, 2 * qs * 16, (2 * qs + 1) * 16, two, kt * 16, lane);     // element jj = dS[q(jj)][key]
                pdb[qs] = trfrag2<P2>(pdh, pdl, 2 * qs * 16, (2 * qs + 1) * 16, two, kt * 16, lane);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 ok = {0.f, 0.f, 0.f, 0.f}, ov = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int qs = 0; qs < NQS; ++qs) {
                    const bool two = 2 * qs + 1 < NQT;
                    const pair8 qa = trfrag2(qsh, qsl, 2 * qs * 16, (2 * qs + 1) * 16, two, dt * 16, lane);
                    const pair8 oa = trfrag2(osh, osl, 2 * qs * 16, (2 * qs + 1) * 16, two, dt * 16, lane);
                    ok = mfma3(qa, dsb[qs], ok);   // dK^T[d][key] = sum_q Q^T[d][q] dS[q][key]
                    ov = mfma3(oa, pdb[qs], ov);   // dV^T[d][key] = sum_q dO^T[d][q] Pd[q][key]
                }
                if (key < Lk) {
                    float v1[4] = {ok[0], ok[1], ok[2], ok[3]}, v2[4] = {ov[0], ov[1], ov[2], ov[3]};
                    store4(dK + (size_t)key * a.lddk + dt * 16 + 4 * g, v1);
                    store4(dV + (size_t)key * a.lddv + dt * 16 + 4 * g, v2);
                }
            }
        }
    }
}

// ============================================================================ host side
template <int NQT, int NKT>
__global__ __launch_bounds__(64) void attn_fwd_x3_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NKT * 16 * ROWB];
    attn_fwd_x3_body<NQT, NKT>(a, blockIdx.x, smem);
}
template <int NQT, int NKT>
__global__ __launch_bounds__(64) void attn_bwd_x3_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[attn_bwd_x3_smem<NQT, NKT>()];
    attn_bwd_x3_body<NQT, NKT>(a, blockIdx.x, smem);
}
template <int Q0, int K0, int Q1, int K1>
__global__ __launch_bounds__(64) void attn_fwd_x3_pair_kernel(const AttnArgs a0, const AttnArgs a1, const int n0) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (K0 > K1 ? K0 : K1) * 16 * ROWB];
    if ((int)blockIdx.x < n0) attn_fwd_x3_body<Q0, K0>(a0, blockIdx.x, smem);
    else attn_fwd_x3_body<Q1, K1>(a1, blockIdx.x - n0, smem);
}
template <int Q0, int K0, int Q1, int K1>
__global__ __launch_bounds__(64) void attn_bwd_x3_pair_kernel(const AttnArgs a0, const AttnArgs a1, const int n0) {
    constexpr int S0 = attn_bwd_x3_smem<Q0, K0>(), S1 = attn_bwd_x3_smem<Q1, K1>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[S0 > S1 ? S0 : S1];
    if ((int)blockIdx.x < n0) attn_bwd_x3_body<Q0, K0>(a0, blockIdx.x, smem);
    else attn_bwd_x3_body<Q1, K1>(a1, blockIdx.x - n0, smem);
}

static int x3_check(const AttnArgs& a, bool bwd) {
    RGQA_REQUIRE(a.dh == 64, "x3 attention: head size must be 64 (got %d)", a.dh);
    RGQA_REQUIRE(a.B > 0 && a.nh > 0 && a.Lq > 0 && a.Lk > 0 && a.Lq <= 64 && a.Lk <= 64, "x3 attention: Lq/Lk must be in 1..64 (got %d %d)", a.Lq, a.Lk);
    RGQA_REQUIRE(a.q && a.k && a.v && (a.ldq % 32) == 0 && (a.ldk % 32) == 0 && (a.ldv % 32) == 0, "x3 attention: null operand or row stride not a multiple of 32");
    RGQA_REQUIRE(((uintptr_t)a.q % 128) == 0 && ((uintptr_t)a.k % 128) == 0 && ((uintptr_t)a.v % 128) == 0, "x3 attention: operands must start on a 128-byte line");
    if (bwd) {
        RGQA_REQUIRE(a.dout && a.dq && a.dk && a.dv && a.lse, "x3 attention bwd: null operand");
        RGQA_REQUIRE((a.lddo % 32) == 0 && (a.lddq % 32) == 0 && (a.lddk % 32) == 0 && (a.lddv % 32) == 0, "x3 attention bwd: bad row strides");
        RGQA_REQUIRE(((uintptr_t)a.dout % 128) == 0 && ((uintptr_t)a.dq % 128) == 0 && ((uintptr_t)a.dk % 128) == 0 && ((uintptr_t)a.dv % 128) == 0, "x3 attention bwd: operands must start on a 128-byte line");
    } else {
        RGQA_REQUIRE(a.out && (a.ldo % 32) == 0 && ((uintptr_t)a.out % 128) == 0, "x3 attention: null output / bad stride");
    }
    return RGQA_OK;
}

#define DISPATCH_TILES(KERNEL, nqt, nkt)                                                                          \
    switch ((nqt) * 8 + (nkt)) {                                                                                  \
        case 1 * 8 + 1: hipLaunchKernelGGL((KERNEL<1, 1>), grid, dim3(64), 0, s, a); break;                        \
        case 1 * 8 + 2: hipLaunchKernelGGL((KERNEL<1, 2>), grid, dim3(64), 0, s, a); break;                        \
        case 1 * 8 + 3: hipLaunchKernelGGL((KERNEL<1, 3>), grid, dim3(64), 0, s, a); break;                        \
        case 1 * 8 + 4: hipLaunchKernelGGL((KERNEL<1, 4>), grid, dim3(64), 0, s, a); break;                        \
        case 2 * 8 + 1: hipLaunchKernelGGL((KERNEL<2, 1>), grid, dim3(64), 0, s, a); break;                        \
        case 2 * 8 + 2: hipLaunchKernelGGL((KERNEL<2, 2>), grid, dim3(64), 0, s, a); break;                        \
        case 2 * 8 + 3: hipLaunchKernelGGL((KERNEL<2, 3>), grid, dim3(64), 0, s, a); break;                        \
        case 2 * 8 + 4: hipLaunchKernelGGL((KERNEL<2, 4>), grid, dim3(64), 0, s, a); break;                        \
        case 3 * 8 + 1: hipLaunchKernelGGL((KERNEL<3, 1>), grid, dim3(64), 0, s, a); break;                        \
        case 3 * 8 + 2: hipLaunchKernelGGL((KERNEL<3, 2>), grid, dim3(64), 0, s, a); break;                        \
        case 3 * 8 + 3: hipLaunchKernelGGL((KERNEL<3, 3>), grid, dim3(64), 0, s, a); break;                        \
        case 3 * 8 + 4: hipLaunchKernelGGL((KERNEL<3, 4>), grid, dim3(64), 0, s, a); break;                        \
        case 4 * 8 + 1: hipLaunchKernelGGL((KERNEL<4, 1>), grid, dim3(64), 0, s, a); break;                        \
        case 4 * 8 + 2: hipLaunchKernelGGL((KERNEL<4, 2>), grid, dim3(64), 0, s, a); break;                        \
        case 4 * 8 + 3: hipLaunchKernelGGL((KERNEL<4, 3>), grid, dim3(64), 0, s, a); break;                        \
        default: hipLaunchKernelGGL((KERNEL<4, 4>), grid, dim3(64), 0, s, a); break;                               \
    }

int k_attn_fwd_x3(const AttnArgs& a, hipStream_t s) {
    int r = x3_check(a, false);
    if (r) return r;
    const int nqt = cdiv(a.Lq, 16), nkt = cdiv(a.Lk, 16);
    dim3 grid(a.B * a.nh);
    DISPATCH_TILES(attn_fwd_x3_kernel, nqt, nkt)
    RGQA_LAUNCH_CHECK("attn_fwd_x3_kernel");
    return RGQA_OK;
}

int k_attn_bwd_x3(const AttnArgs& a, hipStream_t s) {
    int r = x3_check(a, true);
    if (r) return r;
    const int nqt = cdiv(a.Lq, 16), nkt = cdiv(a.Lk, 16);
    dim3 grid(a.B * a.nh);
    DISPATCH_TILES(attn_bwd_x3_kernel, nqt, nkt)
    RGQA_LAUNCH_CHECK("attn_bwd_x3_kernel");
    return RGQA_OK;
}

// two problems of a stage in one launch (see attn_mfma.hip): 1 = launched, 0 = launch them separately, < 0 = error
static int pair_shape(const AttnArgs& a0, const AttnArgs& a1) {
    const int q0 = cdiv(a0.Lq, 16), k0 = cdiv(a0.Lk, 16), q1 = cdiv(a1.Lq, 16), k1 = cdiv(a1.Lk, 16);
    if (a0.nh != a1.nh || a0.dh != a1.dh) return 0;
    if (q0 == 3 && k0 == 3 && q1 == 2 && k1 == 2) return 1;
    if (q0 == 3 && k0 == 2 && q1 == 2 && k1 == 3) return 2;
    return 0;
}
int k_attn_fwd_x3_pair(const AttnArgs& x, const AttnArgs& y, hipStream_t s) {
    const bool swap = x.Lq * x.Lk < y.Lq * y.Lk || (x.Lq * x.Lk == y.Lq * y.Lk && x.Lq < y.Lq);
    const AttnArgs& a0 = swap ? y : x; const AttnArgs& a1 = swap ? x : y;
    const int shape = pair_shape(a0, a1);
    if (shape == 0) return 0;
    int r = x3_check(a0, false); if (r) return r;
    r = x3_check(a1, false); if (r) return r;
    const int n0 = a0.B * a0.nh;
    dim3 grid(n0 + a1.B * a1.nh);
    if (shape == 1) hipLaunchKernelGGL((attn_fwd_x3_pair_kernel<3, 3, 2, 2>), grid, dim3(64), 0, s, a0, a1, n0);
    else hipLaunchKernelGGL((attn_fwd_x3_pair_kernel<3, 2, 2, 3>), grid, dim3(64), 0, s, a0, a1, n0);
    RGQA_LAUNCH_CHECK("attn_fwd_x3_pair_kernel");
    return 1;
}
int k_attn_bwd_x3_pair(const AttnArgs& x, const AttnArgs& y, hipStream_t s) {
    const bool swap = x.Lq * x.Lk < y.Lq * y.Lk || (x.Lq * x.Lk == y.Lq * y.Lk && x.Lq < y.Lq);
    const AttnArgs& a0 = swap ? y : x; const AttnArgs& a1 = swap ? x : y;
    const int shape = pair_shape(a0, a1);
    if (shape == 0) return 0;
    int r = x3_check(a0, true); if (r) return r;
    r = x3_check(a1, true); if (r) return r;
    const int n0 = a0.B * a0.nh;
    dim3 grid(n0 + a1.B * a1.nh);
    if (shape == 1) hipLaunchKernelGGL((attn_bwd_x3_pair_kernel<3, 3, 2, 2>), grid, dim3(64), 0, s, a0, a1, n0);
    else hipLaunchKernelGGL((attn_bwd_x3_pair_kernel<3, 2, 2, 3>), grid, dim3(64), 0, s, a0, a1, n0);
    RGQA_LAUNCH_CHECK("attn_bwd_x3_pair_kernel");
    return 1;
}
