// MFMA attention core for gfx950 (bf16 operands, f32 softmax), head size 64, Lq, Lk <= 64.
// Reference semantics: BertAttention.forward, lxrt/modeling.py:326-346.
//
// One 64-lane wave (= one workgroup) per (sample, head).  Scores are produced TRANSPOSED,
//   S^T[key][query] = K Q^T   via v_mfma_f32_16x16x32_bf16(a = K fragment, b = Q fragment)
// so that each lane owns ONE query column and 4 keys per 16-key tile: the softmax reduction is lane-local plus
// two xor-shuffles (lane groups 16/32), and the probability tile is, without any data movement, the B operand of
//   ctx^T[d][query] = V^T P^T
// whose A operand (V transposed) comes from a row-major LDS image of V through ds_read_b64_tr_b16.
// K and Q fragments are loaded straight from HBM/L2 (16 B per lane, rows are 128-B head slices).
// Nothing of size [Lq,Lk] ever leaves registers.
//
// Backward recomputes P from the saved log-sum-exp, in both orientations (lane = query for dQ; lane = key for
// dK / dV), so every gradient is again a chain of MFMAs fed from accumulators: 5 products as in flash-style
// backward, no atomics, each output element written exactly once.
#include "kernels.h"

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#define ROWB 144  // LDS row pitch in bytes for a [rows][64] bf16 image (128 + 16: spreads 8 consecutive rows over banks)

// Rows past the sample's length are read from its LAST valid row instead of being zero-filled under a branch (4 v_mov + an exec-mask
// branch per fragment: 8-9 % of these kernels' instructions): whatever such a row holds is finite and only ever multiplies a probability
// that is exactly 0 (masked key: exp(-inf); query row past Lq: forced to 0 in the backward kernels, never stored in the forward one).
__device__ __forceinline__ bf16x8 ldfrag(const bf16_t* base, int row, int nrows, int ld, int s, int g) {
    const int r = row < nrows ? row : nrows - 1;
    return *reinterpret_cast<const bf16x8*>(base + (size_t)r * ld + s * 32 + g * 8);
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
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 r;
    r[0] = (bf16_t)a[0]; r[1] = (bf16_t)a[1]; r[2] = (bf16_t)a[2]; r[3] = (bf16_t)a[3];
    r[4] = (bf16_t)b[0]; r[5] = (bf16_t)b[1]; r[6] = (bf16_t)b[2]; r[7] = (bf16_t)b[3];
    return r;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// ============================================================================ forward
// (body + two kernels: one problem per launch, or TWO problems - the language and the vision side of a paired stage, or the two directions of
// a cross-attention stage - in one launch, the heavier one's blocks first: round 3)
template <int NQT, int NKT>
__device__ __forceinline__ void attn_fwd_mfma_body(const AttnArgs& a, const int blk, unsigned char* vs /* NKT * 16 * ROWB bytes */) {
    const int lane = threadIdx.x, fr = lane & 15, g = lane >> 4;
    const int b = blk / a.nh, h = blk % a.nh;
    ATTN_SAMPLE_ROWS(a, b)      // q0, k0: first row of this sample; Lq, Lk: its valid row counts (<= a.Lq, a.Lk)
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + q0 * a.ldq + h * 64;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + k0 * a.ldk + h * 64;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + k0 * a.ldv + h * 64;
    // one global-load phase: Q, K, V fragments and the mask requested back to back; the V image for the transposed reads is
    // written from the V registers
    bf16x8 qf[NQT][2], kf[NKT][2], vf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) { kf[kt][s] = ldfrag(K, kt * 16 + fr, Lk, a.ldk, s, g); vf[kt][s] = ldfrag(V, kt * 16 + fr, Lk, a.ldv, s, g); }
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[qt][s] = ldfrag(Q, qt * 16 + fr, Lq, a.ldq, s, g);
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
        for (int s = 0; s < 2; ++s) *reinterpret_cast<bf16x8*>(vs + (kt * 16 + fr) * ROWB + s * 64 + g * 16) = vf[kt][s];
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const bf16x8 kf0 = kf[kt][0], kf1 = kf[kt][1];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            c = MFMA(kf0, qf[qt][0], c);
            c = MFMA(kf1, qf[qt][1], c);
            acc[kt][qt] = c;
        }
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
        for (int kt = 0; kt < NKT; ++kt)
        {
            // four consecutive keys of one query row: with an even row length the index is even and one hash serves two elements
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
    bf16_t* O = reinterpret_cast<bf16_t*>(a.out) + q0 * a.ldo + h * 64;
    constexpr int NKS = (NKT + 1) / 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        bf16x8 va[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) va[ks] = trfrag(vs, 2 * ks * 16, (2 * ks + 1) * 16, 2 * ks + 1 < NKT, dt * 16, lane);
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                bf16x8 pb = pack8(acc[2 * ks][qt], (2 * ks + 1 < NKT) ? acc[(2 * ks + 1 < NKT) ? 2 * ks + 1 : 0][qt] : zero4);
                o = MFMA(va[ks], pb, o);
            }
            const int q = qt * 16 + fr;
            if (q < Lq) {
                float v[4] = {o[0], o[1], o[2], o[3]};
                store4(O + (size_t)q * a.ldo + dt * 16 + 4 * g, v);
            }
        }
    }
}

// ============================================================================ backward, probabilities computed once
// Same products, but the key-major pass no longer recomputes S, the softmax, the dropout draws and dP: the query-major pass parks
// dS[q][key] and Pd[q][key] (bf16, exactly the values it feeds its own dQ MFMAs / the values the key-major pass would recompute) in two
// row-major LDS images, and the key-major pass fetches them as MFMA B operands with the transposed reads it already uses for Q^T and
// dO^T.  The kernel is instruction-issue bound (SQ_ACTIVE_INST_ANY 0.36 per wave at two waves per SIMD): the key-major pass shrinks
// from ~2200 to ~800 instructions.  dS takes the K image's place once every dQ is done (held packed in registers until then); the Pd
// image is the only new LDS (pitch NKT * 32 + 16 bytes).
template <int NQT, int NKT>
__host__ __device__ constexpr int attn_bwd1_smem() {
    return (NKT * 16 * ROWB > NQT * 16 * (NKT * 32 + 16) ? NKT * 16 * ROWB : NQT * 16 * (NKT * 32 + 16)) + 2 * NQT * 16 * ROWB + NQT * 16 * (NKT * 32 + 16);
}
template <int NQT, int NKT>
__device__ __forceinline__ void attn_bwd1_mfma_body(const AttnArgs& a, const int blk, unsigned char* smem /* attn_bwd1_smem<NQT, NKT>() bytes */) {
    constexpr int P2 = NKT * 32 + 16;                  // pitch of the [query][key] images
    constexpr int KS_BYTES = NKT * 16 * ROWB > NQT * 16 * P2 ? NKT * 16 * ROWB : NQT * 16 * P2;
    unsigned char* ks_ = smem;                                   // K image; after the query-major pass: dS[q][key]
    unsigned char* qs_ = ks_ + KS_BYTES;
    unsigned char* os_ = qs_ + NQT * 16 * ROWB;
    unsigned char* pd_ = os_ + NQT * 16 * ROWB;                  // Pd[q][key] = dropout(P)
    const int lane = threadIdx.x, fr = lane & 15, g = lane >> 4;
    const int b = blk / a.nh, h = blk % a.nh;
    ATTN_SAMPLE_ROWS(a, b)
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + q0 * a.ldq + h * 64;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + k0 * a.ldk + h * 64;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v) + k0 * a.ldv + h * 64;
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dout) + q0 * a.lddo + h * 64;
    const float* lse = a.lse + ((size_t)b * a.nh + h) * a.Lq;
    // ONE global-load phase: every operand fragment (both passes use the same 16-B-per-lane row pieces), the log-sum-exp and
    // the key mask are requested back to back; the three LDS images the transposed reads need are then written from those
    // registers.  (The first version re-read K/Q/dO from global for the images and again per pass: ~6 dependent L2/HBM
    // round trips on a 1-wave block.)
    bf16x8 kf[NKT][2], vf[NKT][2], qf[NQT][2], of[NQT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) { kf[kt][s] = ldfrag(K, kt * 16 + fr, Lk, a.ldk, s, g); vf[kt][s] = ldfrag(V, kt * 16 + fr, Lk, a.ldv, s, g); }
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int s = 0; s < 2; ++s) { qf[qt][s] = ldfrag(Q, qt * 16 + fr, Lq, a.ldq, s, g); of[qt][s] = ldfrag(dO, qt * 16 + fr, Lq, a.lddo, s, g); }
    float lse_t[NQT], mk[NKT][4];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        lse_t[qt] = qt * 16 + fr < Lq ? lse[qt * 16 + fr] : 0.f;
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            mk[kt][r] = key < Lk ? (a.mask ? a.mask[(size_t)b * a.Lk + key] : 0.f) : -INFINITY;
        }
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) *reinterpret_cast<bf16x8*>(ks_ + (kt * 16 + fr) * ROWB + s * 64 + g * 16) = kf[kt][s];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            *reinterpret_cast<bf16x8*>(qs_ + (qt * 16 + fr) * ROWB + s * 64 + g * 16) = qf[qt][s];
            *reinterpret_cast<bf16x8*>(os_ + (qt * 16 + fr) * ROWB + s * 64 + g * 16) = of[qt][s];
        }
    __syncthreads();
    DropCfg dc = a.drop; dc.seed_hi ^= a.drop_site;
    const uint32_t idx0 = (uint32_t)((b * a.nh + h) * a.Lq) * (uint32_t)a.Lk;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int NKS = (NKT + 1) / 2, NQS = (NQT + 1) / 2;

    // ---------------- pass T: lane = query, registers = keys  ->  dQ; Pd -> pd_, dS held packed
    bf16x4 dsh[NQT][NKT];
    {
        bf16_t* dQ = reinterpret_cast<bf16_t*>(a.dq) + q0 * a.lddq + h * 64;
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            const int q = qt * 16 + fr;
            const bf16x8 qf0 = qf[qt][0], qf1 = qf[qt][1];
            const bf16x8 of0 = of[qt][0], of1 = of[qt][1];
            const float lq = lse_t[qt];
            f32x4 pp[NKT], dpp[NKT];
            float delta = 0.f;
            unsigned char* pdrow = pd_ + q * P2 + 8 * g;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp4 = {0.f, 0.f, 0.f, 0.f};
                s4 = MFMA(kf[kt][0], qf0, s4); s4 = MFMA(kf[kt][1], qf1, s4);
                dp4 = MFMA(vf[kt][0], of0, dp4); dp4 = MFMA(vf[kt][1], of1, dp4);
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
                    bf16x4 pk;
                    pk[0] = (bf16_t)keep4[0]; pk[1] = (bf16_t)keep4[1]; pk[2] = (bf16_t)keep4[2]; pk[3] = (bf16_t)keep4[3];
                    *reinterpret_cast<bf16x4*>(pdrow + kt * 32) = pk;      // keys kt*16 + 4g .. +3 of query row q
                }
            }
            delta += __shfl_xor(delta, 16, 64);
            delta += __shfl_xor(delta, 32, 64);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pp[kt][r] = pp[kt][r] * (dpp[kt][r] - delta) * a.scale;   // dS^T
                dsh[qt][kt][0] = (bf16_t)pp[kt][0]; dsh[qt][kt][1] = (bf16_t)pp[kt][1]; dsh[qt][kt][2] = (bf16_t)pp[kt][2]; dsh[qt][kt][3] = (bf16_t)pp[kt][3];
            }
            // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const bf16x8 ka = trfrag(ks_, 2 * ks * 16, (2 * ks + 1) * 16, 2 * ks + 1 < NKT, dt * 16, lane);
                    const bf16x8 db = pack8(pp[2 * ks], (2 * ks + 1 < NKT) ? pp[(2 * ks + 1 < NKT) ? 2 * ks + 1 : 0] : zero4);
                    o = MFMA(ka, db, o);
                }
                if (q < Lq) {
                    float v[4] = {o[0], o[1], o[2], o[3]};
                    store4(dQ + (size_t)q * a.lddq + dt * 16 + 4 * g, v);
                }
            }
        }
    }
    __syncthreads();   // every dQ product has read the K image
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) *reinterpret_cast<bf16x4*>(ks_ + (qt * 16 + fr) * P2 + kt * 32 + 8 * g) = dsh[qt][kt];
    __syncthreads();
    // ---------------- pass N: lane = key  ->  dK, dV from the parked dS / Pd
    {
        bf16_t* dK = reinterpret_cast<bf16_t*>(a.dk) + k0 * a.lddk + h * 64;
        bf16_t* dV = reinterpret_cast<bf16_t*>(a.dv) + k0 * a.lddv + h * 64;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int key = kt * 16 + fr;
            bf16x8 dsb[NQS], pdb[NQS];
#pragma unroll
            for (int qs = 0; qs < NQS; ++qs) {
                const bool two = 2 * qs + 1 < NQT;
                dsb[qs] = trfrag<P2>(ks_, 2 * qs * 16, (2 * qs + 1) * 16, two, kt * 16, lane);     // element jj = dS[q(jj)][key]
                pdb[qs] = trfrag<P2>(pd_, 2 * qs * 16, (2 * qs + 1) * 16, two, kt * 16, lane);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 ok = {0.f, 0.f, 0.f, 0.f}, ov = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int qs = 0; qs < NQS; ++qs) {
                    const bool two = 2 * qs + 1 < NQT;
                    const bf16x8 qa = trfrag(qs_, 2 * qs * 16, (2 * qs + 1) * 16, two, dt * 16, lane);
                    const bf16x8 oa = trfrag(os_, 2 * qs * 16, (2 * qs + 1) * 16, two, dt * 16, lane);
                    ok = MFMA(qa, dsb[qs], ok);   // dK^T[d][key] = sum_q Q^T[d][q] dS[q][key]
                    ov = MFMA(oa, pdb[qs], ov);   // dV^T[d][key] = sum_q dO^T[d][q] Pd[q][key]
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
static int mfma_check(const AttnArgs& a, bool bwd) {
    RGQA_REQUIRE(a.dh == 64, "mfma attention: head size must be 64 (got %d)", a.dh);
    RGQA_REQUIRE(a.B > 0 && a.nh > 0 && a.Lq > 0 && a.Lk > 0 && a.Lq <= 64 && a.Lk <= 64, "mfma attention: Lq/Lk must be in 1..64 (got %d %d)", a.Lq, a.Lk);
    RGQA_REQUIRE(a.q && a.k && a.v && (a.ldq % 8) == 0 && (a.ldk % 8) == 0 && (a.ldv % 8) == 0, "mfma attention: null operand or row stride not a multiple of 8");
    RGQA_REQUIRE(((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 && ((uintptr_t)a.v % 16) == 0, "mfma attention: operands must be 16-byte aligned");
    if (bwd) {
        RGQA_REQUIRE(a.dout && a.dq && a.dk && a.dv && a.lse, "mfma attention bwd: null operand");
        RGQA_REQUIRE((a.lddo % 8) == 0 && (a.lddq % 4) == 0 && (a.lddk % 4) == 0 && (a.lddv % 4) == 0, "mfma attention bwd: bad row strides");
    } else {
        RGQA_REQUIRE(a.out && (a.ldo % 4) == 0, "mfma attention: null output / bad stride");
    }
    return RGQA_OK;
}

template <int NQT, int NKT>
__global__ __launch_bounds__(64) void attn_fwd_mfma_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[NKT * 16 * ROWB];
    attn_fwd_mfma_body<NQT, NKT>(a, blockIdx.x, smem);
}
template <int NQT, int NKT>
__global__ __launch_bounds__(64) void attn_bwd1_mfma_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[attn_bwd1_smem<NQT, NKT>()];
    attn_bwd1_mfma_body<NQT, NKT>(a, blockIdx.x, smem);
}
// two problems in one launch: blocks [0, n0) work on a0 with <Q0, K0> tiles, the rest on a1 with <Q1, K1> (block-uniform branch)
template <int Q0, int K0, int Q1, int K1>
__global__ __launch_bounds__(64) void attn_fwd_mfma_pair_kernel(const AttnArgs a0, const AttnArgs a1, const int n0) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(K0 > K1 ? K0 : K1) * 16 * ROWB];
    if ((int)blockIdx.x < n0) attn_fwd_mfma_body<Q0, K0>(a0, blockIdx.x, smem);
    else attn_fwd_mfma_body<Q1, K1>(a1, blockIdx.x - n0, smem);
}
template <int Q0, int K0, int Q1, int K1>
__global__ __launch_bounds__(64) void attn_bwd1_mfma_pair_kernel(const AttnArgs a0, const AttnArgs a1, const int n0) {
    constexpr int S0 = attn_bwd1_smem<Q0, K0>(), S1 = attn_bwd1_smem<Q1, K1>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[S0 > S1 ? S0 : S1];
    if ((int)blockIdx.x < n0) attn_bwd1_mfma_body<Q0, K0>(a0, blockIdx.x, smem);
    else attn_bwd1_mfma_body<Q1, K1>(a1, blockIdx.x - n0, smem);
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

int k_attn_fwd_mfma(const AttnArgs& a, hipStream_t s) {
    int r = mfma_check(a, false);
    if (r) return r;
    const int nqt = cdiv(a.Lq, 16), nkt = cdiv(a.Lk, 16);
    dim3 grid(a.B * a.nh);
    DISPATCH_TILES(attn_fwd_mfma_kernel, nqt, nkt)
    RGQA_LAUNCH_CHECK("attn_fwd_mfma_kernel");
    return RGQA_OK;
}

int k_attn_bwd_mfma(const AttnArgs& a, hipStream_t s) {
    int r = mfma_check(a, true);
    if (r) return r;
    const int nqt = cdiv(a.Lq, 16), nkt = cdiv(a.Lk, 16);
    dim3 grid(a.B * a.nh);
    DISPATCH_TILES(attn_bwd1_mfma_kernel, nqt, nkt)
    RGQA_LAUNCH_CHECK("attn_bwd_mfma_kernel");
    return RGQA_OK;
}

// The two attention problems of a stage in ONE launch when their tile shapes are the GQA ones (questions of 17..32 tokens, 33..48 regions):
// self-attention <2,2> + <3,3>, cross-attention <2,3> + <3,2>; the heavier problem's blocks first.  Returns 1 when it launched, 0 when the
// caller should launch the two problems separately, < 0 on error.
static int pair_shape(const AttnArgs& a0, const AttnArgs& a1) {
    const int q0 = cdiv(a0.Lq, 16), k0 = cdiv(a0.Lk, 16), q1 = cdiv(a1.Lq, 16), k1 = cdiv(a1.Lk, 16);
    if (a0.nh != a1.nh || a0.dh != a1.dh) return 0;
    if (q0 == 3 && k0 == 3 && q1 == 2 && k1 == 2) return 1;
    if (q0 == 3 && k0 == 2 && q1 == 2 && k1 == 3) return 2;
    return 0;
}
int k_attn_fwd_mfma_pair(const AttnArgs& x, const AttnArgs& y, hipStream_t s) {
    const bool swap = x.Lq * x.Lk < y.Lq * y.Lk || (x.Lq * x.Lk == y.Lq * y.Lk && x.Lq < y.Lq);
    const AttnArgs& a0 = swap ? y : x; const AttnArgs& a1 = swap ? x : y;
    const int shape = pair_shape(a0, a1);
    if (shape == 0) return 0;
    int r = mfma_check(a0, false); if (r) return r;
    r = mfma_check(a1, false); if (r) return r;
    const int n0 = a0.B * a0.nh;
    dim3 grid(n0 + a1.B * a1.nh);
    if (shape == 1) hipLaunchKernelGGL((attn_fwd_mfma_pair_kernel<3, 3, 2, 2>), grid, dim3(64), 0, s, a0, a1, n0);
    else hipLaunchKernelGGL((attn_fwd_mfma_pair_kernel<3, 2, 2, 3>), grid, dim3(64), 0, s, a0, a1, n0);
    RGQA_LAUNCH_CHECK("attn_fwd_mfma_pair_kernel");
    return 1;
}
int k_attn_bwd_mfma_pair(const AttnArgs& x, const AttnArgs& y, hipStream_t s) {
    const bool swap = x.Lq * x.Lk < y.Lq * y.Lk || (x.Lq * x.Lk == y.Lq * y.Lk && x.Lq < y.Lq);
    const AttnArgs& a0 = swap ? y : x; const AttnArgs& a1 = swap ? x : y;
    const int shape = pair_shape(a0, a1);
    if (shape == 0) return 0;
    int r = mfma_check(a0, true); if (r) return r;
    r = mfma_check(a1, true); if (r) return r;
    const int n0 = a0.B * a0.nh;
    dim3 grid(n0 + a1.B * a1.nh);
    if (shape == 1) hipLaunchKernelGGL((attn_bwd1_mfma_pair_kernel<3, 3, 2, 2>), grid, dim3(64), 0, s, a0, a1, n0);
    else hipLaunchKernelGGL((attn_bwd1_mfma_pair_kernel<3, 2, 2, 3>), grid, dim3(64), 0, s, a0, a1, n0);
    RGQA_LAUNCH_CHECK("attn_bwd1_mfma_pair_kernel");
    return 1;
}
