// Grouped GEMM descriptors + the shared epilogue (used by the bf16 MFMA kernels and the f32 kernel).
#pragma once
#include "common.h"

enum GemmEpi {
    EPI_BIAS = 0,        // C = acc + bias
    EPI_GELU = 1,        // pre = acc + bias; C = gelu(pre); C2 = gelu'(pre) (if C2)     (BertIntermediate; C2 feeds EPI_DGELU)
    EPI_TANH = 2,        // C = tanh(acc + bias)                                        (BertPooler)
    EPI_RESID_DROP = 3,  // C = dropout(acc + bias) + aux                               (BertAttOutput / BertOutput, pre-LN)
    EPI_DGELU = 4,       // C = acc * aux, aux = the gelu'(pre) saved by EPI_GELU       (dgrad through GeLU)
    EPI_ADD = 5,         // C = acc + aux                                               (dgrad + residual-path gradient)
    EPI_ACCUM = 6,       // C += acc  (C must be f32)                                   (wgrad accumulate)
    EPI_DTANH = 7,       // C = acc * (1 - aux^2)                                       (dgrad through the pooler's tanh)
    EPI_RELU = 8,        // C = relu(acc + bias)                                        (BUTD MLP, butd.py:8-26)
    EPI_RELU_DROP = 9,   // C = dropout(relu(acc + bias))                               (BUTD classifier hidden, butd.py:170-178)
    EPI_DRELU_DROP = 10, // C = acc * (aux > 0 ? 1/(1-p) : 0), aux = the EPI_RELU_DROP output (dgrad through dropout + ReLU)
};

// One problem: C[M,N] = op(A)[M,K] * op(B)[K,N] (+ epilogue).  Layout depends on the kernel:
//   NT kernel: A [M,K] row-major (lda), B = W [N,K] row-major (ldb)        -> x @ W^T
//   TN kernel: A = P [K,M] row-major (lda), B = Q [K,N] row-major (ldb)    -> P^T @ Q   (wgrad: dW[n,k] = dY^T X)
struct GemmProblem {
    const void* A;
    const void* B;
    void* C;
    void* C2;           // optional second output (pre-activation), same type / ld as C (bf16 when c2_lp is set)
    void* Cb;           // split-f32 results only: optional bf16 image of C at the same element offsets (ldc), or null - what the bf16 backward of the
                        // bf16x3_fwd precision reads (the image IS the hi part of every element: one more 16-byte store per lane, no arithmetic)
    int c2_lp;          // split-f32 results only: C2 is a bf16 buffer (gelu' is read by the backward alone)
    const float* bias;  // [N] or null
    const void* aux;    // [M,N] residual or pre-activation, act type
    int M, N, K;
    int lda, ldb, ldc, ldaux;
    int tile_start;     // first linear tile id of this problem in the grouped launch
    int tiles_n;        // number of tiles along N
    uint32_t drop_site; // mixed into the dropout stream for EPI_RESID_DROP
    float* colsum_out;  // TN kernels only: colsum_out[m] (+)= sum_k A[k][m]  (bias gradient riding the wgrad GEMM), or null
    int epi;            // GemmEpi
    // bf16 LDS-DMA NT kernels, EPI_RESID_DROP, N == 768 (round 5): the LayerNorm that follows the dense layer (BertAttOutput / BertOutput,
    // lxrt/modeling.py:350-361, 404-415), done by the workgroup that finishes the LAST of a row block's N / 256 tiles - ln_tk != null switches it on:
    // ln_y[m, :] = (C[m, :] - mean) * rstd * ln_g + ln_b (ld = ldc), ln_mean / ln_rstd [M] for the backward pass; ln_tk: one zeroed int per row block
    // of this problem (the last arriver resets it)
    const float* ln_g; const float* ln_b; void* ln_y; float* ln_mean; float* ln_rstd; int* ln_tk; float ln_eps;
};

// 32 (3,888 bytes of kernel arguments): two or three layers' weight-gradient problems go into one launch (engine.hip, flush_wgrad)
#define GEMM_MAX_PROBLEMS 32
// NT launches (forward, dgrad) group at most 12: their kernels look a tile's problem up in a loop over this many entries, once per tile
#define GEMM_NT_MAX_PROBLEMS 12
// NP problems; the NT kernels take the 12-problem prefix of a group as their argument (1.5 KB of kernel arguments instead of 3.9), the
// TN kernels the whole group
template <int NP>
struct GemmGroupT {
    int count;
    int total_tiles;
    int a_f32;          // NT only: A operand is f32 in memory (converted to bf16 while staging)
    int b_kn;           // bf16 NT LDS-DMA kernels only: EVERY problem's B operand is stored [K, N] row-major (ldb = row pitch) instead of [N, K]: C = A B.
                        // The dgrad GEMMs on the weight as it lies ([out, in]: the contraction runs over its rows) - no transposed copy.  K % 64 == 0
    int stagger;        // persistent NT kernels (round 6, rgqa_debug_set key 22; 0 = off): a block that walks one tile FEWER than the busiest blocks starts late by
                        // stagger / 16 x (its K-step count) x ~2 us - inside the slack it has anyway - so that its K loops run while the others store their tiles
    const void* zeros;  // TN LDS-DMA kernel: >= 16 zero bytes on the device (source of the contraction tail's A rows)
    unsigned long long* stamps;   // clock probe only (rgqa_probe_gemm: separately instantiated, stamped kernels): 8 words per block
    float* splitk_ws;   // bf16 NT launches: scratch for the split-K path of skinny problems (gemm_mfma256.hip), splitk_floats floats, or null
    size_t splitk_floats;
    DropCfg drop;
    GemmProblem p[NP];
};
using GemmGroup = GemmGroupT<GEMM_MAX_PROBLEMS>;
using GemmGroupNT = GemmGroupT<GEMM_NT_MAX_PROBLEMS>;
static inline const GemmGroupNT& nt_prefix(const GemmGroup& g) { return reinterpret_cast<const GemmGroupNT&>(g); }

// Epilogue on 4 consecutive columns n0..n0+3 of row m. OutT = float or bf16_t; AuxT = act type.
template <typename OutT, typename AuxT>
__device__ __forceinline__ void gemm_epilogue4_e(const GemmProblem& P, const int epi, const DropCfg& drop, int m, int n0, float v[4]) {
    if (m >= P.M || n0 >= P.N) return;
    const int nvalid = (P.N - n0) >= 4 ? 4 : (P.N - n0);
    if (P.bias != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < nvalid) v[i] += P.bias[n0 + i];
    }
    OutT* crow = reinterpret_cast<OutT*>(P.C) + (size_t)m * P.ldc + n0;
    float pre[4] = {v[0], v[1], v[2], v[3]};
    if (epi == EPI_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = gelu_f(pre[i]); pre[i] = dgelu_f(pre[i]); }
    } else if (epi == EPI_TANH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = tanhf(pre[i]);
    } else if (epi == EPI_RELU || epi == EPI_RELU_DROP) {
        DropCfg d = drop; d.seed_hi ^= P.drop_site;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = fmaxf(pre[i], 0.f);
            if (epi == EPI_RELU_DROP) v[i] = drop_apply(d, (uint32_t)m * (uint32_t)P.N + (uint32_t)(n0 + i), v[i]);
        }
    } else if (epi == EPI_RESID_DROP || epi == EPI_DGELU || epi == EPI_ADD || epi == EPI_DTANH || epi == EPI_DRELU_DROP) {
        const AuxT* arow = reinterpret_cast<const AuxT*>(P.aux) + (size_t)m * P.ldaux + n0;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (nvalid == 4) load4(arow, a);
        else for (int i = 0; i < nvalid; ++i) a[i] = ld_elem(arow + i);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (epi == EPI_RESID_DROP) {
                uint32_t idx = (uint32_t)m * (uint32_t)P.N + (uint32_t)(n0 + i);
                DropCfg d = drop; d.seed_hi ^= P.drop_site;
                v[i] = drop_apply(d, idx, v[i]) + a[i];
            } else if (epi == EPI_DGELU) {
                v[i] = v[i] * a[i];
            } else if (epi == EPI_DTANH) {
                v[i] = v[i] * (1.0f - a[i] * a[i]);
            } else if (epi == EPI_DRELU_DROP) {
                v[i] = a[i] > 0.f ? v[i] * drop.scale : 0.f;
            } else {
                v[i] = v[i] + a[i];
            }
        }
    } else if (epi == EPI_ACCUM) {
        float* c = reinterpret_cast<float*>(P.C) + (size_t)m * P.ldc + n0;
        if (nvalid == 4) {
            float o[4]; load4(c, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += v[i];
            store4(c, o);
        } else for (int i = 0; i < nvalid; ++i) c[i] += v[i];
        return;
    }
    if (nvalid == 4) {
        store4(crow, v);
        if (P.Cb != nullptr) store4(reinterpret_cast<bf16_t*>(P.Cb) + (size_t)m * P.ldc + n0, v);      // split-f32 results (bf16x3_fwd): the bf16 image beside them
        if (epi == EPI_GELU && P.C2 != nullptr) {
            if (P.c2_lp) store4(reinterpret_cast<bf16_t*>(P.C2) + (size_t)m * P.ldc + n0, pre);
            else store4(reinterpret_cast<OutT*>(P.C2) + (size_t)m * P.ldc + n0, pre);
        }
    } else {
        for (int i = 0; i < nvalid; ++i) {
            st_elem(crow + i, v[i]);
            if (epi == EPI_GELU && P.C2 != nullptr) st_elem(reinterpret_cast<OutT*>(P.C2) + (size_t)m * P.ldc + n0 + i, pre[i]);
        }
    }
}

template <typename OutT, typename AuxT>
__device__ __forceinline__ void gemm_epilogue4(const GemmProblem& P, const DropCfg& drop, int m, int n0, float v[4]) {
    gemm_epilogue4_e<OutT, AuxT>(P, P.epi, drop, m, n0, v);
}

// ---- two-phase epilogue for the MFMA kernels: all bias / aux operands of a lane are fetched up front (one exposed
// memory round trip) instead of one dependent load per 4 outputs (loads cannot be hoisted over the stores in between).
struct Aux4 { float a[4]; };

__host__ __device__ __forceinline__ bool epi_needs_aux(int epi) { return epi == EPI_RESID_DROP || epi == EPI_DGELU || epi == EPI_ADD || epi == EPI_DTANH || epi == EPI_DRELU_DROP; }

__device__ __forceinline__ void epi_fetch_bias(const GemmProblem& P, int n0, float b[4]) {
    b[0] = b[1] = b[2] = b[3] = 0.f;
    if (P.bias == nullptr || n0 >= P.N) return;
    if (n0 + 3 < P.N) load4(P.bias + n0, b);
    else for (int i = 0; i < P.N - n0; ++i) b[i] = P.bias[n0 + i];
}

// raw (packed) aux fetch: 4 consecutive AuxT elements as a uint2 (bf16) — converted at use
template <typename AuxT> struct AuxRaw;
template <> struct AuxRaw<bf16_t> { uint2 r; };
template <> struct AuxRaw<float> { float4 r; };

__device__ __forceinline__ void epi_fetch_aux(const GemmProblem& P, int epi, int m, int n0, AuxRaw<bf16_t>& out) {
    out.r = make_uint2(0u, 0u);
    if (!epi_needs_aux(epi) || m >= P.M || n0 >= P.N) return;
    const bf16_t* arow = reinterpret_cast<const bf16_t*>(P.aux) + (size_t)m * P.ldaux + n0;
    if (n0 + 3 < P.N) out.r = *reinterpret_cast<const uint2*>(arow);
    else {
        bf16x4 t = {0, 0, 0, 0};
        for (int i = 0; i < P.N - n0; ++i) t[i] = arow[i];
        out.r = *reinterpret_cast<uint2*>(&t);
    }
}
__device__ __forceinline__ void aux_unpack(const AuxRaw<bf16_t>& in, float a[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(&in.r);
    a[0] = (float)t[0]; a[1] = (float)t[1]; a[2] = (float)t[2]; a[3] = (float)t[3];
}

// finish: v = acc (4 consecutive columns of row m); bias and aux already in registers
template <typename OutT>
__device__ __forceinline__ void epi_finish(const GemmProblem& P, const int epi, const DropCfg& drop, int m, int n0, const float b[4], const float a[4], float v[4]) {
    if (m >= P.M || n0 >= P.N) return;
    const int nvalid = (P.N - n0) >= 4 ? 4 : (P.N - n0);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += b[i];
    float pre[4] = {v[0], v[1], v[2], v[3]};
    if (epi == EPI_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = gelu_f(pre[i]); pre[i] = dgelu_f(pre[i]); }
    } else if (epi == EPI_TANH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = tanhf(pre[i]);
    } else if (epi == EPI_RELU || epi == EPI_RELU_DROP) {
        DropCfg d = drop; d.seed_hi ^= P.drop_site;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = fmaxf(pre[i], 0.f);
            if (epi == EPI_RELU_DROP) v[i] = drop_apply(d, (uint32_t)m * (uint32_t)P.N + (uint32_t)(n0 + i), v[i]);
        }
    } else if (epi == EPI_DRELU_DROP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = a[i] > 0.f ? v[i] * drop.scale : 0.f;
    } else if (epi == EPI_RESID_DROP) {
        DropCfg d = drop; d.seed_hi ^= P.drop_site;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = drop_apply(d, (uint32_t)m * (uint32_t)P.N + (uint32_t)(n0 + i), v[i]) + a[i];
    } else if (epi == EPI_DGELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] * a[i];
    } else if (epi == EPI_DTANH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] * (1.0f - a[i] * a[i]);
    } else if (epi == EPI_ADD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] + a[i];
    } else if (epi == EPI_ACCUM) {
        float* c = reinterpret_cast<float*>(P.C) + (size_t)m * P.ldc + n0;
        if (nvalid == 4) {
            float o[4]; load4(c, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += v[i];
            store4(c, o);
        } else for (int i = 0; i < nvalid; ++i) c[i] += v[i];
        return;
    }
    OutT* crow = reinterpret_cast<OutT*>(P.C) + (size_t)m * P.ldc + n0;
    if (nvalid == 4) {
        store4(crow, v);
        if (epi == EPI_GELU && P.C2 != nullptr) store4(reinterpret_cast<OutT*>(P.C2) + (size_t)m * P.ldc + n0, pre);
    } else {
        for (int i = 0; i < nvalid; ++i) {
            crow[i] = from_f32<OutT>(v[i]);
            if (epi == EPI_GELU && P.C2 != nullptr) (reinterpret_cast<OutT*>(P.C2) + (size_t)m * P.ldc + n0)[i] = from_f32<OutT>(pre[i]);
        }
    }
}

// host launchers (gemm_mfma.hip / gemm_f32.hip). out_f32: C is float (else bf16). All return RGQA_* codes.
int launch_gemm_nt_bf16(GemmGroup& g, int out_f32, hipStream_t s);
int launch_gemm_tn_bf16(GemmGroup& g, int out_f32, hipStream_t s);
// bf16x3 precision (gemm_x3.hip): A / B (and C, aux unless out_f32) in the split-f32 layout
int launch_gemm_nt_x3(GemmGroup& g, int out_f32, hipStream_t s);
int launch_gemm_tn_x3(GemmGroup& g, hipStream_t s);
// f32 generic: A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; everything f32.
int launch_gemm_f32(GemmGroup& g, int trans_a, int trans_b, hipStream_t s);
void gemm_group_finalize(GemmGroup& g, int bm, int bn);
