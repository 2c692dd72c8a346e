// Fused attention core (reference BertAttention.forward, lxrt/modeling.py:326-346) for tiny sequences
// (Lq, Lk <= 64): scores = Q K^T * scale + additive key mask -> softmax (f32) -> dropout -> P V, heads
// merged in place; no [B,12,Lq,Lk] tensor ever reaches HBM.  One workgroup per (sample, head).
//
// This file holds the type-generic LDS/VALU implementation used by the f32 parity mode (and as the on-GPU
// cross-check of the MFMA kernels in attn_mfma.hip).
#include "kernels.h"

template <typename T>
__global__ __launch_bounds__(64) void attn_fwd_ref_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / a.nh, h = blockIdx.x % a.nh;
    ATTN_SAMPLE_ROWS(a, b)
    const int dh = a.dh, dp = dh + 1, kp = Lk + 1;
    float* Qs = sm;
    float* Ks = Qs + Lq * dp;
    float* Vs = Ks + Lk * dp;
    float* Ps = Vs + Lk * dp;
    const T* q = reinterpret_cast<const T*>(a.q) + q0 * a.ldq + h * dh;
    const T* k = reinterpret_cast<const T*>(a.k) + k0 * a.ldk + h * dh;
    const T* v = reinterpret_cast<const T*>(a.v) + k0 * a.ldv + h * dh;
    const int tid = threadIdx.x;
    for (int x = tid; x < Lq * dh; x += 64) Qs[(x / dh) * dp + x % dh] = ld_elem(q + (size_t)(x / dh) * a.ldq + x % dh);
    for (int x = tid; x < Lk * dh; x += 64) {
        Ks[(x / dh) * dp + x % dh] = ld_elem(k + (size_t)(x / dh) * a.ldk + x % dh);
        Vs[(x / dh) * dp + x % dh] = ld_elem(v + (size_t)(x / dh) * a.ldv + x % dh);
    }
    __syncthreads();
    for (int x = tid; x < Lq * Lk; x += 64) {
        const int i = x / Lk, j = x % Lk;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s = fmaf(Qs[i * dp + d], Ks[j * dp + d], s);
        s *= a.scale;
        if (a.mask) s += a.mask[(size_t)b * a.Lk + j];
        Ps[i * kp + j] = s;
    }
    __syncthreads();
    DropCfg dc = a.drop; dc.seed_hi ^= a.drop_site;
    for (int i = tid; i < Lq; i += 64) {
        float m = -INFINITY;
        for (int j = 0; j < Lk; ++j) m = fmaxf(m, Ps[i * kp + j]);
        float sum = 0.f;
        for (int j = 0; j < Lk; ++j) { float e = __expf(Ps[i * kp + j] - m); Ps[i * kp + j] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int j = 0; j < Lk; ++j) {
            uint32_t idx = (uint32_t)(((b * a.nh + h) * a.Lq + i) * a.Lk + j);
            Ps[i * kp + j] = drop_apply(dc, idx, Ps[i * kp + j] * inv);
        }
        if (a.lse) a.lse[((size_t)b * a.nh + h) * a.Lq + i] = m + __logf(sum);
    }
    __syncthreads();
    T* o = reinterpret_cast<T*>(a.out) + q0 * a.ldo + h * dh;
    for (int x = tid; x < Lq * dh; x += 64) {
        const int i = x / dh, d = x % dh;
        float s = 0.f;
        for (int j = 0; j < Lk; ++j) s = fmaf(Ps[i * kp + j], Vs[j * dp + d], s);
        st_elem(o + (size_t)i * a.ldo + d, s);
    }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_ref_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / a.nh, h = blockIdx.x % a.nh;
    ATTN_SAMPLE_ROWS(a, b)
    const int dh = a.dh, dp = dh + 1, kp = Lk + 1;
    float* Qs = sm;
    float* Ks = Qs + Lq * dp;
    float* Vs = Ks + Lk * dp;
    float* Os = Vs + Lk * dp;   // dO
    float* Ps = Os + Lq * dp;   // dropped probabilities (what multiplied V)
    float* Ds = Ps + Lq * kp;   // dS
    const T* q = reinterpret_cast<const T*>(a.q) + q0 * a.ldq + h * dh;
    const T* k = reinterpret_cast<const T*>(a.k) + k0 * a.ldk + h * dh;
    const T* v = reinterpret_cast<const T*>(a.v) + k0 * a.ldv + h * dh;
    const T* dO = reinterpret_cast<const T*>(a.dout) + q0 * a.lddo + h * dh;
    const int tid = threadIdx.x;
    for (int x = tid; x < Lq * dh; x += 64) {
        Qs[(x / dh) * dp + x % dh] = ld_elem(q + (size_t)(x / dh) * a.ldq + x % dh);
        Os[(x / dh) * dp + x % dh] = ld_elem(dO + (size_t)(x / dh) * a.lddo + x % dh);
    }
    for (int x = tid; x < Lk * dh; x += 64) {
        Ks[(x / dh) * dp + x % dh] = ld_elem(k + (size_t)(x / dh) * a.ldk + x % dh);
        Vs[(x / dh) * dp + x % dh] = ld_elem(v + (size_t)(x / dh) * a.ldv + x % dh);
    }
    __syncthreads();
    DropCfg dc = a.drop; dc.seed_hi ^= a.drop_site;
    // P (undropped) into Ds temporarily, dP (through the dropout mask) into Ps
    for (int x = tid; x < Lq * Lk; x += 64) {
        const int i = x / Lk, j = x % Lk;
        float s = 0.f, dp_ = 0.f;
        for (int d = 0; d < dh; ++d) {
            s = fmaf(Qs[i * dp + d], Ks[j * dp + d], s);
            dp_ = fmaf(Os[i * dp + d], Vs[j * dp + d], dp_);
        }
        s *= a.scale;
        if (a.mask) s += a.mask[(size_t)b * a.Lk + j];
        const float p = __expf(s - a.lse[((size_t)b * a.nh + h) * a.Lq + i]);
        uint32_t idx = (uint32_t)(((b * a.nh + h) * a.Lq + i) * a.Lk + j);
        const float keep = drop_apply(dc, idx, 1.0f);   // 0 or 1/(1-p)
        Ds[i * kp + j] = p;
        Ps[i * kp + j] = dp_ * keep;                     // dL/dp_ij
    }
    __syncthreads();
    for (int i = tid; i < Lq; i += 64) {
        float delta = 0.f;
        for (int j = 0; j < Lk; ++j) delta = fmaf(Ds[i * kp + j], Ps[i * kp + j], delta);
        for (int j = 0; j < Lk; ++j) {
            const float p = Ds[i * kp + j], dpv = Ps[i * kp + j];
            uint32_t idx = (uint32_t)(((b * a.nh + h) * a.Lq + i) * a.Lk + j);
            Ds[i * kp + j] = p * (dpv - delta) * a.scale;   // dS (w.r.t. Q K^T before scaling folded in)
            Ps[i * kp + j] = drop_apply(dc, idx, p);         // dropped P for dV
        }
    }
    __syncthreads();
    T* dq = reinterpret_cast<T*>(a.dq) + q0 * a.lddq + h * dh;
    T* dk = reinterpret_cast<T*>(a.dk) + k0 * a.lddk + h * dh;
    T* dv = reinterpret_cast<T*>(a.dv) + k0 * a.lddv + h * dh;
    for (int x = tid; x < Lq * dh; x += 64) {
        const int i = x / dh, d = x % dh;
        float s = 0.f;
        for (int j = 0; j < Lk; ++j) s = fmaf(Ds[i * kp + j], Ks[j * dp + d], s);
        st_elem(dq + (size_t)i * a.lddq + d, s);
    }
    for (int x = tid; x < Lk * dh; x += 64) {
        const int j = x / dh, d = x % dh;
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < Lq; ++i) {
            s1 = fmaf(Ds[i * kp + j], Qs[i * dp + d], s1);
            s2 = fmaf(Ps[i * kp + j], Os[i * dp + d], s2);
        }
        st_elem(dk + (size_t)j * a.lddk + d, s1);
        st_elem(dv + (size_t)j * a.lddv + d, s2);
    }
}

static int attn_check(const AttnArgs& a, bool bwd) {
    RGQA_REQUIRE(a.B > 0 && a.nh > 0 && a.Lq > 0 && a.Lk > 0 && a.dh > 0, "attention: empty problem");
    RGQA_REQUIRE(a.Lq <= 64 && a.Lk <= 64 && a.dh <= 64, "attention: Lq/Lk/dh must be <= 64 (got %d %d %d)", a.Lq, a.Lk, a.dh);
    RGQA_REQUIRE(a.q && a.k && a.v, "attention: null operand");
    if (bwd) RGQA_REQUIRE(a.dout && a.dq && a.dk && a.dv && a.lse, "attention bwd: null operand");
    else RGQA_REQUIRE(a.out != nullptr, "attention: null output");
    return RGQA_OK;
}

template <typename T>
int k_attn_fwd_ref(const AttnArgs& a, hipStream_t s) {
    int r = attn_check(a, false);
    if (r) return r;
    size_t sh = ((size_t)a.Lq * (a.dh + 1) + 2 * (size_t)a.Lk * (a.dh + 1) + (size_t)a.Lq * (a.Lk + 1)) * sizeof(float);
    if (sh > 48 * 1024) RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_ref_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(attn_fwd_ref_kernel<T>, dim3(a.B * a.nh), dim3(64), sh, s, a);
    RGQA_LAUNCH_CHECK("attn_fwd_ref_kernel");
    return RGQA_OK;
}

template <typename T>
int k_attn_bwd_ref(const AttnArgs& a, hipStream_t s) {
    int r = attn_check(a, true);
    if (r) return r;
    size_t sh = (2 * (size_t)a.Lq * (a.dh + 1) + 2 * (size_t)a.Lk * (a.dh + 1) + 2 * (size_t)a.Lq * (a.Lk + 1)) * sizeof(float);
    if (sh > 48 * 1024) RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_ref_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(attn_bwd_ref_kernel<T>, dim3(a.B * a.nh), dim3(64), sh, s, a);
    RGQA_LAUNCH_CHECK("attn_bwd_ref_kernel");
    return RGQA_OK;
}

// Attention probabilities of one attention call, written out for inspection (reference lxrt_vis/modeling.py:337,347-348:
// `output_attention=True` returns softmax(QK^T/sqrt(d) + mask)): out[B, nh, a.Lq, a.Lk] f32, recomputed from the Q / K the
// forward pass left in its qkv buffer.  Rows / columns outside a packed sample's window are written as 0 (a padded key's
// probability is exp(-10000) = 0 in the reference too; padded QUERY rows are simply not computed in the packed layout).
template <typename T>
__global__ __launch_bounds__(64) void attn_probs_kernel(const AttnArgs a, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / a.nh, h = blockIdx.x % a.nh;
    ATTN_SAMPLE_ROWS(a, b)
    const int dh = a.dh, dp = dh + 1, kp = Lk + 1;
    float* Qs = sm;
    float* Ks = Qs + Lq * dp;
    float* Ps = Ks + Lk * dp;
    const T* q = reinterpret_cast<const T*>(a.q) + q0 * a.ldq + h * dh;
    const T* k = reinterpret_cast<const T*>(a.k) + k0 * a.ldk + h * dh;
    const int tid = threadIdx.x;
    for (int x = tid; x < Lq * dh; x += 64) Qs[(x / dh) * dp + x % dh] = ld_elem(q + (size_t)(x / dh) * a.ldq + x % dh);
    for (int x = tid; x < Lk * dh; x += 64) Ks[(x / dh) * dp + x % dh] = ld_elem(k + (size_t)(x / dh) * a.ldk + x % dh);
    __syncthreads();
    for (int x = tid; x < Lq * Lk; x += 64) {
        const int i = x / Lk, j = x % Lk;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s = fmaf(Qs[i * dp + d], Ks[j * dp + d], s);
        s *= a.scale;
        if (a.mask) s += a.mask[(size_t)b * a.Lk + j];
        Ps[i * kp + j] = s;
    }
    __syncthreads();
    float* o = out + ((size_t)b * a.nh + h) * a.Lq * a.Lk;
    for (int i = tid; i < a.Lq; i += 64) {
        if (i < Lq) {
            float m = -INFINITY;
            for (int j = 0; j < Lk; ++j) m = fmaxf(m, Ps[i * kp + j]);
            float sum = 0.f;
            for (int j = 0; j < Lk; ++j) { float e = expf(Ps[i * kp + j] - m); Ps[i * kp + j] = e; sum += e; }
            const float inv = 1.f / sum;
            for (int j = 0; j < a.Lk; ++j) o[(size_t)i * a.Lk + j] = j < Lk ? Ps[i * kp + j] * inv : 0.f;
        } else {
            for (int j = 0; j < a.Lk; ++j) o[(size_t)i * a.Lk + j] = 0.f;
        }
    }
}

template <typename T>
int k_attn_probs(const AttnArgs& a, float* out, hipStream_t s) {
    RGQA_REQUIRE(a.q != nullptr && a.k != nullptr && out != nullptr && a.B > 0 && a.nh > 0 && a.Lq > 0 && a.Lk > 0 && a.dh > 0, "attention probabilities: bad arguments");
    size_t sh = ((size_t)a.Lq * (a.dh + 1) + (size_t)a.Lk * (a.dh + 1) + (size_t)a.Lq * (a.Lk + 1)) * sizeof(float);
    RGQA_REQUIRE(sh <= 160 * 1024, "attention probabilities: %d x %d x %d does not fit LDS", a.Lq, a.Lk, a.dh);
    if (sh > 48 * 1024) RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_probs_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(attn_probs_kernel<T>, dim3(a.B * a.nh), dim3(64), sh, s, a, out);
    RGQA_LAUNCH_CHECK("attn_probs_kernel");
    return RGQA_OK;
}
template int k_attn_probs<float>(const AttnArgs&, float*, hipStream_t);
template int k_attn_probs<bf16_t>(const AttnArgs&, float*, hipStream_t);
template int k_attn_probs<sf32>(const AttnArgs&, float*, hipStream_t);

template int k_attn_fwd_ref<float>(const AttnArgs&, hipStream_t);
template int k_attn_fwd_ref<bf16_t>(const AttnArgs&, hipStream_t);
template int k_attn_fwd_ref<sf32>(const AttnArgs&, hipStream_t);
template int k_attn_bwd_ref<float>(const AttnArgs&, hipStream_t);
template int k_attn_bwd_ref<bf16_t>(const AttnArgs&, hipStream_t);
template int k_attn_bwd_ref<sf32>(const AttnArgs&, hipStream_t);
