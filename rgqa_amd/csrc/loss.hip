// Fused BCE-with-logits loss forward + backward (reference tasks/gqa_conf.py:197-198):
//   loss = BCEWithLogitsLoss()(z, t) * NA = (1/B) * sum_{b,n} [max(z,0) - z t + log1p(exp(-|z|))]
//   dz   = (sigmoid(z) - t) / B
#include "kernels.h"

// rows != null: the sample's share of the loss goes to rows[b] and bce_fold_kernel adds the B shares in a fixed order (the engines: a train step
// reports the same loss bit for bit every run); null: the shares meet in *loss by float atomics (the stand-alone operator: no scratch in its signature)
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ z, int ldl, const float* __restrict__ t, int ldt, float* __restrict__ loss,
                                                  float* __restrict__ dz, int lddl, int B, int NA, int NAp, float inv_b, float gscale, float* __restrict__ rows) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int n = threadIdx.x; n < NAp; n += 256) {
        float g = 0.f;
        if (n < NA) {
            const float x = z[(size_t)b * ldl + n], y = t[(size_t)b * ldt + n];
            acc += fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x)));
            const float sg = 1.f / (1.f + __expf(-x));
            g = (sg - y) * inv_b * gscale;
        }
        if (dz) dz[(size_t)b * lddl + n] = g;   // padded columns get exact zeros (they feed K-padded GEMMs)
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (red[0] + red[1] + red[2] + red[3]) * inv_b;
        if (rows) rows[b] = v;
        else if (loss) atomicAdd(loss, v);
    }
}

__global__ __launch_bounds__(256) void bce_fold_kernel(const float* __restrict__ rows, int B, float* __restrict__ loss) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) acc += rows[b];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (red[0] + red[1]) + (red[2] + red[3]);
}

int k_bce_fwd_bwd(const float* logits, int ldl, const float* target, int ldt, float* loss_out, float* dlogits, int lddl, int B, int NA, int NAp, float grad_scale, hipStream_t s,
                  float* row_scratch) {
    RGQA_REQUIRE(B > 0 && NA > 0 && NAp >= NA, "bce: bad shape");
    if (loss_out == nullptr) row_scratch = nullptr;
    if (loss_out && !row_scratch) RGQA_HIP(hipMemsetAsync(loss_out, 0, sizeof(float), s));
    hipLaunchKernelGGL(bce_kernel, dim3(B), dim3(256), 0, s, logits, ldl, target, ldt, loss_out, dlogits, lddl, B, NA, NAp, 1.0f / (float)B, grad_scale, row_scratch);
    RGQA_LAUNCH_CHECK("bce_kernel");
    if (row_scratch) {
        hipLaunchKernelGGL(bce_fold_kernel, dim3(1), dim3(256), 0, s, row_scratch, B, loss_out);
        RGQA_LAUNCH_CHECK("bce_fold_kernel");
    }
    return RGQA_OK;
}
