// Test-time scoring of answer logits: what every RVQA test script of the reference computes on the
// [B, num_answers] logits, fused into one pass over each row.
//   max_score, label = torch.sigmoid(logit / temperature).max(1)     tasks/gqa_conf.py:344, gqa_energy.py:184,204; gqa_odin.py:130-131
//   energy           = torch.log(1 + torch.exp(logit)).sum(1)        tasks/gqa_energy.py:135,185
//   topk             = logit.topk(k)                                 tasks/gqa_energy.py:205, gqa_check_topk_preds.py:189
//   topk_energy      = torch.log(1 + torch.exp(topk.values)).sum(1)  tasks/gqa_energy.py:206
// One wave per row (1842 answers = 29 elements per lane), HBM-bound: 4 B per logit read once (the k selection passes re-read
// the 7-KiB row from L1/L2).  Formulas are kept as the reference writes them (the naive softplus overflows to +inf above
// logit 88.7 exactly as torch's does); the max runs over the SIGMOID values, so saturated ties resolve to the first index
// as on the reference's CPU path.
#include "kernels.h"

__device__ __forceinline__ float softplus_naive(float x) { return logf(1.0f + expf(x)); }

__global__ __launch_bounds__(256) void score_rows_kernel(const float* __restrict__ logits, int ld, int B, int NA, float temperature, int k,
                                                         float* __restrict__ max_score, int64_t* __restrict__ label, float* __restrict__ energy,
                                                         float* __restrict__ topk_val, int64_t* __restrict__ topk_idx, float* __restrict__ topk_energy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;                               // wave-uniform
    const float* x = logits + (size_t)row * ld;
    float e = 0.f, best = -1.0f;                        // sigmoid values are >= 0
    int besti = 0x7fffffff;
    for (int j = lane; j < NA; j += 64) {
        const float v = x[j];
        e += softplus_naive(v);
        const float sg = 1.0f / (1.0f + expf(-(v / temperature)));
        if (sg > best) { best = sg; besti = j; }         // ascending j per lane: the first maximum stays
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        e += __shfl_xor(e, o);
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(besti, o);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) {
        if (energy) energy[row] = e;
        if (max_score) max_score[row] = best;
        if (label) label[row] = besti;
    }
    if (k <= 0) return;
    // top-k of the raw logits, descending, ties by ascending index: k selection passes in (value desc, index asc) order
    float last_v = INFINITY, te = 0.f;
    int last_i = -1;
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int j = lane; j < NA; j += 64) {
            const float v = x[j];
            const bool after = v < last_v || (v == last_v && j > last_i);      // not selected yet
            if (after && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        last_v = bv; last_i = bi;
        te += softplus_naive(bv);
        if (lane == 0) {
            if (topk_val) topk_val[(size_t)row * k + r] = bv;
            if (topk_idx) topk_idx[(size_t)row * k + r] = bi;
        }
    }
    if (lane == 0 && topk_energy) topk_energy[row] = te;
}

int k_score_rows(const float* logits, int ld, int B, int NA, float temperature, int k, float* max_score, int64_t* label, float* energy,
                 float* topk_val, int64_t* topk_idx, float* topk_energy, hipStream_t s) {
    RGQA_REQUIRE(logits != nullptr && B > 0 && NA > 0 && ld >= NA, "score_rows: bad shape B=%d NA=%d ld=%d", B, NA, ld);
    RGQA_REQUIRE(k >= 0 && k <= NA, "score_rows: k=%d outside [0, %d]", k, NA);
    RGQA_REQUIRE(temperature > 0.f, "score_rows: temperature must be positive");
    hipLaunchKernelGGL(score_rows_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, logits, ld, B, NA, temperature, k, max_score, label, energy, topk_val, topk_idx, topk_energy);
    RGQA_LAUNCH_CHECK("score_rows_kernel");
    return RGQA_OK;
}
