// Input path: what the reference's GQATorchDataset.__getitem__ does per sample on the host
// (tasks/gqa_data.py:173-238) done per BATCH on the device, fed from a binary feature store instead of the base64 TSV
// (utils.py:16-54):
//   feats  [B,O,F]  f16 or f32 in the store  -> f32 (the engine's input type); f16 halves the host->device bytes (147 KB/sample)
//   boxes  [B,O,4]  pixel coordinates        -> boxes[:, (0,2)] /= img_w; boxes[:, (1,3)] /= img_h        (gqa_data.py:197-200)
//   labels {answer: score} per sample (CSR)  -> target = zeros(num_answers); target[ans2label[ans]] = score (gqa_data.py:213-217)
// HBM-bound: one read of the staged batch, one write of the f32 batch.  IEEE f32 division, so boxes are bit-identical to numpy's.
#include <hip/hip_fp16.h>
#include "kernels.h"

__global__ __launch_bounds__(256) void feats_f16_to_f32_kernel(const __half* __restrict__ in, float* __restrict__ out, size_t n8) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const uint4 raw = reinterpret_cast<const uint4*>(in)[i];
        const __half* h = reinterpret_cast<const __half*>(&raw);
        float4 a = make_float4(__half2float(h[0]), __half2float(h[1]), __half2float(h[2]), __half2float(h[3]));
        float4 b = make_float4(__half2float(h[4]), __half2float(h[5]), __half2float(h[6]), __half2float(h[7]));
        reinterpret_cast<float4*>(out)[2 * i] = a;
        reinterpret_cast<float4*>(out)[2 * i + 1] = b;
    }
}

// one thread per box: x1, y1, x2, y2 (pixels) / (w, h, w, h)
__global__ __launch_bounds__(256) void boxes_normalize_kernel(const float* __restrict__ in, const int32_t* __restrict__ img_hw, float* __restrict__ out, int B, int O) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * O) return;
    const int b = i / O;
    const float h = (float)img_hw[2 * b], w = (float)img_hw[2 * b + 1];
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(v.x / w, v.y / h, v.z / w, v.w / h);
}

// one block per sample: clear the row, then scatter its (label, score) pairs in list order (a later duplicate wins, as the
// reference's sequential assignment does)
__global__ __launch_bounds__(256) void targets_build_kernel(const int32_t* __restrict__ offsets, const int32_t* __restrict__ labels, const float* __restrict__ scores,
                                                            float* __restrict__ target, int NA, int ld) {
    const int b = blockIdx.x;
    float* row = target + (size_t)b * ld;
    for (int n = threadIdx.x; n < NA; n += blockDim.x) row[n] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = offsets[b]; k < offsets[b + 1]; ++k) {
            const int l = labels[k];
            if (l >= 0 && l < NA) row[l] = scores[k];          // answers outside ans2label are skipped (gqa_data.py:216)
        }
}

int k_batch_prepare(const void* feats_in, int feats_f16, float* feats_out, const float* boxes_in, const int32_t* img_hw, float* boxes_out,
                    const int32_t* offsets, const int32_t* labels, const float* scores, float* target, int ld_target,
                    int B, int O, int F, int NA, hipStream_t s) {
    RGQA_REQUIRE(B > 0 && O > 0 && F > 0, "batch_prepare: bad shape B=%d O=%d F=%d", B, O, F);
    if (feats_in != nullptr) {
        RGQA_REQUIRE(feats_out != nullptr, "batch_prepare: feats_out is null");
        const size_t n = (size_t)B * O * F;
        if (feats_f16) {
            RGQA_REQUIRE(n % 8 == 0, "batch_prepare: B*O*F must be a multiple of 8 for the f16 store");
            const size_t n8 = n / 8;
            size_t blocks = (n8 + 255) / 256; if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(feats_f16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const __half*>(feats_in), feats_out, n8);
            RGQA_LAUNCH_CHECK("feats_f16_to_f32_kernel");
        } else if (feats_in != feats_out) {
            RGQA_HIP(hipMemcpyAsync(feats_out, feats_in, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    }
    if (boxes_in != nullptr) {
        RGQA_REQUIRE(boxes_out != nullptr && img_hw != nullptr, "batch_prepare: boxes_out / img_hw is null");
        hipLaunchKernelGGL(boxes_normalize_kernel, dim3(cdiv(B * O, 256)), dim3(256), 0, s, boxes_in, img_hw, boxes_out, B, O);
        RGQA_LAUNCH_CHECK("boxes_normalize_kernel");
    }
    if (target != nullptr) {
        RGQA_REQUIRE(offsets != nullptr && NA > 0 && ld_target >= NA, "batch_prepare: target needs offsets and NA <= ld");
        hipLaunchKernelGGL(targets_build_kernel, dim3(B), dim3(256), 0, s, offsets, labels, scores, target, NA, ld_target);
        RGQA_LAUNCH_CHECK("targets_build_kernel");
    }
    return RGQA_OK;
}
