// BUTD path kernel launchers (butd_kernels.hip).
#pragma once
#include "common.h"
// one linear layer for the grouped weight-norm launches (kb_wn_forward_group / kb_wn_backward_group): V [out, in] (+ scalar g and ||V||^2 slot when
// weight-normed), effective copies eff [op, kp] / efft [kp, op] (element type of the engine), optional f32 copy of the effective weight, and for the
// backward pass dW [out, lddw] (the wgrad GEMM's result), dV / dg in the gradient arena.  blk0 / nblk: this layer's blocks in the reduction /
// element-wise launches; tile0 / tiles_x / tiles_y: its 32 x 32 tiles in the effective-weight launch.
struct WnDesc {
    const float* v; const float* g; float* sumsq; void* eff; void* efft; float* eff_f32;
    const float* dw; int lddw; float* dv; float* dgp;
    int nsplit; size_t split_stride;      // dW = sum of nsplit partial results (the long contractions are cut along the rows), split_stride floats apart
    const float* bpart; float* db;        // nsplit > 1: the bias gradient's partial column sums [nsplit][out] and where their sum goes
    int out, in, kp, op;
    int blk0, nblk, tile0, tiles_x, tiles_y;
};
int kb_wn_forward_group(const WnDesc* descs_dev, int nd, int total_blocks, int total_tiles, float* partial, int elem, hipStream_t s);
int kb_wn_backward_group(const WnDesc* descs_dev, int nd, int total_blocks, float* partial, hipStream_t s);
template <typename T> int kb_embed_fwd(const int64_t* toks, const float* table, T* out, int rows, int E, int Ep, hipStream_t s);
template <typename T> int kb_embed_bwd(const int64_t* toks, const T* dx, float* dtable, int rows, int E, int Ep, int pad_idx, hipStream_t s);
template <typename T> int kb_gru_fwd(const T* gi, long ldgi, const T* gh, const T* hprev, T* hnew, T* rs, T* zs, T* ns, T* ghn, int B, int H, hipStream_t s);
template <typename T> int kb_gru_bwd(const T* dh, const T* hprev, const T* rs, const T* zs, const T* ns, const T* ghn, T* dgi, long lddgi, T* dgh, T* dhprev, int B, int H, hipStream_t s);
template <typename T> int kb_concat(const float* feat, const float* pos, T* out, int rows, int F, int Pd, int Dp, hipStream_t s);
template <typename T> int kb_attend_fwd(const T* ip, const T* qp, const float* wlin, const float* blin, const T* imgf, float* att, T* img_enc, int B, int O, int H, int Dp, DropCfg drop, hipStream_t s);
template <typename T> int kb_attend_bwd(const T* dimg, const T* imgf, const float* att, const T* ip, const T* qp, const float* wlin, T* dip, T* dqp, float* dw_part, float* db_part,
                                        int B, int O, int H, int Dp, DropCfg drop, hipStream_t s);
template <typename T> int kb_mul_fwd(const T* a, const T* b, T* out, size_t n, hipStream_t s);
template <typename T> int kb_mul_relu_bwd(const T* dj, const T* a, const T* b, T* da, T* db, size_t n, hipStream_t s);
// effective weight of a (weight-normed when g != null) linear: w [N, ldo] (+ transposed wt [K, ldt] when wt != null)
template <typename T> int kb_wn_eff(const float* v, const float* g, const float* sumsq, T* w, int ldo, T* wt, int ldt, int N, int K, hipStream_t s);
// gradient of V (and g) from the gradient of the effective weight; plain copy when g == null. partial: >= 256 floats
int kb_wn_bwd(const float* dw, int lddw, const float* v, const float* g, const float* sumsq, float* partial, float* dv, float* dg, int N, int K, int accumulate, hipStream_t s);

// the GRU recurrence as one persistent launch per direction (butd_gru.hip): bf16, H = 1024, B <= 256; counters: gru_persist_counter_ints(B, L) ints
// of device memory owned by these launches (the error word behind the counters is zeroed once by the owner)
bool gru_persist_ok(int B, int H);
size_t gru_persist_counter_ints(int B, int L);
int k_gru_fwd_persist(const bf16_t* GI, long ldgi, const bf16_t* W, int ldw, const float* bhh, bf16_t* Hall, bf16_t* Rg, bf16_t* Zg, bf16_t* Ng, bf16_t* GHN,
                      int B, int L, int H, int* counters, hipStream_t s);
int k_gru_bwd_persist(const bf16_t* dH, const bf16_t* Hall, const bf16_t* Rg, const bf16_t* Zg, const bf16_t* Ng, const bf16_t* GHN, bf16_t* dGI, long lddgi, bf16_t* dGH,
                      const bf16_t* WT, int ldwt, int B, int L, int H, int* counters, hipStream_t s);
