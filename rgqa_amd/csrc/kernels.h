// Internal kernel launchers (host side). Every function enqueues work on `s` and returns an RGQA_* code;
// none allocates, frees or synchronises.  T = float (exact-f32 mode), bf16_t (throughput mode) or sf32 (split f32: bf16x3 precision).
#pragma once
#include "common.h"
#include "gemm.h"

// ---- norm.hip
#define FIN_MAXQ 24
struct FinOut { float* p[FIN_MAXQ]; int stride[FIN_MAXQ]; int qsrc[FIN_MAXQ], b0[FIN_MAXQ], b1[FIN_MAXQ]; };   // qsrc / b0 / b1: filled by k_colsum_finalize, or by the caller of _ranges
int k_colsum_finalize(const float* part, int nblk, int nq, int N, const FinOut& fo, int accumulate, hipStream_t s);
int k_colsum_finalize_ranges(const float* part, int nq_part, int nout, int N, const FinOut& fo, int accumulate, hipStream_t s);
// Deferred finalisation of LayerNorm-backward column sums (dgamma / dbeta / the producing GEMM's bias gradient): the backward kernels of one
// encoder layer park their per-block partials behind each other in `base` (3 quantities per block, width N) and append their outputs here;
// ONE colsum_finalize launch folds them all, on whichever stream the owner chooses (fin_flush) - nothing downstream on the main stream reads them.
struct FinDefer {
    float* base = nullptr; int cap_blocks = 0, N = 0;       // partials region: cap_blocks * 3 * N floats
    int blk = 0, nout = 0; FinOut fo = {};
    void begin(float* b, int cap, int n) { base = b; cap_blocks = cap; N = n; blk = 0; nout = 0; fo = FinOut{}; }
    bool room(int nblk, int outs, int n) const { return base != nullptr && n == N && blk + nblk <= cap_blocks && nout + outs <= FIN_MAXQ; }
    float* take(int nblk) { float* q = base + (size_t)blk * 3 * N; blk += nblk; return q; }
    void add(float* out, int qsrc, int b0, int b1) { if (out) { fo.p[nout] = out; fo.stride[nout] = 1; fo.qsrc[nout] = qsrc; fo.b0[nout] = b0; fo.b1[nout] = b1; ++nout; } }
};
int fin_flush(FinDefer& d, int accumulate, hipStream_t s);
// y_b (split-f32 rows only): optional bf16 image of y at the same element offsets (the bf16x3_fwd precision's backward reads it)
template <typename T>
int k_ln_fwd(const T* x, int ldx, const float* gamma, const float* beta, T* y, int ldy, float* mean, float* rstd, int M, int N, float eps, hipStream_t s, bf16_t* y_b = nullptr);
// rows [0, split) use (gamma, beta), rows [split, M) use (gamma2, beta2): two modules' LayerNorms over adjacent row ranges in one launch
template <typename T>
int k_ln_fwd2(const T* x, int ldx, const float* gamma, const float* beta, const float* gamma2, const float* beta2, int split, T* y, int ldy, float* mean, float* rstd,
              int M, int N, float eps, hipStream_t s, bf16_t* y_b = nullptr);
int ln_bwd_blocks(int M, int N);
// part: workspace of ln_bwd_blocks(M,N)*3*N floats (or null: no column sums). dzd may be null.
template <typename T>
int k_ln_bwd(const T* dy, int lddy, const T* z, int ldz, const float* gamma, const float* mean, const float* rstd, T* dz, T* dzd, int lddz,
             float* part, float* dgamma, float* dbeta, float* dbias, int accumulate, int M, int N, DropCfg drop, DropCfg drop_in, float dy_scale, hipStream_t s,
             FinDefer* defer = nullptr, int z_split = 0 /* T = bf16_t, N = 768: z points at a split-f32 tensor (ldz in its elements) whose hi parts are read in place */);
// two adjacent row segments of the same buffers (language | vision), each with its own module parameters, gradients and dropout site
template <typename T>
int k_ln_bwd2(const T* dy, int lddy, const T* z, int ldz, const float* mean, const float* rstd, T* dz, T* dzd, int lddz, float* part, int N, int accumulate,
              int M0, const float* gamma0, float* dgamma0, float* dbeta0, float* dbias0, DropCfg drop0,
              int M1, const float* gamma1, float* dgamma1, float* dbeta1, float* dbias1, DropCfg drop1, hipStream_t s, FinDefer* defer = nullptr, int z_split = 0);
// out[n] (+)= sum_m x[m][n]; part: workspace of 256*N floats
template <typename T>
int k_colsum(const T* x, int ldx, float* part, float* out, int accumulate, int M, int N, hipStream_t s);

// ---- attn.hip
struct AttnArgs {
    const void* q; const void* k; const void* v;  // element (row 0, head 0, dim 0) of each operand
    int ldq, ldk, ldv;                             // row strides in elements
    void* out; int ldo;                            // fwd: context [B*Lq, nh*dh]
    void* out_b;                                   // split-f32 forward only: optional bf16 image of `out` (same ldo), or null
    const float* mask;                             // additive key mask [B, Lk] (0 / -10000) or null
    float* lse;                                    // [B, nh, Lq] log-sum-exp of the masked, scaled scores
    // backward only
    const void* dout; int lddo;
    void* dq; void* dk; void* dv; int lddq, lddk, lddv;
    int B, nh, Lq, Lk, dh;                         // Lq / Lk: per-sample row counts (the maxima when cu_q / cu_k are set)
    // unpadded language rows (engine varlen mode): sample b owns rows cu[b] .. cu[b+1]-1 of the packed operand; null = b*L
    const int* cu_q; const int* cu_k;
    float scale;
    DropCfg drop;                                  // attention-probability dropout (reference modeling.py:341)
    uint32_t drop_site;
};
// per-sample row window of the (possibly packed) q and k/v operands: declares q0, k0 (first row) and Lq, Lk (valid rows)
#define ATTN_SAMPLE_ROWS(a, b)                                                                   \
    size_t q0 = (size_t)(b) * (a).Lq, k0 = (size_t)(b) * (a).Lk;                                 \
    int Lq = (a).Lq, Lk = (a).Lk;                                                                \
    if ((a).cu_q) { const int c0_ = (a).cu_q[(b)]; q0 = (size_t)c0_; Lq = (a).cu_q[(b) + 1] - c0_; } \
    if ((a).cu_k) { const int c0_ = (a).cu_k[(b)]; k0 = (size_t)c0_; Lk = (a).cu_k[(b) + 1] - c0_; }
template <typename T> int k_attn_fwd_ref(const AttnArgs& a, hipStream_t s);
template <typename T> int k_attn_bwd_ref(const AttnArgs& a, hipStream_t s);
template <typename T> int k_attn_probs(const AttnArgs& a, float* out /* [B, nh, Lq, Lk] */, hipStream_t s);
int k_attn_fwd_mfma(const AttnArgs& a, hipStream_t s);   // bf16 only
int k_attn_bwd_mfma(const AttnArgs& a, hipStream_t s);   // bf16 only
int k_attn_fwd_x3(const AttnArgs& a, hipStream_t s);     // split f32 (attn_x3.hip): every product as hi*hi + hi*lo + lo*hi on the bf16 matrix pipe
int k_attn_bwd_x3(const AttnArgs& a, hipStream_t s);
// the two attention problems of a stage (language | vision, or the two cross directions) in ONE launch: 1 = launched, 0 = shapes not covered
// (launch them separately), < 0 = error
int k_attn_fwd_mfma_pair(const AttnArgs& a0, const AttnArgs& a1, hipStream_t s);
int k_attn_bwd_mfma_pair(const AttnArgs& a0, const AttnArgs& a1, hipStream_t s);
int k_attn_fwd_x3_pair(const AttnArgs& a0, const AttnArgs& a1, hipStream_t s);
int k_attn_bwd_x3_pair(const AttnArgs& a0, const AttnArgs& a1, hipStream_t s);

// ---- embed.hip
// lang[b*T+t] = dropout(LN(word[ids] + pos[t] + type[seg]))      (reference BertEmbeddings, modeling.py:278-292)
template <typename T>
int k_embed_fwd(const int64_t* ids, const int64_t* seg, const int* row_src, const int* row_dst /* output row map or null */, int rows, const float* word, const float* pos, const float* type, const float* gamma, const float* beta,
                T* out, int ldo, T* zsave, float* mean, float* rstd, int B, int Tn, int H, int vocab, int type_vocab, float eps, DropCfg drop, hipStream_t s);
// de [B*T, H] f32 (gradient w.r.t. the pre-LN embedding sum) scattered into the three tables; row 0 of each
// table gets no gradient (padding_idx=0, modeling.py:269-271)
template <typename T>
int k_embed_scatter(const T* de, const int64_t* ids, const int64_t* seg, const int* row_src, int rows, float* dword, float* dpos, float* dtype, int B, int Tn, int H, int type_vocab,
                    int pad0_all, int accumulate, int* keys, float* scratch, size_t scratch_floats, hipStream_t s);
// the embedding tables' dense gradients without float atomics: every table row summed in an order the batch alone fixes (embed.hip).  kw / kp / kt: word / position /
// token-type key per row (kp, kt null: a lone word table); iscratch: rows + 132 ints; fscratch: (np + nt + 64) * ceil(rows / 256) * H floats
template <typename T>
int k_embed_table_grads(const T* de, int ldde, const int* kw, const int* kp, const int* kt, int rows, float* dword, float* dpos, float* dtype, int H, int np, int nt,
                        int pad_key, int pad0_all, int accumulate, int* iscratch, float* fscratch, size_t fscratch_floats, hipStream_t s);
int k_embed_keys(const int64_t* ids, int rows, int* keys, hipStream_t s);
// additive key mask (1 - m) * -10000 from the 0/1 int64 attention mask (modeling.py:857-865)
int k_make_mask(const int64_t* input_mask, float* out, int n, hipStream_t s);

// ---- uniter.hip (UNITER embedding front-end, reference uniter/modeling.py:560-635)
int k_uniter_dst(const int* tcu, const int* jcu, int B, int O, int* text_dst, int* img_dst, hipStream_t s);
int k_uniter_mask(const int64_t* input_mask, float* out, int B, int T, int O, hipStream_t s);
template <typename T> int k_pos_proj(const float* pos, int pd, const float* Wp, const float* bp, T* out, int ldo, int M, int H, hipStream_t s);
template <typename T> int k_pos_wgrad(const T* dzp, int ld, const float* pos, int pd, float* part, float* dWp, int accumulate, int M, int H, hipStream_t s);
template <typename T>
int k_sum3_ln_fwd(const T* a, const T* b, int ld, const float* trow, const float* gamma, const float* beta, const int* dst, T* out, int ldo, T* xsave, float* mean, float* rstd,
                  int M, int H, float eps, DropCfg drop, hipStream_t s);

// ---- visn.hip
// out = dropout((LN(zf) + LN(boxes Wb^T + bb)) / 2)            (reference VisualFeatEncoder, modeling.py:507-517)
template <typename T>
int k_visn_combine_fwd(const T* zf, int ldz, const float* boxes, const float* Wb, const float* bb, const float* g1, const float* b1, const float* g2,
                       const float* b2, T* out, int ldo, float* stats /*[M,4]: mean1,rstd1,mean2,rstd2*/, int M, int H, int pos_dim, float eps, DropCfg drop, hipStream_t s);
// backward of the above: dzf (gradient entering visn_fc's output) + all small parameter gradients
template <typename T>
int k_visn_combine_bwd(const T* dout, int lddo, const T* zf, int ldz, const float* boxes, const float* Wb, const float* bb, const float* g1, const float* g2,
                       const float* stats, T* dzf, int lddz, float* part, float* dg1, float* db1, float* dg2, float* db2, float* dbias_fc,
                       float* dWb, float* dbb, int accumulate, int M, int H, int pos_dim, DropCfg drop, float* dboxes /* [M,pos_dim] or null */, hipStream_t s);

// ---- loss.hip
// loss = mean_b sum_n BCEWithLogits(z, t)  (= BCEWithLogitsLoss() * NA, gqa_conf.py:197-198); dz = (sigmoid(z)-t)/B * grad_scale
int k_bce_fwd_bwd(const float* logits, int ldl, const float* target, int ldt, float* loss_out, float* dlogits, int lddl, int B, int NA, int NAp, float grad_scale, hipStream_t s,
                  float* row_scratch = nullptr /* B floats: the loss is then folded in a fixed order (no float atomics) */);

// ---- optim.hip
int k_sumsq(const float* g, size_t n, float* partial /* >= 1025 floats, any content ([1024] = ticket word, zeroed on the stream by the call) */, float* out_sumsq, int accumulate_into_out, hipStream_t s);
// several owned ranges in one launch (each with partials / ticket / output slot of its own, as k_sumsq_owned; never accumulating)
#define SUMSQ_GROUP_MAX 8
struct SumsqRange { const float* g; size_t n; float* partial; float* out; int nblk; };
struct SumsqGroup { int count; SumsqRange r[SUMSQ_GROUP_MAX]; };
int k_sumsq_owned_group(SumsqGroup& gr, hipStream_t s);
int k_sumsq_owned(const float* g, size_t n, float* partial /* >= 1025 floats touched by nothing else, [1024] zeroed once by the owner */, float* out_sumsq, int accumulate_into_out, hipStream_t s);
struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    void* p_lp;            // optional low-precision operand copy at the same element offsets (null in f32 mode)
    int lp_split;          // p_lp holds split-f32 pairs (bf16x3 precision) instead of bf16
    size_t n;
    float lr_t, b1, b2, eps, wd;
    const float* sumsq;    // device scalar: sum of squared grads (for clip); null = no clipping
    float max_norm;
    float grad_prescale;   // multiplies g before everything (1/world_size for DP sum-reduce), 1 otherwise
};
int k_bertadam(const AdamArgs& a, hipStream_t s);
int k_clip_scale(float* g, size_t n, const float* sumsq, float max_norm, hipStream_t s);
// dst_t[k][n] = (bf16) src[n][k] for each listed [N,K] matrix; desc on device: {src_off, dst_off, N, K, tile_start}
#define TRANSPOSE_TILE 64   // tile_start counts cdiv(ld_dst, TRANSPOSE_TILE) * cdiv(K, TRANSPOSE_TILE) tiles per matrix
struct TransDesc { long src_off, dst_off; int N, K, ld_dst, tile_start; };
int k_cast_transpose(const void* src, int src_is_bf16, void* dst, int dst_split, const TransDesc* desc_dev, int ndesc, int total_tiles, hipStream_t s);
int k_cast_bf16(const float* src, void* dst, size_t n, hipStream_t s);
int k_cast_split(const float* src, void* dst_split, size_t n, hipStream_t s, bf16_t* img = nullptr /* also dst's bf16 image (= its hi parts) at the same element offsets */);
int k_sum_bf16_parts(const void* parts, size_t stride, int nparts, float* dst, size_t n, hipStream_t s);
int k_sum_parts(const void* parts, int parts_f32, size_t stride, int nparts, float* dst, size_t n, float* sq_ws /* >= 1025 floats or null */, float* sq_out /* += sum(dst^2), or null */, hipStream_t s);

// ---- misc.hip
// RoI-mixup gather (gqa_mixup_vis.py:134-181): rows [B,2B) of feats/boxes built from partner + positive rows
// device-side batch preparation from the binary feature store (loader.hip)
int k_batch_prepare(const void* feats_in, int feats_f16, float* feats_out, const float* boxes_in, const int32_t* img_hw, float* boxes_out,
                    const int32_t* offsets, const int32_t* labels, const float* scores, float* target, int ld_target,
                    int B, int O, int F, int NA, hipStream_t s);
// test-time scoring of logits rows (score.hip)
int k_score_rows(const float* logits, int ld, int B, int NA, float temperature, int k, float* max_score, int64_t* label, float* energy,
                 float* topk_val, int64_t* topk_idx, float* topk_energy, hipStream_t s);
int k_mixup_gather(float* feats, float* boxes, const int32_t* partner, const uint8_t* take_pos /*[B,O]*/, int B, int O, int F, int mode_v3, hipStream_t s);
int k_mixup_perturb(float* feats, float* boxes, const int32_t* perm, int B, int O, int F, hipStream_t s);
int k_mixup_weighted_sum(float* feats, float* boxes, const int32_t* partner, const float* p, const float* q, int B, int O, int F, hipStream_t s);
int k_scale_rows(float* target, const float* prop, int B, int NA, int ld, int row0, hipStream_t s);
template <typename T> int k_fill_rows(T* dst, int ld, const T* src, int lds, int rows, int cols, hipStream_t s);
template <typename T> int k_cast_pad(const float* src, int lds, T* dst, int ldd, int rows, int cols, float scale, hipStream_t s);
template <typename T> int k_to_f32(const T* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s);
int k_sum_partials(const float* part, int S, size_t n, float* out, int accumulate, hipStream_t s);
// varlen language rows: lengths (host, passed as kernel arguments) -> lens/cu[B+1]/row_src[sum] on the device
int k_set_lengths(const int* lens_host, int B, int Tn, int* lens_dev, int* cu_dev, int* row_src_dev, hipStream_t s);
// row(b) = cu ? cu[b] : b * stride_rows
template <typename T> int k_gather_rows(const T* src, int lds, const int* cu, int stride_rows, T* dst, int ldd, int rows, int cols, hipStream_t s);
template <typename T> int k_scatter_rows(const T* src, int lds, T* dst, int ldd, const int* cu, int stride_rows, int rows, int cols, hipStream_t s);
// dst[r][c] = bf16(src[r][c]) = the hi part of every split-f32 element: the bf16 image of a tensor no kernel writes twice (embeddings, gathered rows)
int k_sf_image(const sf32* src, int lds, bf16_t* dst, int ldd, int rows, int cols, hipStream_t s);
template <typename T> int k_dgelu_mul(const T* dy, const T* pre, T* out, size_t n, hipStream_t s);
template <typename T> int k_dtanh_mul(const T* dy, const T* y, T* out, size_t n, hipStream_t s);
