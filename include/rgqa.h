/* rgqa.h — C ABI of librgqa_hip.so (MI355X / gfx950).
 *
 * The reference (chihhuiho/RGQA) has no FFI or operator registry: its boundary for this path is a pair of
 * Python classes.  This library is what the Python mirror of those classes
 * (rgqa_amd/lxrt/entry.py, rgqa_amd/tasks/gqa_model.py) binds through ctypes.  Each entry point names the
 * reference code it replaces (paths relative to the reference's src/).
 *
 * Conventions: plain pointers and sizes only (no torch types); every function returns 0 on success or a
 * negative RGQA_ERR_* code and never throws; rgqa_last_error_string() describes the last failure on the
 * calling thread; nothing here allocates or frees caller memory — parameter arenas and the workspace are
 * caller-owned device buffers; all work is enqueued on the given hipStream_t (passed as void*) and is
 * stream-ordered.  Device pointers unless stated otherwise.
 */
#ifndef RGQA_H
#define RGQA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGQA_PRECISION_F32 0    /* exact-f32 FMA arithmetic on the vector ALU: the on-device numerical reference (slow) */
#define RGQA_PRECISION_BF16 1   /* bf16 MFMA operands, f32 accumulate / statistics: BASELINE config 3's mode; its logits are OUTSIDE
                                   the 1e-3 bound (bf16 rounding through 19 blocks, ~5e-2) */
#define RGQA_PRECISION_BF16X3 2 /* split-f32 ("bf16x3"): every activation / weight operand a bf16 pair hi + lo (16-17 significant bits),
                                   products on the bf16 matrix pipe as hi*hi + hi*lo + lo*hi with f32 accumulation: the fast path whose
                                   logits stay within 1e-3 of the reference CPU path (lxrt/modeling.py:309-346 is f32 end to end); arch 0 (LXMERT)
                                   and 2 (UNITER), head size 64, hidden / inter / feat_dim multiples of 32 */
#define RGQA_PRECISION_BF16X3_FWD 3 /* the forward pass of RGQA_PRECISION_BF16X3 (the SAME kernels: logits within 1e-3 of the reference CPU path)
                                   with the backward pass of RGQA_PRECISION_BF16 (BASELINE config 3 prescribes a bf16 backward): the forward
                                   kernels leave a bf16 image of every tensor the backward reads beside the split-f32 one.  Gradients carry bf16
                                   rounding (held to the bf16 mode's loss / gradient-norm / sampled-gradient gates), weights, moments and the
                                   optimizer stay f32.  Operand copies: params_lp split f32 (4 B / element), params_lp_t bf16 (2 B / element). */

typedef struct rgqa_config {
    int32_t vocab_size, hidden, heads, inter, max_pos, type_vocab; /* BertConfig, lxrt/modeling.py:172-258 */
    int32_t l_layers, x_layers, r_layers;                         /* VISUAL_CONFIG, lxrt/entry.py:74-77   */
    int32_t feat_dim, pos_dim;                                    /* VisualConfig, lxrt/modeling.py:141-169 */
    int32_t num_answers;                                          /* tasks/gqa_model.py:15,26 */
    int32_t precision;                                            /* RGQA_PRECISION_* */
    float ln_eps;                                                 /* 1e-12 everywhere in the reference */
    float hidden_dropout, attn_dropout;                           /* 0.1 / 0.1, lxrt/modeling.py:182-183 */
    int32_t arch;     /* 0 = LXMERT-GQA (default); 1 = BUTD-GQA (butd/butd.py:108-221): hidden = 1024, vocab_size = ntoken+1,
                         hidden_dropout = answer dropout 0.5, attn_dropout = attention dropout 0.2; layer counts / heads / inter unused;
                         2 = UNITER-GQA (uniter/modeling.py:560-655, uniter/uniter.py:15-44): l_layers BertLayers over one sequence
                         [text ; regions] per sample (x_layers = r_layers = 0), pos_dim = 7, state_dict keys `encoder.model.uniter.*`;
                         `boxes` carries the 7-d position features, set_lengths takes the TEXT token counts */
    int32_t emb_dim;  /* BUTD word-embedding size (300) */
} rgqa_config;

typedef struct rgqa_engine rgqa_engine;

const char* rgqa_last_error_string(void);
int rgqa_version(void);
/* test switches: key 0: 1 forces the 128x128 register-staged GEMM kernels everywhere; key 1: forces the NT tile height (16-row
 * m-tiles per wave: 2, 4..8; 0 = cost model); key 2: 1 runs the deferred weight-gradient launches on the caller's stream instead of
 * the side stream; key 4: forces the wgrad (TN) tile height: 4 = 128 rows / 3-slot ring, 8 = 256 rows / 2 slots (one tile per block);
 * key 5 (measurement only): 1 skips the deferred weight-gradient launches; key 6: periods of backward whose weight-gradient problems go into
 * one launch (1..4; 0 = default); key 7: 0 = the [CLS]-row GEMMs (K >= 1536) run whole instead of split along K; key 9: NT tile numbering: 0 = row-major, -1 = panels of N-tiles sized to the L2 (default), n = panel width n; key 8: 0 computes
 * the last language FFN on every row (as the reference does), 1 on the [CLS] rows only (default), -1 = environment RGQA_CLS_TAIL;
 * key 14: 1 = the bf16 engine's dgrad GEMMs read the weights as they lie ([K, N] operand form; two transposed copies are kept: -400 MB, same
 * step time), 0 (default) = every dgrad on a transposed bf16 copy; takes effect at the next rgqa_engine_sync_weights / optimizer step;
 * key 16: 0 launches the two attention problems of a stage separately, 1 as one launch (default); key 17: gradient-buffer sets planned by
 * the NEXT rgqa_engine_bind (2 x key 6 .. 8; 0 = that minimum); key 18: 0 = the BUTD engine's GRU recurrence as one GEMM + one gate kernel per
 * token from the host, 1 = one persistent launch per direction (bf16, hidden 1024, B <= 256): 64-sample row groups x 16-unit slices, W_hh slice
 * in LDS, 4 waves, 2 = the same with 8 waves, 3 (default) = 32-sample row groups x 32-unit slices, weights in registers, 8 waves;
 * key 19: 1 = the bf16 engine's LayerNorms behind the attention-output / FFN-output projections are done inside the projection's launch by the
 * workgroup that finishes a row block last, 0 (default: same step time) = separate LayerNorm launches; bit-identical results either way;
 * key 20: cap on the grid of the (grid-stride) BertAdam kernel, default 2048 workgroups (with the update beside the forward pass a smaller grid
 * only delays the weights the forward waits for: 2048 10.98, 512 11.15, 128 11.61 ms per step);
 * key 21: bf16x3_fwd precision, 1 (default) = the LayerNorm backward reads the hi parts of the split-f32 pre-LayerNorm sums in place, 0 = a bf16 image of them is
 * stored by the projections' epilogues (round 5); bit-identical gradients;
 * key 22: persistent NT GEMM launches: the blocks that walk one tile fewer than the busiest ones start late by value / 16 x K-steps x ~2 us (inside their slack), so that
 * their K loops run beside the other blocks' tile stores; 0 = every block starts at once (rounds 1-5); default 8; same results. */
int rgqa_debug_set(int key, int value);
/* The library runs the deferred weight-gradient GEMMs on ONE side stream per device, shared by every engine of the process.  By default it makes
 * that stream itself when the first engine is bound; a caller that knows better hands one in BEFORE that (the Python binding does: HIP maps streams
 * onto a handful of hardware queues, and a side stream that shares a queue with the launch stream serialises with it - rgqa_amd/streams.py picks
 * streams that demonstrably run beside the caller's).  The stream must outlive every engine. */
int rgqa_set_side_stream(int device, void* stream);

/* ---- host text path: replaces the per-batch Python loop convert_sents_to_features (lxrt/entry.py:36-71) over
 * BertTokenizer.tokenize (lxrt/tokenization.py:174-348) for pure-ASCII sentences. vocab_path: one wordpiece per line
 * (tokenization.py:48-60). encode() writes, per sentence i, ids[i*T .. i*T+T) = [CLS] pieces.. [SEP] 0.., mask likewise 1../0..,
 * lengths[i] = number of real tokens (what rgqa_engine_set_lengths takes); a sentence containing a byte >= 0x80 gets
 * needs_python[i] = 1 and an all-zero row: Unicode normalisation stays with the Python implementation of the same rules.
 * All buffers are host memory. */
typedef struct rgqa_tokenizer rgqa_tokenizer;
int rgqa_tokenizer_create(const char* vocab_path, int do_lower_case, rgqa_tokenizer** out);
void rgqa_tokenizer_destroy(rgqa_tokenizer* t);
int rgqa_tokenizer_vocab_size(const rgqa_tokenizer* t, int64_t* out);
int rgqa_tokenizer_encode(const rgqa_tokenizer* t, const char* const* sents, int n, int max_seq_length, int64_t* ids, int64_t* mask,
                          int32_t* lengths, uint8_t* needs_python);

/* ---- engine: replaces GQAModel.__init__/forward (tasks/gqa_model.py:14-43), LXRTEncoder.forward after
 * tokenisation (lxrt/entry.py:113-120) and everything below it in lxrt/modeling.py. */
int rgqa_engine_create(const rgqa_config* cfg, rgqa_engine** out);
void rgqa_engine_destroy(rgqa_engine* e);
/* flat-arena layout: number of f32 elements, number of tensors, and per-tensor (state_dict key, offset, shape) */
int rgqa_engine_arena_elems(const rgqa_engine* e, size_t* out);
int rgqa_engine_num_params(const rgqa_engine* e, int* out);
int rgqa_engine_param_info(const rgqa_engine* e, int index, char* name, size_t name_cap, size_t* offset,
                           int64_t shape[2], int* ndim, int* flags /* bit0 linear weight (has low-precision operand copies), bit1 dead in mode 'x',
                                                                       bit2 the forward reads this tensor from the f32 master arena in every precision
                                                                       (biases, LayerNorm, embedding tables, the K = 4 box projection) */);
/* element range [begin,end) of the parameters that never receive gradients in mode 'x' */
int rgqa_engine_dead_range(const rgqa_engine* e, size_t* begin, size_t* end);
int rgqa_engine_workspace_bytes(rgqa_engine* e, int B, int T, int O, size_t* out);
/* params / grads: f32 arenas of arena_elems; params_lp / params_lp_t: the operand copies of the linear weights and their
 * transposes at the same element offsets - bf16 arenas of arena_elems in bf16 precision, 4-byte-per-element split-f32 arenas
 * (128-byte aligned) in bf16x3 precision, may be null in f32 precision; workspace: ws_bytes of scratch.  Must be called again
 * when B, T or O change. */
int rgqa_engine_bind(rgqa_engine* e, float* params, float* grads, void* params_lp, void* params_lp_t,
                     void* workspace, size_t ws_bytes, int B, int T, int O);
/* refresh the low-precision weight copies from the f32 master arena (after load_state_dict / external updates) */
int rgqa_engine_sync_weights(rgqa_engine* e, void* stream);
/* same, but only the transposed copies: for use after rgqa_bertadam_step was given p_lp and already wrote the direct copy */
int rgqa_engine_sync_transposed(rgqa_engine* e, void* stream);
/* forward: feats [B,O,feat_dim] f32, boxes [B,O,pos_dim] f32, ids/seg/mask [B,T] i64 (seg may be null = zeros)
 * -> pooled [B,hidden] f32 (may be null), logits [B,num_answers] f32 with row stride ld_logits. train != 0
 * applies dropout (counter-based, keyed by seed) and keeps what backward needs. */
int rgqa_engine_forward(rgqa_engine* e, const float* feats, const float* boxes, const int64_t* input_ids,
                        const int64_t* segment_ids, const int64_t* input_mask, float* pooled, float* logits,
                        int ld_logits, int train, uint64_t seed, void* stream);
/* BCE-with-logits x NA loss (tasks/gqa_conf.py:197-198) + full backward (loss.backward(), :200) into the grad arena.
 * grad_scale multiplies dL/dlogits (1 for a plain step). accumulate = 0 overwrites the gradient arena. */
int rgqa_engine_loss_backward(rgqa_engine* e, const float* target, int ld_target, float* loss_out, float grad_scale,
                              int accumulate, void* stream);
/* backward from a caller-supplied dL/dlogits [B,num_answers] f32 (autograd integration) */
int rgqa_engine_backward(rgqa_engine* e, const float* dlogits, int ld, int accumulate, void* stream);
/* backward from dL/dpooled [B,hidden] f32: LXRTEncoder used under a caller-owned head (tasks/vqa_model.py, nlvr2_model.py) */
int rgqa_engine_backward_pooled(rgqa_engine* e, const float* dpooled, int ld, int accumulate, void* stream);
/* debug / parity: copy a saved activation ("embed_lang", "embed_visn", "l3", "r1", "x2_lang", "x2_visn", "pooled") as f32 */
int rgqa_engine_get_activation(rgqa_engine* e, const char* name, float* out, size_t cap_elems, void* stream);
/* Cross-attention probabilities of cross-modality layer `layer` after a forward pass (reference lxrt_vis/modeling.py:337,
 * 347-348, 458-462: `output_attention=True`): direction 0 = l2v, language queries over vision keys, out [B, heads, T, O];
 * direction 1 = v2l, vision queries over language keys, out [B, heads, O, T]. f32, softmax(QK^T/sqrt(d) + mask) before
 * dropout (what eval mode returns). Padded language positions come out as 0 (a padded key's probability is exactly 0 in
 * the reference as well; padded query rows are not computed when language rows are packed). The vision-query direction of
 * the LAST layer feeds nothing in mode 'x' and the forward pass skips it; its queries / keys are projected on demand here. */
int rgqa_engine_get_cross_attention(rgqa_engine* e, int layer, int direction, float* out, size_t cap_elems,
                                    void* stream);
/* Unpadded language rows for the FOLLOWING forward passes. lengths (HOST array, n = B of the bound shape) holds each
 * sample's real token count ([CLS] .. [SEP]; what sum(input_mask[b]) is for the prefix masks convert_sents_to_features
 * builds, lxrt/entry.py:37-79). The reference computes all max_seq_length positions and masks the padding with -10000
 * (entry.py:119, modeling.py:336): padded positions then carry probability exactly 0 as keys and the pooler reads token 0,
 * so they reach neither the logits nor any gradient. With lengths set the engine packs only the real rows (GEMM /
 * LayerNorm rows B*T -> sum(lengths); attention windows per sample) - same logits and gradients, less work; input_mask is
 * then not read and language activations returned by get_activation are the packed rows. NULL / n = 0 restores the
 * padded layout (the default after bind). The array is consumed before the call returns. */
int rgqa_engine_set_lengths(rgqa_engine* e, const int32_t* lengths, int n);
/* Input gradients for the FOLLOWING backward calls (the reference's ODIN scorer differentiates w.r.t. the RoI features and boxes,
 * tasks/gqa_odin.py:97-121): dfeats [B*O, feat_dim] f32, dboxes [B*O, pos_dim] f32 device buffers, either may be NULL (not computed,
 * the default). */
int rgqa_engine_set_input_grads(rgqa_engine* e, float* dfeats, float* dboxes);
/* Per-segment sum of squared gradients for clip_grad_norm_ (tasks/gqa_conf.py:201): slots (device, n >= rgqa_engine_num_grad_segments
 * floats, or null to switch off) receives, during every following backward call, sum(g^2) of gradient segment k as soon as that segment
 * is final - on the stream that finished it, beside the rest of backward - so the optimizer can add n numbers instead of re-reading
 * the whole gradient arena. Valid for the gradients as backward leaves them (single GPU; after a data-parallel all-reduce the norm
 * must be taken from the reduced arena with rgqa_grad_sumsq). */
int rgqa_engine_set_grad_sumsq_slots(rgqa_engine* e, float* slots, int n);

/* data-parallel overlap: the gradient arena becomes final range by range while backward runs (head first, embeddings
 * last). grad_segment k = element range [begin,end) + the id of the event recorded on the backward stream once that
 * range is final; wait_grad_event makes `stream` wait for it (hipStreamWaitEvent), so an all-reduce of the range can be
 * enqueued on a side stream before backward has finished on the GPU. Dead parameters are in no segment. */
int rgqa_engine_num_grad_segments(const rgqa_engine* e, int* out);
int rgqa_engine_grad_segment(const rgqa_engine* e, int k, size_t* begin, size_t* end, int* event);
int rgqa_engine_wait_grad_event(rgqa_engine* e, int event, void* stream);
/* the opposite direction (the sharded data-parallel exchange all-gathers the updated weights chunk by chunk on a side stream WHILE the next forward
 * pass already runs; replaces the per-step weight re-broadcast of nn.DataParallel, lxrt/entry.py:102-103): the first launch of the next forward pass
 * that reads the weights of gradient segment `segment_event` (ids as in rgqa_engine_grad_segment) waits for hip_event (a hipEvent_t recorded by the
 * caller once those weights - operand copy and f32 masters alike - are in place); the next backward pass waits for the event given to
 * set_backward_event before its first launch (the transposed dgrad operand copies).  One-shot: consumed by the pass that waits; NULL clears.  The
 * event must stay alive until that pass has been enqueued. */
int rgqa_engine_set_weight_event(rgqa_engine* e, int segment_event, void* hip_event);
int rgqa_engine_set_backward_event(rgqa_engine* e, void* hip_event);
/* how many segment events the engine's forward pass honours (0: this engine waits for none - the BUTD engine re-derives every effective weight from
 * the f32 masters at the start of each pass -, so its caller must have the weights in place on the pass's stream before it calls forward) */
int rgqa_engine_num_weight_segments(const rgqa_engine* e, int* out);

/* measurement: time every GEMM / attention launch with HIP events on the launch stream. profile_read synchronises
 * on the recorded events; categories: 0 gemm NT forward, 1 gemm TN (wgrad), 2 attention fwd, 3 attention bwd,
 * 4 layernorm, 5 other, 6 gemm NT dgrad (with 0 until round 5). flops / bytes are algorithmic (2*M*N*K - one per f32 product whatever
 * the number of bf16 MFMA products behind it; operand + result bytes in the launch's own element type). ncat >= 7. */
int rgqa_engine_profile(rgqa_engine* e, int enable);
int rgqa_engine_profile_read(rgqa_engine* e, double* ms, double* flops, double* bytes, int64_t* launches, int ncat);
/* the same records of the last profile_read, split by model block: 0 input embeddings (text + visual), 1 language / vision
 * single-modality layers, 2 cross-modality layers (LXRTXLayer, lxrt/modeling.py:439-488: the block the north-star roofline target
 * names), 3 pooler + answer head + loss; ms = sum of kernel durations, flops = GEMM + attention FLOPs */
int rgqa_engine_profile_blocks(rgqa_engine* e, double* ms, double* flops, int nblock);
/* per category of the last profile_read: the bytes of the GEMM operands alone (A + B + C), i.e. `bytes` without the operands of the fused
 * epilogues (residual / gelu' read, second output written): bench.py reports its traffic ratio against both denominators */
int rgqa_engine_profile_operand_bytes(rgqa_engine* e, double* bytes, int ncat);

/* GEMM probe (roofline.peak_sustained in bench.py; tools/nt_stamps.py): `launches` back-to-back launches of the bf16 NT GEMM
 * C[M,N] = A[M,K] W[N,K]^T (K % 64 == 0, N % 8 == 0; dense row-major bf16; gelu != 0: GELU epilogue, C2 = its second output) on separately
 * instantiated, stamped copies of the product kernels at tile height 32 * mt (8, 7: the persistent loop; 5, 2: the deep ring; 0 = 8).
 * stamps (device, >= 8 * blocks uint64; blocks = min(tiles, CUs) for mt 8 / 7, tiles otherwise) receives, from the LAST launch, 8 words per
 * block: [0] shader cycles and [1] 100-MHz ticks at entry, [2] / [3] at exit ((c_out - c_in) / (t_out - t_in) x 100 MHz = the shader clock
 * held under the dense MFMA loop), [4] ticks when the block's first operands have landed, [5] at the end of its first tile's K loop,
 * [6] after that tile's epilogue (stores issued), [7] tiles walked.  No product kernel executes a stamp. */
int rgqa_probe_gemm(const void* A, const void* W, void* C, void* C2, int M, int N, int K, int mt, int gelu, int launches,
                    unsigned long long* stamps, void* stream);

/* ---- optimizer: replaces nn.utils.clip_grad_norm_(params, max_norm) (tasks/gqa_conf.py:201) followed by
 * BertAdam.step (lxrt/optimization.py:101-180) over arena ranges. */
int rgqa_grad_sumsq(const float* grads, size_t n, float* partial_ws /* >= 1025 f32 of scratch, any content ([1024] is a ticket word the call zeroes on the stream) */, float* sumsq_out,
                    int accumulate, void* stream);
/* the in-place half of clip_grad_norm_ for callers that clip and step in two calls (the drop-in BertAdam): grads *= max_norm /
 * (sqrt(*sumsq) + 1e-6) if that is < 1; no memory traffic otherwise */
int rgqa_clip_scale(float* grads, size_t n, const float* sumsq /* device scalar */, float max_norm, void* stream);
int rgqa_bertadam_step(float* p, const float* g, float* m, float* v, void* p_lp /* operand copy or null */,
                       int lp_split /* 0: p_lp is bf16, 1: split f32 (bf16x3 precision) */, size_t n,
                       float lr_t, float b1, float b2, float eps, float weight_decay,
                       const float* sumsq /* device scalar or null */, float max_norm, float grad_prescale, void* stream);

/* ---- data-parallel gradient exchange helpers (rgqa_amd/parallel.py; replaces what nn.DataParallel's gather/reduce does at
 * lxrt/entry.py:102-103).  rgqa_cast_bf16: dst[i] = bf16(src[i]) (the gradient payload that goes on the wire).
 * rgqa_sum_bf16_parts: dst[i] = sum over r < nparts of f32(parts[r * part_stride + i]), r ascending: the f32 accumulation,
 * at the rank that owns the range, of the bf16 shards it received from every rank (part_stride % 8 == 0). */
int rgqa_cast_bf16(const float* src, void* dst_bf16, size_t n, void* stream);
/* split-f32 storage (bf16x3 precision; layout in rgqa_amd/csrc/common.h): dst slot i = (hi, lo) of src[i]; dst 128-byte aligned, 4 bytes
 * per element; and back: dst[i] = hi + lo.  n % 32 == 0.  Used by the kernel parity tests to build / read operands. */
int rgqa_split_f32(const float* src, void* dst_split, size_t n, void* stream);
int rgqa_unsplit_f32(const void* src_split, float* dst, size_t n, void* stream);
int rgqa_sum_bf16_parts(const void* parts_bf16, size_t part_stride, int nparts, float* dst, size_t n, void* stream);
/* The general form: parts are bf16 (parts_f32 = 0: the payload of bf16 / bf16x3_fwd engines) or f32 (1: f32 / bf16x3 engines, whose gradients
 * are exact beyond bf16); with sq_ws (>= 1025 floats of scratch, any content) and sumsq_accum (device scalar) the kernel also ADDS
 * sum(dst^2) to *sumsq_accum - the owner's share of clip_grad_norm_'s norm (tasks/gqa_conf.py:201) without a second pass. */
int rgqa_sum_parts(const void* parts, int parts_f32, size_t part_stride, int nparts, float* dst, size_t n, float* sq_ws, float* sumsq_accum,
                   void* stream);

/* ---- peer-to-peer exchange over hipIpc buffers (rgqa_amd/csrc/peer.hip): the hand-written fallback for RCCL's all-to-all / all-gather in the gradient
 * exchange that replaces nn.DataParallel's reduce / broadcast (lxrt/entry.py:102-103), for a node whose library collectives do not drive all seven
 * xGMI links of a GPU (bench.py's `dp_wire` probe).  One process per GPU; every rank owns ONE staging buffer - library-owned fine-grained device
 * memory, the only allocation this library makes besides its streams - exported as an IPC handle and mapped by every other rank.  Data moves by PULL:
 *   rgqa_peer_pull   dst + r * dst_stride_bytes <- rank r's staging buffer [src_off_bytes, src_off_bytes + bytes), for every rank r (own buffer
 *                    included), ONE launch whose workgroups are dealt over the peers: a rank reads from all of its peers at once.
 * A reduce-scatter is: stage the payload, barrier, pull part `rank` of every rank, rgqa_sum_parts; an all-gather: stage the owned range, barrier,
 * pull.  The barrier between a rank's staging kernel and its peers' pulls (and between the pulls and the next overwrite) is the CALLER's and is
 * stream-ordered (a tiny collective on the exchange's stream; rgqa_amd/parallel.py PeerShardedExchange): nothing here spins on the device.
 * create: world <= 16, stage_bytes of staging memory on the current device; export writes RGQA_PEER_HANDLE_BYTES bytes; connect takes the world
 * handles in rank order (the own slot is ignored) and maps the peers' buffers; offsets / sizes of a pull are multiples of 16 bytes. */
#define RGQA_PEER_HANDLE_BYTES 64
typedef struct rgqa_peer_comm rgqa_peer_comm;
int rgqa_peer_comm_create(int rank, int world, size_t stage_bytes, rgqa_peer_comm** out);
int rgqa_peer_comm_stage(const rgqa_peer_comm* c, void** ptr, size_t* bytes);
int rgqa_peer_comm_export(rgqa_peer_comm* c, void* handle);
int rgqa_peer_comm_connect(rgqa_peer_comm* c, const void* handles);
int rgqa_peer_comm_is_fine_grained(const rgqa_peer_comm* c);
void rgqa_peer_comm_destroy(rgqa_peer_comm* c);
int rgqa_peer_pull(rgqa_peer_comm* c, size_t src_off_bytes, size_t bytes, void* dst, size_t dst_stride_bytes, void* stream);

/* ---- batch construction: replaces the host loop of RoI-mixup (tasks/gqa_mixup_vis.py:134-181).
 * feats [2B,O,F] / boxes [2B,O,4] with rows [0,B) filled; partner [B] i32; take_pos [B,O] u8 (1 = RoI taken from
 * the positive sample). Writes rows [B,2B). */
int rgqa_mixup_gather(float* feats, float* boxes, const int32_t* partner, const uint8_t* take_pos, int B, int O,
                      int F, int mode_v3, void* stream);
/* 'perturb' (gqa_mixup_vis.py:124-133): rows [B,2B) = the features again; boxes[B+j][o] = boxes[j][perm[o]], perm [O] i32 = the
 * batch's one torch.randperm draw.  (targets of the second half are zero: the caller's buffer.) */
int rgqa_mixup_perturb(float* feats, float* boxes, const int32_t* perm, int B, int O, int F, void* stream);
/* 'weighted_sum_v1/v2' (gqa_mixup_vis.py:217-244): feats[B+j] = feats[j] * prop[j] + feats[partner[j]] * one_minus_prop[j], each product
 * rounded to f32 before the sum (bit-identical to the reference's torch expression); one_minus_prop[j] = (float)(1.0 - prop_double);
 * boxes[B+j] = boxes[j]. */
int rgqa_mixup_weighted_sum(float* feats, float* boxes, const int32_t* partner, const float* prop, const float* one_minus_prop,
                            int B, int O, int F, void* stream);
/* target[B+j,:] = target[j,:] * prop[j]   (mixup_v1 / v3 / weighted_sum_v1 soft targets, gqa_mixup_vis.py:170-171, 233-234) */
int rgqa_scale_rows(float* target, const float* prop, int B, int NA, int ld, int row0, void* stream);

/* ---- input path: a staged batch from the binary feature store -> engine inputs, on the device ----------
 * Replaces, per batch, what GQATorchDataset.__getitem__ does per sample on the host (tasks/gqa_data.py:173-238):
 *   feats_in [B,O,F] f16 (feats_f16 = 1) or f32 -> feats_out [B,O,F] f32                (the store keeps utils.py:16-54's features)
 *   boxes_in [B,O,4] pixels, img_hw [B,2] int32 (img_h, img_w) -> boxes_out: x / img_w, y / img_h   (gqa_data.py:197-200)
 *   labels in CSR form (offsets [B+1], labels = ans2label[ans] or -1, scores) -> target [B,NA] f32, zeros elsewhere
 *                                                                                        (gqa_data.py:213-217; also :224-228, :235-239)
 * Any of the three parts is skipped when its input pointer (feats_in / boxes_in / target) is null. Device pointers. */
int rgqa_batch_prepare(const void* feats_in, int feats_f16, float* feats_out, const float* boxes_in, const int32_t* img_hw,
                       float* boxes_out, const int32_t* offsets, const int32_t* labels, const float* scores, float* target,
                       int ld_target, int B, int O, int F, int NA, void* stream);

/* Host helper of the same path: dst[i] = src[rows[i]] for n rows of row_bytes each, copied by `threads` host threads (the
 * batch gather out of the mmap'ed store into pinned memory; replaces the per-sample `.copy()` of gqa_data.py:189-190). */
int rgqa_host_gather_rows(const void* src, size_t row_bytes, size_t n_src_rows, const int64_t* rows, int n, void* dst,
                          int threads);

/* ---- test-time scoring of the answer logits -------------------------------------------------
 * One fused pass per row of logits [B, NA] (f32, row stride ld) for what the reference's RVQA test scripts compute:
 *   max_score[B], label[B] = torch.sigmoid(logit / temperature).max(1)     tasks/gqa_conf.py:344, gqa_energy.py:184,204,
 *                                                                            gqa_odin.py:130-131 (temperature), gqa_dropout.py:109
 *   energy[B]              = torch.log(1 + torch.exp(logit)).sum(1)         tasks/gqa_energy.py:135,185
 *   topk_val/topk_idx[B,k] = logit.topk(k)  (descending, ties by index)     tasks/gqa_energy.py:205, gqa_check_topk_preds.py:189
 *   topk_energy[B]         = torch.log(1 + torch.exp(topk values)).sum(1)   tasks/gqa_energy.py:206
 * Any output pointer may be null; k = 0 skips the top-k part. All pointers are device pointers. */
int rgqa_score_rows(const float* logits, int ld, int B, int NA, float temperature, int k, float* max_score,
                    int64_t* label, float* energy, float* topk_val, int64_t* topk_idx, float* topk_energy,
                    void* stream);

/* ---- stand-alone operators (unit parity tests; the engine calls the same kernels internally) ------------- */
/* dtype of the stand-alone operators: 0 f32, 1 bf16, 2 split f32 (every activation / weight operand in the split layout, ld % 32 == 0).
 * C[M,N] = A[M,K] W[N,K]^T + bias, epilogue 0 none / 1 gelu / 2 tanh (A, W, C all dtype) */
int rgqa_op_linear(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw,
                   int ldc, int epilogue, int dtype, void* stream);
/* bf16 (dtype 1) or split f32 (dtype 2), any epilogue of the grouped NT GEMM (rgqa_amd/csrc/gemm.h GemmEpi: 0 bias, 1 gelu (+ C2 = gelu'), 2 tanh,
 * 3 dropout(x)+aux, 4 x*aux, 5 x+aux, 7 x*(1-aux^2), 8 relu, 9 dropout(relu), 10 relu/dropout gradient); aux / C2 may be
 * NULL when the epilogue does not use them.  Kernel parity tests and tools/lab only. */
int rgqa_op_linear_ex(const void* A, const void* W, const float* bias, const void* aux, void* C, void* C2, int M, int N,
                      int K, int lda, int ldw, int ldc, int ldaux, int epilogue, float drop_p, int dtype, void* stream);
/* C[M,N] = A[M,K] B[K,N] (+ epilogue 0 bias-less plain, 4 x*aux, 5 x+aux, 7 x*(1-aux^2)) with the B operand stored [K, N] row-major: the form the bf16
 * engine's dgrad GEMMs take on the weight as it lies ([out, in]: no transposed copy).  bf16, K % 64 == 0; out_f32 != 0: C is float (epilogue 0);
 * ws (optional, >= 12 * min(M, 256) * N floats): split-K scratch as in rgqa_op_linear_splitk.  Kernel parity tests. */
int rgqa_op_linear_kn(const void* A, const void* Bkn, const void* aux, void* C, int M, int N, int K, int lda, int ldb, int ldc, int ldaux,
                      int epilogue, int out_f32, float* ws, size_t ws_floats, void* stream);
/* The same bf16 problem with split-K scratch: problems of K >= 1536 are cut into <= 12 slices along the contraction (the count depends on K
 * alone), run as one grouped launch into f32 partial tiles in ws (>= 12 * min(M, 256) * N floats) and folded in slice order with the epilogue
 * applied once; more than 256 rows run as row groups of 256, so a row's arithmetic does not depend on M.
 * out_f32 != 0: C is float (epilogue 0 only). */
int rgqa_op_linear_splitk(const void* A, const void* W, const float* bias, const void* aux, void* C, void* C2, int M, int N,
                          int K, int lda, int ldw, int ldc, int ldaux, int epilogue, float drop_p, int out_f32, float* ws,
                          size_t ws_floats, void* stream);
/* C[M,N] f32 = A[K,M]^T B[K,N]   (wgrad form); dtype of A and B */
int rgqa_op_matmul_tn(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                      int dtype, void* stream);
/* One grouped wgrad launch as the engine issues it (<= 32 problems): C_i[M_i,N_i] (+)= A_i[K_i,M_i]^T B_i[K_i,N_i] and, where colsum_i is not
 * NULL, colsum_i[m] (+)= sum_k A_i[k][m] (the bias gradient); dtype 1 bf16 / 2 split f32.  Kernel parity tests and tools/wgrad_lab.py. */
int rgqa_op_matmul_tn_group(int count, const void* const* A, const void* const* B, float* const* C, float* const* colsum,
                            const int* M, const int* N, const int* K, const int* lda, const int* ldb, const int* ldc,
                            int accumulate, int dtype, void* stream);
int rgqa_op_layernorm(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int M, int N, float eps, int dtype, void* stream);
int rgqa_op_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                          void* dx, float* dgamma, float* dbeta, float* ws /* 512*3*N f32 */, int M, int N, int dtype,
                          void* stream);
/* fused attention core on a packed QKV buffer [B*L, 3*nh*dh] (self-attention), impl 0 generic / 1 MFMA (bf16 and split f32) */
int rgqa_op_attention(const void* qkv, const float* mask /* [B,L] additive or null */, void* out, float* lse, int B,
                      int nh, int L, int dh, int dtype, int impl, void* stream);
int rgqa_op_attention_bwd(const void* qkv, const float* mask, const float* lse, const void* dout, void* dqkv, int B,
                          int nh, int L, int dh, int dtype, int impl, void* stream);
int rgqa_op_bce(const float* logits, const float* target, float* loss, float* dlogits, int B, int NA, void* stream);

#ifdef __cplusplus
}
#endif
#endif
