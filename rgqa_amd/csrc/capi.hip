// extern "C" surface of librgqa_hip.so (declared in include/rgqa.h).
#include "engine.h"
#include <string.h>
#include <new>

struct rgqa_engine { EngineBase* impl; };
EngineBase* make_butd_engine(const rgqa_config& cfg);
int launch_gemm_nt256_probe(GemmGroup& g, int mt, hipStream_t s);      // gemm_mfma256.hip

#define S(x) reinterpret_cast<hipStream_t>(x)
#define NEED(e) do { if ((e) == nullptr || (e)->impl == nullptr) { rgqa_set_error("null engine handle"); return RGQA_ERR_ARG; } } while (0)

extern "C" {

int rgqa_version(void) { return 100; }

extern int g_rgqa_force_gemm128;
extern int g_rgqa_force_mt;
extern int g_rgqa_wgrad_serial;
extern int g_rgqa_tn_mtw;
extern int g_rgqa_cls_tail;
extern int g_rgqa_attn_pair;
extern int g_rgqa_skip_wgrad;
extern int g_rgqa_wgrad_merge;
extern int g_rgqa_nt_splitk;
extern int g_rgqa_nt_panel;
extern int g_rgqa_wgrad_sets;
extern int g_rgqa_dgrad_nn;
extern int g_rgqa_butd_gru_persist;
extern int g_rgqa_ln_fuse;
extern int g_rgqa_adam_blocks;
extern int g_rgqa_z_in_place;
extern int g_rgqa_nt_stagger;
int rgqa_set_side_stream(int device, void* stream) {
    RGQA_REQUIRE(device >= 0 && stream != nullptr, "set_side_stream: bad argument");
    std::lock_guard<std::mutex> lk(rgqa_side_stream_mutex());
    auto& m = rgqa_side_streams();
    RGQA_REQUIRE(m.find(device) == m.end() || m[device] == reinterpret_cast<hipStream_t>(stream), "set_side_stream: device %d already has a side stream (set it before the first engine is bound)", device);
    m[device] = reinterpret_cast<hipStream_t>(stream);
    return RGQA_OK;
}
// debug / test switches (include/rgqa.h)
int rgqa_debug_set(int key, int value) {
    if (key == 0) { g_rgqa_force_gemm128 = value; return RGQA_OK; }
    if (key == 1) { g_rgqa_force_mt = value; return RGQA_OK; }
    if (key == 2) { g_rgqa_wgrad_serial = value; return RGQA_OK; }
    if (key == 4) { g_rgqa_tn_mtw = value; return RGQA_OK; }
    if (key == 5) { g_rgqa_skip_wgrad = value; return RGQA_OK; }
    if (key == 6) { g_rgqa_wgrad_merge = value; return RGQA_OK; }
    if (key == 7) { g_rgqa_nt_splitk = value; return RGQA_OK; }
    if (key == 8) { g_rgqa_cls_tail = value; return RGQA_OK; }
    if (key == 9) { g_rgqa_nt_panel = value; return RGQA_OK; }
    if (key == 14) { g_rgqa_dgrad_nn = value; return RGQA_OK; }
    if (key == 16) { g_rgqa_attn_pair = value; return RGQA_OK; }
    if (key == 17) { g_rgqa_wgrad_sets = value; return RGQA_OK; }
    if (key == 18) { g_rgqa_butd_gru_persist = value; return RGQA_OK; }
    if (key == 19) { g_rgqa_ln_fuse = value; return RGQA_OK; }
    if (key == 20) { g_rgqa_adam_blocks = value; return RGQA_OK; }
    if (key == 21) { g_rgqa_z_in_place = value; return RGQA_OK; }
    if (key == 22) { g_rgqa_nt_stagger = value; return RGQA_OK; }
    rgqa_set_error("debug_set: unknown key %d", key);
    return RGQA_ERR_ARG;
}

int rgqa_engine_create(const rgqa_config* cfg, rgqa_engine** out) {
    RGQA_REQUIRE(cfg != nullptr && out != nullptr, "engine_create: null argument");
    if (cfg->arch == 1) {
        RGQA_REQUIRE(cfg->hidden > 0 && cfg->hidden % 64 == 0 && cfg->emb_dim > 0 && cfg->vocab_size > 1 && cfg->num_answers > 0 && cfg->feat_dim % 4 == 0 &&
                     cfg->pos_dim >= 1 && cfg->pos_dim <= 4, "engine_create (BUTD): hidden %% 64, emb_dim, vocab_size, num_answers, feat_dim %% 4 required");
        RGQA_REQUIRE(cfg->precision == RGQA_PRECISION_F32 || cfg->precision == RGQA_PRECISION_BF16 || cfg->precision == RGQA_PRECISION_BF16X3,
                     "engine_create (BUTD): precision %d unsupported (f32, bf16, bf16x3)", cfg->precision);
        if (cfg->precision == RGQA_PRECISION_BF16X3) RGQA_REQUIRE(cfg->hidden % 64 == 0, "engine_create (BUTD): bf16x3 precision needs hidden %% 64 == 0");
        RGQA_REQUIRE(cfg->hidden_dropout >= 0.f && cfg->hidden_dropout < 1.f && cfg->attn_dropout >= 0.f && cfg->attn_dropout < 1.f, "engine_create: dropout out of range");
        rgqa_engine* eb = new (std::nothrow) rgqa_engine;
        if (!eb) { rgqa_set_error("engine_create: out of host memory"); return RGQA_ERR_STATE; }
        eb->impl = make_butd_engine(*cfg);
        *out = eb;
        return RGQA_OK;
    }
    RGQA_REQUIRE(cfg->arch == 0 || cfg->arch == 2, "engine_create: unknown arch %d", cfg->arch);
    if (cfg->arch == 2)      // UNITER: one stream of BertLayers over [text ; regions] (uniter/modeling.py:615-635)
        RGQA_REQUIRE(cfg->x_layers == 0 && cfg->r_layers == 0 && cfg->l_layers >= 1 && cfg->type_vocab >= 2 && cfg->pos_dim <= 8,
                     "engine_create (UNITER): layers go in l_layers (x_layers = r_layers = 0), type_vocab >= 2, pos_dim <= 8");
    RGQA_REQUIRE(cfg->hidden > 0 && cfg->heads > 0 && cfg->hidden % cfg->heads == 0,
                 "The hidden size (%d) is not a multiple of the number of attention heads (%d)", cfg->hidden, cfg->heads);  // modeling.py:298-301
    RGQA_REQUIRE(cfg->hidden % 64 == 0 && cfg->hidden <= 1024, "engine_create: hidden (%d) must be a multiple of 64 and <= 1024", cfg->hidden);
    RGQA_REQUIRE(cfg->hidden / cfg->heads <= 64, "engine_create: head size %d > 64 unsupported", cfg->hidden / cfg->heads);
    RGQA_REQUIRE(cfg->inter % 8 == 0 && cfg->feat_dim % 8 == 0, "engine_create: intermediate (%d) and feature (%d) sizes must be multiples of 8", cfg->inter, cfg->feat_dim);
    RGQA_REQUIRE(cfg->pos_dim >= 1 && cfg->pos_dim <= (cfg->arch == 2 ? 8 : 4), "engine_create: pos_dim %d unsupported", cfg->pos_dim);
    RGQA_REQUIRE(cfg->vocab_size > 0 && cfg->max_pos > 0 && cfg->type_vocab > 0 && cfg->num_answers > 0, "engine_create: empty table");
    RGQA_REQUIRE(cfg->l_layers >= 0 && cfg->x_layers >= 0 && cfg->r_layers >= 0, "engine_create: negative layer count");
    RGQA_REQUIRE(cfg->precision == RGQA_PRECISION_F32 || cfg->precision == RGQA_PRECISION_BF16 || cfg->precision == RGQA_PRECISION_BF16X3 || cfg->precision == RGQA_PRECISION_BF16X3_FWD,
                 "engine_create: unknown precision %d", cfg->precision);
    if (cfg->precision == RGQA_PRECISION_BF16X3 || cfg->precision == RGQA_PRECISION_BF16X3_FWD)
        RGQA_REQUIRE(cfg->hidden % 64 == 0 && cfg->hidden / cfg->heads == 64 && cfg->inter % 32 == 0 && cfg->feat_dim % 32 == 0,
                     "engine_create: bf16x3 precision needs head size 64 and hidden / intermediate / feature sizes that are multiples of 32");
    RGQA_REQUIRE(cfg->hidden_dropout >= 0.f && cfg->hidden_dropout < 1.f && cfg->attn_dropout >= 0.f && cfg->attn_dropout < 1.f, "engine_create: dropout out of range");
    rgqa_engine* e = new (std::nothrow) rgqa_engine;
    if (!e) { rgqa_set_error("engine_create: out of host memory"); return RGQA_ERR_STATE; }
    e->impl = make_engine(*cfg);
    *out = e;
    return RGQA_OK;
}
void rgqa_engine_destroy(rgqa_engine* e) { if (e) { delete e->impl; delete e; } }

int rgqa_engine_arena_elems(const rgqa_engine* e, size_t* out) { NEED(e); *out = e->impl->arena_elems; return RGQA_OK; }
int rgqa_engine_num_params(const rgqa_engine* e, int* out) { NEED(e); *out = (int)e->impl->params.size(); return RGQA_OK; }
int rgqa_engine_param_info(const rgqa_engine* e, int index, char* name, size_t name_cap, size_t* offset, int64_t shape[2], int* ndim, int* flags) {
    NEED(e);
    RGQA_REQUIRE(index >= 0 && index < (int)e->impl->params.size(), "param_info: index %d out of range", index);
    const ParamInfo& p = e->impl->params[index];
    RGQA_REQUIRE(name_cap > p.name.size(), "param_info: name buffer too small");
    strcpy(name, p.name.c_str());
    *offset = p.offset; shape[0] = p.shape[0]; shape[1] = p.shape[1]; *ndim = p.ndim;
    *flags = (p.is_linear_weight ? 1 : 0) | (p.dead_in_x_mode ? 2 : 0) | ((!p.is_linear_weight || p.f32_master_read) ? 4 : 0);
    return RGQA_OK;
}
int rgqa_engine_dead_range(const rgqa_engine* e, size_t* b, size_t* en) { NEED(e); *b = e->impl->dead_begin; *en = e->impl->dead_end; return RGQA_OK; }
int rgqa_engine_workspace_bytes(rgqa_engine* e, int B, int T, int O, size_t* out) {
    NEED(e);
    RGQA_REQUIRE(B > 0 && T > 0 && O > 0, "workspace_bytes: bad shape");
    *out = e->impl->workspace_bytes(B, T, O);
    return RGQA_OK;
}
int rgqa_engine_bind(rgqa_engine* e, float* params, float* grads, void* plp, void* plpt, void* ws, size_t ws_bytes, int B, int T, int O) {
    NEED(e);
    return e->impl->bind(params, grads, plp, plpt, ws, ws_bytes, B, T, O);
}
int rgqa_engine_sync_weights(rgqa_engine* e, void* stream) { NEED(e); return e->impl->sync_weights(S(stream)); }
int rgqa_engine_sync_transposed(rgqa_engine* e, void* stream) { NEED(e); return e->impl->sync_transposed(S(stream)); }
int rgqa_engine_forward(rgqa_engine* e, const float* feats, const float* boxes, const int64_t* ids, const int64_t* seg, const int64_t* mask,
                        float* pooled, float* logits, int ld_logits, int train, uint64_t seed, void* stream) {
    NEED(e);
    return e->impl->forward(feats, boxes, ids, seg, mask, pooled, logits, ld_logits, train, seed, S(stream));
}
int rgqa_engine_loss_backward(rgqa_engine* e, const float* target, int ldt, float* loss_out, float grad_scale, int accumulate, void* stream) {
    NEED(e);
    RGQA_REQUIRE(target != nullptr, "loss_backward: null target");
    return e->impl->loss_backward(target, ldt, loss_out, grad_scale, accumulate, S(stream));
}
int rgqa_engine_backward(rgqa_engine* e, const float* dlogits, int ld, int accumulate, void* stream) {
    NEED(e);
    return e->impl->backward(dlogits, ld, accumulate, S(stream));
}
int rgqa_engine_backward_pooled(rgqa_engine* e, const float* dpooled, int ld, int accumulate, void* stream) {
    NEED(e);
    return e->impl->backward_pooled(dpooled, ld, accumulate, S(stream));
}
int rgqa_engine_get_cross_attention(rgqa_engine* e, int layer, int direction, float* out, size_t cap, void* stream) {
    NEED(e);
    RGQA_REQUIRE(out != nullptr, "get_cross_attention: null argument");
    return e->impl->get_cross_attention(layer, direction, out, cap, S(stream));
}
int rgqa_engine_get_activation(rgqa_engine* e, const char* name, float* out, size_t cap, void* stream) {
    NEED(e);
    RGQA_REQUIRE(name && out, "get_activation: null argument");
    return e->impl->get_activation(name, out, cap, S(stream));
}

int rgqa_engine_set_grad_sumsq_slots(rgqa_engine* e, float* slots, int n) {
    NEED(e);
    RGQA_REQUIRE(slots == nullptr || n >= (int)e->impl->grad_segs.size(), "set_grad_sumsq_slots: %d slots for %d gradient segments", n, (int)e->impl->grad_segs.size());
    e->impl->sumsq_slots = slots;
    return RGQA_OK;
}
int rgqa_engine_set_input_grads(rgqa_engine* e, float* dfeats, float* dboxes) {
    NEED(e);
    return e->impl->set_input_grads(dfeats, dboxes);
}

int rgqa_engine_set_lengths(rgqa_engine* e, const int32_t* lengths, int n) {
    NEED(e);
    return e->impl->set_lengths(lengths, n);
}

int rgqa_engine_num_grad_segments(const rgqa_engine* e, int* out) { NEED(e); *out = (int)e->impl->grad_segs.size(); return RGQA_OK; }
int rgqa_engine_grad_segment(const rgqa_engine* e, int k, size_t* begin, size_t* end, int* event) {
    NEED(e);
    RGQA_REQUIRE(k >= 0 && k < (int)e->impl->grad_segs.size(), "grad_segment: index %d out of range", k);
    *begin = e->impl->grad_segs[k].begin; *end = e->impl->grad_segs[k].end; *event = e->impl->grad_segs[k].event;
    return RGQA_OK;
}
int rgqa_engine_wait_grad_event(rgqa_engine* e, int event, void* stream) {
    NEED(e);
    RGQA_REQUIRE(event >= 0 && event < (int)e->impl->seg_events.size(), "wait_grad_event: event %d has not been recorded (run backward first)", event);
    RGQA_HIP(hipStreamWaitEvent(S(stream), e->impl->seg_events[event], 0));
    return RGQA_OK;
}
int rgqa_engine_set_weight_event(rgqa_engine* e, int segment_event, void* hip_event) {
    NEED(e);
    const int n = e->impl->num_weight_segments();
    RGQA_REQUIRE(segment_event >= 0 && segment_event < n, "set_weight_event: segment event %d outside 0..%d (this engine's forward waits for none)", segment_event, n - 1);
    if ((int)e->impl->wready.size() < n) e->impl->wready.resize(n, nullptr);
    e->impl->wready[segment_event] = reinterpret_cast<hipEvent_t>(hip_event);
    return RGQA_OK;
}
int rgqa_engine_num_weight_segments(const rgqa_engine* e, int* out) {
    NEED(e);
    RGQA_REQUIRE(out != nullptr, "num_weight_segments: null argument");
    *out = e->impl->num_weight_segments();
    return RGQA_OK;
}
int rgqa_engine_set_backward_event(rgqa_engine* e, void* hip_event) {
    NEED(e);
    RGQA_REQUIRE(e->impl->num_weight_segments() > 0, "set_backward_event: not supported by this engine");
    e->impl->bwd_wait = reinterpret_cast<hipEvent_t>(hip_event);
    return RGQA_OK;
}
int rgqa_engine_profile(rgqa_engine* e, int enable) { NEED(e); e->impl->profiling = enable != 0; return RGQA_OK; }
int rgqa_engine_profile_read(rgqa_engine* e, double* ms, double* flops, double* bytes, int64_t* launches, int ncat) {
    NEED(e);
    RGQA_REQUIRE(ncat >= PC_COUNT, "profile_read: need room for %d categories", PC_COUNT);
    ProfSummary ps; e->impl->prof_collect(ps);
    for (int i = 0; i < PC_COUNT; ++i) { ms[i] = ps.ms[i]; flops[i] = ps.flops[i]; bytes[i] = ps.bytes[i]; launches[i] = ps.launches[i]; }
    return RGQA_OK;
}
int rgqa_engine_profile_blocks(rgqa_engine* e, double* ms, double* flops, int nblock) {
    NEED(e);
    RGQA_REQUIRE(ms && flops && nblock >= PB_COUNT, "profile_blocks: need room for %d blocks", PB_COUNT);
    for (int i = 0; i < PB_COUNT; ++i) { ms[i] = e->impl->last_blk_ms[i]; flops[i] = e->impl->last_blk_flops[i]; }
    return RGQA_OK;
}
int rgqa_engine_profile_operand_bytes(rgqa_engine* e, double* bytes, int ncat) {
    NEED(e);
    RGQA_REQUIRE(bytes && ncat >= PC_COUNT, "profile_operand_bytes: need room for %d categories", PC_COUNT);
    for (int i = 0; i < PC_COUNT; ++i) bytes[i] = e->impl->last_cat_obytes[i];
    return RGQA_OK;
}
int rgqa_grad_sumsq(const float* grads, size_t n, float* partial_ws, float* sumsq_out, int accumulate, void* stream) {
    RGQA_REQUIRE(grads && partial_ws && sumsq_out, "grad_sumsq: null argument");
    return k_sumsq(grads, n, partial_ws, sumsq_out, accumulate, S(stream));
}
int rgqa_clip_scale(float* grads, size_t n, const float* sumsq, float max_norm, void* stream) {
    RGQA_REQUIRE(grads && sumsq, "clip_scale: null argument");
    return k_clip_scale(grads, n, sumsq, max_norm, S(stream));
}
int rgqa_bertadam_step(float* p, const float* g, float* m, float* v, void* p_lp, int lp_split, size_t n, float lr_t, float b1, float b2, float eps, float wd,
                       const float* sumsq, float max_norm, float grad_prescale, void* stream) {
    RGQA_REQUIRE(p && g && m && v, "bertadam_step: null argument");
    AdamArgs a; a.p = p; a.g = g; a.m = m; a.v = v; a.p_lp = p_lp; a.lp_split = lp_split; a.n = n; a.lr_t = lr_t; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = wd;
    a.sumsq = sumsq; a.max_norm = max_norm; a.grad_prescale = grad_prescale;
    return k_bertadam(a, S(stream));
}
int rgqa_cast_bf16(const float* src, void* dst_bf16, size_t n, void* stream) {
    RGQA_REQUIRE(src && dst_bf16, "cast_bf16: null argument");
    RGQA_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst_bf16 % 8) == 0, "cast_bf16: src must be 16-byte, dst 8-byte aligned");
    return k_cast_bf16(src, dst_bf16, n, S(stream));
}
int rgqa_split_f32(const float* src, void* dst_split, size_t n, void* stream) {
    RGQA_REQUIRE(src && dst_split && (n % 32) == 0 && (((uintptr_t)dst_split) & 127) == 0, "split_f32: n %% 32 and a 128-byte aligned destination required");
    return k_cast_split(src, dst_split, n, S(stream));
}
int rgqa_unsplit_f32(const void* src_split, float* dst, size_t n, void* stream) {
    RGQA_REQUIRE(src_split && dst && (n % 32) == 0 && (((uintptr_t)src_split) & 127) == 0, "unsplit_f32: n %% 32 and a 128-byte aligned source required");
    if (n == 0) return RGQA_OK;
    const size_t rows = (n + 32767) / 32768;           // rows of 32768 elements (whole lines), the last one shorter
    for (size_t r = 0; r < rows; ++r) {
        const size_t o = r * 32768, c = n - o < 32768 ? n - o : 32768;
        if (int rc = k_to_f32<sf32>(reinterpret_cast<const sf32*>(src_split) + o, (int)c, dst + o, (int)c, 1, (int)c, S(stream))) return rc;
    }
    return RGQA_OK;
}
int rgqa_sum_bf16_parts(const void* parts_bf16, size_t part_stride, int nparts, float* dst, size_t n, void* stream) {
    RGQA_REQUIRE(parts_bf16 && dst && nparts >= 1, "sum_bf16_parts: bad argument");
    return k_sum_bf16_parts(parts_bf16, part_stride, nparts, dst, n, S(stream));
}
int rgqa_sum_parts(const void* parts, int parts_f32, size_t part_stride, int nparts, float* dst, size_t n, float* sq_ws, float* sumsq_accum, void* stream) {
    RGQA_REQUIRE(parts && dst && nparts >= 1, "sum_parts: bad argument");
    return k_sum_parts(parts, parts_f32, part_stride, nparts, dst, n, sq_ws, sumsq_accum, S(stream));
}
int rgqa_mixup_gather(float* feats, float* boxes, const int32_t* partner, const uint8_t* take_pos, int B, int O, int F, int mode_v3, void* stream) {
    RGQA_REQUIRE(feats && boxes && partner && take_pos, "mixup_gather: null argument");
    return k_mixup_gather(feats, boxes, partner, take_pos, B, O, F, mode_v3, S(stream));
}
int rgqa_mixup_perturb(float* feats, float* boxes, const int32_t* perm, int B, int O, int F, void* stream) {
    RGQA_REQUIRE(feats && boxes && perm, "mixup_perturb: null argument");
    return k_mixup_perturb(feats, boxes, perm, B, O, F, S(stream));
}
int rgqa_mixup_weighted_sum(float* feats, float* boxes, const int32_t* partner, const float* prop, const float* one_minus_prop, int B, int O, int F, void* stream) {
    RGQA_REQUIRE(feats && boxes && partner && prop && one_minus_prop, "mixup_weighted_sum: null argument");
    return k_mixup_weighted_sum(feats, boxes, partner, prop, one_minus_prop, B, O, F, S(stream));
}
int rgqa_scale_rows(float* target, const float* prop, int B, int NA, int ld, int row0, void* stream) {
    RGQA_REQUIRE(target && prop, "scale_rows: null argument");
    return k_scale_rows(target, prop, B, NA, ld, row0, S(stream));
}

// ---------------------------------------------------------------------------- stand-alone operators
int rgqa_op_linear(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw, int ldc, int epilogue, int dtype, void* stream) {
    GemmGroup g; memset(&g, 0, sizeof g);
    g.count = 1; g.drop = make_drop(0.f, 0, 0);
    GemmProblem& p = g.p[0];
    p.A = A; p.B = W; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = ldc; p.epi = epilogue;
    RGQA_REQUIRE(epilogue >= 0 && epilogue <= 2, "op_linear: epilogue must be 0..2");
    if (dtype == 2) return launch_gemm_nt_x3(g, 0, S(stream));
    return dtype == 1 ? launch_gemm_nt_bf16(g, 0, S(stream)) : launch_gemm_f32(g, 0, 0, S(stream));
}
// GEMM probe: `launches` back-to-back launches of the bf16 NT GEMM C = A W^T on a stamped instantiation of the product kernels (tile height
// 32 * mt: 8 / 7 = the persistent loop, 5 / 2 = the deep ring; gelu != 0: the GELU epilogue with C2 as its second output).
// stamps[8 * b + 0..7] of the LAST launch: (s_memtime, s_memrealtime) at entry and at exit of block b, s_memrealtime at first operands
// landed / end of the first tile's K loop / after its epilogue, tiles walked.
int rgqa_probe_gemm(const void* A, const void* W, void* C, void* C2, int M, int N, int K, int mt, int gelu, int launches, unsigned long long* stamps, void* stream) {
    RGQA_REQUIRE(A && W && C && stamps && launches >= 1, "probe_gemm: null argument");
    for (int i = 0; i < launches; ++i) {
        GemmGroup g; memset(&g, 0, sizeof g);
        g.count = 1; g.drop = make_drop(0.f, 0, 0); g.stamps = stamps;
        GemmProblem& p = g.p[0];
        p.A = A; p.B = W; p.C = C; p.C2 = gelu ? C2 : nullptr; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.epi = gelu ? EPI_GELU : EPI_BIAS;
        if (int r = launch_gemm_nt256_probe(g, mt, S(stream))) return r;
    }
    return RGQA_OK;
}
// bf16 / split f32: every epilogue of the grouped NT GEMM on one problem (kernel parity tests, tools/lab).  epilogue = GemmEpi value.
int rgqa_op_linear_ex(const void* A, const void* W, const float* bias, const void* aux, void* C, void* C2, int M, int N, int K, int lda, int ldw, int ldc,
                      int ldaux, int epilogue, float drop_p, int dtype, void* stream) {
    RGQA_REQUIRE(dtype == 1 || dtype == 2, "op_linear_ex: dtype 1 (bf16) or 2 (split f32)");
    RGQA_REQUIRE(epilogue >= 0 && epilogue <= EPI_DRELU_DROP && epilogue != EPI_ACCUM, "op_linear_ex: bad epilogue %d", epilogue);
    RGQA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "op_linear_ex: dropout out of range");
    GemmGroup g; memset(&g, 0, sizeof g);
    g.count = 1; g.drop = make_drop(drop_p, 0x1234567ull, 0);
    GemmProblem& p = g.p[0];
    p.A = A; p.B = W; p.C = C; p.C2 = C2; p.bias = bias; p.aux = aux; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = ldc; p.ldaux = ldaux;
    p.epi = epilogue; p.drop_site = 17u;
    return dtype == 2 ? launch_gemm_nt_x3(g, 0, S(stream)) : launch_gemm_nt_bf16(g, 0, S(stream));
}
// C[M,N] = A[M,K] B[K,N] (+ epilogue): the dgrad form on the weight as it lies (csrc/gemm_nt256.h NN); bf16, K % 64 == 0; ws: optional split-K scratch
int rgqa_op_linear_kn(const void* A, const void* Bkn, const void* aux, void* C, int M, int N, int K, int lda, int ldb, int ldc, int ldaux, int epilogue,
                      int out_f32, float* ws, size_t ws_floats, void* stream) {
    RGQA_REQUIRE(epilogue == EPI_BIAS || epilogue == EPI_DGELU || epilogue == EPI_ADD || epilogue == EPI_DTANH, "op_linear_kn: epilogue %d has no [K, N]-operand kernel (0, 4, 5, 7)", epilogue);
    RGQA_REQUIRE(K % 64 == 0 && K >= 64, "op_linear_kn: K must be a whole number of 64-row K-steps (got %d)", K);
    GemmGroup g; memset(&g, 0, sizeof g);
    g.count = 1; g.b_kn = 1; g.drop = make_drop(0.f, 0, 0); g.splitk_ws = ws; g.splitk_floats = ws_floats;
    GemmProblem& p = g.p[0];
    p.A = A; p.B = Bkn; p.C = C; p.aux = aux; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux; p.epi = epilogue;
    return launch_gemm_nt_bf16(g, out_f32, S(stream));
}
// the same bf16 problem with split-K scratch (ws_floats floats): problems of M <= 256 rows and K >= 1536 are cut along the contraction
// (csrc/gemm_mfma256.hip); out_f32: C is float (plain-bias epilogue only)
int rgqa_op_linear_splitk(const void* A, const void* W, const float* bias, const void* aux, void* C, void* C2, int M, int N, int K, int lda, int ldw, int ldc,
                          int ldaux, int epilogue, float drop_p, int out_f32, float* ws, size_t ws_floats, void* stream) {
    RGQA_REQUIRE(epilogue >= 0 && epilogue <= EPI_DRELU_DROP && epilogue != EPI_ACCUM, "op_linear_splitk: bad epilogue %d", epilogue);
    RGQA_REQUIRE(drop_p >= 0.f && drop_p < 1.f && ws != nullptr, "op_linear_splitk: bad argument");
    GemmGroup g; memset(&g, 0, sizeof g);
    g.count = 1; g.drop = make_drop(drop_p, 0x1234567ull, 0); g.splitk_ws = ws; g.splitk_floats = ws_floats;
    GemmProblem& p = g.p[0];
    p.A = A; p.B = W; p.C = C; p.C2 = C2; p.bias = bias; p.aux = aux; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = ldc; p.ldaux = ldaux;
    p.epi = epilogue; p.drop_site = 17u;
    return launch_gemm_nt_bf16(g, out_f32, S(stream));
}
int rgqa_op_matmul_tn(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int dtype, void* stream) {
    GemmGroup g; memset(&g, 0, sizeof g);
    g.count = 1; g.drop = make_drop(0.f, 0, 0);
    GemmProblem& p = g.p[0];
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.epi = EPI_BIAS;
    if (dtype == 2) return launch_gemm_tn_x3(g, S(stream));
    return dtype == 1 ? launch_gemm_tn_bf16(g, 1, S(stream)) : launch_gemm_f32(g, 1, 1, S(stream));
}
// grouped wgrad launch as the engine issues it: C_i[M_i,N_i] (+)= A_i[K_i,M_i]^T B_i[K_i,N_i], colsum_i[m] (+)= sum_k A_i[k][m] (or null)
int rgqa_op_matmul_tn_group(int count, const void* const* A, const void* const* B, float* const* C, float* const* colsum, const int* M, const int* N, const int* K,
                            const int* lda, const int* ldb, const int* ldc, int accumulate, int dtype, void* stream) {
    RGQA_REQUIRE(count >= 1 && count <= GEMM_MAX_PROBLEMS && A && B && C && M && N && K && lda && ldb && ldc, "op_matmul_tn_group: bad argument (1..%d problems)", GEMM_MAX_PROBLEMS);
    RGQA_REQUIRE(dtype == 1 || dtype == 2, "op_matmul_tn_group: dtype 1 (bf16) or 2 (split f32)");
    GemmGroup g; memset(&g, 0, sizeof g);
    g.count = count; g.drop = make_drop(0.f, 0, 0);
    for (int i = 0; i < count; ++i) {
        GemmProblem& p = g.p[i];
        p.A = A[i]; p.B = B[i]; p.C = C[i]; p.colsum_out = colsum ? colsum[i] : nullptr; p.M = M[i]; p.N = N[i]; p.K = K[i];
        p.lda = lda[i]; p.ldb = ldb[i]; p.ldc = ldc[i]; p.epi = accumulate ? EPI_ACCUM : EPI_BIAS;
    }
    return dtype == 2 ? launch_gemm_tn_x3(g, S(stream)) : launch_gemm_tn_bf16(g, 1, S(stream));
}
int rgqa_op_layernorm(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int N, float eps, int dtype, void* stream) {
    if (dtype == 2) return k_ln_fwd<sf32>((const sf32*)x, N, gamma, beta, (sf32*)y, N, mean, rstd, M, N, eps, S(stream));
    if (dtype == 1) return k_ln_fwd<bf16_t>((const bf16_t*)x, N, gamma, beta, (bf16_t*)y, N, mean, rstd, M, N, eps, S(stream));
    return k_ln_fwd<float>((const float*)x, N, gamma, beta, (float*)y, N, mean, rstd, M, N, eps, S(stream));
}
int rgqa_op_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                          float* ws, int M, int N, int dtype, void* stream) {
    const DropCfg nd = make_drop(0.f, 0, 0);
    if (dtype == 2) return k_ln_bwd<sf32>((const sf32*)dy, N, (const sf32*)x, N, gamma, mean, rstd, (sf32*)dx, nullptr, N, ws, dgamma, dbeta, nullptr, 0, M, N, nd, nd, 1.f, S(stream));
    if (dtype == 1) return k_ln_bwd<bf16_t>((const bf16_t*)dy, N, (const bf16_t*)x, N, gamma, mean, rstd, (bf16_t*)dx, nullptr, N, ws, dgamma, dbeta, nullptr, 0, M, N, nd, nd, 1.f, S(stream));
    return k_ln_bwd<float>((const float*)dy, N, (const float*)x, N, gamma, mean, rstd, (float*)dx, nullptr, N, ws, dgamma, dbeta, nullptr, 0, M, N, nd, nd, 1.f, S(stream));
}
static void fill_attn(AttnArgs& a, const void* qkv, const float* mask, int B, int nh, int L, int dh, size_t esz) {
    memset(&a, 0, sizeof a);
    const int H = nh * dh;
    a.q = qkv; a.k = (const char*)qkv + (size_t)H * esz; a.v = (const char*)qkv + (size_t)2 * H * esz;
    a.ldq = a.ldk = a.ldv = 3 * H; a.mask = mask; a.B = B; a.nh = nh; a.Lq = a.Lk = L; a.dh = dh;
    a.scale = 1.0f / sqrtf((float)dh); a.drop = make_drop(0.f, 0, 0);
}
int rgqa_op_attention(const void* qkv, const float* mask, void* out, float* lse, int B, int nh, int L, int dh, int dtype, int impl, void* stream) {
    AttnArgs a; fill_attn(a, qkv, mask, B, nh, L, dh, dtype == 1 ? 2 : 4);
    a.out = out; a.ldo = nh * dh; a.lse = lse;
    if (dtype == 2) return impl == 1 ? k_attn_fwd_x3(a, S(stream)) : k_attn_fwd_ref<sf32>(a, S(stream));
    if (dtype == 1) return impl == 1 ? k_attn_fwd_mfma(a, S(stream)) : k_attn_fwd_ref<bf16_t>(a, S(stream));
    return k_attn_fwd_ref<float>(a, S(stream));
}
int rgqa_op_attention_bwd(const void* qkv, const float* mask, const float* lse, const void* dout, void* dqkv, int B, int nh, int L, int dh, int dtype, int impl, void* stream) {
    const size_t esz = dtype == 1 ? 2 : 4;
    AttnArgs a; fill_attn(a, qkv, mask, B, nh, L, dh, esz);
    const int H = nh * dh;
    a.lse = const_cast<float*>(lse); a.dout = dout; a.lddo = H;
    a.dq = dqkv; a.dk = (char*)dqkv + (size_t)H * esz; a.dv = (char*)dqkv + (size_t)2 * H * esz; a.lddq = a.lddk = a.lddv = 3 * H;
    if (dtype == 2) return impl == 1 ? k_attn_bwd_x3(a, S(stream)) : k_attn_bwd_ref<sf32>(a, S(stream));
    if (dtype == 1) return impl == 1 ? k_attn_bwd_mfma(a, S(stream)) : k_attn_bwd_ref<bf16_t>(a, S(stream));
    return k_attn_bwd_ref<float>(a, S(stream));
}
int rgqa_batch_prepare(const void* feats_in, int feats_f16, float* feats_out, const float* boxes_in, const int32_t* img_hw, float* boxes_out,
                       const int32_t* offsets, const int32_t* labels, const float* scores, float* target, int ld_target,
                       int B, int O, int F, int NA, void* stream) {
    return k_batch_prepare(feats_in, feats_f16, feats_out, boxes_in, img_hw, boxes_out, offsets, labels, scores, target, ld_target, B, O, F, NA, S(stream));
}
int rgqa_score_rows(const float* logits, int ld, int B, int NA, float temperature, int k, float* max_score, int64_t* label, float* energy,
                    float* topk_val, int64_t* topk_idx, float* topk_energy, void* stream) {
    return k_score_rows(logits, ld, B, NA, temperature, k, max_score, label, energy, topk_val, topk_idx, topk_energy, S(stream));
}
int rgqa_op_bce(const float* logits, const float* target, float* loss, float* dlogits, int B, int NA, void* stream) {
    return k_bce_fwd_bwd(logits, NA, target, NA, loss, dlogits, NA, B, NA, NA, 1.0f, S(stream));
}

}  // extern "C"
