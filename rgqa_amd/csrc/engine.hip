// Engine implementation: parameter table, workspace plan, forward and backward launch sequences.
#include "engine.h"
#include <map>
#include <mutex>

int g_rgqa_cls_tail = -1;   // rgqa_debug_set key 8: 1 / 0 = last language FFN on the [CLS] rows only / on every row; -1 = env RGQA_CLS_TAIL (default on)
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

static inline size_t rup(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ============================================================================ parameter table
struct ModelParams {
    size_t word, pos, type; LNp emb_ln;
    Lin visn_fc; LNp visn_ln; Lin box_fc; LNp box_ln;
    LNp img_ln;      // UNITER only: UniterImageEmbeddings.LayerNorm
    std::vector<AttP> l_att, r_att; std::vector<FfnP> l_ffn, r_ffn;
    std::vector<AttP> x_cross, x_latt, x_vatt; std::vector<FfnP> x_lffn, x_vffn;
    Lin pooler, head0; LNp head_ln; Lin head3;
};

struct TableBuilder {
    std::vector<ParamInfo>& out;
    size_t cur = 0;
    bool dead = false;
    explicit TableBuilder(std::vector<ParamInfo>& o) : out(o) {}
    size_t add(const std::string& name, long d0, long d1, int ndim, size_t reserve, int is_w) {
        cur = rup(cur, 64);
        ParamInfo p; p.name = name; p.offset = cur; p.ndim = ndim; p.shape[0] = d0; p.shape[1] = d1;
        p.is_linear_weight = is_w; p.dead_in_x_mode = dead ? 1 : 0;
        out.push_back(p);
        size_t n = (size_t)d0 * (ndim == 2 ? (size_t)d1 : 1);
        size_t off = cur;
        cur += n > reserve ? n : reserve;
        return off;
    }
    void read_as_f32(size_t off) { for (auto& p : out) if (p.offset == off) p.f32_master_read = 1; }
    Lin lin(const std::string& name, int o, int i) {
        Lin l; l.out = o; l.in = i; l.ldt = (int)rup(o, 64);
        l.w = add(name + ".weight", o, i, 2, (size_t)i * l.ldt, 1);
        l.b = add(name + ".bias", o, 0, 1, rup(o, 64), 0);
        return l;
    }
    LNp ln(const std::string& name, int n) {
        LNp l; l.n = n;
        l.w = add(name + ".weight", n, 0, 1, 0, 0);
        l.b = add(name + ".bias", n, 0, 1, 0, 0);
        return l;
    }
    // query/key/value stay separate state_dict tensors but sit back-to-back: one fused [3H,H] projection
    AttP att(const std::string& self_name, const std::string& out_name, int H) {
        AttP a;
        a.qkv.out = 3 * H; a.qkv.in = H; a.qkv.ldt = 3 * H;
        a.qkv.w = add(self_name + ".query.weight", H, H, 2, 0, 2);
        add(self_name + ".key.weight", H, H, 2, 0, 2);
        add(self_name + ".value.weight", H, H, 2, 0, 2);
        a.qkv.b = add(self_name + ".query.bias", H, 0, 1, 0, 0);
        add(self_name + ".key.bias", H, 0, 1, 0, 0);
        add(self_name + ".value.bias", H, 0, 1, 0, 0);
        a.o = lin(out_name + ".dense", H, H);
        a.ln = ln(out_name + ".LayerNorm", H);
        return a;
    }
    FfnP ffn(const std::string& inter, const std::string& outp, int H, int I) {
        FfnP f;
        f.up = lin(inter + ".dense", I, H);
        f.down = lin(outp + ".dense", H, I);
        f.ln = ln(outp + ".LayerNorm", H);
        return f;
    }
};

static void build_params(const rgqa_config& c, std::vector<ParamInfo>& tab, ModelParams& mp, size_t& total, size_t& dead_b, size_t& dead_e) {
    TableBuilder tb(tab);
    const int H = c.hidden, I = c.inter;
    if (c.arch == 2) {
        // GQAUNITER (uniter/uniter.py:15-44): encoder = UniterEncoder, .model = UniterFeatureExtraction, .uniter = UniterModel
        // (uniter/modeling.py:615-655); 12 BertLayers named encoder.layer.N like BERT's
        const std::string pre = "encoder.model.uniter.";
        mp.word = tb.add(pre + "embeddings.word_embeddings.weight", c.vocab_size, H, 2, 0, 0);
        mp.pos = tb.add(pre + "embeddings.position_embeddings.weight", c.max_pos, H, 2, 0, 0);
        mp.type = tb.add(pre + "embeddings.token_type_embeddings.weight", c.type_vocab, H, 2, 0, 0);
        mp.emb_ln = tb.ln(pre + "embeddings.LayerNorm", H);
        mp.visn_fc = tb.lin(pre + "img_embeddings.img_linear", H, c.feat_dim);
        mp.visn_ln = tb.ln(pre + "img_embeddings.img_layer_norm", H);
        mp.box_ln = tb.ln(pre + "img_embeddings.pos_layer_norm", H);
        mp.box_fc = tb.lin(pre + "img_embeddings.pos_linear", H, c.pos_dim); tb.read_as_f32(mp.box_fc.w);
        mp.img_ln = tb.ln(pre + "img_embeddings.LayerNorm", H);
        char b2[64];
        for (int i = 0; i < c.l_layers; ++i) {
            snprintf(b2, sizeof b2, "encoder.layer.%d", i);
            std::string n = pre + b2;
            mp.l_att.push_back(tb.att(n + ".attention.self", n + ".attention.output", H));
            mp.l_ffn.push_back(tb.ffn(n + ".intermediate", n + ".output", H, I));
        }
        dead_b = dead_e = 0;
        mp.pooler = tb.lin(pre + "pooler.dense", H, H);
        mp.head0 = tb.lin("logit_fc.0", 2 * H, H);
        mp.head_ln = tb.ln("logit_fc.2", 2 * H);
        mp.head3 = tb.lin("logit_fc.3", c.num_answers, 2 * H);
        total = rup(tb.cur, 64);
        return;
    }
    const std::string pre = "lxrt_encoder.model.bert.";
    mp.word = tb.add(pre + "embeddings.word_embeddings.weight", c.vocab_size, H, 2, 0, 0);
    mp.pos = tb.add(pre + "embeddings.position_embeddings.weight", c.max_pos, H, 2, 0, 0);
    mp.type = tb.add(pre + "embeddings.token_type_embeddings.weight", c.type_vocab, H, 2, 0, 0);
    mp.emb_ln = tb.ln(pre + "embeddings.LayerNorm", H);
    const std::string enc = pre + "encoder.";
    mp.visn_fc = tb.lin(enc + "visn_fc.visn_fc", H, c.feat_dim);
    mp.visn_ln = tb.ln(enc + "visn_fc.visn_layer_norm", H);
    mp.box_fc = tb.lin(enc + "visn_fc.box_fc", H, c.pos_dim); tb.read_as_f32(mp.box_fc.w);
    mp.box_ln = tb.ln(enc + "visn_fc.box_layer_norm", H);
    char buf[64];
    for (int i = 0; i < c.l_layers; ++i) {
        snprintf(buf, sizeof buf, "layer.%d", i);
        std::string n = enc + buf;
        mp.l_att.push_back(tb.att(n + ".attention.self", n + ".attention.output", H));
        mp.l_ffn.push_back(tb.ffn(n + ".intermediate", n + ".output", H, I));
    }
    for (int i = 0; i < c.r_layers; ++i) {
        snprintf(buf, sizeof buf, "r_layers.%d", i);
        std::string n = enc + buf;
        mp.r_att.push_back(tb.att(n + ".attention.self", n + ".attention.output", H));
        mp.r_ffn.push_back(tb.ffn(n + ".intermediate", n + ".output", H, I));
    }
    dead_b = dead_e = 0;
    for (int i = 0; i < c.x_layers; ++i) {
        snprintf(buf, sizeof buf, "x_layers.%d", i);
        std::string n = enc + buf;
        mp.x_cross.push_back(tb.att(n + ".visual_attention.att", n + ".visual_attention.output", H));
        mp.x_latt.push_back(tb.att(n + ".lang_self_att.self", n + ".lang_self_att.output", H));
        mp.x_lffn.push_back(tb.ffn(n + ".lang_inter", n + ".lang_output", H, I));
        const bool last = (i == c.x_layers - 1);
        if (last) { tb.cur = rup(tb.cur, 64); dead_b = tb.cur; tb.dead = true; }
        mp.x_vatt.push_back(tb.att(n + ".visn_self_att.self", n + ".visn_self_att.output", H));
        mp.x_vffn.push_back(tb.ffn(n + ".visn_inter", n + ".visn_output", H, I));
        if (last) { tb.cur = rup(tb.cur, 64); dead_e = tb.cur; tb.dead = false; }
    }
    mp.pooler = tb.lin(pre + "pooler.dense", H, H);
    mp.head0 = tb.lin("logit_fc.0", 2 * H, H);
    mp.head_ln = tb.ln("logit_fc.2", 2 * H);
    mp.head3 = tb.lin("logit_fc.3", c.num_answers, 2 * H);
    total = rup(tb.cur, 64);
}

// ============================================================================ engine
// Gradient-buffer sets of the deferred wgrad launches: launch period k uses set k % wgrad_sets().  Two sets let the main stream run ONE period
// ahead of the side stream; behind the cross-modality layers the four language-only layers have short chains, and the main stream then
// sat 0.17 ms per step waiting for the set of two periods ago (profiles/r02_timeline_b256.txt); with four it runs on and the side stream
// catches up beside the longer chains of the paired layers that follow.
#define NPAR 8
#define WGRAD_MERGE_MAX 4      // periods of backward whose weight-gradient problems may share one launch (NPAR >= 2 * WGRAD_MERGE_MAX)
#define SUMSQ_WS_STRIDE 1088     // k_sumsq_owned: 1024 block partials + the ticket word, per gradient segment
#define LNPART_BLOCKS 1536     // per layer: <= 3 LayerNorm-backward launches of <= 512 blocks
int g_rgqa_wgrad_serial = 0;   // rgqa_debug_set(2, v): run the deferred wgrad launches on the main stream
int g_rgqa_skip_wgrad = 0;     // rgqa_debug_set(5, v): MEASUREMENT ONLY - the deferred weight-gradient launches are not issued (gradients are then wrong)
int g_rgqa_wgrad_merge = 0;    // rgqa_debug_set(6, v): periods per weight-gradient launch (1 .. WGRAD_MERGE_MAX); 0 = default
#define WGRAD_MERGE_DEFAULT 3
extern int g_rgqa_force_gemm128;
std::mutex& rgqa_side_stream_mutex() { static std::mutex m; return m; }
std::map<int, hipStream_t>& rgqa_side_streams() { static std::map<int, hipStream_t> m; return m; }
int g_rgqa_ln_fuse = 0;        // rgqa_debug_set(19, v): 1 = the bf16 engine's LayerNorms behind the attention-output / FFN-output projections ride in the GEMM launch
                               // (gemm256_dev.h nt256_ln_after_tile), bit-identical to the separate launches (tests/test_gpu_engine.py).  Default 0: measured, no gain -
                               // the rows come back from the memory side of the L2s either way, so the launch grows by 7 us where the separate kernel took 10
                               // (serial sums: gemm_nt +0.25 ms, layernorm -0.35 ms per step) and the step, whose LayerNorm launches already overlap the
                               // side stream's weight-gradient GEMMs, does not move: 11.404 against 11.406 ms (profiles/r05_ln_fuse_ab.txt)
int g_rgqa_dgrad_nn = 0;       // rgqa_debug_set(14, v): 1 = the bf16 engine's dgrad GEMMs read the weight as it lies ([K, N] operand form; only the visual projection's and the
                               // answer layer's transposed copies are kept: 9 MB instead of 410, their re-cast 0.16 ms shorter), 0 = every dgrad on the transposed bf16
                               // copy (default).  Takes effect at the next rgqa_engine_sync_weights / optimizer step (the copies are re-made by the table the switch
                               // selects).  Bit-identical results (tests/test_gpu_ops.py::test_linear_kn_...); measured on the train step, four interleaved rounds on
                               // one box: 11.176 ms with it, 11.161 without (profiles/r05_dgrad_nn_ab.txt) - the transposed fragment reads (two ds_read_b64_tr_b16
                               // per fragment) cost the 70 dgrad launches what the re-cast saved, so the default stays; the switch is for memory, not for time
int g_rgqa_wgrad_sets = 0;     // rgqa_debug_set(17, v): gradient-buffer sets planned at the next bind (2 * periods-per-launch .. NPAR); 0 = default (the minimum)
int g_rgqa_z_in_place = 1;   // rgqa_debug_set(21, v): bf16x3_fwd precision: 1 (default) = the LayerNorm backward reads the pre-LayerNorm sums' hi parts out of the split-f32 tensor, 0 = out of a bf16
                               // image the projections' epilogues store beside it (round 5); the same bits either way; read at the next forward pass
int g_rgqa_attn_pair = 1;      // rgqa_debug_set(16, v): 0 = the two attention problems of a stage as two launches (the bit-identity test's other arm)

// RGQA_ATTN_REF (test switch: the plain attention kernels instead of the MFMA ones), read once
static bool attn_ref_forced() { static const bool v = getenv("RGQA_ATTN_REF") != nullptr; return v; }
// attention kernels by element type (forward: the engine's activation type; backward: its gradient type)
template <typename U> static int attn_fwd_any(const AttnArgs& a, hipStream_t s) {
    if constexpr (std::is_same<U, sf32>::value) { if (a.dh == 64 && !attn_ref_forced()) return k_attn_fwd_x3(a, s); }
    else if constexpr (std::is_same<U, bf16_t>::value) { if (a.dh == 64 && !attn_ref_forced()) return k_attn_fwd_mfma(a, s); }
    return k_attn_fwd_ref<U>(a, s);
}
template <typename U> static int attn_bwd_any(const AttnArgs& a, hipStream_t s) {
    if constexpr (std::is_same<U, sf32>::value) { if (a.dh == 64 && !attn_ref_forced()) return k_attn_bwd_x3(a, s); }
    else if constexpr (std::is_same<U, bf16_t>::value) { if (a.dh == 64 && !attn_ref_forced()) return k_attn_bwd_mfma(a, s); }
    return k_attn_bwd_ref<U>(a, s);
}
template <typename U> static int attn_fwd_pair_any(const AttnArgs& a0, const AttnArgs& a1, hipStream_t s) {
    if constexpr (std::is_same<U, sf32>::value) return (a0.dh == 64 && !attn_ref_forced()) ? k_attn_fwd_x3_pair(a0, a1, s) : 0;
    else if constexpr (std::is_same<U, bf16_t>::value) return (a0.dh == 64 && !attn_ref_forced()) ? k_attn_fwd_mfma_pair(a0, a1, s) : 0;
    else return 0;
}
template <typename U> static int attn_bwd_pair_any(const AttnArgs& a0, const AttnArgs& a1, hipStream_t s) {
    if constexpr (std::is_same<U, sf32>::value) return (a0.dh == 64 && !attn_ref_forced()) ? k_attn_bwd_x3_pair(a0, a1, s) : 0;
    else if constexpr (std::is_same<U, bf16_t>::value) return (a0.dh == 64 && !attn_ref_forced()) ? k_attn_bwd_mfma_pair(a0, a1, s) : 0;
    else return 0;
}

// T = type of the activations the FORWARD pass computes in (and of the direct weight operand copy).  MIXED (T = sf32 only; rgqa.h
// RGQA_PRECISION_BF16X3_FWD): the backward pass runs on the bf16 kernels - TB = bf16_t is then the type of every gradient buffer, of the
// transposed weight copy, and of a bf16 IMAGE of every forward tensor the backward reads.  The image lives in a mirror of the workspace at
// half its size: the element at byte offset o of the split-f32 workspace has its bf16 image at byte offset o / 2 of the mirror (img()), so
// no second set of pointers is planned; the forward kernels write both (the image is the hi part of every element: one more store, no
// arithmetic; GEMM epilogues, LayerNorm, attention context), small one-off tensors get theirs from k_sf_image.
template <typename T, bool MIXED = false>
class Engine : public EngineBase {
public:
    static_assert(!MIXED || std::is_same<T, sf32>::value, "the mixed precision is a split-f32 forward with a bf16 backward");
    using TB = typename std::conditional<MIXED, bf16_t, T>::type;
    static constexpr bool LP = !std::is_same<T, float>::value;       // operand copies of the weights (direct + transposed) exist
    static constexpr bool X3 = std::is_same<T, sf32>::value;         // split-f32 activations and operand copies (bf16x3 precision)
    template <typename U> static int nt_gemm_t(GemmGroup& g, int out_f32, int trans_b, hipStream_t s) {
        if constexpr (std::is_same<U, sf32>::value) return launch_gemm_nt_x3(g, out_f32, s);
        else if constexpr (std::is_same<U, bf16_t>::value) return launch_gemm_nt_bf16(g, out_f32, s);
        else return launch_gemm_f32(g, 0, trans_b, s);
    }
    static int nt_gemm(GemmGroup& g, int out_f32, int trans_b, hipStream_t s) { return nt_gemm_t<T>(g, out_f32, trans_b, s); }
    static int nt_gemm_b(GemmGroup& g, int out_f32, int trans_b, hipStream_t s) { return nt_gemm_t<TB>(g, out_f32, trans_b, s); }      // dgrad
    static int tn_gemm(GemmGroup& g, hipStream_t s) {
        if constexpr (std::is_same<TB, sf32>::value) return launch_gemm_tn_x3(g, s);
        else if constexpr (std::is_same<TB, bf16_t>::value) return launch_gemm_tn_bf16(g, 1, s);
        else return launch_gemm_f32(g, 1, 1, s);
    }
    // bf16 image of a forward tensor (MIXED), for the kernels that write it (null otherwise) ...
    bf16_t* img(const void* p) const {
        if constexpr (MIXED) return (p == nullptr || ws == nullptr) ? nullptr : reinterpret_cast<bf16_t*>(ws + mirror_off + (((const char*)p - ws) >> 1));
        else return nullptr;
    }
    // ... and what the backward pass reads in place of forward tensor p
    const TB* sv(const void* p) const {
        if constexpr (MIXED) return img(p);
        else return reinterpret_cast<const TB*>(p);
    }
    size_t mirror_off = 0, fwd_end = 0;
    // MIXED, hidden = 768 (round 6): the pre-LayerNorm sums have no bf16 image - the LayerNorm backward kernel reads the hi parts of the split-f32 tensor in
    // place (norm.hip ln_bwd16_kernel<.., ZSF>), and the projections' epilogues write 8 instead of 10 bytes per element
    bool z_in_place() const { return MIXED && cfg.hidden == 768 && fwd_z_in_place; }
    bool fwd_z_in_place = true;      // latched by forward(): backward follows the recorded pass
    const TB* svz(const void* z) const { return z_in_place() ? reinterpret_cast<const TB*>(z) : sv(z); }
    int image_of(const void* p, int ld, int rows, int cols, hipStream_t s) {        // one-off tensors: the image by a copy kernel
        if constexpr (MIXED) return k_sf_image(reinterpret_cast<const sf32*>(p), ld, img(p), ld, rows, cols, s);
        else return RGQA_OK;
    }
    // f32 -> the activation type of the LDS-DMA GEMM operands (the RoI features)
    static int cast_lp(const float* src, T* dst, size_t n, hipStream_t s, bf16_t* image = nullptr) {
        if constexpr (X3) return k_cast_split(src, dst, n, s, image);
        else return k_cast_bf16(src, dst, n, s);
    }
    ModelParams mp;
    std::vector<Stage> stages;
    // bound memory
    float* P = nullptr; float* G = nullptr; T* Pb = nullptr; TB* PbT = nullptr;
    char* ws = nullptr; size_t ws_bytes_ = 0, ws_used = 0;
    int B = 0, Tn = 0, O = 0, Rl = 0, Rv = 0, R = 0, NAp = 0;
    bool dry = false;
    // inputs of the last forward (caller-owned, must stay alive until backward returns)
    const float* in_feats = nullptr; const float* in_boxes = nullptr; const int64_t* in_ids = nullptr; const int64_t* in_seg = nullptr;
    int last_train = 0; uint64_t last_seed = 0; bool have_fwd = false;
    // workspace pieces
    float* maskf = nullptr;
    T *emb_out = nullptr, *emb_z = nullptr; float *emb_mean = nullptr, *emb_rstd = nullptr;
    T *zf = nullptr, *visn_out = nullptr, *feats_lp = nullptr; float* visn_stats = nullptr;
    T *pooled = nullptr, *h1pre = nullptr, *h1 = nullptr, *h2 = nullptr; float *hd_mean = nullptr, *hd_rstd = nullptr;
    float* logits = nullptr; TB* dlogits = nullptr; float* loss_dev = nullptr;
    TB *gA = nullptr, *gB = nullptr, *gctx = nullptr;
    TB *gz_s[NPAR][3] = {}, *gzd_s[NPAR][3] = {}, *gqkv_s[NPAR][3] = {}, *gh_s[NPAR][3] = {};   // [ring position of the launch period][stage slot]
    hipStream_t s_w = nullptr;                 // side stream: the deferred weight-gradient GEMMs of a layer run beside the next layer's chain
    // Made at bind(), not at the first backward: HIP hands streams to its hardware queues in creation order, and which queue a stream shares with
    // which other decides whether two streams really run side by side (round 5: with a RCCL communicator in the process and two to four other
    // streams made first, this stream - or the update stream - landed on a queue where it serialised with the launch stream: 18-20 ms per
    // step instead of 11; tools/rccl_presence3.py).  Engine streams first, in a fixed order, whatever the caller makes later.
    // ONE side stream per device for all engines of the process (a process makes several: bench.py's legs, a trainer's train / eval models): they
    // run one after the other anyway, and every further stream is one more chance to share a hardware queue with the launch stream.
    int make_side_stream() {
        if (s_w != nullptr) return RGQA_OK;
        int dev = 0;
        RGQA_HIP(hipGetDevice(&dev));
        {
            std::lock_guard<std::mutex> lk(rgqa_side_stream_mutex());
            auto& shared = rgqa_side_streams();
            auto it = shared.find(dev);
            if (it == shared.end()) {       // nobody handed one in (rgqa_set_side_stream): make one
                hipStream_t st = nullptr;
                RGQA_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                it = shared.emplace(dev, st).first;
            }
            s_w = it->second;
        }
        for (int i = 0; i < NPAR; ++i) {
            RGQA_HIP(hipEventCreateWithFlags(&ev_chain[i], hipEventDisableTiming));
            RGQA_HIP(hipEventCreateWithFlags(&ev_wdone[i], hipEventDisableTiming));
        }
        return RGQA_OK;
    }
    hipEvent_t ev_chain[NPAR] = {}, ev_wdone[NPAR] = {};
    bool wdone_valid[NPAR] = {};
    TB* gemb = nullptr;
    TB *gp1 = nullptr, *gp2 = nullptr, *gp3 = nullptr;
    static constexpr bool ln_merge = true;   // one LayerNorm launch over [language | vision] rows
    float* lnpart_s[NPAR] = {}; FinDefer fin; int fin_accumulate = 0;   // LayerNorm-backward column sums of the open layer (finalised with its wgrad launch)
    size_t part_floats = 0;
    float* part = nullptr; TransDesc* tdesc = nullptr; int n_tdesc = 0, tdesc_tiles = 0;
    void* lang_final = nullptr;
    std::vector<TransDesc> tdesc_host;

    explicit Engine(const rgqa_config& c) {
        cfg = c;
        joint = cfg.arch == 2;
        build_params(cfg, params, mp, arena_elems, dead_begin, dead_end);
        build_transpose_table();
        build_grad_segments();
    }

    // Completion order of backward: head+pooler, x-layers last..first, then l/r layers last..first, then the embeddings.
    // Every layer occupies one contiguous arena range, so each segment is an element range + the event recorded when
    // its last gradient kernel has been enqueued.
    int n_seg_events = 0;
    int num_weight_segments() const override { return n_seg_events; }
    void build_grad_segments() {
        grad_segs.clear();
        int ev = 0;
        auto seg = [&](size_t b, size_t e, int event) {
            if (dead_end > dead_begin && b < dead_end && e > dead_begin) {      // cut the never-written dead range out
                if (b < dead_begin) grad_segs.push_back({b, dead_begin, event});
                if (e > dead_end) grad_segs.push_back({dead_end, e, event});
            } else if (e > b) grad_segs.push_back({b, e, event});
        };
        seg(mp.pooler.w, arena_elems, ev++);
        auto att_begin = [](const AttP& a) { return a.qkv.w; };
        for (int i = cfg.x_layers - 1; i >= 0; --i) {
            const size_t b = att_begin(mp.x_cross[i]);
            const size_t e = (i + 1 < cfg.x_layers) ? att_begin(mp.x_cross[i + 1]) : mp.pooler.w;
            seg(b, e, ev++);
        }
        const int nlr = cfg.l_layers > cfg.r_layers ? cfg.l_layers : cfg.r_layers;
        const size_t l_end = cfg.r_layers ? att_begin(mp.r_att[0]) : (cfg.x_layers ? att_begin(mp.x_cross[0]) : mp.pooler.w);
        const size_t r_end = cfg.x_layers ? att_begin(mp.x_cross[0]) : mp.pooler.w;
        for (int i = nlr - 1; i >= 0; --i) {
            if (i < cfg.l_layers) seg(att_begin(mp.l_att[i]), (i + 1 < cfg.l_layers) ? att_begin(mp.l_att[i + 1]) : l_end, ev);
            if (i < cfg.r_layers) seg(att_begin(mp.r_att[i]), (i + 1 < cfg.r_layers) ? att_begin(mp.r_att[i + 1]) : r_end, ev);
            ev++;
        }
        const size_t first_layer = cfg.l_layers ? att_begin(mp.l_att[0]) : (cfg.r_layers ? att_begin(mp.r_att[0]) : (cfg.x_layers ? att_begin(mp.x_cross[0]) : mp.pooler.w));
        seg(0, first_layer, ev++);
        n_seg_events = ev;
    }
    // Launches the collected weight-gradient GEMMs of one layer on the side stream, ordered after everything the main
    // stream has enqueued for that layer; records the DP segment event there (the segment is final once both the main
    // stream's bias / LayerNorm gradients and these GEMMs are done).
    // A layer's launch groups problems whose contraction lengths differ 4x (3,140 language rows, 9,216 vision rows, 12,356 rows of the shared
    // cross-attention weights) and whose 256 x 256 output tiles number about one per CU: the launch lasts as long as its longest contraction
    // (193 K-steps) while the CUs of the short ones idle (344 us for work that, spread evenly, is 150 us per CU).  With the problems of two
    // periods in ONE launch the dispatcher hands the second period's tiles to the CUs the first one's short contractions free: 260 us per
    // layer (tools/wgrad_lab.py, profiles/r04_wgrad_lab.txt).  The periods wait in `wgm` until the launch is due; their gradient-buffer
    // sets stay untouched meanwhile (NPAR sets, used round-robin).
    // (Round 4 also measured stream-K scheduling of these launches - equal shares of (tile, K-step) units per CU, tiles cut by a share's
    // boundary summed through partial slabs in a fixed order: correct and bit-repeatable, but 423 us per layer instead of 344: with all 256
    // CUs streaming unshared operand rows the loop is bound by what the fabric delivers, 22 GB/s per CU instead of 46; docs/MEASUREMENTS.md.)
    GemmGroup wg_head;          // head / pooler weight-gradient problems, handed to the encoder's first deferred launch
    GemmGroup wgm; int pend_n = 0, pend_marks = 0; int pend_par[WGRAD_MERGE_MAX] = {}; FinDefer pend_fin[WGRAD_MERGE_MAX]; int pend_acc[WGRAD_MERGE_MAX] = {};
    // periods per weight-gradient launch: what the debug key asks for, within the gradient-buffer sets the workspace was planned with (two per period
    // of a launch: the main stream fills one launch's sets while the side stream still reads the previous launch's)
    static int wgrad_merge_wanted() { const int m = g_rgqa_wgrad_merge > 0 ? g_rgqa_wgrad_merge : WGRAD_MERGE_DEFAULT; return m > WGRAD_MERGE_MAX ? WGRAD_MERGE_MAX : m; }
    int nsets = 2 * WGRAD_MERGE_DEFAULT;
    int wgrad_merge() const { const int m = wgrad_merge_wanted(); return m > nsets / 2 ? nsets / 2 : m; }
    bool set_pending(int par) const { for (int k = 0; k < pend_n; ++k) if (pend_par[k] == par) return true; return false; }
    // Collects the weight-gradient GEMMs of one period; once enough periods wait (or `force`), launches them on the side stream, ordered after
    // everything the main stream has enqueued so far; records the DP segment events there (a segment is final once both the main stream's
    // bias / LayerNorm gradients and these GEMMs are done).
    int flush_wgrad(GemmGroup& wg, int par, hipStream_t s, bool layer_done = true, bool force = false) {
        if (pend_n == 0) gg_init(wgm);
        RGQA_REQUIRE(wgm.count + wg.count <= GEMM_MAX_PROBLEMS && pend_n < WGRAD_MERGE_MAX, "flush_wgrad: too many pending problems (%d + %d)", wgm.count, wg.count);
        for (int i = 0; i < wg.count; ++i) wgm.p[wgm.count++] = wg.p[i];
        gg_init(wg);
        pend_par[pend_n] = par; pend_fin[pend_n] = fin; pend_acc[pend_n] = fin_accumulate; ++pend_n;
        fin.blk = 0; fin.nout = 0; fin.fo = FinOut{};        // handed over: the period's column sums are folded with the launch
        if (layer_done) ++pend_marks;
        // a period adds at most 10 problems (paired FFN + two attention stages)
        if (!force && pend_n < wgrad_merge() && wgm.count + 10 <= GEMM_MAX_PROBLEMS) return RGQA_OK;
        return launch_pending(s);
    }
    int launch_pending(hipStream_t s) {
        if (pend_n == 0) {       // nothing collected (an encoder without stages): only the segment marks are due
            for (int k = 0; k < pend_marks; ++k) if (int r = mark_segment(s)) return r;
            pend_marks = 0;
            return RGQA_OK;
        }
        static const bool serial = getenv("RGQA_WGRAD_SERIAL") != nullptr;
        const bool on_main = serial || g_rgqa_wgrad_serial || profiling;
        hipStream_t st = s;
        if (!on_main) {
            if (int r = make_side_stream()) return r;
            const int par = pend_par[pend_n - 1];
            RGQA_HIP(hipEventRecord(ev_chain[par], s));
            RGQA_HIP(hipStreamWaitEvent(s_w, ev_chain[par], 0));
            st = s_w;
        }
        for (int k = 0; k < pend_n; ++k) if (int r = fin_flush(pend_fin[k], pend_acc[k], st)) return r;
        if (int r = run_wgrad(wgm, st)) return r;
        gg_init(wgm);
        sq_defer = true;
        for (int k = 0; k < pend_marks; ++k) if (int r = mark_segment(st)) { sq_defer = false; return r; }
        sq_defer = false;
        if (int r = k_sumsq_owned_group(sq_batch, st)) return r;
        if (!on_main)
            for (int k = 0; k < pend_n; ++k) {
                RGQA_HIP(hipEventRecord(ev_wdone[pend_par[k]], s_w));
                wdone_valid[pend_par[k]] = true;
            }
        pend_n = 0; pend_marks = 0;
        return RGQA_OK;
    }
    int wgrad_sets() const { return nsets; }
    // every launch on the side stream has joined `s` (the side stream is in order: the newest event covers the older ones); `next` = the set the
    // next period would use
    int join_wgrad(int next, hipStream_t s) {
        const int n = wgrad_sets();
        for (int k = 1; k <= n; ++k) {
            const int p = (next + n - k) % n;          // newest first
            if (wdone_valid[p]) { RGQA_HIP(hipStreamWaitEvent(s, ev_wdone[p], 0)); break; }
        }
        for (int p = 0; p < NPAR; ++p) wdone_valid[p] = false;
        return RGQA_OK;
    }
    // the main stream must not overwrite a gradient-buffer set while an older wgrad launch still reads it
    int wait_wgrad(int par, hipStream_t s) {
        if (wdone_valid[par]) { RGQA_HIP(hipStreamWaitEvent(s, ev_wdone[par], 0)); wdone_valid[par] = false; }
        return RGQA_OK;
    }
    int seg_cursor = 0;
    int mark_segment(hipStream_t s) {
        if ((int)seg_events.size() < n_seg_events) {
            seg_events.resize(n_seg_events);
            for (auto& e : seg_events) RGQA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if (seg_cursor < n_seg_events) {
            const int ev = seg_cursor;
            RGQA_HIP(hipEventRecord(seg_events[seg_cursor++], s));
            // clip_grad_norm_ needs sum(g^2) over everything: take each segment's share here, on the stream that finished it (the
            // weight-gradient side stream for the layers), instead of one 0.8-GB read of the whole arena after backward
            if (sumsq_slots != nullptr && sumsq_ws != nullptr)
                for (int k = 0; k < (int)grad_segs.size() && k < sumsq_ws_segs; ++k)
                    if (grad_segs[k].event == ev) {
                        if (sq_batch.count == SUMSQ_GROUP_MAX) { int r = k_sumsq_owned_group(sq_batch, s); if (r) return r; }
                        sq_batch.r[sq_batch.count++] = SumsqRange{G + grad_segs[k].begin, grad_segs[k].end - grad_segs[k].begin, sumsq_ws + (size_t)k * SUMSQ_WS_STRIDE, sumsq_slots + k, 0};
                    }
            if (!sq_defer) return k_sumsq_owned_group(sq_batch, s);
        }
        return RGQA_OK;
    }
    // the shares of every segment one weight-gradient launch finalises go out as ONE launch (launch_pending brackets its marks with sq_defer)
    SumsqGroup sq_batch{}; bool sq_defer = false;

    // dgrad in the [K, N] operand form (the weight as it lies, csrc/gemm_nt256.h NN): the pure bf16 engine, whole 64-row K-steps of weight rows
    static bool dgrad_nn_on() { return std::is_same<T, bf16_t>::value && !MIXED && g_rgqa_dgrad_nn != 0; }
    static bool dgrad_nn_shape(const Lin& l, int wrows) { return wrows % 64 == 0 && wrows >= 64 && l.in % 8 == 0 && l.in >= 64; }
    std::vector<TransDesc> tdesc_min_host; int n_tdesc_min = 0, tdesc_min_tiles = 0; TransDesc* tdesc_min = nullptr;
    void build_transpose_table() {
        // every [out,in] linear weight gets a transposed low-precision copy at the same arena offset (full table); with the [K, N]-form dgrads only the
        // weights that still need one: the visual projection (the f32 input-gradient GEMM of the ODIN scorer) and layers whose row count is not a
        // whole number of K-steps (the answer layer: 1842 rows)
        auto add = [&](const Lin& l) {
            TransDesc d; d.src_off = (long)l.w; d.dst_off = (long)l.w; d.N = l.out; d.K = l.in; d.ld_dst = l.ldt; d.tile_start = tdesc_tiles;
            tdesc_tiles += cdiv(l.ldt, TRANSPOSE_TILE) * cdiv(l.in, TRANSPOSE_TILE);
            tdesc_host.push_back(d);
            if (&l == &mp.visn_fc || !dgrad_nn_shape(l, l.out)) {
                d.tile_start = tdesc_min_tiles;
                tdesc_min_tiles += cdiv(l.ldt, TRANSPOSE_TILE) * cdiv(l.in, TRANSPOSE_TILE);
                tdesc_min_host.push_back(d);
            }
        };
        add(mp.visn_fc);
        auto addatt = [&](const AttP& a) { add(a.qkv); add(a.o); };
        auto addffn = [&](const FfnP& f) { add(f.up); add(f.down); };
        for (auto& a : mp.l_att) addatt(a);
        for (auto& f : mp.l_ffn) addffn(f);
        for (auto& a : mp.r_att) addatt(a);
        for (auto& f : mp.r_ffn) addffn(f);
        for (auto& a : mp.x_cross) addatt(a);
        for (auto& a : mp.x_latt) addatt(a);
        for (auto& a : mp.x_vatt) addatt(a);
        for (auto& f : mp.x_lffn) addffn(f);
        for (auto& f : mp.x_vffn) addffn(f);
        add(mp.pooler); add(mp.head0); add(mp.head3);
        n_tdesc = (int)tdesc_host.size();
        n_tdesc_min = (int)tdesc_min_host.size();
    }

    // ------------------------------------------------------------------ workspace plan
    template <typename U> U* take(size_t n) {
        size_t bytes = rup(n * sizeof(U), 256);
        U* p = dry ? nullptr : reinterpret_cast<U*>(ws + ws_used);
        ws_used += bytes;
        return p;
    }
    static void* rows(void* base, size_t row, size_t width) { return base ? (void*)((T*)base + row * width) : nullptr; }

    // Buffers are sized for the padded row count (RlC = B*T language rows); the ACTIVE language row count Rl (<= RlC: the
    // packed rows of varlen mode, set_lengths()) only moves the [lang; visn] split inside them, so re-planning for another
    // Rl is pointer arithmetic on the same workspace.
    void plan(int B_, int T_, int O_, int rl_active = -1) {
        if (joint && O_ > 0) { Tt = T_; Oi = O_; T_ = Tt + Oi; O_ = 0; jstate = 0; }       // a new shape; re-plans pass (Tn, 0)
        B = B_; Tn = T_; O = O_; Rv = B * O;
        const int RlC = B * Tn, RC = RlC + Rv;
        Rl = rl_active < 0 ? RlC : rl_active; R = Rl + Rv;
        NAp = (int)rup(cfg.num_answers, 64);
        const int H = cfg.hidden, I = cfg.inter, nh = cfg.heads;
        ws_used = 0;
        stages.clear();
        maskf = take<float>(RlC);
        lens_dev = take<int>(B); cu_dev = take<int>(B + 1); row_src_dev = take<int>(RlC); emb_keys = take<int>(4 * ((size_t)RlC + 4) + 128); cls_rows = take<T>((size_t)B * H); tail_x = take<T>((size_t)B * H); tail_dx = take<TB>((size_t)B * H);
        emb_out = take<T>((size_t)RC * H); emb_z = take<T>((size_t)RlC * H); emb_mean = take<float>(RlC); emb_rstd = take<float>(RlC);
        zf = take<T>((size_t)Rv * H); visn_stats = take<float>((size_t)Rv * 4);
        feats_lp = LP ? take<T>((size_t)Rv * cfg.feat_dim) : nullptr;   // bf16 copy of the RoI features: read by visn_fc forward AND its wgrad
        visn_out = emb_out ? emb_out + (size_t)Rl * H : nullptr;   // [lang; visn] contiguous
        void* cur[2] = {emb_out, visn_out};
        uint32_t site = 16;
        auto new_stage = [&](int kind, bool al, bool av) -> Stage& {
            Stage st; memset(&st, 0, sizeof st);
            st.kind = kind; st.active[0] = al; st.active[1] = av; st.site = site; site += 8;
            T* y = take<T>((size_t)RC * H);
            T* z = take<T>((size_t)RC * H);
            float* mean = take<float>(RC); float* rstd = take<float>(RC);
            T *qkv = nullptr, *ctx = nullptr, *hpre = nullptr, *h = nullptr; float* lse = nullptr;
            if (kind == ST_FFN) { hpre = take<T>((size_t)RC * I); h = take<T>((size_t)RC * I); }
            else { qkv = take<T>((size_t)RC * 3 * H); ctx = take<T>((size_t)RC * H); lse = take<float>((size_t)B * nh * (Tn + O)); }
            for (int m = 0; m < 2; ++m) {
                const size_t r0 = m == 0 ? 0 : Rl;
                SegBuf& s = st.sb[m];
                s.x_in = cur[m];
                s.y = rows(y, r0, H); s.z = rows(z, r0, H);
                s.mean = mean ? mean + r0 : nullptr; s.rstd = rstd ? rstd + r0 : nullptr;
                s.qkv = rows(qkv, r0, 3 * H); s.ctx = rows(ctx, r0, H);
                s.hpre = rows(hpre, r0, I); s.h = rows(h, r0, I);
                s.lse = lse ? lse + (m == 0 ? 0 : (size_t)B * nh * Tn) : nullptr;
                if (st.active[m]) cur[m] = s.y;
            }
            stages.push_back(st);
            return stages.back();
        };
        const int nlr = cfg.l_layers > cfg.r_layers ? cfg.l_layers : cfg.r_layers;
        for (int i = 0; i < nlr; ++i) {
            const bool al = i < cfg.l_layers, av = i < cfg.r_layers;
            Stage& a = new_stage(ST_ATT_SELF, al, av);
            if (al) a.att[0] = &mp.l_att[i];
            if (av) a.att[1] = &mp.r_att[i];
            a.slot = 0; a.layer_first = 1; a.seg_event = 1 + cfg.x_layers + (nlr - 1 - i);
            Stage& f = new_stage(ST_FFN, al, av);
            if (al) f.ffn[0] = &mp.l_ffn[i];
            if (av) f.ffn[1] = &mp.r_ffn[i];
            f.seg_event = a.seg_event;
            f.slot = 2; f.layer_first = 0;      // not 1: with RGQA_WGRAD_PHASE=ffn a launch period spans [self-attention (slot 1) of the first x-layer, this FFN]
        }
        // the cross-modality layers need [lang; visn] contiguous: copy-free when both chains end in one stage,
        // otherwise the engine gathers them into x0 (one row copy of the shorter chain's output)
        x0_needed = (cfg.x_layers > 0) && (cfg.l_layers != cfg.r_layers);
        if (x0_needed) {
            x0 = take<T>((size_t)RC * H);
            x0_src[0] = cur[0]; x0_src[1] = cur[1];
            cur[0] = x0; cur[1] = rows(x0, Rl, H);
            // the last stage of each chain writes its output straight into x0 (no row copy); a chain without layers still copies
            // its embedding output
            for (int m = 0; m < 2; ++m)
                for (int si = (int)stages.size() - 1; si >= 0; --si)
                    if (stages[si].active[m]) { stages[si].sb[m].y = cur[m]; x0_src[m] = nullptr; break; }
        }
        for (int i = 0; i < cfg.x_layers; ++i) {
            const bool last = (i == cfg.x_layers - 1);
            Stage& c = new_stage(ST_ATT_CROSS, true, !last);
            c.att[0] = c.att[1] = &mp.x_cross[i]; c.last_dead = last; c.slot = 0; c.layer_first = 1; c.seg_event = 1 + (cfg.x_layers - 1 - i);
            Stage& a = new_stage(ST_ATT_SELF, true, !last);
            a.att[0] = &mp.x_latt[i]; a.att[1] = &mp.x_vatt[i]; a.last_dead = last; a.slot = 1; a.layer_first = 0; a.seg_event = c.seg_event;
            Stage& f = new_stage(ST_FFN, true, !last);
            f.ffn[0] = &mp.x_lffn[i]; f.ffn[1] = &mp.x_vffn[i]; f.last_dead = last; f.slot = 2; f.layer_first = 0; f.seg_event = c.seg_event;
        }
        lang_final = cur[0];
        visn_final = cur[1];
        pooled = take<T>((size_t)B * H); h1pre = take<T>((size_t)B * 2 * H); h1 = take<T>((size_t)B * 2 * H); h2 = take<T>((size_t)B * 2 * H);
        hd_mean = take<float>(B); hd_rstd = take<float>(B);
        logits = take<float>((size_t)B * NAp);
        if (joint) {      // UNITER front-end: its forward tensors (they have bf16 images under MIXED) before the mark below
            const size_t ni = (size_t)B * Oi;
            u_zf = take<T>(ni * H); u_zp = take<T>(ni * H); u_a1 = take<T>(ni * H); u_a2 = take<T>(ni * H); u_x3 = take<T>(ni * H);
            feats_lp = LP ? take<T>(ni * cfg.feat_dim) : nullptr;
        }
        fwd_end = ws_used;        // everything img() / sv() may be asked for lies below: the bf16 image mirror covers [0, fwd_end) only
        dlogits = take<TB>((size_t)B * NAp); loss_dev = take<float>(64);
        gA = take<TB>((size_t)RC * H); gB = take<TB>((size_t)RC * H); gctx = take<TB>((size_t)RC * H); gemb = take<TB>((size_t)RC * H);
        // one set of per-stage gradient buffers per stage slot of a layer: the weight-gradient GEMMs of a whole layer are
        // deferred into ONE grouped launch (432-504 tiles: fills the 256 CUs), so their operands must outlive the stage
        // ... and two such sets (layer parity): layer i's wgrad launch reads its set on the side stream while layer i-1
        // already overwrites the other one on the main stream
        // two per period of a launch - the main stream fills the next launch's sets while the side stream reads the previous launch's - plus one
        // spare pair: with exactly 2 x 3 the main stream waits now and then for the launch in flight (A/B on one box, rgqa_debug_set key 17,
        // three interleaved rounds: 6 sets 11.556 ms per step, 8 sets 11.516; profiles/r05_wgrad_sets_ab.txt)
        const int min_sets = 2 * wgrad_merge_wanted();
        nsets = min_sets + 2 < NPAR ? min_sets + 2 : NPAR;
        if (g_rgqa_wgrad_sets >= min_sets) nsets = g_rgqa_wgrad_sets < NPAR ? g_rgqa_wgrad_sets : NPAR;
        // (8 sets were planned whatever the merge depth: +2.4 GB bf16 / +4.8 GB split f32 at B = 256)
        for (int par = 0; par < nsets; ++par)
            for (int k = 0; k < 3; ++k) {
                gz_s[par][k] = take<TB>((size_t)RC * H); gzd_s[par][k] = take<TB>((size_t)RC * H);
                gqkv_s[par][k] = take<TB>((size_t)RC * 3 * H); gh_s[par][k] = take<TB>((size_t)RC * I);
            }
        gp1 = take<TB>((size_t)B * 2 * H); gp2 = take<TB>((size_t)B * 2 * H); gp3 = take<TB>((size_t)B * 2 * H);
        size_t pw = 2 * (size_t)H; if ((size_t)I > pw) pw = I; if (3 * (size_t)H > pw) pw = 3 * (size_t)H; if ((size_t)NAp > pw) pw = NAp;
        part_floats = (size_t)512 * 10 * pw;
        part = take<float>(part_floats);
        for (int par = 0; par < nsets; ++par) lnpart_s[par] = take<float>((size_t)LNPART_BLOCKS * 3 * H);
        sumsq_ws_segs = (int)grad_segs.size(); sumsq_ws = take<float>((size_t)sumsq_ws_segs * SUMSQ_WS_STRIDE);
        if (joint) {
            const size_t ni = (size_t)B * Oi, nt = (size_t)B * Tt;
            tlens_dev = take<int>(B); tcu_dev = take<int>(B + 1); trow_src_dev = take<int>(nt); text_dst_dev = take<int>(nt); img_dst_dev = take<int>(ni);
            u_g = take<TB>(ni * H); u_dx3 = take<TB>(ni * H); u_dz = take<TB>(ni * H); u_gt = take<TB>(nt * H); u_de = take<TB>(nt * H);
            u_st = take<float>(6 * ni);
        }
        tdesc = take<TransDesc>(n_tdesc + 1);
        tdesc_min = take<TransDesc>(n_tdesc_min + 1);
        ln_tk = take<int>(2 * LN_TK_PER_PROBLEM);        // tickets of the LayerNorms fused into the projections' launches: per problem of a launch, one per row block
        if (MIXED) {          // the bf16 images: a half-size mirror of the forward tensors planned above (img()); gradient buffers, f32 scratch and
            mirror_off = rup(ws_used, 256);      // index arrays have no image (the mirror used to cover the whole plan, +40 %)
            ws_used = mirror_off + rup(fwd_end / 2, 256);
        }
    }
    int* emb_keys = nullptr;      // word / position / token-type key of every packed row (embedding backward)
    int *lens_dev = nullptr, *cu_dev = nullptr, *row_src_dev = nullptr; T* cls_rows = nullptr; T* tail_x = nullptr; TB* tail_dx = nullptr; T* pool_in = nullptr;
    // UNITER (arch 2): ONE sequence per sample, [text tokens ; image regions], laid out as the engine's language modality with
    // Tn = Tt + Oi rows per sample (packed: real text tokens + Oi) and no vision modality; only the embedding front-end differs.
    bool joint = false; int Tt = 0, Oi = 0, jstate = 0, n_text = 0;      // jstate: index arrays built for 1 = padded / 2 = packed rows
    std::vector<int> tlens_host;
    int *tlens_dev = nullptr, *tcu_dev = nullptr, *trow_src_dev = nullptr, *text_dst_dev = nullptr, *img_dst_dev = nullptr;
    T *u_zf = nullptr, *u_zp = nullptr, *u_a1 = nullptr, *u_a2 = nullptr, *u_x3 = nullptr; TB *u_g = nullptr, *u_dx3 = nullptr, *u_dz = nullptr, *u_gt = nullptr, *u_de = nullptr;
    float *u_st = nullptr;     // [6][B*Oi]: mean / rstd of img_layer_norm, pos_layer_norm, LayerNorm
    bool varlen = false, lens_dirty = false, fwd_varlen = false;   // fwd_varlen: layout of the recorded forward pass
    int n_lang = 0; std::vector<int> lens_host;
    bool x0_needed = false; T* x0 = nullptr; void* x0_src[2] = {nullptr, nullptr}; void* visn_final = nullptr;

    size_t workspace_bytes(int B_, int T_, int O_) override {
        dry = true;
        plan(B_, T_, O_);
        dry = false;
        return ws_used + 256;
    }

    int bind(float* p, float* g, void* plp, void* plpt, void* w, size_t wb, int B_, int T_, int O_) override {
        RGQA_REQUIRE(p != nullptr && w != nullptr, "bind: null parameter arena or workspace");
        RGQA_REQUIRE(B_ > 0 && T_ > 0 && O_ > 0 && T_ <= 64 && O_ <= 64, "bind: B=%d T=%d O=%d unsupported (T, O <= 64)", B_, T_, O_);
        RGQA_REQUIRE(T_ <= cfg.max_pos, "bind: T=%d exceeds max_position_embeddings=%d", T_, cfg.max_pos);
        if (LP) RGQA_REQUIRE(plp != nullptr && plpt != nullptr, "bind: bf16 / bf16x3 precision needs the operand-copy arenas");
        if (X3) RGQA_REQUIRE(((uintptr_t)plp % 128) == 0 && ((uintptr_t)plpt % 128) == 0, "bind: the split-f32 arenas must be 128-byte aligned");      // (MIXED: the transposed copy is bf16; the same alignment costs nothing)
        RGQA_REQUIRE(((uintptr_t)p % 256) == 0 && ((uintptr_t)w % 256) == 0, "bind: arenas must be 256-byte aligned");
        size_t need = workspace_bytes(B_, T_, O_);
        if (wb < need) { rgqa_set_error("bind: workspace too small (%zu < %zu bytes)", wb, need); return RGQA_ERR_WORKSPACE; }
        P = p; G = g; Pb = (T*)plp; PbT = (TB*)plpt; ws = (char*)w; ws_bytes_ = wb;
        if (int r = make_side_stream()) return r;
        plan(B_, T_, O_);
        RGQA_HIP(hipMemset(sumsq_ws, 0, sizeof(float) * (size_t)sumsq_ws_segs * SUMSQ_WS_STRIDE));       // the ticket words of k_sumsq_owned start at zero
        RGQA_HIP(hipMemset(ln_tk, 0, sizeof(int) * 2 * LN_TK_PER_PROBLEM));                               // ... and so do the fused LayerNorms' (the last arriver of a row block re-zeroes its own)
        varlen = false; lens_dirty = false;
        have_fwd = false;
        tdesc_uploaded = false;
        return RGQA_OK;
    }
    bool tdesc_uploaded = false;
    bool nn_synced = false;        // the transposed copies were last re-made by the minimal table: the other weights' copies are stale

    // gradients w.r.t. the inputs (f32 [B*O, feat_dim] / [B*O, pos_dim]) written by the following backward calls; null = not computed
    float* dfeats_out = nullptr; float* dboxes_out = nullptr;
    int set_input_grads(float* dfeats, float* dboxes) override { dfeats_out = dfeats; dboxes_out = dboxes; return RGQA_OK; }

    // Unpadded language rows: lens[b] = number of real tokens of sample b ([CLS] .. [SEP], a PREFIX of its T slots - what
    // convert_sents_to_features builds, lxrt/entry.py:37-79).  Padded positions never reach the logits or any gradient
    // (-10000 key mask -> probability exactly 0, pooler reads token 0), so the engine then packs only the real rows:
    // GEMM / LayerNorm rows shrink from B*T to sum(lens), attention windows follow cu[].  null / n == 0: padded layout.
    int set_lengths(const int* lens, int n) override {
        if (lens == nullptr || n == 0) { varlen = false; return RGQA_OK; }
        RGQA_REQUIRE(ws != nullptr, "set_lengths: engine not bound");
        RGQA_REQUIRE(n == B, "set_lengths: %d lengths for a batch of %d", n, B);
        long tot = 0;
        if (joint) {       // lengths count TEXT tokens; a sample's row window is its text tokens followed by its Oi regions
            for (int i = 0; i < n; ++i) {
                RGQA_REQUIRE(lens[i] >= 1 && lens[i] <= Tt, "set_lengths: lengths[%d] = %d outside 1..%d", i, lens[i], Tt);
                tot += lens[i] + Oi;
            }
            tlens_host.assign(lens, lens + n);
            lens_host.resize(n);
            for (int i = 0; i < n; ++i) lens_host[i] = lens[i] + Oi;
            n_lang = (int)tot; varlen = true; lens_dirty = true;
            return RGQA_OK;
        }
        for (int i = 0; i < n; ++i) {
            RGQA_REQUIRE(lens[i] >= 1 && lens[i] <= Tn, "set_lengths: lengths[%d] = %d outside 1..%d", i, lens[i], Tn);
            tot += lens[i];
        }
        lens_host.assign(lens, lens + n);
        n_lang = (int)tot; varlen = true; lens_dirty = true;
        return RGQA_OK;
    }

    int sync_weights(hipStream_t s) override {
        RGQA_REQUIRE(P != nullptr, "sync_weights: engine not bound");
        if (!LP) return RGQA_OK;
        int r = cast_lp(P, Pb, arena_elems, s);
        if (r) return r;
        return sync_transposed(s);
    }
    // only the transposed bf16 copies (the optimizer kernel already wrote the direct bf16 copy)
    int sync_transposed(hipStream_t s) override {
        RGQA_REQUIRE(P != nullptr, "sync_transposed: engine not bound");
        if (!LP) return RGQA_OK;
        if (!tdesc_uploaded) {
            RGQA_HIP(hipMemcpyAsync(tdesc, tdesc_host.data(), sizeof(TransDesc) * n_tdesc, hipMemcpyHostToDevice, s));
            if (n_tdesc_min) RGQA_HIP(hipMemcpyAsync(tdesc_min, tdesc_min_host.data(), sizeof(TransDesc) * n_tdesc_min, hipMemcpyHostToDevice, s));
            tdesc_uploaded = true;
        }
        nn_synced = dgrad_nn_on();
        if (nn_synced) return k_cast_transpose(Pb, 1, PbT, 0, tdesc_min, n_tdesc_min, tdesc_min_tiles, s);      // 2 of the 98 copies: 9 MB instead of 410
        if constexpr (MIXED) return k_cast_transpose(P, 0, PbT, 0, tdesc, n_tdesc, tdesc_tiles, s);      // bf16 transposed copy (the dgrad operand) from the f32 masters
        else if constexpr (X3) return k_cast_transpose(P, 0, PbT, 1, tdesc, n_tdesc, tdesc_tiles, s);
        else return k_cast_transpose(Pb, 1, PbT, 0, tdesc, n_tdesc, tdesc_tiles, s);     // from the bf16 copy (the optimizer kernel / sync_weights wrote it): half the read bytes
    }

    // ------------------------------------------------------------------ GEMM helpers
    DropCfg drop_base(float p) const { return make_drop(last_train ? p : 0.f, last_seed, 0); }
    DropCfg drop_site(float p, uint32_t site) const { DropCfg d = drop_base(p); d.seed_hi ^= site; return d; }

    struct GG { GemmGroup g; };
    static void gg_init(GemmGroup& g) { memset(&g, 0, sizeof g); }
    // y[rows, out_cols] = x[rows, in] @ W[row0.., :]^T (+bias) ; FWD
    void add_fwd(GemmGroup& g, const void* x, int ldx, const Lin& l, int wrow0, int wrows, void* y, int ldy, int M, int epi, const void* aux, int ldaux, void* c2, uint32_t site, bool bias = true) {
        GemmProblem& p = g.p[g.count++];
        memset(&p, 0, sizeof p);
        p.A = x; p.lda = ldx; p.M = M; p.N = wrows; p.K = l.in; p.C = y; p.ldc = ldy; p.C2 = c2;
        p.B = LP ? (const void*)(Pb + l.w + (size_t)wrow0 * l.in) : (const void*)(P + l.w + (size_t)wrow0 * l.in);
        p.ldb = l.in;
        p.bias = bias ? P + l.b + wrow0 : nullptr;
        p.aux = aux; p.ldaux = ldaux; p.epi = epi; p.drop_site = site;
        if (MIXED) {      // the bf16 image of the result beside it; gelu' (read by the backward alone) only as its image
            p.Cb = (epi == EPI_RESID_DROP && z_in_place()) ? nullptr : img(y);      // (a pre-LayerNorm sum: read in place by the LayerNorm backward)
            if (c2 != nullptr) { p.C2 = img(c2); p.c2_lp = 1; }
        }
    }
    // The LayerNorm behind the projection just added to g (EPI_RESID_DROP, N = hidden = 768) rides in its launch: the workgroup that finishes the last
    // of a row block's three tiles normalises it (gemm.h ln_*; gemm256_dev.h nt256_ln_after_tile).  bf16 engine only: the split-f32 rows of the
    // bf16x3 precisions would need 3 x 512 B per row and lane-exact agreement with their own LayerNorm kernel - not built.
    static constexpr int LN_TK_PER_PROBLEM = 512;      // row blocks of >= 64 rows: up to 32768 rows per problem
    int* ln_tk = nullptr;
    bool ln_fuse_on() const { return std::is_same<T, bf16_t>::value && !MIXED && g_rgqa_ln_fuse != 0 && !g_rgqa_force_gemm128 && cfg.hidden == 768 && R <= 64 * LN_TK_PER_PROBLEM; }
    void fuse_ln(GemmGroup& g, const LNp& ln, void* y, float* mean, float* rstd) {
        GemmProblem& p = g.p[g.count - 1];
        p.ln_g = P + ln.w; p.ln_b = P + ln.b; p.ln_y = y; p.ln_mean = mean; p.ln_rstd = rstd; p.ln_eps = cfg.ln_eps;
        p.ln_tk = ln_tk + (size_t)(g.count - 1) * LN_TK_PER_PROBLEM;
    }
    // dx[rows, in] = dy[rows, cols] @ W[wrow0 : wrow0+cols, :]      ; DGRAD
    bool dgrad_mixed = false;      // a group mixing [K, N] and [N, K] weight operands was built (never for the encoder's layer shapes): run_dgrad refuses it
    void add_dgrad(GemmGroup& g, const void* dy, int lddy, const Lin& l, int wrow0, int wrows, void* dx, int lddx, int M, int epi, const void* aux, int ldaux, bool allow_nn = true) {
        GemmProblem& p = g.p[g.count++];
        memset(&p, 0, sizeof p);
        p.A = dy; p.lda = lddy; p.M = M; p.N = l.in; p.C = dx; p.ldc = lddx;
        const bool nn = allow_nn && nn_synced && dgrad_nn_shape(l, wrows) && &l != &mp.visn_fc;
        if (LP && nn && (g.count == 1 || g.b_kn)) {
            // the weight as stored, [out, in]: rows wrow0 .. wrow0 + wrows are the contraction - no transposed copy involved (gemm.h b_kn)
            if constexpr (std::is_same<T, bf16_t>::value) { g.b_kn = 1; p.B = Pb + l.w + (size_t)wrow0 * l.in; p.ldb = l.in; p.K = wrows; }
        } else if (LP) {
            // the transposed copy ([in, ldt]; it and dy are zero-padded up to ldt = round_up(out, 64): a whole number of K-steps for the LDS-DMA kernel).
            // Under the [K, N] regime only the minimal table's copies are current: anything else here is a bug of the caller
            const bool has_copy = !nn_synced || &l == &mp.visn_fc || !dgrad_nn_shape(l, l.out);
            if (g.b_kn || !has_copy) dgrad_mixed = true;
            p.B = PbT + l.w + wrow0; p.ldb = l.ldt; p.K = (wrow0 + wrows == l.out) ? (int)rup(wrows, 64) : wrows; if (p.K > l.ldt - wrow0) p.K = l.ldt - wrow0;
        }
        else { p.B = P + l.w + (size_t)wrow0 * l.in; p.ldb = l.in; p.K = wrows; }
        p.aux = aux; p.ldaux = ldaux; p.epi = epi;
    }
    // dW[wrow0 : wrow0+cols, :] (+)= dy[rows, cols]^T @ x[rows, in]   ; WGRAD
    void add_wgrad(GemmGroup& g, const void* dy, int lddy, const Lin& l, int wrow0, int wrows, const void* x, int ldx, int M, int accumulate, bool with_bias = false) {
        GemmProblem& p = g.p[g.count++];
        memset(&p, 0, sizeof p);
        if (with_bias) p.colsum_out = G + l.b + wrow0;
        p.A = dy; p.lda = lddy; p.B = x; p.ldb = ldx; p.K = M; p.M = wrows; p.N = l.in;
        p.C = G + l.w + (size_t)wrow0 * l.in; p.ldc = l.in; p.epi = accumulate ? EPI_ACCUM : EPI_BIAS;
    }
    // per-launch dump tag (RGQA_PROF_DUMP): problem count, epilogue, M0 x N0 x K0 [+ M1]
    static const char* gemm_tag(const GemmGroup& g, char (&buf)[48]) {
        if (g.count > 1) snprintf(buf, sizeof buf, "n%d_e%d_%dx%dx%d+%d", g.count, g.p[0].epi, g.p[0].M, g.p[0].N, g.p[0].K, g.p[1].M);
        else snprintf(buf, sizeof buf, "n1_e%d_%dx%dx%d", g.p[0].epi, g.p[0].M, g.p[0].N, g.p[0].K);
        return buf;
    }
    double last_obytes = 0;        // gemm_work: the operands alone (A + B + C) of the group it was last called for
    void gemm_work(const GemmGroup& g, double& flops, double& bytes) { gemm_work_t<T>(g, flops, bytes); }
    // U = element type of the launch's operands (T forward; TB dgrad / wgrad: bf16 under MIXED, where round 5 priced them at 4 bytes)
    template <typename U> void gemm_work_t(const GemmGroup& g, double& flops, double& bytes) {
        flops = 0; bytes = 0;
        for (int i = 0; i < g.count; ++i) {
            const GemmProblem& p = g.p[i];
            flops += 2.0 * p.M * p.N * p.K;
            bytes += sizeof(U) * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N);
            last_obytes = (i == 0 ? 0.0 : last_obytes) + sizeof(U) * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N);
            // the operands of the fused epilogue are algorithmic bytes of the launch too: the residual / gelu' / activation it reads and
            // the second output it writes would be moved by a separate element-wise kernel otherwise (twice: that kernel would re-read C)
            if (epi_needs_aux(p.epi) && p.aux != nullptr) bytes += sizeof(U) * (double)p.M * p.N;
            if (p.C2 != nullptr) bytes += (p.c2_lp ? 2.0 : (double)sizeof(U)) * (double)p.M * p.N;
            if (p.Cb != nullptr) bytes += 2.0 * (double)p.M * p.N;
            if (p.ln_tk != nullptr) bytes += 2.0 * sizeof(U) * (double)p.M * p.N;      // the fused LayerNorm: the sum read back, the normalised rows written
        }
    }
    // cls_rows: the launch is one of those whose rows are the B [CLS] rows (tail of the last layer, pooler, answer head): skinny whatever the
    // batch - it may take the split-K path.  Chosen by call site, not by shape: a sample's arithmetic must not depend on the batch it is in.
    int run_fwd(GemmGroup& g, hipStream_t s, int out_f32 = 0, int a_f32 = 0, bool cls_rows = false) {
        if (g.count == 0) return RGQA_OK;
        g.a_f32 = a_f32;
        if (out_f32) for (int i = 0; i < g.count; ++i) g.p[i].Cb = nullptr;      // f32 results (the logits) have no image
        double f, b; gemm_work(g, f, b);
        char tg[48];
        prof_begin(PC_GEMM_NT, f, b, s, profiling ? gemm_tag(g, tg) : "", last_obytes);
        if constexpr (LP) { if (cls_rows) { g.splitk_ws = part; g.splitk_floats = part_floats; } }      // (split f32 since round 6: gemm_x3.hip)
        int r = nt_gemm(g, out_f32, 0, s);
        prof_end(s);
        return r;
    }
    int run_dgrad(GemmGroup& g, hipStream_t s, bool cls_rows = false) {
        if (g.count == 0) return RGQA_OK;
        RGQA_REQUIRE(!dgrad_mixed, "dgrad: a launch mixes [K, N] and transposed weight operands, or names a transposed copy that is not kept");
        double f, b; gemm_work_t<TB>(g, f, b);
        char tg[48];
        prof_begin(PC_GEMM_NT_D, f, b, s, profiling ? gemm_tag(g, tg) : "", last_obytes);
        if constexpr (LP) { if (cls_rows) { g.splitk_ws = part; g.splitk_floats = part_floats; } }
        int r = nt_gemm_b(g, 0, 1, s);
        prof_end(s);
        return r;
    }
    int run_wgrad(GemmGroup& g, hipStream_t s, int b_f32 = 0) {
        if (g.count == 0 || g_rgqa_skip_wgrad) return RGQA_OK;
        g.a_f32 = b_f32;
        double f, b; gemm_work_t<TB>(g, f, b);
        char tg[48];
        prof_begin(PC_GEMM_TN, f, b, s, profiling ? gemm_tag(g, tg) : "");
        int r = tn_gemm(g, s);
        prof_end(s);
        return r;
    }
    int attn_fwd_dispatch(const AttnArgs& a, hipStream_t s) { return attn_fwd_any<T>(a, s); }
    int attn_bwd_dispatch(const AttnArgs& a, hipStream_t s) { return attn_bwd_any<TB>(a, s); }
    // both attention problems of a stage in one launch: 1 = launched, 0 = not covered (launch them separately), < 0 = error
    int attn_fwd_pair_dispatch(const AttnArgs& a0, const AttnArgs& a1, hipStream_t s) { return attn_fwd_pair_any<T>(a0, a1, s); }
    int attn_bwd_pair_dispatch(const AttnArgs& a0, const AttnArgs& a1, hipStream_t s) { return attn_bwd_pair_any<TB>(a0, a1, s); }
    static bool attn_pair_wanted() { return g_rgqa_attn_pair != 0; }

#define CK(x) do { int _r = (x); if (_r) return _r; } while (0)
// CK + HIP-event timing of the call under profiling (non-GEMM kernels: they count towards the per-block times)
#define CKP(cat, x) do { prof_begin((cat), 0.0, 0.0, s); int _r = (x); prof_end(s); if (_r) return _r; } while (0)

    // The pooler reads only token 0 of the final language output (modeling.py:575-581) and nothing else reads that output in mode
    // 'x': the FFN sub-block of the LAST cross-modality layer runs on the B [CLS] rows only (gathered in, kept compact: rows 0..B-1
    // of the stage buffers), forward and backward.  RGQA_CLS_TAIL=0 computes every row as the reference does.
    bool fwd_cls_tail = true;          // latched by forward(): backward and get_activation follow the layout of the recorded pass
    static bool cls_tail_wanted() {
        static const bool env_on = []() { const char* e = getenv("RGQA_CLS_TAIL"); return !(e && e[0] == '0'); }();
        return g_rgqa_cls_tail < 0 ? env_on : g_rgqa_cls_tail != 0;       // rgqa_debug_set key 8 (tests: both paths in one process)
    }
    bool cls_tail(const Stage& st) const {
        return fwd_cls_tail && cfg.arch == 0 && st.kind == ST_FFN && st.last_dead && st.active[0] && !st.active[1] && &st == &stages.back();
    }
    int seg_rows(int m) const { return m == 0 ? Rl : Rv; }
    int seg_len(int m) const { return m == 0 ? Tn : O; }

    // ------------------------------------------------------------------ forward
    // one stage (self-attention / cross-attention / FFN sub-block) of the forward pass for the modalities st.active[] marks, on stream s
    int forward_stage(Stage& st, const int* cu, hipStream_t s) {
        const int H = cfg.hidden, I = cfg.inter, nh = cfg.heads, dh = H / nh;
        CK(wait_wready(st.seg_event, s));             // this layer's weights may still be arriving from their owner rank (sharded exchange)
        const float pd = cfg.hidden_dropout, pa = cfg.attn_dropout;
        if (st.kind == ST_FFN && cls_tail(st)) {
            GemmGroup g; gg_init(g);
            CKP(PC_OTHER, k_gather_rows<T>((const T*)st.sb[0].x_in, H, cu, Tn, tail_x, H, B, H, s));      // the [CLS] row of every sample
            CK(image_of(tail_x, H, B, H, s));
            add_fwd(g, tail_x, H, st.ffn[0]->up, 0, I, st.sb[0].h, I, B, EPI_GELU, nullptr, 0, st.sb[0].hpre, 0);
            CK(run_fwd(g, s, 0, 0, true));
            gg_init(g); g.drop = drop_base(pd);
            add_fwd(g, st.sb[0].h, I, st.ffn[0]->down, 0, H, st.sb[0].z, H, B, EPI_RESID_DROP, tail_x, H, nullptr, st.site + 1);
            CK(run_fwd(g, s, 0, 0, true));
            CKP(PC_LN, k_ln_fwd<T>((T*)st.sb[0].z, H, P + st.ffn[0]->ln.w, P + st.ffn[0]->ln.b, (T*)st.sb[0].y, H, st.sb[0].mean, st.sb[0].rstd, B, H, cfg.ln_eps, s, img(st.sb[0].y)));
            return RGQA_OK;
        }
        if (st.kind == ST_FFN) {
            GemmGroup g; gg_init(g);
            for (int m = 0; m < 2; ++m) if (st.active[m])
                add_fwd(g, st.sb[m].x_in, H, st.ffn[m]->up, 0, I, st.sb[m].h, I, seg_rows(m), EPI_GELU, nullptr, 0, st.sb[m].hpre, 0);
            CK(run_fwd(g, s));
            gg_init(g); g.drop = drop_base(pd);
            const bool lnf = ln_fuse_on();
            for (int m = 0; m < 2; ++m) if (st.active[m]) {
                add_fwd(g, st.sb[m].h, I, st.ffn[m]->down, 0, H, st.sb[m].z, H, seg_rows(m), EPI_RESID_DROP, st.sb[m].x_in, H, nullptr, st.site + m * 4 + 1);
                if (lnf) fuse_ln(g, st.ffn[m]->ln, st.sb[m].y, st.sb[m].mean, st.sb[m].rstd);
            }
            CK(run_fwd(g, s));
            if (lnf) return RGQA_OK;
            if (ln_merge && st.active[0] && st.active[1] && Rl > 0 && Rv > 0 && (T*)st.sb[1].y == (T*)st.sb[0].y + (size_t)Rl * H) {
                // language | vision rows are adjacent in the stage buffers: one launch, per-segment module parameters
                CKP(PC_LN, k_ln_fwd2<T>((T*)st.sb[0].z, H, P + st.ffn[0]->ln.w, P + st.ffn[0]->ln.b, P + st.ffn[1]->ln.w, P + st.ffn[1]->ln.b, Rl,
                                        (T*)st.sb[0].y, H, st.sb[0].mean, st.sb[0].rstd, R, H, cfg.ln_eps, s, img(st.sb[0].y)));
                return RGQA_OK;
            }
            for (int m = 0; m < 2; ++m) if (st.active[m])
                CKP(PC_LN, k_ln_fwd<T>((T*)st.sb[m].z, H, P + st.ffn[m]->ln.w, P + st.ffn[m]->ln.b, (T*)st.sb[m].y, H, st.sb[m].mean, st.sb[m].rstd, seg_rows(m), H, cfg.ln_eps, s, img(st.sb[m].y)));
            return RGQA_OK;
        }
        // ---- attention stages
        const bool cross = st.kind == ST_ATT_CROSS;
        {
            GemmGroup g; gg_init(g);
            if (cross) {
                const AttP& ap = *st.att[0];
                if (st.active[1]) add_fwd(g, st.sb[0].x_in, H, ap.qkv, 0, 3 * H, st.sb[0].qkv, 3 * H, R, EPI_BIAS, nullptr, 0, nullptr, 0);
                else {   // final x-layer: only lang queries and visn keys/values are live
                    add_fwd(g, st.sb[0].x_in, H, ap.qkv, 0, H, st.sb[0].qkv, 3 * H, Rl, EPI_BIAS, nullptr, 0, nullptr, 0);
                    add_fwd(g, st.sb[1].x_in, H, ap.qkv, H, 2 * H, (T*)st.sb[1].qkv + H, 3 * H, Rv, EPI_BIAS, nullptr, 0, nullptr, 0);
                }
            } else {
                for (int m = 0; m < 2; ++m) if (st.active[m])
                    add_fwd(g, st.sb[m].x_in, H, st.att[m]->qkv, 0, 3 * H, st.sb[m].qkv, 3 * H, seg_rows(m), EPI_BIAS, nullptr, 0, nullptr, 0);
            }
            CK(run_fwd(g, s));
        }
        {
            AttnArgs aa[2];
            for (int m = 0; m < 2; ++m) if (st.active[m]) {
                const int km = cross ? 1 - m : m;    // modality that provides keys / values
                AttnArgs& a = aa[m]; memset(&a, 0, sizeof a);
                a.q = st.sb[m].qkv; a.ldq = 3 * H;
                a.k = (T*)st.sb[km].qkv + H; a.v = (T*)st.sb[km].qkv + 2 * H; a.ldk = a.ldv = 3 * H;
                a.out = st.sb[m].ctx; a.ldo = H; a.out_b = img(st.sb[m].ctx);
                a.mask = (km == 0 && !fwd_varlen) ? maskf : nullptr;      // only language keys carry a padding mask (entry.py:119)
                a.cu_q = m == 0 ? cu : nullptr; a.cu_k = km == 0 ? cu : nullptr;  // packed language rows: the window IS the mask
                a.lse = st.sb[m].lse;
                a.B = B; a.nh = nh; a.Lq = seg_len(m); a.Lk = seg_len(km); a.dh = dh;
                a.scale = 1.0f / sqrtf((float)dh);
                a.drop = drop_base(pa); a.drop_site = st.site + m * 4;
            }
            auto fl = [&](const AttnArgs& a) { return 4.0 * B * nh * a.Lq * a.Lk * dh; };
            auto by = [&](const AttnArgs& a) { return sizeof(T) * (double)B * nh * dh * (2.0 * a.Lq + 2.0 * a.Lk); };
            int paired = 0;
            if (st.active[0] && st.active[1] && attn_pair_wanted()) {      // language | vision (or the two cross directions) in ONE launch
                prof_begin(PC_ATTN_FWD, fl(aa[0]) + fl(aa[1]), by(aa[0]) + by(aa[1]), s);
                paired = attn_fwd_pair_dispatch(aa[0], aa[1], s);
                if (paired != 0) prof_end(s); else prof_cancel();
                if (paired < 0) return paired;
            }
            if (paired == 0)
                for (int m = 0; m < 2; ++m) if (st.active[m]) {
                    prof_begin(PC_ATTN_FWD, fl(aa[m]), by(aa[m]), s);
                    int ra = attn_fwd_dispatch(aa[m], s);
                    prof_end(s);
                    CK(ra);
                }
        }
        {
            GemmGroup g; gg_init(g); g.drop = drop_base(pd);
            const bool lnf = ln_fuse_on();
            if (cross && st.active[1]) {
                add_fwd(g, st.sb[0].ctx, H, st.att[0]->o, 0, H, st.sb[0].z, H, R, EPI_RESID_DROP, st.sb[0].x_in, H, nullptr, st.site + 1);
                if (lnf) fuse_ln(g, st.att[0]->ln, st.sb[0].y, st.sb[0].mean, st.sb[0].rstd);
            } else {
                for (int m = 0; m < 2; ++m) if (st.active[m]) {
                    add_fwd(g, st.sb[m].ctx, H, st.att[m]->o, 0, H, st.sb[m].z, H, seg_rows(m), EPI_RESID_DROP, st.sb[m].x_in, H, nullptr, st.site + m * 4 + 1);
                    if (lnf) fuse_ln(g, st.att[m]->ln, st.sb[m].y, st.sb[m].mean, st.sb[m].rstd);
                }
            }
            CK(run_fwd(g, s));
            if (lnf) return RGQA_OK;
        }
        if (cross && st.active[1]) {
            CKP(PC_LN, k_ln_fwd<T>((T*)st.sb[0].z, H, P + st.att[0]->ln.w, P + st.att[0]->ln.b, (T*)st.sb[0].y, H, st.sb[0].mean, st.sb[0].rstd, R, H, cfg.ln_eps, s, img(st.sb[0].y)));
        } else if (ln_merge && st.active[0] && st.active[1] && Rl > 0 && Rv > 0 && (T*)st.sb[1].y == (T*)st.sb[0].y + (size_t)Rl * H) {
            CKP(PC_LN, k_ln_fwd2<T>((T*)st.sb[0].z, H, P + st.att[0]->ln.w, P + st.att[0]->ln.b, P + st.att[1]->ln.w, P + st.att[1]->ln.b, Rl,
                                    (T*)st.sb[0].y, H, st.sb[0].mean, st.sb[0].rstd, R, H, cfg.ln_eps, s, img(st.sb[0].y)));
        } else {
            for (int m = 0; m < 2; ++m) if (st.active[m])
                CKP(PC_LN, k_ln_fwd<T>((T*)st.sb[m].z, H, P + st.att[m]->ln.w, P + st.att[m]->ln.b, (T*)st.sb[m].y, H, st.sb[m].mean, st.sb[m].rstd, seg_rows(m), H, cfg.ln_eps, s, img(st.sb[m].y)));
        }
            return RGQA_OK;
    }

    // UNITER embeddings (uniter/modeling.py:560-612, 628-631) written straight into the joint row layout
    int forward_joint_embeddings(const float* feats, const float* boxes, const int64_t* ids, const int64_t* seg, const int64_t* mask, hipStream_t s) {
        const int H = cfg.hidden, ni = B * Oi;
        const float pd = cfg.hidden_dropout;
        if (!fwd_varlen) CKP(PC_OTHER, k_uniter_mask(mask, maskf, B, Tt, Oi, s));
        // text: word + position + token type -> LayerNorm -> dropout; pre-LN sums and statistics kept in text order for the backward
        CKP(PC_OTHER, k_embed_fwd<T>(ids, seg, trow_src_dev, text_dst_dev, n_text, P + mp.word, P + mp.pos, P + mp.type, P + mp.emb_ln.w, P + mp.emb_ln.b, emb_out, H, emb_z,
                                     emb_mean, emb_rstd, B, Tt, H, cfg.vocab_size, cfg.type_vocab, cfg.ln_eps, drop_site(pd, 1), s));
        CK(image_of(emb_z, H, n_text, H, s));
        // image: LN(img_linear(feat)) + LN(pos_linear(pos)) + type_emb[1] -> LayerNorm -> dropout
        GemmGroup g; gg_init(g);
        if (LP) CKP(PC_OTHER, cast_lp(feats, feats_lp, (size_t)ni * cfg.feat_dim, s, img(feats_lp)));
        add_fwd(g, LP ? (const void*)feats_lp : (const void*)feats, cfg.feat_dim, mp.visn_fc, 0, H, u_zf, H, ni, EPI_BIAS, nullptr, 0, nullptr, 0);
        CK(run_fwd(g, s));
        CKP(PC_LN, k_ln_fwd<T>(u_zf, H, P + mp.visn_ln.w, P + mp.visn_ln.b, u_a1, H, u_st, u_st + ni, ni, H, cfg.ln_eps, s));
        CKP(PC_OTHER, k_pos_proj<T>(boxes, cfg.pos_dim, P + mp.box_fc.w, P + mp.box_fc.b, u_zp, H, ni, H, s));
        CKP(PC_LN, k_ln_fwd<T>(u_zp, H, P + mp.box_ln.w, P + mp.box_ln.b, u_a2, H, u_st + 2 * ni, u_st + 3 * ni, ni, H, cfg.ln_eps, s));
        CKP(PC_LN, k_sum3_ln_fwd<T>(u_a1, u_a2, H, P + mp.type + H, P + mp.img_ln.w, P + mp.img_ln.b, img_dst_dev, emb_out, H, u_x3, u_st + 4 * ni, u_st + 5 * ni,
                                    ni, H, cfg.ln_eps, drop_site(pd, 2), s));
        // images of what the backward pass reads: the joint embedding rows (first layer's wgrad operand) and the pre-LayerNorm sums
        CK(image_of(emb_out, H, R, H, s)); CK(image_of(u_zp, H, ni, H, s)); CK(image_of(u_x3, H, ni, H, s));
        return RGQA_OK;
    }
    // dy = gradient w.r.t. the joint embedding rows
    int backward_joint_embeddings(const TB* dy, int accumulate, hipStream_t s) {
        const int H = cfg.hidden, ni = B * Oi;
        const float pd = cfg.hidden_dropout;
        const DropCfg nodrop = make_drop(0.f, 0, 0);
        RGQA_REQUIRE(dfeats_out == nullptr && dboxes_out == nullptr, "backward: input gradients are not available for the UNITER backbone");
        // text rows -> text order, LayerNorm backward, scatter into the tables (only the word table has padding_idx, :563-568)
        CKP(PC_OTHER, k_gather_rows<TB>(dy, H, text_dst_dev, 0, u_gt, H, n_text, H, s));
        CKP(PC_LN, k_ln_bwd<TB>(u_gt, H, sv(emb_z), H, P + mp.emb_ln.w, emb_mean, emb_rstd, u_de, nullptr, H, part, G + mp.emb_ln.w, G + mp.emb_ln.b, nullptr, accumulate, n_text, H,
                               nodrop, drop_site(pd, 1), 1.0f, s));
        CKP(PC_OTHER, k_embed_scatter<TB>(u_de, in_ids, in_seg, trow_src_dev, n_text, G + mp.word, G + mp.pos, G + mp.type, B, Tt, H, cfg.type_vocab, 0, accumulate, emb_keys, part, part_floats, s));
        // image rows: final LayerNorm, then the same gradient enters both branch LayerNorms and the type-1 embedding row
        CKP(PC_OTHER, k_gather_rows<TB>(dy, H, img_dst_dev, 0, u_g, H, ni, H, s));
        CKP(PC_LN, k_ln_bwd<TB>(u_g, H, sv(u_x3), H, P + mp.img_ln.w, u_st + 4 * ni, u_st + 5 * ni, u_dx3, nullptr, H, part, G + mp.img_ln.w, G + mp.img_ln.b, nullptr, accumulate, ni, H,
                               nodrop, drop_site(pd, 2), 1.0f, s));
        CKP(PC_OTHER, k_colsum<TB>(u_dx3, H, part, G + mp.type + H, 1, ni, H, s));       // the tables were zeroed (or hold the accumulated sum)
        CKP(PC_LN, k_ln_bwd<TB>(u_dx3, H, sv(u_zp), H, P + mp.box_ln.w, u_st + 2 * ni, u_st + 3 * ni, u_dz, nullptr, H, part, G + mp.box_ln.w, G + mp.box_ln.b, G + mp.box_fc.b, accumulate, ni, H,
                               nodrop, nodrop, 1.0f, s));
        CKP(PC_OTHER, k_pos_wgrad<TB>(u_dz, H, in_boxes, cfg.pos_dim, part, G + mp.box_fc.w, accumulate, ni, H, s));
        CKP(PC_LN, k_ln_bwd<TB>(u_dx3, H, sv(u_zf), H, P + mp.visn_ln.w, u_st, u_st + ni, u_dz, nullptr, H, part, G + mp.visn_ln.w, G + mp.visn_ln.b, G + mp.visn_fc.b, accumulate, ni, H,
                               nodrop, nodrop, 1.0f, s));
        GemmGroup g; gg_init(g);
        add_wgrad(g, u_dz, H, mp.visn_fc, 0, H, LP ? (const void*)sv(feats_lp) : (const void*)in_feats, cfg.feat_dim, ni, accumulate);
        return run_wgrad(g, s);
    }

    int forward(const float* feats, const float* boxes, const int64_t* ids, const int64_t* seg, const int64_t* mask,
                float* pooled_out, float* logits_out, int ld_logits, int train, uint64_t seed, hipStream_t s) override {
        RGQA_REQUIRE(P != nullptr && ws != nullptr, "forward: engine not bound");
        RGQA_REQUIRE(feats && boxes && ids && (mask || varlen), "forward: null input");
        const int H = cfg.hidden, I = cfg.inter, nh = cfg.heads, dh = H / nh;
        in_feats = feats; in_boxes = boxes; in_ids = ids; in_seg = seg; last_train = train; last_seed = seed;
        const float pd = cfg.hidden_dropout, pa = cfg.attn_dropout;
        prof_block = PB_EMBED;
        {   // language row layout of this pass: packed (varlen) or padded
            const int want = varlen ? n_lang : B * Tn;
            if (want != Rl) plan(B, Tn, O, want);
            if (joint) {
                // the row maps of the joint layout are needed in both layouts (padded = every text position is a row)
                if (!varlen && jstate != 1) { tlens_host.assign(B, Tt); lens_host.assign(B, Tn); lens_dirty = true; }
                if (lens_dirty) {
                    CK(k_set_lengths(lens_host.data(), B, Tn, lens_dev, cu_dev, row_src_dev, s));
                    CK(k_set_lengths(tlens_host.data(), B, Tt, tlens_dev, tcu_dev, trow_src_dev, s));
                    CK(k_uniter_dst(tcu_dev, cu_dev, B, Oi, text_dst_dev, img_dst_dev, s));
                    n_text = 0;
                    for (int v : tlens_host) n_text += v;
                    lens_dirty = false; jstate = varlen ? 2 : 1;
                }
            } else
            if (varlen && lens_dirty) { CK(k_set_lengths(lens_host.data(), B, Tn, lens_dev, cu_dev, row_src_dev, s)); lens_dirty = false; }
            fwd_varlen = varlen;
            fwd_cls_tail = cls_tail_wanted();
            fwd_z_in_place = g_rgqa_z_in_place != 0;
        }
        const int* cu = fwd_varlen ? cu_dev : nullptr;
        CK(wait_wready(n_seg_events - 1, s));         // embedding tables, visual projection (the segment backward finishes last)
        if (joint) CK(forward_joint_embeddings(feats, boxes, ids, seg, mask, s));
        else {
        if (!fwd_varlen) CKP(PC_OTHER, k_make_mask(mask, maskf, Rl, s));
        CKP(PC_OTHER, k_embed_fwd<T>(ids, seg, fwd_varlen ? row_src_dev : nullptr, nullptr, Rl, P + mp.word, P + mp.pos, P + mp.type, P + mp.emb_ln.w, P + mp.emb_ln.b, emb_out, H, emb_z,
                          emb_mean, emb_rstd, B, Tn, H, cfg.vocab_size, cfg.type_vocab, cfg.ln_eps, drop_site(pd, 1), s));
        CK(image_of(emb_out, H, Rl, H, s)); CK(image_of(emb_z, H, Rl, H, s));
        {   // VisualFeatEncoder: GEMM on the RoI features (one f32->bf16 cast pass in bf16 precision, so that this GEMM and
            // its weight-gradient GEMM run on the LDS-DMA kernels), then the fused LN/LN/avg tail
            GemmGroup g; gg_init(g);
            if (LP) CKP(PC_OTHER, cast_lp(feats, feats_lp, (size_t)Rv * cfg.feat_dim, s, img(feats_lp)));      // (MIXED: + the bf16 wgrad's operand, same pass)
            add_fwd(g, LP ? (const void*)feats_lp : (const void*)feats, cfg.feat_dim, mp.visn_fc, 0, H, zf, H, Rv, EPI_BIAS, nullptr, 0, nullptr, 0);
            CK(run_fwd(g, s));
            CKP(PC_OTHER, k_visn_combine_fwd<T>(zf, H, boxes, P + mp.box_fc.w, P + mp.box_fc.b, P + mp.visn_ln.w, P + mp.visn_ln.b, P + mp.box_ln.w, P + mp.box_ln.b,
                                     visn_out, H, visn_stats, Rv, H, cfg.pos_dim, cfg.ln_eps, drop_site(pd, 2), s));
            CK(image_of(visn_out, H, Rv, H, s));
        }
        }
        bool gathered = false;
        const size_t n_lr_stages = 2 * (size_t)(cfg.l_layers > cfg.r_layers ? cfg.l_layers : cfg.r_layers);
        for (size_t si = 0; si < stages.size(); ++si) {
            Stage& st = stages[si];
            prof_block = si < n_lr_stages ? PB_LR : PB_X;
            if (st.kind == ST_ATT_CROSS && x0_needed && !gathered) {
                if (x0_src[0] && Rl > 0) CK(rgqa_check_hip(hipMemcpyAsync(x0, x0_src[0], (size_t)Rl * H * sizeof(T), hipMemcpyDeviceToDevice, s), "x0 gather lang"));
                if (x0_src[1]) CK(rgqa_check_hip(hipMemcpyAsync(x0 + (size_t)Rl * H, x0_src[1], (size_t)Rv * H * sizeof(T), hipMemcpyDeviceToDevice, s), "x0 gather visn"));
                if (x0_src[0] && Rl > 0) CK(image_of(x0, H, Rl, H, s));
                if (x0_src[1]) CK(image_of(x0 + (size_t)Rl * H, H, Rv, H, s));
                gathered = true;
            }
            CK(forward_stage(st, cu, s));
        }
        // ---- BertPooler (modeling.py:575-581) + answer head (gqa_model.py:22-27)
        prof_block = PB_HEAD;
        CK(wait_wready(0, s));                        // pooler + answer head
        {
            GemmGroup g; gg_init(g);
            pool_in = cls_rows;
            if (!stages.empty() && cls_tail(stages.back())) pool_in = (T*)lang_final;      // already the B compact [CLS] rows
            else { CKP(PC_OTHER, k_gather_rows<T>((const T*)lang_final, H, cu, Tn, cls_rows, H, B, H, s)); CK(image_of(cls_rows, H, B, H, s)); }     // the [CLS] row of every sample
            add_fwd(g, pool_in, H, mp.pooler, 0, H, pooled, H, B, EPI_TANH, nullptr, 0, nullptr, 0);
            CK(run_fwd(g, s, 0, 0, true));
            gg_init(g);
            add_fwd(g, pooled, H, mp.head0, 0, 2 * H, h1, 2 * H, B, EPI_GELU, nullptr, 0, h1pre, 0);
            CK(run_fwd(g, s, 0, 0, true));
            CKP(PC_LN, k_ln_fwd<T>(h1, 2 * H, P + mp.head_ln.w, P + mp.head_ln.b, h2, 2 * H, hd_mean, hd_rstd, B, 2 * H, cfg.ln_eps, s, img(h2)));
            gg_init(g);
            // N = NAp (a multiple of 64): the arena slots of logit_fc.3.weight / .bias are zero-padded up to NAp rows, so the extra
            // logits columns come out as exact zeros and the GEMM runs on the LDS-DMA kernel
            add_fwd(g, h2, 2 * H, mp.head3, 0, LP ? NAp : cfg.num_answers, logits, NAp, B, EPI_BIAS, nullptr, 0, nullptr, 0);
            CK(run_fwd(g, s, 1, 0, true));
        }
        if (pooled_out) CKP(PC_OTHER, k_to_f32<T>(pooled, H, pooled_out, H, B, H, s));
        if (logits_out) CKP(PC_OTHER, k_fill_rows<float>(logits_out, ld_logits, logits, NAp, B, cfg.num_answers, s));
        have_fwd = true;
        return RGQA_OK;
    }

    // ------------------------------------------------------------------ backward
    int loss_backward(const float* target, int ldt, float* loss_out, float grad_scale, int accumulate, hipStream_t s) override {
        RGQA_REQUIRE(have_fwd, "loss_backward: no forward pass recorded");
        RGQA_REQUIRE(G != nullptr, "loss_backward: no gradient arena bound");
        // BCE on the f32 logits; dlogits written as f32 into `logits`' sibling then cast+padded to T
        prof_block = PB_HEAD;
        float* dl32 = part;   // scratch [B, NAp] f32
        RGQA_REQUIRE((size_t)B * NAp + B <= part_floats, "loss_backward: batch %d exceeds the scratch", B);
        CKP(PC_OTHER, k_bce_fwd_bwd(logits, NAp, target, ldt, loss_dev, dl32, NAp, B, cfg.num_answers, NAp, grad_scale, s, part + (size_t)B * NAp));
        if (loss_out) CK(rgqa_check_hip(hipMemcpyAsync(loss_out, loss_dev, sizeof(float), hipMemcpyDeviceToDevice, s), "loss copy"));
        CKP(PC_OTHER, k_cast_pad<TB>(dl32, NAp, dlogits, NAp, B, NAp, 1.0f, s));
        return backward_impl(accumulate, s);
    }
    int backward(const float* dl, int ldd, int accumulate, hipStream_t s) override {
        RGQA_REQUIRE(have_fwd, "backward: no forward pass recorded");
        RGQA_REQUIRE(G != nullptr && dl != nullptr, "backward: null gradient arena / dlogits");
        CKP(PC_OTHER, k_cast_pad<TB>(dl, ldd, dlogits, NAp, B, cfg.num_answers, 1.0f, s));
        return backward_impl(accumulate, s);
    }

    int colsum_bias(const void* dy, int ld, const Lin& l, int col0, int cols, int M, int accumulate, hipStream_t s) {
        return k_colsum<TB>((const TB*)dy + col0, ld, part, G + l.b + col0, accumulate, M, cols, s);
    }

    int backward_impl(int accumulate, hipStream_t s) {
        prof_block = PB_HEAD;
        CK(wait_bwd(s));
        const int H = cfg.hidden, I = cfg.inter, nh = cfg.heads, dh = H / nh, NA = cfg.num_answers;
        const float pd = cfg.hidden_dropout, pa = cfg.attn_dropout;
        const DropCfg nodrop = make_drop(0.f, 0, 0);
        if (!accumulate) {
            // table rows no token of the batch names and parameters that receive no gradient must be zero
            CK(rgqa_check_hip(hipMemsetAsync(G + mp.word, 0, sizeof(float) * (mp.emb_ln.w - mp.word), s), "zero embedding grads"));
            if (dead_end > dead_begin) CK(rgqa_check_hip(hipMemsetAsync(G + dead_begin, 0, sizeof(float) * (dead_end - dead_begin), s), "zero dead grads"));
        }
        GemmGroup g;
        // ---- head
        // padded columns of dlogits are exact zeros and the bias slot reserves round_up(NA, 64) elements
        CKP(PC_OTHER, colsum_bias(dlogits, NAp, mp.head3, 0, NAp, B, accumulate, s));
        // the head's and the pooler's weight gradients ride in the encoder's first deferred launch: their operands (dlogits, gp3, gp1 and
        // forward tensors) are not written again during backward
        gg_init(wg_head);
        add_wgrad(wg_head, dlogits, NAp, mp.head3, 0, NA, sv(h2), 2 * H, B, accumulate);
        gg_init(g); add_dgrad(g, dlogits, NAp, mp.head3, 0, NA, gp1, 2 * H, B, EPI_BIAS, nullptr, 0); CK(run_dgrad(g, s, true));
        CKP(PC_LN, k_ln_bwd<TB>(gp1, 2 * H, sv(h1), 2 * H, P + mp.head_ln.w, hd_mean, hd_rstd, gp2, nullptr, 2 * H, part, G + mp.head_ln.w, G + mp.head_ln.b, nullptr,
                       accumulate, B, 2 * H, nodrop, nodrop, 1.0f, s));
        CKP(PC_OTHER, k_dgelu_mul<TB>(gp2, sv(h1pre), gp3, (size_t)B * 2 * H, s));
        CKP(PC_OTHER, colsum_bias(gp3, 2 * H, mp.head0, 0, 2 * H, B, accumulate, s));
        add_wgrad(wg_head, gp3, 2 * H, mp.head0, 0, 2 * H, sv(pooled), H, B, accumulate);
        gg_init(g); add_dgrad(g, gp3, 2 * H, mp.head0, 0, 2 * H, gp1, H, B, EPI_DTANH, sv(pooled), H); CK(run_dgrad(g, s, true));   // gp1[B,H] = d(pooler pre-tanh)
        return backward_encoder(accumulate, s);
    }

    // encoder-only autograd entry: dL/dpooled supplied by the caller (LXRTEncoder used under a foreign head)
    int backward_pooled(const float* dpooled, int ld, int accumulate, hipStream_t s) override {
        RGQA_REQUIRE(have_fwd, "backward_pooled: no forward pass recorded");
        RGQA_REQUIRE(G != nullptr && dpooled != nullptr, "backward_pooled: null gradient arena / dpooled");
        const int H = cfg.hidden;
        CK(wait_bwd(s));
        if (!accumulate) {
            CK(rgqa_check_hip(hipMemsetAsync(G + mp.word, 0, sizeof(float) * (mp.emb_ln.w - mp.word), s), "zero embedding grads"));
            if (dead_end > dead_begin) CK(rgqa_check_hip(hipMemsetAsync(G + dead_begin, 0, sizeof(float) * (dead_end - dead_begin), s), "zero dead grads"));
            CK(rgqa_check_hip(hipMemsetAsync(G + mp.head0.w, 0, sizeof(float) * (arena_elems - mp.head0.w), s), "zero head grads"));
        }
        CKP(PC_OTHER, k_cast_pad<TB>(dpooled, ld, gp2, H, B, H, 1.0f, s));
        CKP(PC_OTHER, k_dtanh_mul<TB>(gp2, sv(pooled), gp1, (size_t)B * H, s));
        gg_init(wg_head);
        return backward_encoder(accumulate, s);
    }

    int backward_encoder(int accumulate, hipStream_t s) {
        prof_block = PB_HEAD;
        const int n_lr_stages = 2 * (cfg.l_layers > cfg.r_layers ? cfg.l_layers : cfg.r_layers);
        const int H = cfg.hidden, I = cfg.inter, nh = cfg.heads, dh = H / nh;
        const float pd = cfg.hidden_dropout, pa = cfg.attn_dropout;
        const DropCfg nodrop = make_drop(0.f, 0, 0);
        GemmGroup g;
        CKP(PC_OTHER, colsum_bias(gp1, H, mp.pooler, 0, H, B, accumulate, s));
        const int* cu = fwd_varlen ? cu_dev : nullptr;
        add_wgrad(wg_head, gp1, H, mp.pooler, 0, H, sv(pool_in), H, B, accumulate);
        // gradient w.r.t. the final hidden states: zero except the [CLS] rows of lang
        // Gradient w.r.t. the current stage's output, one pointer per modality: a stage moves only the modalities it computes to
        // the other buffer, so a modality that merely passes through (vision under the language-only layers) stays where it is
        // instead of being copied.  Stages that treat [language | vision] as one row range need the two adjacent; single-modality
        // stages come in pairs (attention + FFN), so adjacency is back whenever it is needed - adjacent() re-establishes it otherwise.
        TB* dyp[2] = {gA, gA + (size_t)Rl * H};
        TB* dxp[2] = {gB, gB + (size_t)Rl * H};
        auto adjacent = [&]() -> int {
            if (dyp[1] != dyp[0] + (size_t)Rl * H) {
                TB* want = dyp[0] + (size_t)Rl * H;
                int r = rgqa_check_hip(hipMemcpyAsync(want, dyp[1], (size_t)Rv * H * sizeof(TB), hipMemcpyDeviceToDevice, s), "grad rows adjacent");
                if (r) return r;
                dxp[1] = dyp[1]; dyp[1] = want;
            }
            if (dxp[1] != dxp[0] + (size_t)Rl * H) dxp[1] = dxp[0] + (size_t)Rl * H;      // the free halves always pair up again
            return RGQA_OK;
        };
        CK(rgqa_check_hip(hipMemsetAsync(dyp[0], 0, (size_t)R * H * sizeof(TB), s), "zero dy"));
        gg_init(g); add_dgrad(g, gp1, H, mp.pooler, 0, H, gp2, H, B, EPI_BIAS, nullptr, 0); CK(run_dgrad(g, s, true));
        const bool tail = !stages.empty() && cls_tail(stages.back());
        if (!tail) CKP(PC_OTHER, k_scatter_rows<TB>(gp2, H, dyp[0], H, cu, Tn, B, H, s));     // tail: the last FFN stage takes gp2 [B,H] as it is
        seg_cursor = 0;
        pend_n = 0; gg_init(wgm);      // (a backward pass that failed half-way leaves nothing behind)
        pend_marks = 1;          // head + pooler gradients are final with the first deferred launch, which carries their weight gradients

        // ---- encoder stages in reverse; weight-gradient GEMMs are collected per layer and launched once
        GemmGroup wg = wg_head; gg_init(wg_head);
        int par = 0; bool layer_open = false;
        // Where a layer's deferred wgrad GEMMs are launched: after every FFN stage - the launch then holds the attention wgrads of the layer
        // above and this layer's FFN wgrads, and runs beside this layer's LayerNorm / attention kernels instead of beside the next layer's FFN
        // GEMMs (-0.08 ms per step against one launch per layer, round 2)
        constexpr bool phase_ffn = true;
        int flushes = 0;
        // phase_ffn: the first launch holds the last layer's FFN only (no layer is complete yet); every later one completes the layer above
        auto flush_after = [&](const Stage& st) { return phase_ffn ? st.kind == ST_FFN : st.layer_first != 0; };
        auto flush_layer = [&](hipStream_t ss, bool force = false) -> int {
            int r = flush_wgrad(wg, par, ss, !phase_ffn || flushes > 0, force);
            ++flushes; par = (par + 1) % wgrad_sets(); layer_open = false;
            return r;
        };
        for (int si = (int)stages.size() - 1; si >= 0; --si) {
            Stage& st = stages[si];
            prof_block = si < n_lr_stages ? PB_LR : PB_X;
            if (!layer_open) {     // first stage (in backward order) of a layer
                if (set_pending(par)) CK(launch_pending(s));       // (cannot happen while NPAR >= 2 * WGRAD_MERGE_MAX)
                CK(wait_wgrad(par, s)); layer_open = true;
                fin_accumulate = accumulate;
                fin.begin(lnpart_s[par], LNPART_BLOCKS, H);     // LayerNorm-backward column sums: folded once per layer, with the layer's wgrad launch
            }
            TB* gz = gz_s[par][st.slot]; TB* gzd = gzd_s[par][st.slot]; TB* gqkv = gqkv_s[par][st.slot]; TB* gh = gh_s[par][st.slot];
            const bool cross = st.kind == ST_ATT_CROSS;
            const bool shared_all = cross && st.active[1];
            auto rowp = [&](TB* base, int m, int width) { return base + (size_t)(m == 0 ? 0 : Rl) * width; };
            if (st.kind == ST_FFN && tail && si == (int)stages.size() - 1) {
                // B compact [CLS] rows: LayerNorm / FFN backward on them, then the input gradient goes back to the [CLS] rows of a zeroed buffer
                const FfnP& f = *st.ffn[0];
                DropCfg d = drop_site(pd, st.site + 1);
                CKP(PC_LN, k_ln_bwd<TB>(gp2, H, svz(st.sb[0].z), H, P + f.ln.w, st.sb[0].mean, st.sb[0].rstd, gz, d.thresh ? gzd : nullptr, H, part, G + f.ln.w, G + f.ln.b, G + f.down.b,
                               accumulate, B, H, d, nodrop, 1.0f, s, &fin, z_in_place()));
                TB* gzm = d.thresh ? gzd : gz;
                gg_init(g); add_dgrad(g, gzm, H, f.down, 0, H, gh, I, B, EPI_DGELU, sv(st.sb[0].hpre), I); CK(run_dgrad(g, s, true));
                add_wgrad(wg, gzm, H, f.down, 0, H, sv(st.sb[0].h), I, B, accumulate);
                add_wgrad(wg, gh, I, f.up, 0, I, sv(tail_x), H, B, accumulate, true);
                gg_init(g); add_dgrad(g, gh, I, f.up, 0, I, tail_dx, H, B, EPI_ADD, gz, H); CK(run_dgrad(g, s, true));
                CK(rgqa_check_hip(hipMemsetAsync(dxp[0], 0, (size_t)Rl * H * sizeof(TB), s), "zero tail dx"));
                CKP(PC_OTHER, k_scatter_rows<TB>(tail_dx, H, dxp[0], H, cu, Tn, B, H, s));
                { TB* t = dyp[0]; dyp[0] = dxp[0]; dxp[0] = t; }
                if (flush_after(st)) CK(flush_layer(s));
                continue;
            }
            if (st.kind == ST_FFN) {
                const bool both = ln_merge && st.active[0] && st.active[1] && Rl > 0 && Rv > 0 && dyp[1] == dyp[0] + (size_t)Rl * H;
                if (both) {
                    const FfnP &f0 = *st.ffn[0], &f1 = *st.ffn[1];
                    CKP(PC_LN, k_ln_bwd2<TB>(dyp[0], H, svz(st.sb[0].z), H, st.sb[0].mean, st.sb[0].rstd, gz, gzd, H, part, H, accumulate,
                                            Rl, P + f0.ln.w, G + f0.ln.w, G + f0.ln.b, G + f0.down.b, drop_site(pd, st.site + 1),
                                            Rv, P + f1.ln.w, G + f1.ln.w, G + f1.ln.b, G + f1.down.b, drop_site(pd, st.site + 5), s, &fin, z_in_place()));
                }
                for (int m = 0; m < 2; ++m) if (st.active[m] && !both) {
                    const FfnP& f = *st.ffn[m];
                    DropCfg d = drop_site(pd, st.site + m * 4 + 1);
                    CKP(PC_LN, k_ln_bwd<TB>(dyp[m], H, svz(st.sb[m].z), H, P + f.ln.w, st.sb[m].mean, st.sb[m].rstd, rowp(gz, m, H),
                                   d.thresh ? rowp(gzd, m, H) : nullptr, H, part, G + f.ln.w, G + f.ln.b, G + f.down.b, accumulate, seg_rows(m), H, d, nodrop, 1.0f, s, &fin, z_in_place()));
                }
                TB* gzm = drop_base(pd).thresh ? gzd : gz;
                gg_init(g);
                for (int m = 0; m < 2; ++m) if (st.active[m])
                    add_dgrad(g, rowp(gzm, m, H), H, st.ffn[m]->down, 0, H, rowp(gh, m, I), I, seg_rows(m), EPI_DGELU, sv(st.sb[m].hpre), I);
                CK(run_dgrad(g, s));
                for (int m = 0; m < 2; ++m) if (st.active[m]) {
                    add_wgrad(wg, rowp(gzm, m, H), H, st.ffn[m]->down, 0, H, sv(st.sb[m].h), I, seg_rows(m), accumulate);
                    add_wgrad(wg, rowp(gh, m, I), I, st.ffn[m]->up, 0, I, sv(st.sb[m].x_in), H, seg_rows(m), accumulate, true);
                }
                gg_init(g);
                for (int m = 0; m < 2; ++m) if (st.active[m])
                    add_dgrad(g, rowp(gh, m, I), I, st.ffn[m]->up, 0, I, dxp[m], H, seg_rows(m), EPI_ADD, rowp(gz, m, H), H);
                CK(run_dgrad(g, s));
                // an inactive modality's gradient passes through untouched: its pointers simply do not move
                for (int m = 0; m < 2; ++m) if (st.active[m]) { TB* t = dyp[m]; dyp[m] = dxp[m]; dxp[m] = t; }
                if (flush_after(st)) CK(flush_layer(s));
                continue;
            }
            // ---- attention stage backward
            TB* gzm = drop_base(pd).thresh ? gzd : gz;
            if (shared_all) {
                const AttP& ap = *st.att[0];
                DropCfg d = drop_site(pd, st.site + 1);
                CK(adjacent());
                CKP(PC_LN, k_ln_bwd<TB>(dyp[0], H, svz(st.sb[0].z), H, P + ap.ln.w, st.sb[0].mean, st.sb[0].rstd, gz, d.thresh ? gzd : nullptr, H, part,
                               G + ap.ln.w, G + ap.ln.b, G + ap.o.b, accumulate, R, H, d, nodrop, 1.0f, s, &fin, z_in_place()));
                gg_init(g); add_dgrad(g, gzm, H, ap.o, 0, H, gctx, H, R, EPI_BIAS, nullptr, 0); CK(run_dgrad(g, s));
            } else {
                const bool both = ln_merge && st.active[0] && st.active[1] && Rl > 0 && Rv > 0 && dyp[1] == dyp[0] + (size_t)Rl * H;
                if (both) {
                    const AttP &a0 = *st.att[0], &a1 = *st.att[1];
                    CKP(PC_LN, k_ln_bwd2<TB>(dyp[0], H, svz(st.sb[0].z), H, st.sb[0].mean, st.sb[0].rstd, gz, gzd, H, part, H, accumulate,
                                            Rl, P + a0.ln.w, G + a0.ln.w, G + a0.ln.b, G + a0.o.b, drop_site(pd, st.site + 1),
                                            Rv, P + a1.ln.w, G + a1.ln.w, G + a1.ln.b, G + a1.o.b, drop_site(pd, st.site + 5), s, &fin, z_in_place()));
                }
                for (int m = 0; m < 2; ++m) if (st.active[m] && !both) {
                    const AttP& ap = *st.att[m];
                    DropCfg d = drop_site(pd, st.site + m * 4 + 1);
                    CKP(PC_LN, k_ln_bwd<TB>(dyp[m], H, svz(st.sb[m].z), H, P + ap.ln.w, st.sb[m].mean, st.sb[m].rstd, rowp(gz, m, H),
                                   d.thresh ? rowp(gzd, m, H) : nullptr, H, part, G + ap.ln.w, G + ap.ln.b, G + ap.o.b, accumulate, seg_rows(m), H, d, nodrop, 1.0f, s, &fin, z_in_place()));
                }
                gg_init(g);
                for (int m = 0; m < 2; ++m) if (st.active[m])
                    add_dgrad(g, rowp(gzm, m, H), H, st.att[m]->o, 0, H, rowp(gctx, m, H), H, seg_rows(m), EPI_BIAS, nullptr, 0);
                CK(run_dgrad(g, s));
            }
            // attention core backward -> gqkv (packed like the forward qkv buffer)
            if (cross && !st.active[1]) {
                // dead visn-query direction: lang rows get no dk/dv, visn rows get no dq
                CK(rgqa_check_hip(hipMemsetAsync(gqkv, 0, (size_t)R * 3 * H * sizeof(TB), s), "zero dqkv"));
            }
            {
                AttnArgs aa[2];
                for (int m = 0; m < 2; ++m) if (st.active[m]) {
                    const int km = cross ? 1 - m : m;
                    AttnArgs& a = aa[m]; memset(&a, 0, sizeof a);
                    a.q = sv(st.sb[m].qkv); a.ldq = 3 * H;
                    a.k = sv(st.sb[km].qkv) + H; a.v = sv(st.sb[km].qkv) + 2 * H; a.ldk = a.ldv = 3 * H;
                    a.mask = (km == 0 && !fwd_varlen) ? maskf : nullptr;
                    a.cu_q = m == 0 ? cu : nullptr; a.cu_k = km == 0 ? cu : nullptr;
                    a.lse = st.sb[m].lse;
                    a.dout = rowp(gctx, m, H); a.lddo = H;
                    a.dq = rowp(gqkv, m, 3 * H); a.dk = rowp(gqkv, km, 3 * H) + H; a.dv = rowp(gqkv, km, 3 * H) + 2 * H;
                    a.lddq = a.lddk = a.lddv = 3 * H;
                    a.B = B; a.nh = nh; a.Lq = seg_len(m); a.Lk = seg_len(km); a.dh = dh;
                    a.scale = 1.0f / sqrtf((float)dh);
                    a.drop = drop_base(pa); a.drop_site = st.site + m * 4;
                }
                auto fl = [&](const AttnArgs& a) { return 10.0 * B * nh * a.Lq * a.Lk * dh; };
                auto by = [&](const AttnArgs& a) { return sizeof(TB) * (double)B * nh * dh * (4.0 * a.Lq + 4.0 * a.Lk); };
                int paired = 0;
                if (st.active[0] && st.active[1] && attn_pair_wanted()) {      // the two problems write disjoint rows / columns of dqkv: one launch
                    prof_begin(PC_ATTN_BWD, fl(aa[0]) + fl(aa[1]), by(aa[0]) + by(aa[1]), s);
                    paired = attn_bwd_pair_dispatch(aa[0], aa[1], s);
                    if (paired != 0) prof_end(s); else prof_cancel();
                    if (paired < 0) return paired;
                }
                if (paired == 0)
                    for (int m = 0; m < 2; ++m) if (st.active[m]) {
                        prof_begin(PC_ATTN_BWD, fl(aa[m]), by(aa[m]), s);
                        int ra = attn_bwd_dispatch(aa[m], s);
                        prof_end(s);
                        CK(ra);
                    }
            }
            // weight / bias gradients (GEMMs deferred to the end of the layer)
            if (shared_all) {
                const AttP& ap = *st.att[0];
                add_wgrad(wg, gzm, H, ap.o, 0, H, sv(st.sb[0].ctx), H, R, accumulate);
                add_wgrad(wg, gqkv, 3 * H, ap.qkv, 0, 3 * H, sv(st.sb[0].x_in), H, R, accumulate, true);
            } else if (cross) {
                const AttP& ap = *st.att[0];
                add_wgrad(wg, gzm, H, ap.o, 0, H, sv(st.sb[0].ctx), H, Rl, accumulate);
                add_wgrad(wg, gqkv, 3 * H, ap.qkv, 0, H, sv(st.sb[0].x_in), H, Rl, accumulate, true);
                add_wgrad(wg, rowp(gqkv, 1, 3 * H) + H, 3 * H, ap.qkv, H, 2 * H, sv(st.sb[1].x_in), H, Rv, accumulate, true);
            } else {
                for (int m = 0; m < 2; ++m) if (st.active[m]) {
                    add_wgrad(wg, rowp(gzm, m, H), H, st.att[m]->o, 0, H, sv(st.sb[m].ctx), H, seg_rows(m), accumulate);
                    add_wgrad(wg, rowp(gqkv, m, 3 * H), 3 * H, st.att[m]->qkv, 0, 3 * H, sv(st.sb[m].x_in), H, seg_rows(m), accumulate, true);
                }
            }
            // the first layer's attention wgrads are the launch nothing of the encoder runs beside: start them here, beside this stage's
            // own QKV dgrad and the embedding backward (they read dqkv / dz, which are final), not after the dgrad
            // (-0.02 ms bf16, -0.12 ms bf16x3 per step, round 3)
            if (si == 0 && phase_ffn && layer_open) CK(flush_layer(s, true));
            // input gradient: dx = dqkv @ Wqkv + dz (residual path)
            gg_init(g);
            if (shared_all) {
                add_dgrad(g, gqkv, 3 * H, st.att[0]->qkv, 0, 3 * H, dxp[0], H, R, EPI_ADD, gz, H);
            } else if (cross) {
                add_dgrad(g, gqkv, 3 * H, st.att[0]->qkv, 0, H, dxp[0], H, Rl, EPI_ADD, gz, H);
                CK(run_dgrad(g, s));          // two launches: a grouped launch needs ONE epilogue to stay on the LDS-DMA kernels
                gg_init(g);
                add_dgrad(g, rowp(gqkv, 1, 3 * H) + H, 3 * H, st.att[0]->qkv, H, 2 * H, dxp[1], H, Rv, EPI_BIAS, nullptr, 0);
            } else {
                for (int m = 0; m < 2; ++m) if (st.active[m])
                    add_dgrad(g, rowp(gqkv, m, 3 * H), 3 * H, st.att[m]->qkv, 0, 3 * H, dxp[m], H, seg_rows(m), EPI_ADD, rowp(gz, m, H), H);
            }
            CK(run_dgrad(g, s));
            // a cross stage writes both modalities' input gradients (the dead last layer too: vision keys / values are live)
            for (int m = 0; m < 2; ++m) if (st.active[m] || cross) { TB* t = dyp[m]; dyp[m] = dxp[m]; dxp[m] = t; }
            if (flush_after(st)) CK(flush_layer(s));
        }
        prof_block = PB_LR;
        if (phase_ffn && layer_open) CK(flush_layer(s, true));        // the first layer's attention wgrads: the only launch nothing of the encoder runs beside
        if (wg.count > 0) CK(flush_wgrad(wg, par, s, false, true));   // (an encoder without stages: the head's problems alone)
        CK(launch_pending(s));
        CK(fin_flush(fin, fin_accumulate, s));
        fin.begin(nullptr, 0, 0);
        CK(run_wgrad(wg, s));
        // `par` is now the OLDER gradient-buffer set (its weight-gradient launch precedes the first layer's on the side stream):
        // join it here - its qkv buffer becomes the split-K scratch below - and let the first layer's weight gradients, still
        // running on the side stream, overlap the embedding backward; they are joined at the end.
        CK(wait_wgrad(par, s));
        prof_block = PB_EMBED;
        if (joint) {
            CK(backward_joint_embeddings(dyp[0], accumulate, s));
            CK(mark_segment(s));
            CK(join_wgrad(par, s));
            return RGQA_OK;
        }
        TB* gz = gemb;
        // ---- embeddings: dropout -> LN backward -> scatter-add into the three tables
        {
            DropCfg din = drop_site(pd, 1);
            CKP(PC_LN, k_ln_bwd<TB>(dyp[0], H, sv(emb_z), H, P + mp.emb_ln.w, emb_mean, emb_rstd, gz, nullptr, H, part, G + mp.emb_ln.w, G + mp.emb_ln.b, nullptr, accumulate, Rl, H,
                           nodrop, din, 1.0f, s));
            CKP(PC_OTHER, k_embed_scatter<TB>(gz, in_ids, in_seg, fwd_varlen ? row_src_dev : nullptr, Rl, G + mp.word, G + mp.pos, G + mp.type, B, Tn, H, cfg.type_vocab, 1, accumulate, emb_keys, part, part_floats, s));
        }
        // ---- visual embedding
        {
            TB* dyv = dyp[1];
            TB* dzf = gz + (size_t)Rl * H;
            CKP(PC_OTHER, k_visn_combine_bwd<TB>(dyv, H, sv(zf), H, in_boxes, P + mp.box_fc.w, P + mp.box_fc.b, P + mp.visn_ln.w, P + mp.box_ln.w, visn_stats, dzf, H, part,
                                     G + mp.visn_ln.w, G + mp.visn_ln.b, G + mp.box_ln.w, G + mp.box_ln.b, G + mp.visn_fc.b, G + mp.box_fc.w, G + mp.box_fc.b,
                                     accumulate, Rv, H, cfg.pos_dim, drop_site(pd, 2), dboxes_out, s));
            // dW_visn_fc [H, feat_dim] contracts over all B*O rows but has only (H/256)*(feat_dim/256) = 24 output tiles, and nothing
            // is left to run beside it: split the contraction S ways into f32 partials (one grouped launch, S*24 tiles) and fold
            // them in a fixed order.  The scratch is the qkv-gradient buffer of the older buffer set (joined above).
            const size_t wsz = (size_t)H * cfg.feat_dim;
            int S = Rv / 1024; if (S > 8) S = 8;
            const size_t scratch_bytes = (size_t)(Rl > 0 ? B * Tn + Rv : Rv) * 3 * H * sizeof(TB);
            while (S > 1 && (size_t)S * wsz * sizeof(float) > scratch_bytes) --S;
            if (LP && S >= 2 && (wsz % 4) == 0) {
                float* part_w = reinterpret_cast<float*>(gqkv_s[par][0]);
                const int kc = (Rv / S) / 64 * 64;
                gg_init(g);
                for (int i = 0; i < S; ++i) {
                    GemmProblem& p = g.p[g.count++];
                    memset(&p, 0, sizeof p);
                    const int r0 = i * kc, rows = (i == S - 1) ? Rv - r0 : kc;
                    p.A = dzf + (size_t)r0 * H; p.lda = H; p.B = sv(feats_lp) + (size_t)r0 * cfg.feat_dim; p.ldb = cfg.feat_dim;
                    p.K = rows; p.M = H; p.N = cfg.feat_dim; p.C = part_w + (size_t)i * wsz; p.ldc = cfg.feat_dim; p.epi = EPI_BIAS;
                }
                CK(run_wgrad(g, s));
                CKP(PC_OTHER, k_sum_partials(part_w, S, wsz, G + mp.visn_fc.w, accumulate, s));
            } else {
                gg_init(g);
                add_wgrad(g, dzf, H, mp.visn_fc, 0, H, LP ? (const void*)sv(feats_lp) : (const void*)in_feats, cfg.feat_dim, Rv, accumulate);
                CK(run_wgrad(g, s));
            }
            if (dfeats_out) {       // input gradient dL/dfeats [B*O, feat_dim] f32 = dzf . W_visn_fc   (ODIN, tasks/gqa_odin.py:97-121)
                gg_init(g);
                add_dgrad(g, dzf, H, mp.visn_fc, 0, H, dfeats_out, cfg.feat_dim, Rv, EPI_BIAS, nullptr, 0);
                double f, b; gemm_work_t<TB>(g, f, b);
                prof_begin(PC_GEMM_NT_D, f, b, s);
                int r = nt_gemm_b(g, 1, 1, s);
                prof_end(s);
                CK(r);
            }
        }
        // embeddings + visual embedding: every gradient of this segment was computed on the caller's stream, so its event and its share of the clip
        // norm (23 us for the tables) need not wait for the side stream - they run while the side stream finishes the first layer's shares;
        // only then is everything on the side stream joined to the caller's stream (round 5: the other order left 50 us of the step idle)
        CK(mark_segment(s));
        CK(join_wgrad(par, s));
        return RGQA_OK;
    }

    // cross-attention probabilities of x-layer `layer` (reference lxrt_vis/modeling.py:458-462: l2v = language queries over
    // vision keys, v2l = vision queries over language keys), recomputed from the qkv buffer the last forward left behind
    int get_cross_attention(int layer, int direction, float* out, size_t cap, hipStream_t s) override {
        RGQA_REQUIRE(have_fwd, "get_cross_attention: no forward pass recorded");
        RGQA_REQUIRE(layer >= 0 && layer < cfg.x_layers && (direction == 0 || direction == 1), "get_cross_attention: layer %d / direction %d out of range", layer, direction);
        const int H = cfg.hidden, nh = cfg.heads, dh = H / nh;
        const int nlr = cfg.l_layers > cfg.r_layers ? cfg.l_layers : cfg.r_layers;
        Stage& st = stages[2 * nlr + 3 * layer];
        RGQA_REQUIRE(st.kind == ST_ATT_CROSS, "get_cross_attention: internal stage table mismatch");
        const int m = direction, km = 1 - m;
        RGQA_REQUIRE(cap >= (size_t)B * nh * seg_len(m) * seg_len(km), "get_cross_attention: buffer too small");
        if (direction == 1 && !st.active[1]) {
            // last cross layer, vision queries over language keys: the dead branch of mode 'x' (its output reaches nothing, so the
            // forward pass skips it).  lxrt_vis still returns its probabilities: project the vision queries and the language keys
            // now, into the columns of the stage's qkv buffer that the live directions leave unused.
            const AttP& ap = *st.att[0];
            GemmGroup g; gg_init(g);
            add_fwd(g, st.sb[1].x_in, H, ap.qkv, 0, H, st.sb[1].qkv, 3 * H, Rv, EPI_BIAS, nullptr, 0, nullptr, 0);
            if (Rl > 0) add_fwd(g, st.sb[0].x_in, H, ap.qkv, H, H, (T*)st.sb[0].qkv + H, 3 * H, Rl, EPI_BIAS, nullptr, 0, nullptr, 0);
            int r = run_fwd(g, s);
            if (r) return r;
        }
        AttnArgs a; memset(&a, 0, sizeof a);
        a.q = st.sb[m].qkv; a.ldq = 3 * H;
        a.k = (T*)st.sb[km].qkv + H; a.ldk = 3 * H;
        a.mask = (km == 0 && !fwd_varlen) ? maskf : nullptr;
        const int* cu = fwd_varlen ? cu_dev : nullptr;
        a.cu_q = m == 0 ? cu : nullptr; a.cu_k = km == 0 ? cu : nullptr;
        a.B = B; a.nh = nh; a.Lq = seg_len(m); a.Lk = seg_len(km); a.dh = dh;
        a.scale = 1.0f / sqrtf((float)dh);
        return k_attn_probs<T>(a, out, s);
    }

    int get_activation(const char* name, float* out, size_t cap, hipStream_t s) override {
        RGQA_REQUIRE(have_fwd, "get_activation: no forward pass recorded");
        const int H = cfg.hidden;
        const void* src = nullptr; size_t n = 0;
        std::string nm(name);
        auto stage_out = [&](int idx, int m) { src = stages[idx].sb[m].y; n = (size_t)seg_rows(m) * H; };
        const int nlr = cfg.l_layers > cfg.r_layers ? cfg.l_layers : cfg.r_layers;
        int i = -1;
        if (nm == "embed_lang") { src = emb_out; n = (size_t)Rl * H; }
        else if (nm == "embed_visn") { src = visn_out; n = (size_t)Rv * H; }
        else if (nm == "pooled") { src = pooled; n = (size_t)B * H; }
        else if (sscanf(name, "l%d", &i) == 1 && nm[0] == 'l' && i >= 0 && i < cfg.l_layers) stage_out(2 * i + 1, 0);
        else if (sscanf(name, "r%d", &i) == 1 && nm[0] == 'r' && i >= 0 && i < cfg.r_layers) stage_out(2 * i + 1, 1);
        else if (sscanf(name, "x%d_", &i) == 1 && nm[0] == 'x' && i >= 0 && i < cfg.x_layers) {
            const bool visn = nm.find("_visn") != std::string::npos;
            RGQA_REQUIRE(!(visn && i == cfg.x_layers - 1), "get_activation: %s is the dead branch in mode 'x' and is not computed", name);
            stage_out(2 * nlr + 3 * i + 2, visn ? 1 : 0);
            if (!visn && i == cfg.x_layers - 1 && cls_tail(stages.back())) n = (size_t)B * H;      // only the B [CLS] rows exist (compact)
        }
        RGQA_REQUIRE(src != nullptr, "get_activation: unknown activation '%s'", name);
        RGQA_REQUIRE(cap == n, "get_activation: '%s' holds %zu elements (%zu rows), the buffer %zu - size it from the layout of the recorded pass (packed rows; only the B [CLS] rows of the last language output)", name, n, n / H, cap);
        return k_to_f32<T>((const T*)src, H, out, H, (int)(n / H), H, s);
    }
};

EngineBase* make_engine(const rgqa_config& cfg) {
    if (cfg.precision == RGQA_PRECISION_BF16) return new Engine<bf16_t>(cfg);
    if (cfg.precision == RGQA_PRECISION_BF16X3) return new Engine<sf32>(cfg);
    if (cfg.precision == RGQA_PRECISION_BF16X3_FWD) return new Engine<sf32, true>(cfg);
    return new Engine<float>(cfg);
}
