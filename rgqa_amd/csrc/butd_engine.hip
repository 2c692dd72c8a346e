// BUTD GQA engine (reference src/butd/butd.py:108-221 GQABUTD; BASELINE config 5): same EngineBase /
// C-ABI surface as the LXMERT engine (rgqa_config.arch = 1).  input_ids carries the front-padded dictionary tokens
// [B, L]; segment_ids / input_mask are ignored.  The projections run on the shared GEMM kernels (bf16 MFMA or exact f32),
// weight-norm (scalar g) is folded into an effective-weight copy refreshed by sync_weights().
#include "engine.h"
#include "butd.h"
#include <string.h>

static inline size_t rupb(size_t x, size_t a) { return (x + a - 1) / a * a; }

__global__ void butd_bias_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = accumulate ? dst[i] + src[i] : src[i];
}
__global__ void butd_sum_small_kernel(const float* __restrict__ src, int n, float* __restrict__ dst, int accumulate) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += src[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) { const float v = red[0] + red[1] + red[2] + red[3]; dst[0] = accumulate ? dst[0] + v : v; }
}

struct BLin {            // one linear layer: V (or W) [out, in], optional scalar g, bias [out]
    size_t v, g, b; int out, in, kp, op; bool wn;
    size_t eff, efft;    // byte offsets of the effective-weight copies in the workspace
    size_t dwo;          // byte offset of its dW scratch [nsplit][out, kp] f32 (the grouped weight-gradient launch)
    size_t bpo;          // byte offset of its bias-gradient partials [nsplit][out] (nsplit > 1)
    int nsplit;          // row slices its weight-gradient contraction is cut into (1: whole)
    int norm;            // index of its ||V||^2 scalar
};

template <typename T>
class ButdEngine : public EngineBase {
public:
    static constexpr bool LP = !std::is_same<T, float>::value;
    static constexpr bool X3 = std::is_same<T, sf32>::value;       // split-f32 activations and effective-weight copies (bf16x3 precision)
    // row pitches / padded contraction lengths: multiples of 64 elements for the low-precision engines - whole K-steps of the LDS-DMA GEMM kernels
    // (until round 5 bf16 padded to 8: K = 304 (GloVe), 2056 (RoI features + box) and 1848 (answers) sent four GEMMs of the step to the 128 x 128
    // register-staged kernel, 0.38 ms) and whole 128-byte lines of split f32
    static constexpr int LDM = LP ? 64 : 8;
    static int nt_gemm(GemmGroup& g, int out_f32, int trans_b, hipStream_t s) {
        if constexpr (X3) return launch_gemm_nt_x3(g, out_f32, s);
        else if constexpr (LP) return launch_gemm_nt_bf16(g, out_f32, s);
        else return launch_gemm_f32(g, 0, trans_b, s);
    }
    static int tn_gemm(GemmGroup& g, hipStream_t s) {
        if constexpr (X3) return launch_gemm_tn_x3(g, s);
        else if constexpr (LP) return launch_gemm_tn_bf16(g, 1, s);
        else return launch_gemm_f32(g, 1, 1, s);
    }
    size_t emb = 0;
    BLin wih, whh, ip, qp, lin, qproj, iproj, c0, c3;
    std::vector<BLin*> lins;
    int H, E, Ep, D, Dp, NA, NAp;
    // bound
    float* P = nullptr; float* G = nullptr; char* ws = nullptr; size_t ws_used = 0; bool dry = false;
    int B = 0, L = 0, O = 0;
    const float* in_feats = nullptr; const float* in_boxes = nullptr; const int64_t* in_toks = nullptr;
    int last_train = 0; uint64_t last_seed = 0; bool have_fwd = false;
    // workspace
    float *c3_bias_pad = nullptr;        // bf16x3: ans_classifier.3.bias zero-padded to op entries
    float *sumsq = nullptr, *sumsq_scr = nullptr, *partial = nullptr, *wlin_eff = nullptr, *att = nullptr, *logits = nullptr, *dwscr = nullptr, *dw_part = nullptr, *db_part = nullptr, *loss_dev = nullptr;
    T *X = nullptr, *GI = nullptr, *GH = nullptr, *Hall = nullptr, *Rg = nullptr, *Zg = nullptr, *Ng = nullptr, *GHN = nullptr;
    T *IF = nullptr, *IP = nullptr, *QP = nullptr, *IE = nullptr, *QR = nullptr, *IR = nullptr, *J = nullptr, *C1 = nullptr, *dlogits = nullptr;
    T *dC1 = nullptr, *dJ = nullptr, *dQR = nullptr, *dIR = nullptr, *dIE = nullptr, *dIP = nullptr, *dQP = nullptr, *dq = nullptr, *dHa = nullptr, *dHb = nullptr,
      *dGI = nullptr, *dGH = nullptr, *dX = nullptr, *tmpH = nullptr;

    explicit ButdEngine(const rgqa_config& c) {
        cfg = c;
        H = c.hidden; E = c.emb_dim; Ep = (int)rupb(E, LDM); D = c.feat_dim + c.pos_dim; Dp = (int)rupb(D, LDM); NA = c.num_answers; NAp = (int)rupb(NA, 64);
        size_t cur = 0;
        auto add = [&](const std::string& name, long d0, long d1, int nd) {
            cur = rupb(cur, 64);
            ParamInfo p; p.name = name; p.offset = cur; p.ndim = nd; p.shape[0] = d0; p.shape[1] = d1; p.is_linear_weight = 0; p.dead_in_x_mode = 0;
            params.push_back(p);
            size_t n = nd == 0 ? 1 : (size_t)d0 * (nd == 2 ? (size_t)d1 : 1);
            size_t off = cur; cur += n; return off;
        };
        emb = add("w_emb.emb.weight", c.vocab_size, E, 2);
        int nnorm = 0;
        auto plain = [&](BLin& l, const std::string& wname, const std::string& bname, int o, int i) {
            l.out = o; l.in = i; l.kp = (int)rupb(i, LDM); l.op = (int)rupb(o, 64); l.wn = false; l.g = 0; l.norm = -1;
            l.v = add(wname, o, i, 2);
            (void)bname;
        };
        plain(wih, "q_enc.rnn.weight_ih_l0", "", 3 * H, E);
        plain(whh, "q_enc.rnn.weight_hh_l0", "", 3 * H, H);
        wih.b = add("q_enc.rnn.bias_ih_l0", 3 * H, 0, 1);
        whh.b = add("q_enc.rnn.bias_hh_l0", 3 * H, 0, 1);
        auto wn = [&](BLin& l, const std::string& name, int o, int i) {
            l.out = o; l.in = i; l.kp = (int)rupb(i, LDM); l.op = (int)rupb(o, 64); l.wn = true; l.norm = nnorm++;
            l.b = add(name + ".bias", o, 0, 1);
            l.g = add(name + ".weight_g", 0, 0, 0);
            l.v = add(name + ".weight_v", o, i, 2);
        };
        wn(ip, "att.image_proj.mlp.0", H, D);
        wn(qp, "att.question_proj.mlp.0", H, H);
        wn(lin, "att.linear", 1, H);
        wn(qproj, "q_project.mlp.0", H, H);
        wn(iproj, "img_project.mlp.0", H, D);
        wn(c0, "ans_classifier.0", 2 * H, H);
        wn(c3, "ans_classifier.3", NA, 2 * H);
        arena_elems = rupb(cur, 64);
        dead_begin = dead_end = 0;
        lins = {&wih, &whh, &ip, &qp, &lin, &qproj, &iproj, &c0, &c3};
        grad_segs.push_back({0, arena_elems, 0});
    }

    template <typename U> U* take(size_t n) {
        size_t bytes = rupb(n * sizeof(U), 256);
        U* p = dry ? nullptr : reinterpret_cast<U*>(ws + ws_used);
        ws_used += bytes;
        return p;
    }
    void plan(int B_, int L_, int O_) {
        B = B_; L = L_; O = O_;
        ws_used = 0;
        sumsq = take<float>(64); loss_dev = take<float>(64); sumsq_scr = take<float>(1088);      // k_sumsq's partials + ticket: not shared with the column-sum scratch
        { size_t pw = 3 * (size_t)H; if ((size_t)NAp > pw) pw = NAp; partial_floats = 256 * pw; const size_t eg = (size_t)64 * cdiv(B * L, 256) * E; if (eg > partial_floats) partial_floats = eg;
          partial = take<float>(partial_floats); }      // (column-sum scratch; also the hot words' partial sums of the embedding gradient: csrc/embed.hip)
        for (BLin* l : lins) {
            l->eff = ws_used; take<T>((size_t)l->op * l->kp);          // op >= out rows: the rows past `out` stay zero (bf16x3: the logits GEMM runs over op columns)
            l->efft = ws_used; take<T>((size_t)l->kp * l->op);
        }
        wlin_eff = take<float>(H); c3_bias_pad = take<float>(c3.op);
        X = take<T>((size_t)B * L * Ep); GI = take<T>((size_t)B * L * 3 * H); GH = take<T>((size_t)B * 3 * H);
        Hall = take<T>((size_t)(L + 1) * B * H);
        Rg = take<T>((size_t)L * B * H); Zg = take<T>((size_t)L * B * H); Ng = take<T>((size_t)L * B * H); GHN = take<T>((size_t)L * B * H);
        IF = take<T>((size_t)B * O * Dp); IP = take<T>((size_t)B * O * H); QP = take<T>((size_t)B * H); att = take<float>((size_t)B * O);
        IE = take<T>((size_t)B * Dp); QR = take<T>((size_t)B * H); IR = take<T>((size_t)B * H); J = take<T>((size_t)B * H); C1 = take<T>((size_t)B * 2 * H);
        logits = take<float>((size_t)B * NAp); dlogits = take<T>((size_t)B * NAp);
        dC1 = take<T>((size_t)B * 2 * H); dJ = take<T>((size_t)B * H); dQR = take<T>((size_t)B * H); dIR = take<T>((size_t)B * H); dIE = take<T>((size_t)B * Dp);
        dIP = take<T>((size_t)B * O * H); dQP = take<T>((size_t)B * H); dq = take<T>((size_t)B * H); dHa = take<T>((size_t)B * H); dHb = take<T>((size_t)B * H);
        tmpH = take<T>((size_t)B * H);
        dGI = take<T>((size_t)B * L * 3 * H); dGH = take<T>((size_t)L * B * 3 * H); dX = take<T>((size_t)B * L * Ep);
        size_t mw = 0;
        for (BLin* l : lins) { size_t n = (size_t)l->out * l->kp; if (n > mw) mw = n; }
        dwscr = take<float>(mw); dw_part = take<float>((size_t)B * H); db_part = take<float>(B);
        gru_cnt = take<int>(gru_persist_counter_ints(B, L)); emb_keys = take<int>(2 * ((size_t)B * L + 4) + 128);
        // the three long contractions of the pass - W_hh and W_ih over L * B rows, image_proj over B * O rows - are cut into 4 row slices each: as whole
        // problems their 108 output tiles walked 144-160 K-steps on an otherwise idle chip (one launch of 280 us: as long as its longest chain)
        for (BLin* l : lins) {
            const long rows = (l == &whh || l == &wih) ? (long)L * B : (l == &ip ? (long)B * O : (long)B);
            l->nsplit = (LP && rows >= 4096) ? 4 : 1;
            l->dwo = ws_used; take<float>((size_t)l->nsplit * l->out * l->kp);
            l->bpo = ws_used; take<float>((size_t)l->nsplit * l->out);
        }
        wn_dev = take<WnDesc>(16); wn_partial = take<float>(1024);
        gemm_ws_floats = LP ? (size_t)GEMM_NT_MAX_PROBLEMS * 256 * 2112 : 0;        // split-K scratch of the M = B GEMMs (csrc/gemm_mfma256.hip)
        gemm_ws = LP ? take<float>(gemm_ws_floats) : nullptr;
    }
    WnDesc* wn_dev = nullptr; float* wn_partial = nullptr; std::vector<WnDesc> wn_host; int wn_blocks = 0, wn_tiles = 0; bool wn_uploaded = false;
    float* gemm_ws = nullptr; size_t gemm_ws_floats = 0;
    float* dwp(const BLin& l) const { return reinterpret_cast<float*>(ws + l.dwo); }
    // the whh weight gradient goes straight into the gradient arena (no weight norm, kp == in)
    bool dw_direct(const BLin& l) const { return !l.wn && l.kp == l.in && l.nsplit == 1; }
    float* bpp(const BLin& l) const { return reinterpret_cast<float*>(ws + l.bpo); }
    void build_wn_table() {
        wn_host.clear(); wn_blocks = 0; wn_tiles = 0;
        for (BLin* l : lins) {
            WnDesc d; memset(&d, 0, sizeof d);
            d.v = P + l->v; d.g = l->wn ? P + l->g : nullptr; d.sumsq = l->wn ? sumsq + l->norm : nullptr;
            d.eff = effp(*l); d.efft = LP ? (void*)efftp(*l) : nullptr; d.eff_f32 = (l == &lin) ? wlin_eff : nullptr;
            const bool direct = dw_direct(*l);
            d.dw = (G && direct) ? G + l->v : dwp(*l); d.lddw = direct ? l->in : l->kp; d.dv = G ? G + l->v : nullptr; d.dgp = (G && l->wn) ? G + l->g : nullptr;
            d.out = l->out; d.in = l->in; d.kp = l->kp; d.op = l->op;
            d.nsplit = l->nsplit; d.split_stride = (size_t)l->out * l->kp;
            d.bpart = (l->nsplit > 1) ? bpp(*l) : nullptr; d.db = (G && l->nsplit > 1) ? G + l->b : nullptr;
            const long n = (long)l->out * l->in;
            d.blk0 = wn_blocks; d.nblk = (int)((n + 8191) / 8192); if (d.nblk > 100) d.nblk = 100; if (d.nblk < 1) d.nblk = 1;
            wn_blocks += d.nblk;
            const int kk = l->kp > l->in ? l->kp : l->in, nn = LP ? (l->op > l->out ? l->op : l->out) : l->out;
            d.tile0 = wn_tiles; d.tiles_x = cdiv(kk, 32); d.tiles_y = cdiv(nn, 32);
            wn_tiles += d.tiles_x * d.tiles_y;
            wn_host.push_back(d);
        }
        wn_uploaded = false;
    }
    int* gru_cnt = nullptr; int* emb_keys = nullptr; size_t partial_floats = 0;
    bool gru_persist() const { if constexpr (LP && !X3) return gru_persist_ok(B, H); else return false; }
    size_t workspace_bytes(int B_, int L_, int O_) override { dry = true; plan(B_, L_, O_); dry = false; return ws_used + 256; }
    int bind(float* p, float* g, void*, void*, void* w, size_t wb, int B_, int L_, int O_) override {
        RGQA_REQUIRE(p != nullptr && w != nullptr, "bind: null parameter arena or workspace");
        RGQA_REQUIRE(B_ > 0 && L_ > 0 && O_ > 0 && O_ <= 64, "bind: B=%d L=%d O=%d unsupported (O <= 64)", B_, L_, O_);
        size_t need = workspace_bytes(B_, L_, O_);
        if (wb < need) { rgqa_set_error("bind: workspace too small (%zu < %zu bytes)", wb, need); return RGQA_ERR_WORKSPACE; }
        P = p; G = g; ws = (char*)w;
        plan(B_, L_, O_);
        RGQA_HIP(hipMemset(gru_cnt, 0, sizeof(int) * gru_persist_counter_ints(B, L)));       // incl. the error word of the persistent GRU launches
        build_wn_table();
        RGQA_REQUIRE(wn_blocks <= 1024 && (int)wn_host.size() <= 16, "bind: weight-norm tables too small (%d blocks, %zu layers)", wn_blocks, wn_host.size());
        have_fwd = false; eff_zeroed = false;
        return RGQA_OK;
    }
    bool eff_zeroed = false;
    T* effp(const BLin& l) const { return reinterpret_cast<T*>(ws + l.eff); }
    T* efftp(const BLin& l) const { return reinterpret_cast<T*>(ws + l.efft); }

#define CKB(x) do { int _r = (x); if (_r) return _r; } while (0)
    int sync_weights(hipStream_t s) override {
        RGQA_REQUIRE(P != nullptr, "sync_weights: engine not bound");
        if (!eff_zeroed) {   // padding rows / columns of the effective copies must be exact zeros
            for (BLin* l : lins) {
                CKB(rgqa_check_hip(hipMemsetAsync(effp(*l), 0, sizeof(T) * (size_t)l->op * l->kp, s), "zero eff"));
                CKB(rgqa_check_hip(hipMemsetAsync(efftp(*l), 0, sizeof(T) * (size_t)l->kp * l->op, s), "zero efft"));
            }
            eff_zeroed = true;
        }
        // every layer's ||V||^2 and effective copies in two launches (butd_kernels.hip, "weight norm ... ONE launch per phase")
        if (!wn_uploaded) {
            CKB(rgqa_check_hip(hipMemcpyAsync(wn_dev, wn_host.data(), sizeof(WnDesc) * wn_host.size(), hipMemcpyHostToDevice, s), "weight-norm table"));
            wn_uploaded = true;
        }
        CKB(kb_wn_forward_group(wn_dev, (int)wn_host.size(), wn_blocks, wn_tiles, wn_partial, X3 ? 2 : (LP ? 1 : 0), s));
        if (LP) {
            CKB(rgqa_check_hip(hipMemsetAsync(c3_bias_pad, 0, sizeof(float) * c3.op, s), "zero padded bias"));
            CKB(rgqa_check_hip(hipMemcpyAsync(c3_bias_pad, P + c3.b, sizeof(float) * c3.out, hipMemcpyDeviceToDevice, s), "padded bias"));
        }
        return RGQA_OK;
    }
    // (every forward pass re-derives the effective weights - and ||V|| - from the parameters as they are then: nothing to refresh after an optimizer
    // step.  Until round 5 this ran the whole weight-norm pass a second time per train step.)
    int sync_transposed(hipStream_t) override { return RGQA_OK; }

    // ------------------------------------------------------------------ GEMM wrappers (single problem)
    static void ginit(GemmGroup& g) { memset(&g, 0, sizeof g); g.count = 1; g.drop = make_drop(0.f, 0, 0); }
    int gemm_fwd(const void* x, int ldx, int M, const BLin& l, void* C, int ldc, int epi, int out_f32, DropCfg drop, uint32_t site, hipStream_t s) {
        GemmGroup g; ginit(g); g.drop = drop;
        GemmProblem& p = g.p[0];
        p.A = x; p.lda = ldx; p.B = effp(l); p.ldb = l.kp; p.C = C; p.ldc = ldc; p.M = M; p.N = l.out; p.K = l.kp; p.bias = P + l.b; p.epi = epi; p.drop_site = site;
        if (M == B) { g.splitk_ws = gemm_ws; g.splitk_floats = gemm_ws_floats; }      // the per-sample GEMMs: skinny whatever the batch (call site, not shape)
        return nt_gemm(g, out_f32, 0, s);
    }
    // dx[M, kp] = dy[M, out(op)] @ Weff
    int gemm_dgrad(const void* dy, int lddy, int M, const BLin& l, void* dx, int lddx, int epi, const void* aux, int ldaux, DropCfg drop, hipStream_t s) {
        GemmGroup g; ginit(g); g.drop = drop;
        GemmProblem& p = g.p[0];
        p.A = dy; p.lda = lddy; p.C = dx; p.ldc = lddx; p.M = M; p.N = l.kp; p.epi = epi; p.aux = aux; p.ldaux = ldaux;
        if (LP) {
            p.B = efftp(l); p.ldb = l.op; p.K = (int)rupb(l.out, LDM) > l.op ? l.op : (int)rupb(l.out, LDM);
            if (M == B) { g.splitk_ws = gemm_ws; g.splitk_floats = gemm_ws_floats; }
            return nt_gemm(g, 0, 1, s);
        }
        p.B = effp(l); p.ldb = l.kp; p.K = l.out;
        return launch_gemm_f32(g, 0, 1, s);
    }
    // gradient of the effective weight into scratch, then through the weight-norm into G; bias gradient = column sums of dy
    int gemm_wgrad(const void* dy, int lddy, const void* x, int ldx, int rows, const BLin& l, int accumulate, hipStream_t s) {
        GemmGroup g; ginit(g);
        GemmProblem& p = g.p[0];
        p.A = dy; p.lda = lddy; p.B = x; p.ldb = ldx; p.C = dwscr; p.ldc = l.kp; p.M = l.out; p.N = l.kp; p.K = rows; p.epi = EPI_BIAS;
        CKB(tn_gemm(g, s));
        CKB(kb_wn_bwd(dwscr, l.kp, P + l.v, l.wn ? P + l.g : nullptr, l.wn ? sumsq + l.norm : nullptr, partial, G + l.v, l.wn ? G + l.g : nullptr, l.out, l.in, accumulate, s));
        // bias gradient: column sums of dy (its padded columns are exact zeros) into scratch, then the first `out` of them
        CKB(k_colsum<T>((const T*)dy, lddy, partial, dwscr, 0, rows, (int)rupb(l.out, 4), s));
        hipLaunchKernelGGL(butd_bias_copy_kernel, dim3(cdiv(l.out, 256)), dim3(256), 0, s, dwscr, G + l.b, l.out, accumulate);
        RGQA_LAUNCH_CHECK("butd_bias_copy_kernel");
        return RGQA_OK;
    }

    DropCfg drop_for(float p, uint32_t site) const { DropCfg d = make_drop(last_train ? p : 0.f, last_seed, 0); d.seed_hi ^= site; return d; }
    DropCfg drop_raw(float p) const { return make_drop(last_train ? p : 0.f, last_seed, 0); }

    // ------------------------------------------------------------------ forward
    int forward(const float* feats, const float* boxes, const int64_t* toks, const int64_t*, const int64_t*, float* pooled_out, float* logits_out, int ld_logits,
                int train, uint64_t seed, hipStream_t s) override {
        RGQA_REQUIRE(P != nullptr && ws != nullptr, "forward: engine not bound");
        RGQA_REQUIRE(feats && boxes && toks, "forward: null input");
        in_feats = feats; in_boxes = boxes; in_toks = toks; last_train = train; last_seed = seed;
        const DropCfg nd = make_drop(0.f, 0, 0);
        CKB(sync_weights(s));
        CKB(kb_embed_fwd<T>(toks, P + emb, X, B * L, E, Ep, s));
        CKB(gemm_fwd(X, Ep, B * L, wih, GI, 3 * H, EPI_BIAS, 0, nd, 0, s));
        CKB(rgqa_check_hip(hipMemsetAsync(Hall, 0, sizeof(T) * (size_t)B * H, s), "h0"));
        if constexpr (LP && !X3) {
            if (gru_persist()) {      // the whole recurrence in one launch, W_hh resident in LDS (butd_gru.hip)
                CKB(k_gru_fwd_persist(GI, (long)L * 3 * H, effp(whh), whh.kp, P + whh.b, Hall, Rg, Zg, Ng, GHN, B, L, H, gru_cnt, s));
                goto gru_done;
            }
        }
        for (int t = 0; t < L; ++t) {
            T* hp = Hall + (size_t)t * B * H;
            CKB(gemm_fwd(hp, H, B, whh, GH, 3 * H, EPI_BIAS, 0, nd, 0, s));
            CKB(kb_gru_fwd<T>(GI + (size_t)t * 3 * H, (long)L * 3 * H, GH, hp, hp + (size_t)B * H, Rg + (size_t)t * B * H, Zg + (size_t)t * B * H,
                              Ng + (size_t)t * B * H, GHN + (size_t)t * B * H, B, H, s));
        }
    gru_done:
        const T* q = Hall + (size_t)L * B * H;
        CKB(kb_concat<T>(feats, boxes, IF, B * O, cfg.feat_dim, cfg.pos_dim, Dp, s));
        CKB(gemm_fwd(IF, Dp, B * O, ip, IP, H, EPI_RELU, 0, nd, 0, s));
        CKB(gemm_fwd(q, H, B, qp, QP, H, EPI_RELU, 0, nd, 0, s));
        CKB(kb_attend_fwd<T>(IP, QP, wlin_eff, P + lin.b, IF, att, IE, B, O, H, Dp, drop_for(cfg.attn_dropout, 1), s));
        CKB(gemm_fwd(q, H, B, qproj, QR, H, EPI_RELU, 0, nd, 0, s));
        CKB(gemm_fwd(IE, Dp, B, iproj, IR, H, EPI_RELU, 0, nd, 0, s));
        CKB(kb_mul_fwd<T>(QR, IR, J, (size_t)B * H, s));
        CKB(gemm_fwd(J, H, B, c0, C1, 2 * H, EPI_RELU_DROP, 0, drop_raw(cfg.hidden_dropout), 2, s));
        if (LP) {       // the LDS-DMA kernels want N % 8 == 0: all op = round_up(NA, 64) columns (zero weight rows, zero-padded bias: exact zeros)
            GemmGroup g; ginit(g);
            GemmProblem& p = g.p[0];
            p.A = C1; p.lda = 2 * H; p.B = effp(c3); p.ldb = c3.kp; p.C = logits; p.ldc = NAp; p.M = B; p.N = c3.op; p.K = c3.kp; p.bias = c3_bias_pad; p.epi = EPI_BIAS;
            g.splitk_ws = gemm_ws; g.splitk_floats = gemm_ws_floats;
            CKB(nt_gemm(g, 1, 0, s));
        } else
        CKB(gemm_fwd(C1, 2 * H, B, c3, logits, NAp, EPI_BIAS, 1, nd, 0, s));
        if (pooled_out) CKB(k_to_f32<T>(J, H, pooled_out, H, B, H, s));
        if (logits_out) CKB(k_fill_rows<float>(logits_out, ld_logits, logits, NAp, B, NA, s));
        have_fwd = true;
        return RGQA_OK;
    }

    // ------------------------------------------------------------------ backward
    int loss_backward(const float* target, int ldt, float* loss_out, float grad_scale, int accumulate, hipStream_t s) override {
        RGQA_REQUIRE(have_fwd && G != nullptr, "loss_backward: no forward pass recorded / no gradient arena");
        float* dl32 = dwscr;
        CKB(k_bce_fwd_bwd(logits, NAp, target, ldt, loss_dev, dl32, NAp, B, NA, NAp, grad_scale, s, db_part));      // (db_part: B floats, free until the backward pass below)
        if (loss_out) CKB(rgqa_check_hip(hipMemcpyAsync(loss_out, loss_dev, sizeof(float), hipMemcpyDeviceToDevice, s), "loss copy"));
        CKB(k_cast_pad<T>(dl32, NAp, dlogits, NAp, B, NAp, 1.0f, s));
        return backward_impl(accumulate, s);
    }
    int backward(const float* dl, int ldd, int accumulate, hipStream_t s) override {
        RGQA_REQUIRE(have_fwd && G != nullptr && dl != nullptr, "backward: no forward pass recorded / null argument");
        CKB(k_cast_pad<T>(dl, ldd, dlogits, NAp, B, NA, 1.0f, s));
        return backward_impl(accumulate, s);
    }
    int backward_pooled(const float*, int, int, hipStream_t) override { rgqa_set_error("backward_pooled: not available for the BUTD engine"); return RGQA_ERR_ARG; }

    // Weight gradients.  Low-precision engines, no accumulation: every layer's problem is COLLECTED while the gradient flows back (their operands
    // live in buffers of their own until the end of the pass) and ONE grouped TN launch computes them all - longest contraction first, the bias
    // gradients ride as column sums of the dy operands (GemmProblem::colsum_out) - followed by the two grouped weight-norm launches.  Until round 5:
    // per layer one TN launch of 12-185 us on a mostly idle chip, a dot product, the weight-norm gradient, a column sum and a bias copy (~45 launches).
    GemmGroup wg; bool wg_collect = false;
    int wgrad(const void* dy, int lddy, const void* x, int ldx, int rows, const BLin& l, int accumulate, hipStream_t s) {
        if (!wg_collect) return gemm_wgrad(dy, lddy, x, ldx, rows, l, accumulate, s);
        RGQA_REQUIRE(wg.count + l.nsplit <= GEMM_MAX_PROBLEMS, "butd backward: too many weight-gradient problems");
        const bool direct = dw_direct(l);
        const int S = l.nsplit, kc = S > 1 ? (rows / S) / 64 * 64 : rows;
        for (int i = 0; i < S; ++i) {
            GemmProblem& p = wg.p[wg.count++];
            memset(&p, 0, sizeof p);
            const int r0 = i * kc, n = (i == S - 1) ? rows - r0 : kc;
            p.A = (const T*)dy + (size_t)r0 * lddy; p.lda = lddy; p.B = (const T*)x + (size_t)r0 * ldx; p.ldb = ldx;
            p.C = direct ? G + l.v : dwp(l) + (size_t)i * l.out * l.kp; p.ldc = direct ? l.in : l.kp;
            p.M = l.out; p.N = direct ? l.in : l.kp; p.K = n; p.epi = EPI_BIAS;
            p.colsum_out = S > 1 ? bpp(l) + (size_t)i * l.out : G + l.b;
        }
        return RGQA_OK;
    }

    int backward_impl(int accumulate, hipStream_t s) {
        const DropCfg nd = make_drop(0.f, 0, 0);
        const T* q = Hall + (size_t)L * B * H;
        if (!accumulate) CKB(rgqa_check_hip(hipMemsetAsync(G + emb, 0, sizeof(float) * (size_t)cfg.vocab_size * E, s), "zero embedding grad"));
        wg_collect = LP && !accumulate;
        memset(&wg, 0, sizeof wg); wg.drop = make_drop(0.f, 0, 0);
        // classifier
        CKB(wgrad(dlogits, NAp, C1, 2 * H, B, c3, accumulate, s));
        CKB(gemm_dgrad(dlogits, NAp, B, c3, dC1, 2 * H, EPI_DRELU_DROP, C1, 2 * H, drop_raw(cfg.hidden_dropout), s));
        CKB(wgrad(dC1, 2 * H, J, H, B, c0, accumulate, s));
        CKB(gemm_dgrad(dC1, 2 * H, B, c0, dJ, H, EPI_BIAS, nullptr, 0, nd, s));
        CKB(kb_mul_relu_bwd<T>(dJ, QR, IR, dQR, dIR, (size_t)B * H, s));
        // projections
        CKB(wgrad(dQR, H, q, H, B, qproj, accumulate, s));
        CKB(gemm_dgrad(dQR, H, B, qproj, dq, H, EPI_BIAS, nullptr, 0, nd, s));
        CKB(wgrad(dIR, H, IE, Dp, B, iproj, accumulate, s));
        CKB(gemm_dgrad(dIR, H, B, iproj, dIE, Dp, EPI_BIAS, nullptr, 0, nd, s));
        // attention over regions
        CKB(kb_attend_bwd<T>(dIE, IF, att, IP, QP, wlin_eff, dIP, dQP, dw_part, db_part, B, O, H, Dp, drop_for(cfg.attn_dropout, 1), s));
        if (wg_collect) {
            CKB(k_colsum<float>(dw_part, H, partial, dwp(lin), 0, B, H, s));          // dW of att.linear [1, H]: folded through its weight norm with the others
        } else {
            CKB(k_colsum<float>(dw_part, H, partial, dwscr, 0, B, H, s));
            CKB(kb_wn_bwd(dwscr, H, P + lin.v, P + lin.g, sumsq + lin.norm, partial, G + lin.v, G + lin.g, 1, H, accumulate, s));
        }
        hipLaunchKernelGGL(butd_sum_small_kernel, dim3(1), dim3(256), 0, s, db_part, B, G + lin.b, accumulate);
        RGQA_LAUNCH_CHECK("butd_sum_small_kernel");
        CKB(wgrad(dIP, H, IF, Dp, B * O, ip, accumulate, s));
        CKB(wgrad(dQP, H, q, H, B, qp, accumulate, s));
        CKB(gemm_dgrad(dQP, H, B, qp, dHa, H, EPI_ADD, dq, H, nd, s));          // dHa = d q_enc = both question paths
        // GRU, back through time
        T* dh = dHa; T* dhn = dHb;
        bool gru_host = true;
        if constexpr (LP && !X3) {
            if (gru_persist()) {
                CKB(k_gru_bwd_persist(dHa, Hall, Rg, Zg, Ng, GHN, dGI, (long)L * 3 * H, dGH, efftp(whh), whh.op, B, L, H, gru_cnt, s));
                gru_host = false;
            }
        }
        if (gru_host)
        for (int t = L - 1; t >= 0; --t) {
            const T* hp = Hall + (size_t)t * B * H;
            T* dgh_t = dGH + (size_t)t * B * 3 * H;
            CKB(kb_gru_bwd<T>(dh, hp, Rg + (size_t)t * B * H, Zg + (size_t)t * B * H, Ng + (size_t)t * B * H, GHN + (size_t)t * B * H,
                              dGI + (size_t)t * 3 * H, (long)L * 3 * H, dgh_t, tmpH, B, H, s));
            CKB(gemm_dgrad(dgh_t, 3 * H, B, whh, dhn, H, EPI_ADD, tmpH, H, nd, s));
            T* x = dh; dh = dhn; dhn = x;
        }
        CKB(wgrad(dGH, 3 * H, Hall, H, L * B, whh, accumulate, s));
        CKB(wgrad(dGI, 3 * H, X, Ep, B * L, wih, accumulate, s));
        CKB(gemm_dgrad(dGI, 3 * H, B * L, wih, dX, Ep, EPI_BIAS, nullptr, 0, nd, s));
        // dense gradient of the word table (nn.Embedding(padding_idx = ntoken): that row gets none, butd.py:36), every table row summed by one workgroup
        // in a fixed order (csrc/embed.hip; until round 6 a float-atomic scatter-add: the last bits changed from run to run)
        CKB(k_embed_keys(in_toks, B * L, emb_keys, s));
        {
            const size_t need = (size_t)64 * cdiv(B * L, 256) * E;      // the hot words' partial sums
            RGQA_REQUIRE(need <= partial_floats, "embedding gradients: %zu floats of scratch needed", need);
            CKB(k_embed_table_grads<T>(dX, Ep, emb_keys, nullptr, nullptr, B * L, G + emb, nullptr, nullptr, E, 0, 0, cfg.vocab_size - 1, 0, accumulate,
                                       emb_keys + (((size_t)B * L + 3) & ~(size_t)3), partial, need, s));
        }
        if (wg_collect) {
            CKB(tn_gemm(wg, s));
            CKB(kb_wn_backward_group(wn_dev, (int)wn_host.size(), wn_blocks, wn_partial, s));
            wg_collect = false;
        }
        // the one gradient segment of this engine (the whole arena: every weight gradient lands in the last launches of the pass) is final: the
        // data-parallel exchange waits for this event on its side stream (rgqa_engine_wait_grad_event)
        if (seg_events.empty()) {
            seg_events.resize(1);
            RGQA_HIP(hipEventCreateWithFlags(&seg_events[0], hipEventDisableTiming));
        }
        RGQA_HIP(hipEventRecord(seg_events[0], s));
        return RGQA_OK;
    }

    int get_activation(const char* name, float* out, size_t cap, hipStream_t s) override {
        RGQA_REQUIRE(have_fwd, "get_activation: no forward pass recorded");
        std::string nm(name);
        if (nm == "att") {
            RGQA_REQUIRE(cap >= (size_t)B * O, "get_activation: buffer too small");
            return rgqa_check_hip(hipMemcpyAsync(out, att, sizeof(float) * (size_t)B * O, hipMemcpyDeviceToDevice, s), "att copy");
        }
        if (nm == "q_enc") {
            RGQA_REQUIRE(cap >= (size_t)B * H, "get_activation: buffer too small");
            return k_to_f32<T>(Hall + (size_t)L * B * H, H, out, H, B, H, s);
        }
        rgqa_set_error("get_activation: unknown activation '%s' (BUTD engine: att, q_enc)", name);
        return RGQA_ERR_ARG;
    }
};

EngineBase* make_butd_engine(const rgqa_config& cfg) {
    if (cfg.precision == RGQA_PRECISION_BF16) return new ButdEngine<bf16_t>(cfg);
    if (cfg.precision == RGQA_PRECISION_BF16X3) return new ButdEngine<sf32>(cfg);
    return new ButdEngine<float>(cfg);
}
