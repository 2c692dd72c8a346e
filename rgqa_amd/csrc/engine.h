// LXMERT-GQA encoder engine: owns the parameter layout of the flat arenas, the workspace plan and the
// launch sequence of the whole forward / backward pass (reference path: tasks/gqa_model.py:30-43 ->
// lxrt/entry.py:109-120 -> lxrt/modeling.py:845-886, 546-566).  One engine per process / GPU.
#pragma once
#include <string>
#include <vector>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include "kernels.h"
#include "../../include/rgqa.h"

struct ParamInfo {
    std::string name;   // state_dict key of GQAModel (tasks/gqa_model.py)
    size_t offset;      // element offset in the flat f32 arena
    int ndim;
    long shape[2];
    int is_linear_weight;  // has a transposed low-precision copy
    int dead_in_x_mode;    // never receives a gradient when mode == 'x' 
    int f32_master_read = 0;   // a linear weight the forward nevertheless reads from the f32 master (the K = pos_dim box projection)
};

struct Lin { size_t w, b; int out, in, ldt; };     // ldt: leading dim of the transposed copy (round_up(out, 64))
struct LNp { size_t w, b; int n; };
struct AttP { Lin qkv, o; LNp ln; };
struct FfnP { Lin up, down; LNp ln; };

enum StageKind { ST_ATT_SELF = 0, ST_ATT_CROSS = 1, ST_FFN = 2 };

struct SegBuf {        // per (stage, modality) activation pointers; byte pointers typed at use
    void* x_in; void* y;
    void* qkv; void* ctx; void* z; void* hpre; void* h;
    float* lse; float* mean; float* rstd;
};

struct Stage {
    int kind;
    int active[2];             // [lang, visn]
    const AttP* att[2];        // self: per modality; cross: both point at the shared module
    const FfnP* ffn[2];
    SegBuf sb[2];
    uint32_t site;
    int last_dead;             // 1: visn side is the dead branch of the final x-layer
    int slot;                  // stage index within its layer (selects the per-stage gradient buffers)
    int layer_first;           // 1: first stage of a layer in forward order (deferred wgrads are launched after it in backward)
    int seg_event;             // event id of the layer's gradient segment (build_grad_segments): also indexes the per-segment weight events
};

// Optional per-launch timing with HIP events on the launch stream (bench.py's live roofline figures).
enum ProfCat { PC_GEMM_NT = 0, PC_GEMM_TN = 1, PC_ATTN_FWD = 2, PC_ATTN_BWD = 3, PC_LN = 4, PC_OTHER = 5, PC_GEMM_NT_D = 6 /* the dgrad launches of the NT family (round 6: under bf16x3_fwd they are bf16 kernels, the forward's split f32) */, PC_COUNT = 7 };
enum ProfBlock { PB_EMBED = 0, PB_LR = 1, PB_X = 2, PB_HEAD = 3, PB_COUNT = 4 };   // input embeddings | l/r layers | cross-modality layers | pooler + head + loss
struct ProfRec { hipEvent_t a, b; int cat; int block; double flops, bytes, obytes; char tag[48]; };
struct ProfSummary { double ms[PC_COUNT]; double flops[PC_COUNT]; double bytes[PC_COUNT]; long launches[PC_COUNT]; };

class EngineBase {
public:
    virtual ~EngineBase() {}
    bool profiling = false;
    int prof_block = PB_EMBED;                 // which part of the model the launches being enqueued belong to
    double last_blk_ms[PB_COUNT] = {0, 0, 0, 0}, last_blk_flops[PB_COUNT] = {0, 0, 0, 0};   // filled by prof_collect
    double last_cat_obytes[PC_COUNT] = {0, 0, 0, 0, 0, 0};   // per category: the GEMM operands alone (A + B + C), without the fused epilogues' operands
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    size_t prof_used = 0;
    hipEvent_t prof_event() {
        if (prof_used == prof_pool.size()) { hipEvent_t e; hipEventCreate(&e); prof_pool.push_back(e); }
        return prof_pool[prof_used++];
    }
    void prof_begin(int cat, double flops, double bytes, hipStream_t s, const char* tag = "", double operand_bytes = -1.0) {
        if (!profiling) return;
        ProfRec r; r.a = prof_event(); r.b = prof_event(); r.cat = cat; r.block = prof_block; r.flops = flops; r.bytes = bytes;
        r.obytes = operand_bytes < 0 ? bytes : operand_bytes;
        snprintf(r.tag, sizeof r.tag, "%s", tag);
        hipEventRecord(r.a, s);
        prof_recs.push_back(r);
    }
    void prof_end(hipStream_t s) { if (profiling) hipEventRecord(prof_recs.back().b, s); }
    void prof_cancel() { if (profiling) prof_recs.pop_back(); }      // the bracketed launch did not happen
    // host-synchronising: call after the stream has been synchronised
    void prof_collect(ProfSummary& out) {
        memset(&out, 0, sizeof out);
        for (int i = 0; i < PB_COUNT; ++i) { last_blk_ms[i] = 0; last_blk_flops[i] = 0; }
        for (int i = 0; i < PC_COUNT; ++i) last_cat_obytes[i] = 0;
        const char* dump = getenv("RGQA_PROF_DUMP");          // per-launch records (category, FLOPs, bytes, ms) for offline analysis
        FILE* df = dump ? fopen(dump, "a") : nullptr;
        for (auto& r : prof_recs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                out.ms[r.cat] += ms; out.flops[r.cat] += r.flops; out.bytes[r.cat] += r.bytes; out.launches[r.cat]++;
                last_cat_obytes[r.cat] += r.obytes;
                if (r.block >= 0 && r.block < PB_COUNT) { last_blk_ms[r.block] += ms; last_blk_flops[r.block] += r.flops; }
                if (df) fprintf(df, "%d %d %.6e %.6e %.6f %s\n", r.cat, r.block, r.flops, r.bytes, ms, r.tag[0] ? r.tag : "-");
            }
        }
        if (df) fclose(df);
        prof_recs.clear(); prof_used = 0;
    }
    rgqa_config cfg;
    std::vector<ParamInfo> params;
    size_t arena_elems = 0;
    size_t dead_begin = 0, dead_end = 0;   // element range of the final x-layer's visn_* parameters
    std::string err;
    virtual size_t workspace_bytes(int B, int T, int O) = 0;
    virtual int bind(float* p, float* g, void* plp, void* plpt, void* ws, size_t ws_bytes, int B, int T, int O) = 0;
    virtual int sync_weights(hipStream_t s) = 0;
    virtual int sync_transposed(hipStream_t s) = 0;
    virtual int forward(const float* feats, const float* boxes, const int64_t* ids, const int64_t* seg, const int64_t* mask,
                        float* pooled_out, float* logits_out, int ld_logits, int train, uint64_t seed, hipStream_t s) = 0;
    virtual int loss_backward(const float* target, int ldt, float* loss_out, float grad_scale, int accumulate, hipStream_t s) = 0;
    virtual int backward(const float* dlogits, int ldd, int accumulate, hipStream_t s) = 0;
    virtual int backward_pooled(const float* dpooled, int ld, int accumulate, hipStream_t s) = 0;
    // gradient-arena ranges in the order backward finishes them (for overlapping the DP exchange with backward)
    struct GradSeg { size_t begin, end; int event; };
    std::vector<GradSeg> grad_segs;
    std::vector<hipEvent_t> seg_events;
    // Weights that arrive while the forward pass is already running (the sharded data-parallel exchange all-gathers the updated weights chunk by
    // chunk on a side stream, rgqa_amd/parallel.py): wready[k] = event the first launch that reads the weights of gradient segment k waits for;
    // bwd_wait = event the next backward pass waits for (the transposed dgrad operand copies are refreshed behind the last chunk).  One-shot.
    std::vector<hipEvent_t> wready;
    hipEvent_t bwd_wait = nullptr;
    int wait_wready(int seg, hipStream_t s) {
        if (seg >= 0 && seg < (int)wready.size() && wready[seg] != nullptr) {
            const hipError_t r = hipStreamWaitEvent(s, wready[seg], 0);
            wready[seg] = nullptr;
            if (r != hipSuccess) { rgqa_set_error("forward: waiting for the weights of segment %d: %s", seg, hipGetErrorString(r)); return RGQA_ERR_HIP; }
        }
        return RGQA_OK;
    }
    int wait_bwd(hipStream_t s) {
        if (bwd_wait != nullptr) {
            const hipError_t r = hipStreamWaitEvent(s, bwd_wait, 0);
            bwd_wait = nullptr;
            if (r != hipSuccess) { rgqa_set_error("backward: waiting for the transposed operand copies: %s", hipGetErrorString(r)); return RGQA_ERR_HIP; }
        }
        return RGQA_OK;
    }
    virtual int num_weight_segments() const { return 0; }      // engines whose forward honours wready (0: none)
    virtual int get_activation(const char* name, float* out, size_t cap_elems, hipStream_t s) = 0;
    virtual int get_cross_attention(int, int, float*, size_t, hipStream_t) { rgqa_set_error("get_cross_attention: this engine has no cross-modality layers"); return RGQA_ERR_ARG; }
    // per-segment sum of squared gradients, written as each segment becomes final during backward (null = off)
    float* sumsq_slots = nullptr; float* sumsq_ws = nullptr; int sumsq_ws_segs = 0;
    virtual int set_input_grads(float* dfeats, float* dboxes) { (void)dfeats; (void)dboxes; rgqa_set_error("set_input_grads: not supported by this engine"); return RGQA_ERR_ARG; }
    // per-sample real token counts for the following forward passes (packed language rows); null: padded layout
    virtual int set_lengths(const int* lens, int n) { (void)lens; (void)n; rgqa_set_error("set_lengths: not supported by this engine"); return RGQA_ERR_ARG; }
};

EngineBase* make_engine(const rgqa_config& cfg);
// the weight-gradient side stream, one per device for all engines of the process (engine.hip make_side_stream; rgqa_set_side_stream)
#include <map>
#include <mutex>
std::mutex& rgqa_side_stream_mutex();
std::map<int, hipStream_t>& rgqa_side_streams();
