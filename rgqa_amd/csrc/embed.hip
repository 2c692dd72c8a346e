// BertEmbeddings (reference lxrt/modeling.py:264-292): word + position + token-type gather, LayerNorm, dropout.
// One wave per token; HBM-bound gather (3 table rows in, 1 activation row out).
#include "kernels.h"

// row_src (varlen mode): packed row -> b*Tn + t of the token it holds; null = identity (padded layout)
// row_dst (UNITER joint layout): output row of `out` for this token; the saved pre-LN sum, the statistics and the dropout stream stay
// indexed by the token's own (contiguous) row
template <typename T, int NV>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ seg, const int* __restrict__ row_src, const int* __restrict__ row_dst, const float* __restrict__ word,
                                                        const float* __restrict__ pos, const float* __restrict__ type, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, int ldo, T* __restrict__ zsave,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int rows, int Tn, int H, float eps, DropCfg drop) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int src = row_src ? row_src[row] : row;
    const int t = src % Tn;
    const int64_t id = ids[src], sg = seg ? seg[src] : 0;
    const int nv = H >> 2;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float a[4], b[4], d[4];
            load4(word + (size_t)id * H + c * 4, a);
            load4(pos + (size_t)t * H + c * 4, b);
            load4(type + (size_t)sg * H + c * 4, d);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[i][j] = a[j] + b[j] + d[j]; s += v[i][j]; }
            if (zsave) store4(zsave + (size_t)row * ldo + c * 4, v[i]);
        }
    }
    const float mu = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[i][j] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float g[4], b[4], o[4];
            load4(gamma + c * 4, g);
            load4(beta + c * 4, b);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = drop_apply(drop, (uint32_t)row * (uint32_t)H + (uint32_t)(c * 4 + j), (v[i][j] - mu) * rs * g[j] + b[j]);
            store4(out + (size_t)(row_dst ? row_dst[row] : row) * ldo + c * 4, o);
        }
    }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// ---------------------------------------------------------------------------------------------------------------- embedding gradients
// de [rows, H] -> the dense f32 gradients of the embedding tables (what the reference's nn.Embedding produces, lxrt/modeling.py:264-292; rows with index 0 are
// skipped: padding_idx = 0 on all three tables in LXMERT (pad0_all), on the word table only in UNITER, uniter/modeling.py:563-568; BUTD: nn.Embedding(padding_idx =
// ntoken), butd.py:36).
// Round 6: DETERMINISTIC - no float atomics.  Until round 5 every row was scatter-added with atomicAdd; the order in which the 256 [CLS] rows of a batch met in
// table row 101 changed from run to run, the last bits of the sums with it, and BertAdam's normalised update amplified that over a few steps (every "equal to one
// rank" / "equal to the serial step" test inherited the noise).  Now every table row is summed in an order that depends on the batch alone:
//   plan         one wave per packed row (the keys of all rows straight from L2, 16 bytes per lane): is this the FIRST row that names its word, and how many
//                rows name it?  Words named by more than HOT_MIN rows ([CLS], [SEP], '?': one row per sample) go on a short hot list, filled in row order by a one-workgroup launch.
//   word table   one workgroup per first-occurrence row: a word that occurs once is copied into its table row; a word of <= HOT_MIN rows is summed by one wave in
//                row order.
//   keyed sums   few keys with many rows each - position t (one row per sample), token type, the hot words: partial sums per (key, chunk of 256 rows), a wave per
//                64 rows folded in wave order, then one workgroup per key folds the chunks in chunk order into the table row.
#define EMB_HOT_MIN 8
#define EMB_HOT_MAX 64
__global__ void embed_keys_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ seg, const int* __restrict__ row_src, int rows, int Tn,
                                  int* __restrict__ kw, int* __restrict__ kp, int* __restrict__ kt) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int src = row_src ? row_src[r] : r;
    kw[r] = (int)ids[src];
    kp[r] = src % Tn;
    kt[r] = seg ? (int)seg[src] : 0;
}
__global__ void embed_keys_i64_kernel(const int64_t* __restrict__ ids, int rows, int* __restrict__ keys) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) keys[r] = (int)ids[r];
}

// plan[r] = number of rows that name row r's word if r is the word's first row and the word is not pad_key; else 0.
// One WAVE per row: 64 lanes x 4 keys per trip straight from the key array (L2-resident, four trips' loads in flight), lane-local counts folded at the end.
// (First forms: one workgroup staging all keys in LDS per row - 68 us, bound by the [CLS] / [SEP] rows' 256-row sums; one thread per row scanning LDS - 88-112 us.)
__global__ __launch_bounds__(256) void embed_word_plan_kernel(const int* __restrict__ kw, int rows, int rows4, int pad_key, int* __restrict__ plan) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;                                      // wave-uniform
    const int key = kw[r];
    int before = 0, cnt = 0;
    const int4* k4 = reinterpret_cast<const int4*>(kw);         // (the key array is padded to whole int4s)
    for (int base = 0; base < rows4; base += 256) {             // 4 x 64 int4 per trip
        int4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i4 = base + u * 64 + lane; v[u] = i4 < rows4 ? k4[i4] : make_int4(key - 1, key - 1, key - 1, key - 1); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = (base + u * 64 + lane) * 4;
            const int e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = (e[t] == key) & (j + t < rows);
                before |= m & (j + t < r);
                cnt += m & (j + t >= r);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); before |= __shfl_xor(before, o, 64); }
    if (lane != 0) return;
    plan[r] = (before || key == pad_key) ? 0 : cnt;
}

// The hot list: the first EMB_HOT_MAX first-occurrence rows (in ROW order - which words are hot must not depend on arrival order, or a word's sum would be folded
// one way in one run and another way in the next) whose word is named by more than EMB_HOT_MIN rows: hot[0] = their number, hot[1 ..] = the rows; their plan
// entries become 0 (the keyed partial / fold launches sum them).  One workgroup: every thread counts its contiguous range of rows, an exclusive scan over the 256
// counts, a second walk hands out the slots.
__global__ __launch_bounds__(256) void embed_hot_select_kernel(int* __restrict__ plan, int rows, int* __restrict__ hot) {
    __shared__ int cnts[256];
    const int tid = threadIdx.x, per = (rows + 255) / 256, lo = tid * per, hi = (lo + per < rows) ? lo + per : rows;
    int c = 0;
    for (int r = lo; r < hi; ++r) c += plan[r] > EMB_HOT_MIN;
    cnts[tid] = c;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < tid; ++i) base += cnts[i];
    for (int r = lo; r < hi; ++r)
        if (plan[r] > EMB_HOT_MIN) {
            if (base < EMB_HOT_MAX) { hot[1 + base] = r; plan[r] = 0; }
            ++base;
        }
    if (tid == 255) hot[0] = base < EMB_HOT_MAX ? base : EMB_HOT_MAX;
}

// the de rows lst[0 .. cnt) (offsets from row0) added in list order to acc, eight rows' loads in flight; one wave, lane = 4-column groups lane + 64 i
template <typename T, int NV>
__device__ __forceinline__ void embed_sum_rows(const T* __restrict__ de, int ldde, int H, int row0, const unsigned short* lst, int cnt, int lane, float (&acc)[NV][4]) {
    const int nv = H >> 2;
    constexpr int RPG = 8;
    for (int i0 = 0; i0 < cnt; i0 += RPG) {
        float v[RPG][NV][4];
#pragma unroll
        for (int k = 0; k < RPG; ++k) {
            const bool on = i0 + k < cnt;
            const size_t row = (size_t)(row0 + (on ? (int)lst[i0 + k] : (int)lst[i0]));
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < nv) load4(de + row * ldde + c * 4, v[k][i]);
                else { v[k][i][0] = v[k][i][1] = v[k][i][2] = v[k][i][3] = 0.f; }
            }
        }
#pragma unroll
        for (int k = 0; k < RPG; ++k)
            if (i0 + k < cnt) {
#pragma unroll
                for (int i = 0; i < NV; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += v[k][i][j];
            }
    }
}

// four waves' partial rows -> wave 0's acc, in wave order (fold: [3][H] floats of LDS)
template <int NV>
__device__ __forceinline__ void embed_fold_waves(float* fold, int H, int wave, int lane, float (&acc)[NV][4]) {
    const int nv = H >> 2;
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int c = lane + 64 * i; if (c < nv) store4(fold + (size_t)(wave - 1) * H + c * 4, acc[i]); }
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < nv) { float t[4]; load4(fold + (size_t)w * H + c * 4, t);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += t[j]; }
            }
    }
}

template <int NV>
__device__ __forceinline__ void embed_store_row(float* dst, int H, int lane, int accumulate, float (&acc)[NV][4]) {
    const int nv = H >> 2;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            if (accumulate) { float t[4]; load4(dst + c * 4, t);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += t[j]; }
            store4(dst + c * 4, acc[i]);
        }
    }
}

// One workgroup per packed row r with plan[r] > 0 (the others leave at once).  n == 1: the row is copied (added) into its table row.  n <= HOT_MIN: wave 0 collects the
// word's rows r .. in row order (256 keys per trip) and sums them.  n > HOT_MIN (the hot list was full): four waves, a contiguous quarter of r .. rows-1 each, partial
// rows folded in wave order.
template <typename T, int NV>
__global__ __launch_bounds__(256) void embed_word_grad_kernel(const T* __restrict__ de, int ldde, const int* __restrict__ kw, const int* __restrict__ plan, float* __restrict__ dword,
                                                              int rows, int H, int q_cap, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char emb_lds[];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = plan[r];
    if (n == 0) return;
    const int nv = H >> 2;
    const int key = kw[r];
    float* dst = dword + (size_t)key * H;
    float acc[NV][4];
    if (n <= EMB_HOT_MIN) {
        if (wave != 0) return;
        if (n == 1) {
#pragma unroll
            for (int i = 0; i < NV; ++i) { const int c = lane + 64 * i; if (c < nv) load4(de + (size_t)r * ldde + c * 4, acc[i]); }
            embed_store_row<NV>(dst, H, lane, accumulate, acc);
            return;
        }
        unsigned short* lst = reinterpret_cast<unsigned short*>(emb_lds);            // <= HOT_MIN entries, offsets from row r... in 16 bits: rows <= 65535 (launcher)
        int cnt = 0;
        for (int base = r; base < rows && cnt < n; base += 256) {
            int kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int j = base + u * 64 + lane; kk[u] = j < rows ? kw[j] : key - 1; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool m = kk[u] == key;
                const unsigned long long mask = __ballot(m);
                if (m) lst[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)(base + u * 64 + lane - r);
                cnt += __popcll(mask);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the wave's own LDS writes have landed before other lanes read them
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
        embed_sum_rows<T, NV>(de, ldde, H, r, lst, cnt, lane, acc);
        embed_store_row<NV>(dst, H, lane, accumulate, acc);
        return;
    }
    float* fold = reinterpret_cast<float*>(emb_lds);                                      // [3][H]
    unsigned short* lists = reinterpret_cast<unsigned short*>(emb_lds + (size_t)3 * H * 4);   // [4][q_cap]
    const int span = rows - r, q = (span + 3) >> 2;
    const int lo = r + wave * q, hi = (lo + q < rows) ? lo + q : rows;
    unsigned short* lst = lists + (size_t)wave * q_cap;
    int cnt = 0;
    for (int base = lo; base < hi; base += 256) {
        int kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int j = base + u * 64 + lane; kk[u] = j < hi ? kw[j] : key - 1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool m = kk[u] == key;
            const unsigned long long mask = __ballot(m);
            if (m) lst[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)(base + u * 64 + lane - lo);
            cnt += __popcll(mask);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
    if (lo < hi) embed_sum_rows<T, NV>(de, ldde, H, lo, lst, cnt, lane, acc);
    embed_fold_waves<NV>(fold, H, wave, lane, acc);
    if (wave == 0) embed_store_row<NV>(dst, H, lane, accumulate, acc);
}

// partial[(ky * nchunk + chunk)][H] = sum of the de rows of this 256-row chunk whose key is ky: ky < np: position ky; < np + nt: token type ky - np; else hot word
// number ky - np - nt (the word of row hot[1 + that])
template <typename T, int NV>
__global__ __launch_bounds__(256) void embed_partial_kernel(const T* __restrict__ de, int ldde, const int* __restrict__ kw, const int* __restrict__ kp, const int* __restrict__ kt,
                                                            const int* __restrict__ hot, float* __restrict__ partial, int rows, int np, int nt, int H, int pad0_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char emb_lds[];
    float* fold = reinterpret_cast<float*>(emb_lds);
    unsigned short* lists = reinterpret_cast<unsigned short*>(emb_lds + (size_t)3 * H * 4);   // [4][64]
    const int chunk = blockIdx.x, ky = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* keys; int k;
    if (ky < np) { keys = kp; k = ky; if (k == 0 && pad0_all) return; }
    else if (ky < np + nt) { keys = kt; k = ky - np; if (k == 0 && pad0_all) return; }
    else {
        const int h = ky - np - nt;
        if (h >= hot[0]) return;                                // block-uniform
        keys = kw; k = kw[hot[1 + h]];
    }
    const int lo = chunk * 256 + wave * 64, j = lo + lane;
    const bool m = j < rows && keys[j] == k;
    const unsigned long long mask = __ballot(m);
    unsigned short* lst = lists + wave * 64;
    if (m) lst[__popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)lane;
    const int cnt = __popcll(mask);
    __syncthreads();
    float acc[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
    if (cnt) embed_sum_rows<T, NV>(de, ldde, H, lo, lst, cnt, lane, acc);
    embed_fold_waves<NV>(fold, H, wave, lane, acc);
    if (wave == 0) {
        const int nv = H >> 2;
        float* dst = partial + ((size_t)ky * gridDim.x + chunk) * H;
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int c = lane + 64 * i; if (c < nv) store4(dst + c * 4, acc[i]); }
    }
}

__global__ __launch_bounds__(256) void embed_fold_kernel(const float* __restrict__ partial, int nchunk, const int* __restrict__ kw, const int* __restrict__ hot, float* __restrict__ dword,
                                                         float* __restrict__ dpos, float* __restrict__ dtype, int np, int nt, int H, int pad0_all, int accumulate) {
    const int ky = blockIdx.x;
    float* dst;
    if (ky < np) { if (ky == 0 && pad0_all) return; dst = dpos + (size_t)ky * H; }
    else if (ky < np + nt) { if (ky == np && pad0_all) return; dst = dtype + (size_t)(ky - np) * H; }
    else {
        const int h = ky - np - nt;
        if (h >= hot[0]) return;
        dst = dword + (size_t)kw[hot[1 + h]] * H;
    }
    for (int n = threadIdx.x; n < H; n += 256) {
        float s = 0.f;
        for (int c0 = 0; c0 < nchunk; c0 += 8) {             // eight chunk partials in flight, added in chunk order
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = c0 + k < nchunk ? partial[((size_t)ky * nchunk + c0 + k) * H + n] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (c0 + k < nchunk) s += t[k];
        }
        dst[n] = accumulate ? dst[n] + s : s;
    }
}

__global__ void make_mask_kernel(const int64_t* __restrict__ m, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (1.0f - (float)m[i]) * -10000.0f;
}

template <typename T>
int k_embed_fwd(const int64_t* ids, const int64_t* seg, const int* row_src, const int* row_dst, int rows, const float* word, const float* pos, const float* type, const float* gamma, const float* beta,
                T* out, int ldo, T* zsave, float* mean, float* rstd, int B, int Tn, int H, int vocab, int type_vocab, float eps, DropCfg drop, hipStream_t s) {
    RGQA_REQUIRE(H % 4 == 0 && H <= 2048 && ldo % 4 == 0, "embed: hidden %d unsupported", H);
    RGQA_REQUIRE(rows <= B * Tn, "embed: %d rows exceed B*T = %d", rows, B * Tn);
    if (rows <= 0) return RGQA_OK;
#define EMB(NVV) hipLaunchKernelGGL((embed_fwd_kernel<T, NVV>), dim3(cdiv(rows, 4)), dim3(256), 0, s, ids, seg, row_src, row_dst, word, pos, type, gamma, beta, out, ldo, zsave, mean, rstd, rows, Tn, H, eps, drop)
    const int nvl = cdiv(H / 4, 64);
    if (nvl <= 1) EMB(1); else if (nvl == 2) EMB(2); else if (nvl == 3) EMB(3); else if (nvl == 4) EMB(4); else EMB(8);
#undef EMB
    RGQA_LAUNCH_CHECK("embed_fwd_kernel");
    return RGQA_OK;
}

// The tables' gradients from the keys of the rows.  kw: word key per row (16-byte aligned, readable up to the next multiple of 4 ints); kp / kt: position / token-type key
// per row, or both null (a lone word table: BUTD).  iscratch: rows + 4 + 128 ints; fscratch: (np + nt + EMB_HOT_MAX) * ceil(rows / 256) * H floats.
// accumulate = 0: the tables were zeroed by the caller (rows no token names keep the zeros); 1: the sums are added to what the tables hold.
template <typename T>
int k_embed_table_grads(const T* de, int ldde, const int* kw, const int* kp, const int* kt, int rows, float* dword, float* dpos, float* dtype, int H, int np, int nt,
                        int pad_key, int pad0_all, int accumulate, int* iscratch, float* fscratch, size_t fscratch_floats, hipStream_t s) {
    if (rows <= 0) return RGQA_OK;
    if (kp == nullptr || kt == nullptr) { np = 0; nt = 0; }
    RGQA_REQUIRE(H % 4 == 0 && H <= 2048 && ldde % 4 == 0 && ldde >= H, "embedding gradients: hidden %d / pitch %d unsupported", H, ldde);
    RGQA_REQUIRE(kw != nullptr && iscratch != nullptr && fscratch != nullptr && ((uintptr_t)kw % 16) == 0, "embedding gradients: null or misaligned scratch");
    RGQA_REQUIRE(rows <= 65535, "embedding gradients: %d rows exceed the 16-bit row lists", rows);
    const int nchunk = cdiv(rows, 256), nkeys = np + nt + EMB_HOT_MAX;
    RGQA_REQUIRE((size_t)nkeys * nchunk * H <= fscratch_floats, "embedding gradients: scratch too small (%zu floats for %d keys x %d chunks x %d)", fscratch_floats, nkeys, nchunk, H);
    const size_t rp = ((size_t)rows + 3) & ~(size_t)3;
    int *plan = iscratch, *hot = iscratch + rp;
    const int rows4 = (rows + 3) / 4;
    hipLaunchKernelGGL(embed_word_plan_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, kw, rows, rows4, pad_key, plan);
    hipLaunchKernelGGL(embed_hot_select_kernel, dim3(1), dim3(256), 0, s, plan, rows, hot);
    RGQA_LAUNCH_CHECK("embed_word_plan_kernel / embed_hot_select_kernel");
    const int q_cap = (rows + 3) / 4 + 1;
    const size_t lds_w = (size_t)3 * H * 4 + (size_t)4 * q_cap * 2, lds_p = (size_t)3 * H * 4 + 4 * 64 * 2;
    RGQA_REQUIRE(lds_w <= 160 * 1024, "embedding gradients: %d rows need %zu bytes of LDS", rows, lds_w);
    const int nvl = cdiv(H / 4, 64);
#define EMBW(NVV) do { \
        if (lds_w > 64 * 1024) RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&embed_word_grad_kernel<T, NVV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w)); \
        hipLaunchKernelGGL((embed_word_grad_kernel<T, NVV>), dim3(rows), dim3(256), lds_w, s, de, ldde, kw, plan, dword, rows, H, q_cap, accumulate); \
        hipLaunchKernelGGL((embed_partial_kernel<T, NVV>), dim3(nchunk, nkeys), dim3(256), lds_p, s, de, ldde, kw, kp, kt, hot, fscratch, rows, np, nt, H, pad0_all); } while (0)
    if (nvl <= 1) EMBW(1); else if (nvl == 2) EMBW(2); else if (nvl == 3) EMBW(3); else if (nvl == 4) EMBW(4); else EMBW(8);
#undef EMBW
    RGQA_LAUNCH_CHECK("embed_word_grad_kernel / embed_partial_kernel");
    hipLaunchKernelGGL(embed_fold_kernel, dim3(nkeys), dim3(256), 0, s, fscratch, nchunk, kw, hot, dword, dpos, dtype, np, nt, H, pad0_all, accumulate);
    RGQA_LAUNCH_CHECK("embed_fold_kernel");
    return RGQA_OK;
}

int k_embed_keys(const int64_t* ids, int rows, int* keys, hipStream_t s) {
    if (rows <= 0) return RGQA_OK;
    hipLaunchKernelGGL(embed_keys_i64_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, ids, rows, keys);
    RGQA_LAUNCH_CHECK("embed_keys_i64_kernel");
    return RGQA_OK;
}

// LXMERT / UNITER: keys: 4 * (rows + 4) + 128 ints of scratch, 16-byte aligned; scratch: (Tn + type_vocab + 64) * ceil(rows / 256) * H floats
template <typename T>
int k_embed_scatter(const T* de, const int64_t* ids, const int64_t* seg, const int* row_src, int rows, float* dword, float* dpos, float* dtype, int B, int Tn, int H, int type_vocab,
                    int pad0_all, int accumulate, int* keys, float* scratch, size_t scratch_floats, hipStream_t s) {
    RGQA_REQUIRE(rows <= B * Tn, "embed scatter: %d rows exceed B*T = %d", rows, B * Tn);
    if (rows <= 0) return RGQA_OK;
    RGQA_REQUIRE(keys != nullptr, "embed scatter: null key scratch");
    const size_t rp = ((size_t)rows + 3) & ~(size_t)3;           // every array padded to whole int4s (the plan kernel stages the keys 16 bytes at a time)
    int *kw = keys, *kp = keys + rp, *kt = keys + 2 * rp, *isc = keys + 3 * rp;
    hipLaunchKernelGGL(embed_keys_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, ids, seg, row_src, rows, Tn, kw, kp, kt);
    RGQA_LAUNCH_CHECK("embed_keys_kernel");
    return k_embed_table_grads<T>(de, H, kw, kp, kt, rows, dword, dpos, dtype, H, Tn, type_vocab, 0, pad0_all, accumulate, isc, scratch, scratch_floats, s);
}

int k_make_mask(const int64_t* input_mask, float* out, int n, hipStream_t s) {
    if (n <= 0) return RGQA_OK;
    hipLaunchKernelGGL(make_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, input_mask, out, n);
    RGQA_LAUNCH_CHECK("make_mask_kernel");
    return RGQA_OK;
}

template int k_embed_fwd<float>(const int64_t*, const int64_t*, const int*, const int*, int, const float*, const float*, const float*, const float*, const float*, float*, int, float*, float*, float*, int, int, int, int, int, float, DropCfg, hipStream_t);
template int k_embed_fwd<bf16_t>(const int64_t*, const int64_t*, const int*, const int*, int, const float*, const float*, const float*, const float*, const float*, bf16_t*, int, bf16_t*, float*, float*, int, int, int, int, int, float, DropCfg, hipStream_t);
template int k_embed_fwd<sf32>(const int64_t*, const int64_t*, const int*, const int*, int, const float*, const float*, const float*, const float*, const float*, sf32*, int, sf32*, float*, float*, int, int, int, int, int, float, DropCfg, hipStream_t);
template int k_embed_scatter<sf32>(const sf32*, const int64_t*, const int64_t*, const int*, int, float*, float*, float*, int, int, int, int, int, int, int*, float*, size_t, hipStream_t);
template int k_embed_scatter<float>(const float*, const int64_t*, const int64_t*, const int*, int, float*, float*, float*, int, int, int, int, int, int, int*, float*, size_t, hipStream_t);
template int k_embed_scatter<bf16_t>(const bf16_t*, const int64_t*, const int64_t*, const int*, int, float*, float*, float*, int, int, int, int, int, int, int*, float*, size_t, hipStream_t);
template int k_embed_table_grads<sf32>(const sf32*, int, const int*, const int*, const int*, int, float*, float*, float*, int, int, int, int, int, int, int*, float*, size_t, hipStream_t);
template int k_embed_table_grads<float>(const float*, int, const int*, const int*, const int*, int, float*, float*, float*, int, int, int, int, int, int, int*, float*, size_t, hipStream_t);
template int k_embed_table_grads<bf16_t>(const bf16_t*, int, const int*, const int*, const int*, int, float*, float*, float*, int, int, int, int, int, int, int*, float*, size_t, hipStream_t);
