// The GRU recurrence of the BUTD question encoder (reference src/butd/butd.py:48-73: nn.GRU(300, 1024, 1, batch_first=True) from h0 = 0, output[:, -1];
// BASELINE config 5) as ONE persistent launch per direction, bf16.
//
// Until round 5 the host drove the recurrence: per time step one M = B GEMM (h W_hh^T, then dgh W_hh) and one gate kernel - 160 launches of 5-30 us
// for 40 tokens, 2.2 of the step's 4.5 ms (profiles/r05_butd_kernel_stats_before.md).  Here a launch keeps W_hh on the chip for the whole sequence:
//
//   grid    = (row groups of 64 samples) x (H / 16 column slices); one 256-thread workgroup per CU, all co-resident (the launcher checks)
//   LDS     = the slice's 48 weight rows (forward: rows {r, z, n} x 16 hidden units of W_hh, K = H) or its 16 rows of W_hh^T (backward: K = 3H)
//   a step  = every wave takes the 16 rows x K operand of ITS 16 samples straight from global memory into registers (h_{t-1}; backward: dgh_t),
//             runs 96 MFMAs against the LDS-resident slice, and finishes the gates (backward: the carry dh_{t-1}) for its 16 samples x 16 units in
//             registers: a lane owns the same (sample, 4 units) at every step, so h_{t-1} of the own units (backward: the running dh) never leaves it.
//   hand-off: what the OTHER slices of the row group need - h_t (forward), dgh_t (backward) - is stored write-through (sc1), every wave drains
//             (s_waitcnt vmcnt(0)), the workgroup barrier, then ONE lane adds to the row group's counter of that step; consumers poll that counter
//             with sc1 loads from one lane (bounded), join a workgroup barrier and read the rows with sc1 loads only - MI355X_MICROARCH.md
//             "Valid forms", first table row (hipMalloc memory, one workgroup per CU, 8-byte sc1 stores, 16-byte sc1 loads).  Row groups are
//             independent: a group only ever waits for its own H / 16 workgroups.
// Counters are zeroed by the launcher on the stream before every launch; a poll that gives up (~1 s) raises the error word and the launch still ends.
// Arithmetic: gh = h W_hh^T + b_hh stays in f32 between the MFMAs and the gates (the host loop rounded it to bf16 in between); the saved r, z, n,
// gh_n and every h_t are bf16 as before.
#include "kernels.h"
#include "butd.h"
#include "gemm_nt256.h"      // rgqa_num_cus

typedef __attribute__((address_space(1))) const int gint;

struct GruFwdArgs {
    const bf16_t* GI; long ldgi;              // [B, L, 3H] input projections incl. b_ih; row stride of a sample
    const bf16_t* W; int ldw;                 // effective W_hh [3H, ldw]
    const float* bhh;                         // [3H]
    bf16_t* Hall;                             // [(L + 1), B, H]; Hall[0] = 0 (set by the caller)
    bf16_t *Rg, *Zg, *Ng, *GHN;               // [L, B, H] saved for the backward pass
    int B, L;
    int* cnt;                                 // [row groups][L + 1], zeroed
    int* err;
};
struct GruBwdArgs {
    const bf16_t* dH;                         // [B, H] gradient w.r.t. h_L
    const bf16_t* Hall; const bf16_t *Rg, *Zg, *Ng, *GHN;
    bf16_t* dGI; long lddgi;                  // [B, L, 3H]
    bf16_t* dGH;                              // [L, B, 3H]
    const bf16_t* WT; int ldwt;               // W_hh^T [H, ldwt >= 3H]: row k = hidden unit k, columns = the 3H gate outputs
    int B, L;
    int* cnt;                                 // [row groups][L], zeroed
    int* err;
};

__device__ __forceinline__ float sigm_f(float x) { return 1.f / (1.f + __expf(-x)); }

// one lane polls until *p >= want (sc1 loads; the caller joins a workgroup barrier afterwards)
// returns false when it gave up (the error word is raised; the caller poisons its outputs with NaNs so that the loss / the gradient norm shows it)
__device__ __forceinline__ bool gru_wait(const int* p, int want, int* err) {
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 24)) { __hip_atomic_fetch_add(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    }
    return true;
}

// KSP = 1: 4 waves, each 16 samples x the whole contraction; KSP = 2: 8 waves, waves w and w + 4 share 16 samples and take one half of the contraction
// each (twice the operand loads in flight per CU: the step is bound by the latency of those loads, not by the 96 MFMAs), the upper half hands its
// partial sums over through LDS.
template <int H, int KSP>
__global__ __launch_bounds__(256 * KSP) void gru_fwd_persist_kernel(const GruFwdArgs a) {
    constexpr int KS = H / 32, KSW = KS / KSP, PITCH = 2 * H + 16, CS = H / 16, WBYTES = 48 * PITCH;     // K-steps of 32; LDS row pitch: +16 bytes, conflict-free b128 reads
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int s_bad;                                            // a poll gave up (as in the 32-unit form below: the slice's output is poisoned)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mt = wave & 3, kh = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;
    const int cs = blockIdx.x % CS, rg = blockIdx.x / CS;
    const int B = a.B, L = a.L;
    if (tid == 0) s_bad = 0;
    // the slice's weight rows: LDS row g * 16 + u = W_hh row g * H + cs * 16 + u
    for (int i = tid; i < 48 * (H / 8); i += 256 * KSP) {
        const int r = i / (H / 8), c = i % (H / 8);
        *reinterpret_cast<uint4*>(lds + r * PITCH + c * 16) = *reinterpret_cast<const uint4*>(a.W + (size_t)((r >> 4) * H + cs * 16 + (r & 15)) * a.ldw + c * 8);
    }
    __syncthreads();
    f32x4* xch = reinterpret_cast<f32x4*>(lds + WBYTES);             // [4 m-tiles][3 gates][64 lanes] partial sums of the upper K half
    const int row = rg * 64 + mt * 16 + fr;
    const bool live = row < B && kh == 0;
    const int rowc = row < B ? row : B - 1;
    const int u0 = cs * 16 + fq * 4;                                 // this lane's 4 hidden units
    float bh[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) bh[g][j] = a.bhh[g * H + u0 + j];
    float hown[4] = {0.f, 0.f, 0.f, 0.f};
    const auto rs_h = __builtin_amdgcn_make_buffer_rsrc(a.Hall, 0, (int)((size_t)(L + 1) * B * H * 2), 0x00020000);
    int* cnt = a.cnt + rg * (L + 1);
    bf16x4 rr, zz, nn, gg;
    for (int t = 0; t < L; ++t) {
        const bf16_t* gi = a.GI + (size_t)rowc * a.ldgi + (size_t)t * 3 * H + u0;
        const bf16x4 gir = *reinterpret_cast<const bf16x4*>(gi), giz = *reinterpret_cast<const bf16x4*>(gi + H), gin = *reinterpret_cast<const bf16x4*>(gi + 2 * H);
        if (t > 0) {
            // what the backward pass needs of step t - 1 goes out AFTER that step's signal: these stores are nobody's hand-off
            if (live) {
                const size_t o = ((size_t)(t - 1) * B + row) * H + u0;
                *reinterpret_cast<bf16x4*>(a.Rg + o) = rr; *reinterpret_cast<bf16x4*>(a.Zg + o) = zz;
                *reinterpret_cast<bf16x4*>(a.Ng + o) = nn; *reinterpret_cast<bf16x4*>(a.GHN + o) = gg;
            }
            if (tid == 0 && !gru_wait(cnt + t, CS, a.err)) s_bad = 1;   // h_t of every slice of this row group is in place (Hall[0] is the caller's zero block)
            __syncthreads();
        }
        bf16x8 af[KSW];
        const unsigned hoff = (unsigned)(((size_t)t * B + rowc) * H * 2) + fq * 16 + kh * KSW * 64;
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_h, hoff + ks * 64, 0, 16);      // aux 16 = sc1
            af[ks] = *reinterpret_cast<const bf16x8*>(&v);
        }
        __builtin_amdgcn_sched_barrier(0);       // every operand load of the step is in flight before the first MFMA waits for one
        f32x4 acc[3] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(lds + (g * 16 + fr) * PITCH + ((kh * KSW + ks) * 4 + fq) * 16);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[ks], acc[g], 0, 0, 0);
            }
        if (KSP == 2) {
            if (kh == 1) {
#pragma unroll
                for (int g = 0; g < 3; ++g) xch[(mt * 3 + g) * 64 + lane] = acc[g];
            }
            __syncthreads();
            if (kh == 0) {
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] += xch[(mt * 3 + g) * 64 + lane];
            }
        }
        // gates (gate order r, z, n as torch.nn.GRU): acc[g][j] = (h_{t-1} W_hh^T)[sample fr of this m-tile][unit u0 + j] of gate g
        bf16x4 hn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float r = sigm_f((float)gir[j] + acc[0][j] + bh[0][j]);
            const float z = sigm_f((float)giz[j] + acc[1][j] + bh[1][j]);
            const float gn = acc[2][j] + bh[2][j];
            const float n = tanhf((float)gin[j] + r * gn);
            const float h = (1.f - z) * n + z * hown[j];
            hn[j] = (bf16_t)h; rr[j] = (bf16_t)r; zz[j] = (bf16_t)z; nn[j] = (bf16_t)n; gg[j] = (bf16_t)gn;
            hown[j] = (float)hn[j];                                  // what every other slice reads back
        }
        if (live)
            __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const __attribute__((ext_vector_type(2))) unsigned*>(&hn), rs_h,
                                                  (unsigned)((((size_t)(t + 1) * B + row) * H + u0) * 2), 0, 16);       // write-through
        if (t + 1 < L) {                                             // (the last h is read by later launches only)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt + t + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (live) {
        const size_t o = ((size_t)(L - 1) * B + row) * H + u0;
        *reinterpret_cast<bf16x4*>(a.Rg + o) = rr; *reinterpret_cast<bf16x4*>(a.Zg + o) = zz;
        *reinterpret_cast<bf16x4*>(a.Ng + o) = nn; *reinterpret_cast<bf16x4*>(a.GHN + o) = gg;
        if (s_bad) {             // a poll gave up: the final state of this slice says so (NaNs reach the loss)
            const unsigned long long q = 0x7FC07FC07FC07FC0ull;
            *reinterpret_cast<unsigned long long*>(a.Hall + ((size_t)L * B + row) * H + u0) = q;
        }
    }
}

template <int H, int KSP>
__global__ __launch_bounds__(256 * KSP) void gru_bwd_persist_kernel(const GruBwdArgs a) {
    // CH = 48 K-steps of operand fragments are requested together (4 registers each: 192 of the 256 architectural VGPRs a wave can address; all 96 of
    // a 4-wave block's share spilled 148 registers)
    constexpr int K3 = 3 * H, KS3 = K3 / 32, KSW = KS3 / KSP, CH = 48, NCH = KSW / CH, PITCH = 2 * K3 + 16, CS = H / 16, WBYTES = 16 * PITCH;
    static_assert(KSW % CH == 0, "whole chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mt = wave & 3, kh = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;
    const int cs = blockIdx.x % CS, rg = blockIdx.x / CS;
    const int B = a.B, L = a.L;
    if (tid == 0) s_bad = 0;
    // LDS row u = W_hh^T row cs * 16 + u: the 3H gate-output weights that feed hidden unit u of h_{t-1}
    for (int i = tid; i < 16 * (K3 / 8); i += 256 * KSP) {
        const int r = i / (K3 / 8), c = i % (K3 / 8);
        *reinterpret_cast<uint4*>(lds + r * PITCH + c * 16) = *reinterpret_cast<const uint4*>(a.WT + (size_t)(cs * 16 + r) * a.ldwt + c * 8);
    }
    __syncthreads();
    f32x4* xch = reinterpret_cast<f32x4*>(lds + WBYTES);             // [4 m-tiles][64 lanes]
    const int row = rg * 64 + mt * 16 + fr;
    const bool live = row < B && kh == 0;
    const int rowc = row < B ? row : B - 1;
    const int u0 = cs * 16 + fq * 4;
    float d[4];
    {
        const bf16x4 d0 = *reinterpret_cast<const bf16x4*>(a.dH + (size_t)rowc * H + u0);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = (float)d0[j];
    }
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(a.dGH, 0, (int)((size_t)L * B * K3 * 2), 0x00020000);
    int* cnt = a.cnt + rg * L;
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    size_t o = ((size_t)(L - 1) * B + rowc) * H + u0;
    bf16x4 r4 = *reinterpret_cast<const bf16x4*>(a.Rg + o), z4 = *reinterpret_cast<const bf16x4*>(a.Zg + o), n4 = *reinterpret_cast<const bf16x4*>(a.Ng + o),
           g4 = *reinterpret_cast<const bf16x4*>(a.GHN + o), h4 = *reinterpret_cast<const bf16x4*>(a.Hall + o);
    for (int t = L - 1; t >= 0; --t) {
        bf16x4 qr, qz, qn, qnr;
        float carry[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float r = (float)r4[j], z = (float)z4[j], n = (float)n4[j], gn = (float)g4[j], hp = (float)h4[j];
            const float dn = d[j] * (1.f - z) * (1.f - n * n);
            const float dz = d[j] * (hp - n) * z * (1.f - z);
            const float dr = dn * gn * r * (1.f - r);
            qr[j] = (bf16_t)dr; qz[j] = (bf16_t)dz; qn[j] = (bf16_t)dn; qnr[j] = (bf16_t)(dn * r);
            carry[j] = d[j] * z;                                     // the direct path h_{t-1} -> h_t
        }
        if (live) {
            const unsigned go = (unsigned)((((size_t)t * B + row) * K3 + u0) * 2);
            __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u2*>(&qr), rs_g, go, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u2*>(&qz), rs_g, go + 2 * H, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u2*>(&qnr), rs_g, go + 4 * H, 0, 16);
        }
        if (t > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (live) {                                                  // dgi_t (read by later launches only) and the next step's saved activations: behind the signal
            bf16_t* gi = a.dGI + (size_t)row * a.lddgi + (size_t)t * K3 + u0;
            *reinterpret_cast<bf16x4*>(gi) = qr; *reinterpret_cast<bf16x4*>(gi + H) = qz; *reinterpret_cast<bf16x4*>(gi + 2 * H) = qn;
        }
        if (t == 0) break;                                           // h_0 is the constant zero state: nothing flows further back
        o = ((size_t)(t - 1) * B + rowc) * H + u0;
        r4 = *reinterpret_cast<const bf16x4*>(a.Rg + o); z4 = *reinterpret_cast<const bf16x4*>(a.Zg + o); n4 = *reinterpret_cast<const bf16x4*>(a.Ng + o);
        g4 = *reinterpret_cast<const bf16x4*>(a.GHN + o); h4 = *reinterpret_cast<const bf16x4*>(a.Hall + o);
        if (tid == 0 && !gru_wait(cnt + t, CS, a.err)) s_bad = 1;    // dgh_t of every slice of this row group
        __syncthreads();
        // dh_{t-1}[sample][u0 + j] = carry + sum over the 3H gate outputs of dgh_t[sample][.] W_hh[., u0 + j]
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned aoff = (unsigned)(((size_t)t * B + rowc) * K3 * 2) + fq * 16 + kh * KSW * 64;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            bf16x8 af[CH];
#pragma unroll
            for (int ks = 0; ks < CH; ++ks) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_g, aoff + (c * CH + ks) * 64, 0, 16);
                af[ks] = *reinterpret_cast<const bf16x8*>(&v);
            }
            __builtin_amdgcn_sched_barrier(0);   // the chunk's loads are all in flight before the first MFMA waits for one
#pragma unroll
            for (int ks = 0; ks < CH; ++ks) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(lds + fr * PITCH + ((kh * KSW + c * CH + ks) * 4 + fq) * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[ks], acc, 0, 0, 0);
            }
        }
        if (KSP == 2) {
            if (kh == 1) xch[mt * 64 + lane] = acc;
            __syncthreads();
            if (kh == 0) acc += xch[mt * 64 + lane];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = acc[j] + carry[j];
    }
    __syncthreads();
    if (live && s_bad) {         // a poll gave up: the first token's input gradient of this slice says so
        const unsigned long long q = 0x7FC07FC07FC07FC0ull;
        *reinterpret_cast<unsigned long long*>(a.dGI + (size_t)row * a.lddgi + u0) = q;
    }
}

// ---------------------------------------------------------------------------------------------------------------- 32-unit slices (round 5, second form)
// What a token costs above is the operand every workgroup pulls from the memory side of the L2s (sc1 loads are never L2 hits): 64 workgroups per row
// group each read the group's whole h_{t-1} (128 KB) / dgh_t (384 KB) - 32 / 96 MB per token over the fabric, at 36 GB/s per CU (7 -> 14 us per token
// for 128 -> 384 KB: 3.5 us + 27 ns per KB).  Here a workgroup owns 32 hidden units and 32 samples: 32 slices x 8 row groups, half the bytes per
// workgroup and per token.  The slice's weights (192 KB) do not fit the LDS; they live in REGISTERS: 8 waves split the contraction 8 ways, each wave
// keeps its 96 fragments (16 rows x 32 k each) for the whole sequence - 96 VGPRs -, loads the operand of both 16-sample tiles for its K range only
// (every byte of the operand is requested once per workgroup), and the partial sums meet in LDS (summed in wave order: deterministic).  Waves 0-3 then
// finish the gates for one (sample tile, unit tile) pair each, a lane owning the same (sample, 4 units) at every step as above.  Hand-off: as above.
template <int H>
__global__ __launch_bounds__(512) void gru_fwd_persist32_kernel(const GruFwdArgs a) {
    constexpr int KSW = H / 32 / 8, CS = H / 32;                  // K-steps of 32 per wave; slices per row group
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    f32x4* xch = reinterpret_cast<f32x4*>(lds);                   // [8 waves][12 tiles: (m, gate, u)][64 lanes]
    int* s_bad = reinterpret_cast<int*>(lds + 8 * 12 * 64 * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int cs = blockIdx.x % CS, rg = blockIdx.x / CS;
    const int B = a.B, L = a.L;
    const int kw0 = wave * KSW;
    if (tid == 0) *s_bad = 0;
    // this wave's weight fragments: rows {r, z, n} x 2 unit tiles of W_hh, its K range
    bf16x8 wreg[3][2][KSW];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
                wreg[g][u][ks] = *reinterpret_cast<const bf16x8*>(a.W + (size_t)(g * H + cs * 32 + u * 16 + fr) * a.ldw + (kw0 + ks) * 32 + fq * 8);
    // gate phase (waves 0-3): sample tile gm, unit tile gu
    const int gm = wave & 1, gu = (wave >> 1) & 1;
    const int row = rg * 32 + gm * 16 + fr;
    const bool gate = wave < 4, live = gate && row < B;
    const int rowc = row < B ? row : B - 1;
    const int u0 = cs * 32 + gu * 16 + fq * 4;
    float bh[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) bh[g][j] = a.bhh[g * H + u0 + j];
    float hown[4] = {0.f, 0.f, 0.f, 0.f};
    const auto rs_h = __builtin_amdgcn_make_buffer_rsrc(a.Hall, 0, (int)((size_t)(L + 1) * B * H * 2), 0x00020000);
    int* cnt = a.cnt + rg * (L + 1);
    int lrow[2];                                                   // the operand rows of this lane (clamped: the result of a row >= B is dropped)
#pragma unroll
    for (int m = 0; m < 2; ++m) { const int r = rg * 32 + m * 16 + fr; lrow[m] = r < B ? r : B - 1; }
    bf16x4 rr, zz, nn, gg;
    __syncthreads();
    for (int t = 0; t < L; ++t) {
        bf16x4 gir, giz, gin;
        if (gate) {
            const bf16_t* gi = a.GI + (size_t)rowc * a.ldgi + (size_t)t * 3 * H + u0;
            gir = *reinterpret_cast<const bf16x4*>(gi); giz = *reinterpret_cast<const bf16x4*>(gi + H); gin = *reinterpret_cast<const bf16x4*>(gi + 2 * H);
        }
        if (t > 0) {
            if (live) {          // what the backward pass needs of step t - 1: behind that step's signal
                const size_t o = ((size_t)(t - 1) * B + row) * H + u0;
                *reinterpret_cast<bf16x4*>(a.Rg + o) = rr; *reinterpret_cast<bf16x4*>(a.Zg + o) = zz;
                *reinterpret_cast<bf16x4*>(a.Ng + o) = nn; *reinterpret_cast<bf16x4*>(a.GHN + o) = gg;
            }
            if (tid == 0 && !gru_wait(cnt + t, CS, a.err)) *s_bad = 1;
            __syncthreads();
        }
        bf16x8 af[2][KSW];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_h, (unsigned)((((size_t)t * B + lrow[m]) * H + (kw0 + ks) * 32 + fq * 8) * 2), 0, 16);      // aux 16 = sc1
                af[m][ks] = *reinterpret_cast<const bf16x8*>(&v);
            }
        __builtin_amdgcn_sched_barrier(0);       // every operand load of the step is in flight before the first MFMA waits for one
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KSW; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[g][u][ks], af[m][ks], acc, 0, 0, 0);
                    xch[(wave * 12 + (m * 3 + g) * 2 + u) * 64 + lane] = acc;
                }
        __syncthreads();
        bf16x4 hn;
        if (gate) {
            float gh[3][4];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                f32x4 sum = xch[((gm * 3 + g) * 2 + gu) * 64 + lane];
#pragma unroll
                for (int p = 1; p < 8; ++p) sum += xch[(p * 12 + (gm * 3 + g) * 2 + gu) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) gh[g][j] = sum[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float r = sigm_f((float)gir[j] + gh[0][j] + bh[0][j]);
                const float z = sigm_f((float)giz[j] + gh[1][j] + bh[1][j]);
                const float gn = gh[2][j] + bh[2][j];
                const float n = tanhf((float)gin[j] + r * gn);
                const float h = (1.f - z) * n + z * hown[j];
                hn[j] = (bf16_t)h; rr[j] = (bf16_t)r; zz[j] = (bf16_t)z; nn[j] = (bf16_t)n; gg[j] = (bf16_t)gn;
                hown[j] = (float)hn[j];
            }
            if (live)
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const __attribute__((ext_vector_type(2))) unsigned*>(&hn), rs_h,
                                                      (unsigned)((((size_t)(t + 1) * B + row) * H + u0) * 2), 0, 16);       // write-through
        }
        if (t + 1 < L) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt + t + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (live) {
        const size_t o = ((size_t)(L - 1) * B + row) * H + u0;
        *reinterpret_cast<bf16x4*>(a.Rg + o) = rr; *reinterpret_cast<bf16x4*>(a.Zg + o) = zz;
        *reinterpret_cast<bf16x4*>(a.Ng + o) = nn; *reinterpret_cast<bf16x4*>(a.GHN + o) = gg;
        if (*s_bad) {            // a poll gave up: the final state of this slice says so
            const unsigned long long q = 0x7FC07FC07FC07FC0ull;      // four bf16 NaNs
            *reinterpret_cast<unsigned long long*>(a.Hall + ((size_t)L * B + row) * H + u0) = q;
        }
    }
}

template <int H>
__global__ __launch_bounds__(512) void gru_bwd_persist32_kernel(const GruBwdArgs a) {
    constexpr int K3 = 3 * H, KSW = K3 / 32 / 8, CS = H / 32;   // 12 K-steps of 32 per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    f32x4* xch = reinterpret_cast<f32x4*>(lds);                   // [8 waves][4 tiles: (m, u)][64 lanes]
    int* s_bad = reinterpret_cast<int*>(lds + 8 * 4 * 64 * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int cs = blockIdx.x % CS, rg = blockIdx.x / CS;
    const int B = a.B, L = a.L;
    const int kw0 = wave * KSW;
    if (tid == 0) *s_bad = 0;
    // W_hh^T rows cs * 32 + u * 16 + fr (hidden units), this wave's range of the 3H gate outputs
    bf16x8 wreg[2][KSW];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
            wreg[u][ks] = *reinterpret_cast<const bf16x8*>(a.WT + (size_t)(cs * 32 + u * 16 + fr) * a.ldwt + (kw0 + ks) * 32 + fq * 8);
    const int gm = wave & 1, gu = (wave >> 1) & 1;
    const int row = rg * 32 + gm * 16 + fr;
    const bool gate = wave < 4, live = gate && row < B;
    const int rowc = row < B ? row : B - 1;
    const int u0 = cs * 32 + gu * 16 + fq * 4;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    if (gate) {
        const bf16x4 d0 = *reinterpret_cast<const bf16x4*>(a.dH + (size_t)rowc * H + u0);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = (float)d0[j];
    }
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(a.dGH, 0, (int)((size_t)L * B * K3 * 2), 0x00020000);
    int* cnt = a.cnt + rg * L;
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    int lrow[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { const int r = rg * 32 + m * 16 + fr; lrow[m] = r < B ? r : B - 1; }
    size_t o = ((size_t)(L - 1) * B + rowc) * H + u0;
    bf16x4 r4, z4, n4, g4, h4;
    if (gate) {
        r4 = *reinterpret_cast<const bf16x4*>(a.Rg + o); z4 = *reinterpret_cast<const bf16x4*>(a.Zg + o); n4 = *reinterpret_cast<const bf16x4*>(a.Ng + o);
        g4 = *reinterpret_cast<const bf16x4*>(a.GHN + o); h4 = *reinterpret_cast<const bf16x4*>(a.Hall + o);
    }
    __syncthreads();
    for (int t = L - 1; t >= 0; --t) {
        bf16x4 qr, qz, qn, qnr;
        float carry[4] = {0.f, 0.f, 0.f, 0.f};
        if (gate) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float r = (float)r4[j], z = (float)z4[j], n = (float)n4[j], gn = (float)g4[j], hp = (float)h4[j];
                const float dn = d[j] * (1.f - z) * (1.f - n * n);
                const float dz = d[j] * (hp - n) * z * (1.f - z);
                const float dr = dn * gn * r * (1.f - r);
                qr[j] = (bf16_t)dr; qz[j] = (bf16_t)dz; qn[j] = (bf16_t)dn; qnr[j] = (bf16_t)(dn * r);
                carry[j] = d[j] * z;
            }
            if (live) {
                const unsigned go = (unsigned)((((size_t)t * B + row) * K3 + u0) * 2);
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u2*>(&qr), rs_g, go, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u2*>(&qz), rs_g, go + 2 * H, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u2*>(&qnr), rs_g, go + 4 * H, 0, 16);
            }
        }
        if (t > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (live) {
            bf16_t* gi = a.dGI + (size_t)row * a.lddgi + (size_t)t * K3 + u0;
            *reinterpret_cast<bf16x4*>(gi) = qr; *reinterpret_cast<bf16x4*>(gi + H) = qz; *reinterpret_cast<bf16x4*>(gi + 2 * H) = qn;
        }
        if (t == 0) break;
        if (gate) {
            o = ((size_t)(t - 1) * B + rowc) * H + u0;
            r4 = *reinterpret_cast<const bf16x4*>(a.Rg + o); z4 = *reinterpret_cast<const bf16x4*>(a.Zg + o); n4 = *reinterpret_cast<const bf16x4*>(a.Ng + o);
            g4 = *reinterpret_cast<const bf16x4*>(a.GHN + o); h4 = *reinterpret_cast<const bf16x4*>(a.Hall + o);
        }
        if (tid == 0 && !gru_wait(cnt + t, CS, a.err)) *s_bad = 1;
        __syncthreads();
        bf16x8 af[2][KSW];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (unsigned)((((size_t)t * B + lrow[m]) * K3 + (kw0 + ks) * 32 + fq * 8) * 2), 0, 16);
                af[m][ks] = *reinterpret_cast<const bf16x8*>(&v);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KSW; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[u][ks], af[m][ks], acc, 0, 0, 0);
                xch[(wave * 4 + m * 2 + u) * 64 + lane] = acc;
            }
        __syncthreads();
        if (gate) {
            f32x4 sum = xch[(gm * 2 + gu) * 64 + lane];
#pragma unroll
            for (int p = 1; p < 8; ++p) sum += xch[(p * 4 + gm * 2 + gu) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = sum[j] + carry[j];
        }
    }
    __syncthreads();
    if (live && *s_bad) {        // a poll gave up: the first token's input gradient of this slice says so
        const unsigned long long q = 0x7FC07FC07FC07FC0ull;
        *reinterpret_cast<unsigned long long*>(a.dGI + (size_t)row * a.lddgi + u0) = q;
    }
}

// ---------------------------------------------------------------------------------------------------------------- launchers
int g_rgqa_butd_gru_persist = 3;      // rgqa_debug_set key 18: 0 = the host-driven recurrence (one GEMM + one gate kernel per token); 1 = persistent, 64-sample row groups x
                                      // 16-unit slices, W_hh slice in LDS, 4 waves; 2 = the same with 8 waves (the contraction split over wave pairs); 3 (default) = persistent,
                                      // 32-sample row groups x 32-unit slices, weights in registers, 8 waves (half the operand bytes per workgroup and token)

// the persistent launches apply: bf16, H = 1024, B <= 256 (at most 256 workgroups, every one resident at once: one per CU)
static int gru_row_groups(int B) { return g_rgqa_butd_gru_persist == 3 ? (B + 31) / 32 : (B + 63) / 64; }
static int gru_slices(int H) { return g_rgqa_butd_gru_persist == 3 ? H / 32 : H / 16; }
bool gru_persist_ok(int B, int H) {
    return g_rgqa_butd_gru_persist != 0 && H == 1024 && B >= 1 && B <= 256 && gru_row_groups(B) * gru_slices(H) <= rgqa_num_cus();
}
// [forward counters: row groups x (L + 1)] [backward counters: row groups x L] [error word, zeroed once by the owner] ... - laid out for the finer
// row groups (32 samples) whatever the form in use: the debug key may change between bind and launch
static int gru_rgm(int B) { return (B + 31) / 32; }
size_t gru_persist_counter_ints(int B, int L) { return (size_t)gru_rgm(B) * (2 * L + 1) + 16; }

int k_gru_fwd_persist(const bf16_t* GI, long ldgi, const bf16_t* W, int ldw, const float* bhh, bf16_t* Hall, bf16_t* Rg, bf16_t* Zg, bf16_t* Ng, bf16_t* GHN,
                      int B, int L, int H, int* counters, hipStream_t s) {
    RGQA_REQUIRE(gru_persist_ok(B, H), "gru_fwd_persist: B=%d H=%d not covered", B, H);
    constexpr int HH = 1024, LDS_BYTES = 48 * (2 * HH + 16) + 4 * 3 * 64 * 16, LDS32 = 8 * 12 * 64 * 16 + 16;
    const int RG = gru_row_groups(B), RGM = gru_rgm(B);
    static bool attr_dev[64] = {};     // hipFuncSetAttribute is per device
    int dev = 0;
    RGQA_HIP(hipGetDevice(&dev));
    bool& attr = attr_dev[dev & 63];
    if (!attr) {
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_fwd_persist_kernel<HH, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_fwd_persist_kernel<HH, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_fwd_persist32_kernel<HH>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS32));
        attr = true;
    }
    RGQA_HIP(hipMemsetAsync(counters, 0, sizeof(int) * (size_t)RGM * (2 * L + 1), s));
    GruFwdArgs a{GI, ldgi, W, ldw, bhh, Hall, Rg, Zg, Ng, GHN, B, L, counters, counters + (size_t)RGM * (2 * L + 1)};
    if (g_rgqa_butd_gru_persist == 3) hipLaunchKernelGGL((gru_fwd_persist32_kernel<HH>), dim3(RG * (HH / 32)), dim3(512), LDS32, s, a);
    else if (g_rgqa_butd_gru_persist != 2) hipLaunchKernelGGL((gru_fwd_persist_kernel<HH, 1>), dim3(RG * (HH / 16)), dim3(256), LDS_BYTES, s, a);
    else hipLaunchKernelGGL((gru_fwd_persist_kernel<HH, 2>), dim3(RG * (HH / 16)), dim3(512), LDS_BYTES, s, a);
    RGQA_LAUNCH_CHECK("gru_fwd_persist_kernel");
    return RGQA_OK;
}
int k_gru_bwd_persist(const bf16_t* dH, const bf16_t* Hall, const bf16_t* Rg, const bf16_t* Zg, const bf16_t* Ng, const bf16_t* GHN, bf16_t* dGI, long lddgi, bf16_t* dGH,
                      const bf16_t* WT, int ldwt, int B, int L, int H, int* counters, hipStream_t s) {
    RGQA_REQUIRE(gru_persist_ok(B, H) && ldwt >= 3 * H, "gru_bwd_persist: B=%d H=%d ldwt=%d not covered", B, H, ldwt);
    constexpr int HH = 1024, LDS_BYTES = 16 * (2 * 3 * HH + 16) + 4 * 64 * 16, LDS32 = 8 * 4 * 64 * 16 + 16;
    const int RG = gru_row_groups(B), RGM = gru_rgm(B);
    static bool attr_dev[64] = {};
    int dev = 0;
    RGQA_HIP(hipGetDevice(&dev));
    bool& attr = attr_dev[dev & 63];
    if (!attr) {
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_bwd_persist_kernel<HH, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_bwd_persist_kernel<HH, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        RGQA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_bwd_persist32_kernel<HH>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS32));
        attr = true;
    }
    RGQA_HIP(hipMemsetAsync(counters, 0, sizeof(int) * (size_t)RGM * (2 * L + 1), s));
    GruBwdArgs a{dH, Hall, Rg, Zg, Ng, GHN, dGI, lddgi, dGH, WT, ldwt, B, L, counters + (size_t)RGM * (L + 1), counters + (size_t)RGM * (2 * L + 1)};
    if (g_rgqa_butd_gru_persist == 3) hipLaunchKernelGGL((gru_bwd_persist32_kernel<HH>), dim3(RG * (HH / 32)), dim3(512), LDS32, s, a);
    else if (g_rgqa_butd_gru_persist != 2) hipLaunchKernelGGL((gru_bwd_persist_kernel<HH, 1>), dim3(RG * (HH / 16)), dim3(256), LDS_BYTES, s, a);
    else hipLaunchKernelGGL((gru_bwd_persist_kernel<HH, 2>), dim3(RG * (HH / 16)), dim3(512), LDS_BYTES, s, a);
    RGQA_LAUNCH_CHECK("gru_bwd_persist_kernel");
    return RGQA_OK;
}
