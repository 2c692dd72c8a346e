// Shared device/host helpers for the rgqa HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define RGQA_OK 0
#define RGQA_ERR_ARG (-1)
#define RGQA_ERR_HIP (-2)
#define RGQA_ERR_STATE (-3)
#define RGQA_ERR_WORKSPACE (-4)

void rgqa_set_error(const char* fmt, ...);
int rgqa_check_hip(hipError_t e, const char* what);
#define RGQA_HIP(x) do { int _r = rgqa_check_hip((x), #x); if (_r) return _r; } while (0)
#define RGQA_LAUNCH_CHECK(name) do { int _r = rgqa_check_hip(hipGetLastError(), name); if (_r) return _r; } while (0)
#define RGQA_REQUIRE(cond, ...) do { if (!(cond)) { rgqa_set_error(__VA_ARGS__); return RGQA_ERR_ARG; } } while (0)

// ---------------------------------------------------------------- scalar conversions
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// 4 consecutive elements, 8-/16-byte aligned
__device__ __forceinline__ void load4(const float* p, float v[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float v[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
__device__ __forceinline__ void store4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float v[4]) {
    bf16x4 t; t[0] = (bf16_t)v[0]; t[1] = (bf16_t)v[1]; t[2] = (bf16_t)v[2]; t[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = t;
}

// ---------------------------------------------------------------- split-f32 storage ("bf16x3" precision)
// An f32 value x is kept as two bf16 numbers hi = bf16(x), lo = bf16(x - hi): hi + lo carries 16-17 significant bits
// (relative error <= 2^-17) and products of such pairs are formed on the bf16 matrix pipe as hi*hi + hi*lo + lo*hi
// (three MFMAs, f32 accumulate; the dropped lo*lo term is below 2^-16 relative).  Layout: rows of ld elements, ld % 32 == 0,
// row starts 128-byte aligned; the 32 consecutive elements 32b .. 32b+31 of a row occupy ONE 128-byte line:
// bytes [0,64) their 32 hi parts, bytes [64,128) their 32 lo parts.  A matrix operand therefore reads as a bf16 matrix
// of 2*ld columns whose 64-column K-steps hold [32 hi | 32 lo] of 32 contraction elements: the LDS-DMA GEMM kernels
// stage it unchanged and only their MFMA streams differ.  Storage is 4 bytes per element, so `sf32*` pointer arithmetic
// in ELEMENTS addresses rows and 32-aligned columns exactly as `float*` would; inside a line the address of an
// element's parts is decoded from the slot address itself (sf_hi): no base pointer is needed.
struct sf32 { uint32_t slot; };      // one element slot; never read as a value
__device__ __forceinline__ const unsigned char* sf_hi(const sf32* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    return reinterpret_cast<const unsigned char*>((a & ~(uintptr_t)127) + ((a & 127) >> 1));      // lo part: + 64
}
__device__ __forceinline__ unsigned char* sf_hi(sf32* p) { return const_cast<unsigned char*>(sf_hi(const_cast<const sf32*>(p))); }
// (no contraction here: with x = a * b produced just before, "x - hi" would otherwise fuse into fma(a, b, -hi) in SOME instantiations of a
// kernel and not in others - e.g. when x has a second use - and two epilogue variants of the same GEMM would differ in the last bit of lo)
__device__ __forceinline__ void sf_split(float x, bf16_t& hi, bf16_t& lo) {
#pragma clang fp contract(off)
    hi = (bf16_t)x;
    const float r = x - (float)hi;
    lo = (bf16_t)r;
}
__device__ __forceinline__ float sf_load1(const sf32* p) {
    const unsigned char* h = sf_hi(p);
    return (float)*reinterpret_cast<const bf16_t*>(h) + (float)*reinterpret_cast<const bf16_t*>(h + 64);
}
__device__ __forceinline__ void sf_store1(sf32* p, float x) {
    unsigned char* h = sf_hi(p);
    bf16_t a, b; sf_split(x, a, b);
    *reinterpret_cast<bf16_t*>(h) = a; *reinterpret_cast<bf16_t*>(h + 64) = b;
}
// 4 consecutive elements, first index a multiple of 4
__device__ __forceinline__ void load4(const sf32* p, float v[4]) {
    const unsigned char* h = sf_hi(p);
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(h), b = *reinterpret_cast<const bf16x4*>(h + 64);
    v[0] = (float)a[0] + (float)b[0]; v[1] = (float)a[1] + (float)b[1]; v[2] = (float)a[2] + (float)b[2]; v[3] = (float)a[3] + (float)b[3];
}
__device__ __forceinline__ void store4(sf32* p, const float v[4]) {
    unsigned char* h = sf_hi(p);
    bf16x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { bf16_t x, y; sf_split(v[i], x, y); a[i] = x; b[i] = y; }
    *reinterpret_cast<bf16x4*>(h) = a; *reinterpret_cast<bf16x4*>(h + 64) = b;
}
// 8 consecutive elements, first index a multiple of 8: two 16-byte accesses
__device__ __forceinline__ void sf_load8(const sf32* p, float v[8]) {
    const unsigned char* h = sf_hi(p);
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(h), b = *reinterpret_cast<const bf16x8*>(h + 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i] + (float)b[i];
}
__device__ __forceinline__ void sf_store8(sf32* p, const float v[8]) {
    unsigned char* h = sf_hi(p);
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { bf16_t x, y; sf_split(v[i], x, y); a[i] = x; b[i] = y; }
    *reinterpret_cast<bf16x8*>(h) = a; *reinterpret_cast<bf16x8*>(h + 64) = b;
}
// scalar element access for any activation type (the generic LDS / VALU kernels)
__device__ __forceinline__ float ld_elem(const float* p) { return *p; }
__device__ __forceinline__ float ld_elem(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ float ld_elem(const sf32* p) { return sf_load1(p); }
__device__ __forceinline__ void st_elem(float* p, float x) { *p = x; }
__device__ __forceinline__ void st_elem(bf16_t* p, float x) { *p = (bf16_t)x; }
__device__ __forceinline__ void st_elem(sf32* p, float x) { sf_store1(p, x); }

// raw (unconverted) 4-element loads: lets a kernel issue the next row's loads before it consumes the current row
template <typename T> struct raw4;
template <> struct raw4<float> { float4 v; };
template <> struct raw4<bf16_t> { bf16x4 v; };
template <> struct raw4<sf32> { bf16x4 h, l; };
__device__ __forceinline__ void load_raw4(const sf32* p, raw4<sf32>& r) {
    const unsigned char* a = sf_hi(p);
    r.h = *reinterpret_cast<const bf16x4*>(a); r.l = *reinterpret_cast<const bf16x4*>(a + 64);
}
__device__ __forceinline__ void cvt_raw4(const raw4<sf32>& r, float v[4]) {
    v[0] = (float)r.h[0] + (float)r.l[0]; v[1] = (float)r.h[1] + (float)r.l[1]; v[2] = (float)r.h[2] + (float)r.l[2]; v[3] = (float)r.h[3] + (float)r.l[3];
}
__device__ __forceinline__ void load_raw4(const float* p, raw4<float>& r) { r.v = *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void load_raw4(const bf16_t* p, raw4<bf16_t>& r) { r.v = *reinterpret_cast<const bf16x4*>(p); }
__device__ __forceinline__ void cvt_raw4(const raw4<float>& r, float v[4]) { v[0] = r.v.x; v[1] = r.v.y; v[2] = r.v.z; v[3] = r.v.w; }
__device__ __forceinline__ void cvt_raw4(const raw4<bf16_t>& r, float v[4]) {
    v[0] = (float)r.v[0]; v[1] = (float)r.v[1]; v[2] = (float)r.v[2]; v[3] = (float)r.v[3];
}

// ---------------------------------------------------------------- wave (64-lane) reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- math
// exact-erf GeLU, as the reference's gelu() (lxrt/modeling.py:112-118)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// d/dx gelu = Phi(x) + x*phi(x)
__device__ __forceinline__ float dgelu_f(float x) {
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// gelu(x) and gelu'(x) from one erf / one exp (the exp inside the erf approximation IS the Gaussian pdf term).
// Phi(x) = 1 - h for x >= 0, h for x < 0, with h = 0.5 * erfc(|x|/sqrt 2) = (p(t) * t) * exp(-x^2/2), t = 1/(1 + 0.3275911 |x|/sqrt 2)
// (Abramowitz-Stegun 7.1.26, coefficients pre-multiplied by 0.5; |error| of Phi <= 8e-8).  v_rcp_f32 (1 ulp) instead of an IEEE
// division and a select instead of copysign + two FMAs: 17 VALU slots per element against 26 (the GELU epilogue is VALU-bound,
// ~11 us per 256 x 256 tile with the matrix pipe idle).
__device__ __forceinline__ void gelu_and_grad_fast(float x, float& g, float& dg) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x), 1.0f));
    float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));     // exp(-x^2/2)
    const float h = p * t * e;
    const float cdf = x >= 0.f ? 1.0f - h : h;
    g = x * cdf;
    dg = fmaf(x, 0.39894228040143267794f * e, cdf);
}

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(seed, site, idx): element idx draws the (idx & 1) 16-bit half of ONE 32-bit hash word of (seed, site, idx >> 1), so a
// hash (integer multiplies - the expensive VALU ops of the fused epilogues) serves two neighbouring elements; drop
// probability = round(p * 65536) / 65536.  The same (seed, site, idx) is re-derived in the backward pass, so no mask is
// stored.  `site` (folded into seed_hi) separates the dropout sites of one step.
// (seed, site) enter through ONE multiply on wave-uniform values (scalar ALU, free); the per-element part is the two-multiply
// "lowbias32" finaliser (v_mul_lo_u32 is a quarter-rate instruction: the hash is the most expensive part of the dropout epilogues
// and of the attention kernels, 216 of 4171 instructions in attn_bwd<3,3>).
__device__ __forceinline__ uint32_t rng_hash(uint32_t seed_lo, uint32_t seed_hi, uint32_t idx2) {
    uint32_t x = idx2 ^ (seed_lo + seed_hi * 0x9E3779B1u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
struct DropCfg {
    uint32_t seed_lo, seed_hi;  // seed_hi carries the site id
    uint32_t thresh;            // drop when the element's 16-bit draw < thresh  (thresh = p * 2^16); 0 = no dropout
    float scale;                // 1/(1-p)
};
__device__ __forceinline__ float drop_apply(const DropCfg& d, uint32_t idx, float v) {
    if (d.thresh == 0u) return v;
    const uint32_t h = rng_hash(d.seed_lo, d.seed_hi, idx >> 1);
    return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) < d.thresh ? 0.0f : v * d.scale;
}
// N (even) consecutive elements starting at an EVEN index: N/2 hashes. Same draws as drop_apply element by element.
template <int N>
__device__ __forceinline__ void drop_apply_vec(const DropCfg& d, uint32_t idx0, float (&v)[N]) {
    if (d.thresh == 0u) return;
#pragma unroll
    for (int j = 0; j < N; j += 2) {
        const uint32_t h = rng_hash(d.seed_lo, d.seed_hi, (idx0 >> 1) + (uint32_t)(j >> 1));
        v[j] = (h & 0xFFFFu) < d.thresh ? 0.0f : v[j] * d.scale;
        v[j + 1] = (h >> 16) < d.thresh ? 0.0f : v[j + 1] * d.scale;
    }
}
static inline DropCfg make_drop(float p, uint64_t seed, uint32_t site) {
    DropCfg d;
    d.seed_lo = (uint32_t)seed;
    d.seed_hi = (uint32_t)(seed >> 32) ^ (site * 0x632BE5ABu + 0x7F4A7C15u);
    if (p <= 0.f) { d.thresh = 0u; d.scale = 1.f; }
    else { double t = (double)p * 65536.0 + 0.5; d.thresh = t >= 65535.0 ? 65535u : (t < 1.0 ? 1u : (uint32_t)t); d.scale = 1.0f / (1.0f - p); }
    return d;
}
__host__ __device__ static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
