// device_common.hpp -- shared device helpers for the gfx950 TDOA kernels.
// wave = 64 lanes, LDS 160 KiB/CU; everything here is written for CDNA4 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdoa {

constexpr int kWave = 64;

struct SWDesc {            // one (station, window) unit: raw IQ bytes in HBM
    const uint8_t *base;   // first I byte of the window (2-byte aligned)
    int32_t len;           // samples in the window
    int32_t pad;
};

struct PWDesc {            // one (pair, window) unit
    int32_t sw_a;          // index of the template station-window inside the batch
    int32_t sw_b;          // index of the signal station-window inside the batch
    int32_t out_index;     // slot in the peak array
    int32_t len_a;         // template length (normalisation 1/sqrt(len_a))
};

struct QuadDesc {          // segment form: two template station-windows x two signal station-windows of one window
    int32_t sw_ta, sw_tb;  // templates (batch-relative station-window index; sw_tb = -1: none)
    int32_t sw_sc, sw_sd;  // signals (sw_sd = -1: none)
    int32_t pw[4];         // batch-relative pair-window of (ta,sc), (ta,sd), (tb,sc), (tb,sd); -1 = not wanted
};

// Complex arithmetic on register PAIRS: written on two-element vectors, the compiler emits the packed-f32 instructions
// (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32 -- one instruction per complex add, two per complex product) and folds swaps
// and negations into their op_sel / neg modifiers; written on the two floats of a struct, the same arithmetic came out
// with a third more instructions, most of them moves that rebuild pairs (measured on a 16-point butterfly stage: 115
// against 201).  Every kernel here waits for vector-instruction issue before it waits for memory.
typedef float cplx_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cplx_v cv(float2 a) { return cplx_v{a.x, a.y}; }
__device__ __forceinline__ float2 cf(cplx_v a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return cf(cv(a) + cv(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return cf(cv(a) - cv(b)); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    // (a.x b.x - a.y b.y, a.x b.y + a.y b.x) = b (a.x, a.x) + (-b.y, b.x) (a.y, a.y)
    const cplx_v av = cv(a), bv = cv(b);
    return cf(__builtin_elementwise_fma(cplx_v{-bv.y, bv.x}, av.yy, bv * av.xx));
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b)   // conj(a) * b
{
    // (a.x b.x + a.y b.y, a.x b.y - a.y b.x) = b (a.x, a.x) + (b.y, -b.x) (a.y, a.y)
    const cplx_v av = cv(a), bv = cv(b);
    return cf(__builtin_elementwise_fma(cplx_v{bv.y, -bv.x}, av.yy, bv * av.xx));
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// x + (x of another lane of the same 16-lane row) by a DPP move -- no LDS permute, no address register (__shfl_xor goes
// through ds_bpermute).  CTRL: 0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = quad_perm [2,3,0,1] (lane ^ 2), 0x141 =
// row_half_mirror (the other quad of an 8-lane half row once the quads agree), 0x140 = row_mirror (the other half row).
// (Spelled as one v_add_f32_dpp in inline assembly the pair kernel's FIR spilled: the separate result registers.)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x)
{
    return x + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}
// sum over aligned groups of 4 / of 16 lanes, the result in every lane of the group
__device__ __forceinline__ float quad_sum(float x) { return dpp_add<0x4E>(dpp_add<0xB1>(x)); }
__device__ __forceinline__ float row16_sum(float x) { return dpp_add<0x140>(dpp_add<0x141>(quad_sum(x))); }

// sin / cos of (pi/2)(q + u), |u| <= 1/2 (a hair beyond is harmless), q quarter turns: degree-9 / degree-8 Taylor polynomials
// (truncation < 3e-8) and a rotation by sign flips and one swap
__device__ __forceinline__ float2 unit_root_quarter(float u, int qi, bool positive)
{
    const float v = u * u;
    float s = 0.00016044118478735982f;           // (pi/2)^9 / 9!
    s = __builtin_fmaf(s, v, -0.0046817541353186881f);   // -(pi/2)^7 / 7!
    s = __builtin_fmaf(s, v, 0.079692626246167046f);     //  (pi/2)^5 / 5!
    s = __builtin_fmaf(s, v, -0.64596409750624625f);     // -(pi/2)^3 / 3!
    s = __builtin_fmaf(s, v, 1.5707963267948966f);       //  pi/2
    s = s * u;
    float c = 0.00091926027483942659f;           //  (pi/2)^8 / 8!
    c = __builtin_fmaf(c, v, -0.020863480763352961f);    // -(pi/2)^6 / 6!
    c = __builtin_fmaf(c, v, 0.25366950790104802f);      //  (pi/2)^4 / 4!
    c = __builtin_fmaf(c, v, -1.2337005501361698f);      // -(pi/2)^2 / 2!
    c = __builtin_fmaf(c, v, 1.0f);
    // rotate by q quarter turns: 0 (c,s)  1 (-s,c)  2 (-c,-s)  3 (s,-c)
    const float re = (qi & 1) ? s : c, im = (qi & 1) ? c : s;
    const unsigned int sre = ((unsigned int)(qi + 1) & 2u) << 30;   // negate re for q = 1, 2 (mod 4)
    const unsigned int sim = ((unsigned int)qi & 2u) << 30;         // negate im for q = 2, 3 (mod 4)
    const float rr = __uint_as_float(__float_as_uint(re) ^ sre);
    float ii = __uint_as_float(__float_as_uint(im) ^ sim);
    if (!positive) ii = -ii;
    return make_float2(rr, ii);
}

// exp(sign * 2*pi*i * num / den), den a power of two, 0 <= num < 2^24 (exact in f32).
// x = 2*num/den is an exact dyadic number, so the quadrant reduction x = q/2 + u/2, |u| <= 1/2, is
// exact; about 20 VALU instructions against ~60 for the library sincospif.
__device__ __forceinline__ float2 unit_root(float num, float inv_den_times2, bool positive)
{
    const float x = num * inv_den_times2;        // angle / pi, exact
    const float q = rintf(2.0f * x);             // quarter turns
    const float u = __builtin_fmaf(2.0f, x, -q); // exact, in [-1/2, 1/2]
    return unit_root_quarter(u, (int)q, positive);
}

// The same for a denominator that is NOT a power of two (round 5: transform lengths 5 x 2^k):
// exp(sign 2 pi i num / den), 0 <= num < 2^24 an integer, den a multiple of 4 with den / 4 exact in a float32 (5 x 2^k is).
// num / den is no longer a dyadic number, so the quadrant reduction is done on the INTEGERS: q = round(4 num / den) quarter
// turns -- any rounding of the quotient near a half is fine, the polynomials hold a little beyond |u| = 1/2 --, the
// remainder r = num - q den/4 is exact (one fma: |r| <= den/8 + 1, an integer below 2^24), and u = r / (den/4) is off by one
// rounding of a number below 1/2: 3e-8 of a quarter turn, the size of the polynomials' own truncation.  Four instructions
// more than the dyadic form.
__device__ __forceinline__ float2 unit_root_any(float num, float quarter_den, float inv_quarter_den, bool positive)
{
    const float q = rintf(num * inv_quarter_den);
    const float r = __builtin_fmaf(-q, quarter_den, num);       // exact
    return unit_root_quarter(r * inv_quarter_den, (int)q, positive);
}

// 64-bit peak key: [ |v| bits : 32 ][ (0x7fffffff - rank) : 31 ][ sign : 1 ]
// rank orders lags 0, +1, -1, +2, -2, ... so the larger key is the larger |v|,
// then the smaller |lag|, then the positive lag (processor.go:596-611 order).
__device__ __forceinline__ unsigned long long peak_key(float v, int lag)
{
    unsigned int mag = __float_as_uint(fabsf(v));
    unsigned int a = lag < 0 ? (unsigned int)(-lag) : (unsigned int)lag;
    unsigned int rank = 2u * a - (lag > 0 ? 1u : 0u);
    unsigned int low = ((0x7fffffffu - rank) << 1) | (v < 0.0f ? 1u : 0u);
    return ((unsigned long long)mag << 32) | low;
}

// Zero `n` 64-bit words.  Used instead of hipMemsetAsync inside the captured step: a kernel node like every other node
// of the graph.
__global__ void k_zero_u64(unsigned long long *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0ull;
}

// streaming accesses: data that is read once, or written for a reader that comes after gigabytes of other traffic, should
// not displace what the caches could still serve
typedef float f2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 load_nt(const float2 *p)
{
    const f2v_t x = __builtin_nontemporal_load(reinterpret_cast<const f2v_t *>(p));
    return make_float2(x.x, x.y);
}
__device__ __forceinline__ void store_nt(float2 *p, float2 v)
{
    f2v_t y;
    y.x = v.x;
    y.y = v.y;
    __builtin_nontemporal_store(y, reinterpret_cast<f2v_t *>(p));
}

// A value the compiler must treat as unknown: addresses derived from it are rebuilt where they are used instead of being
// hoisted out of a loop as dozens of invariants (which then spill).
__device__ __forceinline__ int opaque_i(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(k, off, kWave);
        k = o > k ? o : k;
    }
    return k;
}

}  // namespace tdoa
