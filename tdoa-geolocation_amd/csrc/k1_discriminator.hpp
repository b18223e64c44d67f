// k1_discriminator.hpp -- K1: u8 IQ -> phase-difference FM discriminator -> 24-bit phase code, plus the exact window
// statistics (mean / unit-power scale) mode B normalises with.
//
// Replaces (reference file:line): the u8 -> complex64 conversion of processor.go:195-201 and, for the north-star
// pipeline, the instantaneous-frequency demodulation that exists only in the prebuilt processor binary (SURVEY.md
// section 8, row K1): p = x_i conj(x_{i-1}), y_i = atan2(Im p, Re p) in (-pi, +pi], y_0 := y_1.  In units of pi/2^23:
//   a_i    = angle code of one IQ sample: the correctly rounded atan2(Q, I) 2^23/pi (it depends on the 2 bytes only)
//   code_i = the integer congruent to a_i - a_{i-1} modulo 2^24 in (-2^23, +2^23];  code_0 := code_1
// a(-I, -Q) = a(I, Q) -+ 2^23 exactly and collinear samples share a code (the table below holds gcd-reduced first-octant
// directions, octant / quadrant placement is integer arithmetic), while two directions that are NOT collinear stay at
// least 1/(|x_i||x_{i-1}|) >= 7.7e-6 rad = 20 code steps away from a reversal: a_i - a_{i-1} = -+2^23 happens for exactly
// reversed samples only, where atan2(+0, negative) = +pi -- the half-open interval is the whole rule.
// |code| <= 2^23 is exact in a float32, and the step (3.7e-7 rad) is the float32 resolution at that magnitude: this IS
// the float discriminator, with exact integer statistics.  (Rounds 1-2 stored a 16-bit code, step 9.6e-5 rad: on the
// simulators' +-1..3 LSB captures its rounding error is a fixed function of the sample pair and cost up to 4.6e-5 of a
// peak -- outside north_star's 1e-5.)
//
// The table: T[mx (mx + 1) / 2 + mn] = llround(atan2(2 mn' + 1, 2 mx' + 1) 2^23/pi) over the gcd-reduced pair, for
// 0 <= mn <= mx <= 127 (index of the odd magnitudes |2b - 255| = 2 idx + 1): 8256 int32, built by the HOST in float64
// (tdoa_mi355x.hip, k1_build_table_host).  The kernels keep tables DERIVED from it by the host with the integer placement
// rules (|Q| > |I|: 2^22 - c; I < 0: 2^23 - c; Q < 0: -c) in LDS: the first-quadrant table of the column kernels (128 x 128
// entries, scaled by 256, k1_angle_quadrant) and the half-plane table of the streaming pass (32768 entries,
// k1_direct_angle2).  (Round 3 started with the 33 KB first-octant table itself in LDS and a 17-instruction lookup pinned
// in inline assembly; the 11-instruction quadrant lookup replaced it once the column kernels became one workgroup per CU.)
//
// Two ways the codes reach the transforms:
//   * fused (default on the hot plans): k_fm_demod<false> only adds up the window sums (one streaming read of the
//     capture bytes, nothing written); the forward column kernels (fft_radix16.hpp, k_fwd_col256_k1 / k_fwd_col512_k1)
//     read the capture bytes themselves and evaluate the discriminator on the fly.  No code array exists.
//   * materialised: k_fm_demod<true> also writes the codes (int32, held NEGATED: stored = -code in [-2^23, 2^23)) for
//     the consumers that re-read samples many times or post-process them (segment form, k1_smooth, k1_gate, the
//     any-size column kernel).
#pragma once

#include "device_common.hpp"

namespace tdoa {

constexpr int kK1Half = 1 << 23;                 // code units per half turn (pi)
constexpr int kK1TableEntries = 128 * 129 / 2;   // first-octant directions, mn <= mx

// Table read at byte offset `off`.  ABS0: the table is known to start at LDS address 0 (a kernel whose only LDS is its
// dynamic segment; checked once per workgroup by k1_assert_lds0) -- the offset IS the address.  Otherwise the address is
// lut + off, and since the base of a dynamic LDS segment is a link-time symbol that costs a `v_add_u32 v, 0, v` per read.
template <bool ABS0>
__device__ __forceinline__ int k1_table_read(const int *lut, unsigned int off)
{
#if defined(__HIP_DEVICE_COMPILE__)      // (an LDS pointer is 32 bits wide in the device pass only)
    if (ABS0) return *__builtin_bit_cast(const __attribute__((address_space(3))) int *, off);
#endif
    return *reinterpret_cast<const int *>(reinterpret_cast<const char *>(lut) + off);
}

__device__ __forceinline__ void k1_assert_lds0(const int *lut)
{
    if ((unsigned int)(uintptr_t)(const __attribute__((address_space(3))) int *)lut != 0u) __builtin_trap();
}

// Angle code of an IQ sample (I = 2 b_I - 255, Q = 2 b_Q - 255) from the QUADRANT table (128 x 128 entries, Tq[iq][ia] =
// angle of (2 ia + 1, 2 iq + 1), 64 KB; the column kernels: one 1024-thread workgroup per CU).  This runs once per sample
// of the capture inside kernels that are short of vector-instruction issue slots, so the sequence is pinned in inline
// assembly: 11 instructions per sample, no compare, no select (left to itself the compiler turns the masks back into
// v_cmp + v_cndmask pairs plus their wait states on gfx950).
//   x   = index bytes (|2 b - 255| - 1) / 2 = b ^ (b >= 128 ? 0x80 : 0x7f)          (formed per dword by the caller)
//   neg = ~b: bit 7 set = the component is negative
//   each placement step is  c -> K - c  under a condition, written  (c ^ m) + (m & (K + 1))  with m = 0 or -1:
//   I < 0: K = half a turn;  Q < 0: K = 0.      HI = false: the sample in the low half of the dword, true: the high half.
// The quadrant table holds the angle codes SCALED BY 256 (a full
// turn = 2^32): a difference of two scaled angles wraps in the 32-bit subtraction itself, so k1_stored_code's sign
// extension of 24 bits disappears (k1_stored_code_scaled); 256 x code is still exact in a float32 and the factor is
// divided out with the window's scale (k1_normalise_scaled: bit-identical results).
// The look-up in two halves, so that a kernel can put several table reads in flight before it uses the first (round 4: the
// column kernels issue the 8 reads of four rows back to back; one read followed by its placement waits out the LDS
// latency 32 times per tile): byte offset of the entry (3 instructions), then -- on the value read -- the placement (6).
template <bool HI>
__device__ __forceinline__ unsigned int k1_quadrant_offset(unsigned int x)
{
    unsigned int oa, oq, off;
    if (!HI) {
        asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(oa) : "v"(x));
        asm("v_lshlrev_b32_sdwa %0, 9, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(oq) : "v"(x));
    } else {
        asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(oa) : "v"(x));
        asm("v_lshlrev_b32_sdwa %0, 9, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(oq) : "v"(x));
    }
    asm("v_or_b32 %0, %1, %2" : "=v"(off) : "v"(oa), "v"(oq));                    // byte offset 4 (128 iq + ia)
    return off;
}

template <bool HI>
__device__ __forceinline__ int k1_quadrant_place(int c, unsigned int neg)
{
    int k, mi, mq;
    if (!HI) asm("v_bfe_i32 %0, %1, 7, 1" : "=v"(mi) : "v"(neg));               // -1: I < 0
    else asm("v_bfe_i32 %0, %1, 23, 1" : "=v"(mi) : "v"(neg));
    asm("v_and_b32 %0, 0x80000001, %1" : "=v"(k) : "v"(mi));
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(c) : "v"(c), "v"(mi), "v"(k));       // c -> 2^31 - c (half a turn of the scaled code)
    if (!HI) asm("v_bfe_i32 %0, %1, 15, 1" : "=v"(mq) : "v"(neg));              // -1: Q < 0
    else asm("v_ashrrev_i32 %0, 31, %1" : "=v"(mq) : "v"(neg));
    asm("v_xor_b32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(mq));
    asm("v_sub_u32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(mq));                    // c -> -c
    return c;
}

template <bool HI, bool ABS0 = false>
__device__ __forceinline__ int k1_angle_quadrant(unsigned int x, unsigned int neg, const int *qlut)
{
    return k1_quadrant_place<HI>(k1_table_read<ABS0>(qlut, k1_quadrant_offset<HI>(x)), neg);
}

__device__ __forceinline__ unsigned int k1_index_bytes(unsigned int w) { return w ^ (0x7f7f7f7fu + ((w >> 7) & 0x01010101u)); }

// the two samples of a dword w = b_I0 | b_Q0 << 8 | b_I1 << 16 | b_Q1 << 24
template <bool ABS0 = false>
__device__ __forceinline__ void k1_angle2_quadrant(unsigned int w, const int *qlut, int &a0, int &a1)
{
    const unsigned int x = k1_index_bytes(w), neg = ~w;
    a0 = k1_angle_quadrant<false, ABS0>(x, neg, qlut);
    a1 = k1_angle_quadrant<true, ABS0>(x, neg, qlut);
}

// The streaming K1 kernel (k_fm_demod) has the LDS to itself and keeps a DIRECT table instead: the point reflection
// (I, Q) -> (-I, -Q) is b -> 255 - b = ~b on both bytes and changes the angle by exactly half a turn, so 32768 entries
// cover the half plane Q > 0: D[b_I | (b_Q & 0x7f) << 8] = a(I, Q) in (0, 2^23), b_Q >= 128  (128 KB, built by the host
// from the same first-octant codes).  Lookup of the two samples of a dword: 11 instructions, against 2 x 11 above.
// Returns the angle modulo 2^24 (in [0, 2^24)); only differences of angles are ever used.
constexpr int kK1QuadrantEntries = 128 * 128;
constexpr size_t kK1QuadrantBytes = sizeof(int) * kK1QuadrantEntries;     // 65 536
constexpr int kK1DirectEntries = 32768;
constexpr size_t kK1DirectBytes = sizeof(int) * kK1DirectEntries;      // 131 072

__device__ __forceinline__ void k1_direct_angle2(unsigned int w, const int *dlut, int &a0, int &a1)
{
    typedef short short2v __attribute__((ext_vector_type(2)));
    const short2v nq = __builtin_bit_cast(short2v, ~w) >> (short)15;          // per half: 0xffff if Q < 0 (v_pk_ashrrev_i16)
    const unsigned int pm = __builtin_bit_cast(unsigned int, nq);
    const unsigned int fw = w ^ pm;                                            // reflected where Q < 0: now b_Q >= 128
    const int c0 = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(dlut) + ((fw << 2) & 0x1fffcu));
    const int c1 = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(dlut) + ((fw >> 14) & 0x1fffcu));
    a0 = c0 | (int)((pm << 8) & 0x800000u);                                    // + half a turn (mod 2^24) if reflected
    a1 = c1 | (int)((pm >> 8) & 0x800000u);
}

__device__ __forceinline__ int k1_direct_angle(unsigned int s, const int *dlut)
{
    int a0, a1;
    k1_direct_angle2(s | 0x80000000u, dlut, a0, a1);                           // (high half: any sample with Q > 0)
    return a0;
}

// lane i takes the value of lane i - 1 (lane 0 keeps its own): one DPP move
__device__ __forceinline__ int wave_shift_right1(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// stored (negated) phase code of a sample with angle code a_cur after one with a_prev: -(code) in [-2^23, 2^23)
__device__ __forceinline__ int k1_stored_code(int a_cur, int a_prev)
{
    return (int)((unsigned int)(a_prev - a_cur) << 8) >> 8;        // v_bfe_i32: sign-extend the low 24 bits
}

// the same from angle codes scaled by 256 (quadrant table): 256 x stored, by the wrap of the subtraction
__device__ __forceinline__ int k1_stored_code_scaled(int a_cur, int a_prev) { return (int)((unsigned int)a_prev - (unsigned int)a_cur); }

// normalised discriminator sample from a STORED code: (float(code) - mean) * scale with code = -stored
__device__ __forceinline__ float k1_normalise(int stored, float mean, float scale)
{
#pragma clang fp contract(off)
    float d = -(float)stored - mean;
    return d * scale;
}

// Single-look K1 (k1_single_look.hpp): the subtrahend is an integer m0 and the factor a power of two s0, so
// (float(code) - m0) s0 is exact whenever |code - m0| < 2^24 and one fused multiply-add gives the same float:
// w = float(stored) * nscale + off,  nscale = -s0,  off = -m0 s0   (2 instructions per sample instead of 3).  Beyond 2^24
// the two forms may differ by one rounding of a 25-bit difference (2^-24 relative, once).
__device__ __forceinline__ float k1_normalise_fma(int stored, float nscale, float off)
{
    return __builtin_fmaf((float)stored, nscale, off);
}

// window sums of the codes.  code^2 < 2^46, so S2 needs more than 64 bits for long windows: a workgroup's partial sum
// (< 2^63) is added as its low 32 bits into s2a and the rest into s2b; S2 = s2b 2^32 + s2a.
struct StatsPartial {
    long long s1;
    unsigned long long s2a, s2b;
    unsigned long long pad;
};

struct FmStats {          // mirrors tdoa_fm_stats
    long long s1;
    unsigned long long s2_lo, s2_hi;
    float mean, scale;
};

__device__ __forceinline__ void stats_atomic_add(StatsPartial *acc, long long s1, unsigned long long s2)
{
    if (s1) atomicAdd(reinterpret_cast<unsigned long long *>(&acc->s1), (unsigned long long)s1);      // two's complement
    if (s2) {
        atomicAdd(&acc->s2a, s2 & 0xffffffffull);
        atomicAdd(&acc->s2b, s2 >> 32);
    }
}

constexpr int kDemodThreads = 1024;
constexpr int kDemodChunks = 4;      // 8-sample chunks a lane owns per wave step: 4 x 16-byte loads in flight per lane
constexpr int kDemodPiece = 512 * kDemodChunks;   // samples per wave step
constexpr int kDemodItem = 32;       // pieces per workgroup item: 2 steps of 16 waves (one set of atomics per 65536 samples)

// Capture bytes are read through pointers in the GLOBAL address space: a pointer that comes out of a descriptor in
// memory is generic to the compiler, and generic (flat) loads are ordered against LDS traffic -- every one of them was
// followed by s_waitcnt vmcnt(0) lgkmcnt(0), i.e. one load in flight per wave.
typedef const __attribute__((address_space(1))) uint16_t *gptr16;

__device__ __forceinline__ gptr16 k1_global(const uint8_t *base)
{
    return (gptr16)(const uint16_t *)base;
}

// two consecutive IQ samples p[i], p[i + 1] as one dword (windows start on any 2-byte boundary: the load is declared
// 2-byte aligned, global memory takes it as one access)
__device__ __forceinline__ unsigned int k1_fetch2(gptr16 p, long long i)
{
    typedef unsigned int __attribute__((aligned(2))) u32_a2;
    return *(const __attribute__((address_space(1))) u32_a2 *)(p + i);
}

// eight consecutive IQ samples p[i0 .. i0+7] as four dwords, for any 2-byte alignment of the window start:
// one 16-byte load when the address is 4-byte aligned (global memory needs no more), else eight 2-byte loads
__device__ __forceinline__ uint4 k1_fetch8(gptr16 p, int i0)
{
    const uintptr_t a = (uintptr_t)(p + i0);
    if ((a & 3u) == 0) {
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        typedef v4u __attribute__((aligned(4))) v4u_a4;
        const v4u r = *(const __attribute__((address_space(1))) v4u_a4 *)(p + i0);
        return make_uint4(r.x, r.y, r.z, r.w);
    }
    unsigned int h[8];
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = p[i0 + k];
    return make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}

__device__ __forceinline__ void k1_unpack8(uint4 q, unsigned int (&s)[9])
{
    s[1] = q.x & 0xffffu; s[2] = q.x >> 16; s[3] = q.y & 0xffffu; s[4] = q.y >> 16;
    s[5] = q.z & 0xffffu; s[6] = q.z >> 16; s[7] = q.w & 0xffffu; s[8] = q.w >> 16;
}

__device__ __forceinline__ void k1_load8(gptr16 p, int i0, unsigned int (&s)[9]) { k1_unpack8(k1_fetch8(p, i0), s); }

// ---- optional power gate (tdoa_params.k1_gate; the prebuilt binary's preprocessSignal) -----
// mean power p = mean |x|^2 of x = (b - 127.5)/127.5 is M / (65025 len) with the exact integer
// M = sum (2I-255)^2 + (2Q-255)^2; windows with p <= 0.01 take the envelope |x| instead of the discriminator.
// Envelope code = round-half-up(16384 sqrt(m)) = (isqrt(m << 30) + 1) >> 1, m <= 130050 (code <= 5 908 471 < 2^23):
// exact in a float32 like the phase codes, held negated like them, same statistics and normalisation downstream.
__device__ __forceinline__ bool k1_envelope_class(unsigned long long power_sum, int len)
{
    return 100ull * power_sum <= 65025ull * (unsigned long long)len;
}

__device__ __forceinline__ unsigned int k1_sample_power(unsigned int s)     // s = I | Q << 8
{
    const int a = 2 * (int)(s & 0xffu) - 255, b = 2 * (int)(s >> 8) - 255;
    return (unsigned int)(a * a + b * b);
}

__device__ __forceinline__ int k1_envelope_code(unsigned int s)
{
    const unsigned long long x = (unsigned long long)k1_sample_power(s) << 30;
    unsigned long long r = (unsigned long long)__builtin_sqrt((double)x);     // within a unit of isqrt(x)
#pragma unroll
    for (int k = 0; k < 2; k++)
        if (r * r > x) r--;
#pragma unroll
    for (int k = 0; k < 2; k++)
        if ((r + 1) * (r + 1) <= x) r++;
    return (int)((r + 1) >> 1);
}

// M of every station-window.  grid (ceil(maxlen / 2048), n_sw), 256 threads; power zeroed before the launch.
__global__ __launch_bounds__(256) void k_k1_power(const SWDesc *sw, unsigned long long *power)
{
    __shared__ unsigned long long red[4];
    const SWDesc d = sw[blockIdx.y];
    const gptr16 p = k1_global(d.base);
    const int i0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 8;
    unsigned long long m = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (i0 + k < d.len) m += k1_sample_power(p[i0 + k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off, kWave);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = red[0] + red[1] + red[2] + red[3];
        if (t) atomicAdd(&power[blockIdx.y], t);
    }
}

// a thread's eight stored codes -> two 16-byte stores (rows are 32-byte aligned and padded to a multiple of 8)
__device__ __forceinline__ void k1_store8(int *dst, const int (&c)[8])
{
    reinterpret_cast<int4 *>(dst)[0] = make_int4(c[0], c[1], c[2], c[3]);
    reinterpret_cast<int4 *>(dst)[1] = make_int4(c[4], c[5], c[6], c[7]);
}

// The same 8 codes as 24 bytes (PACKED code rows, round 4: the segment form's array -- a stored code has 24 significant
// bits, and that consumer reads every code once or twice, so the fourth byte was a quarter of both kernels' traffic).
// Code i of a row lives at bytes [3 i, 3 i + 3), little-endian; a reader loads the (unaligned) dword at 3 i and keeps its
// low 24 bits sign-extended (k1_code3_at).  dst is 8-byte aligned: rows start on multiples of 24 bytes and a thread's 8
// codes at a multiple of 8 codes.
__device__ __forceinline__ void k1_store8_packed(unsigned char *dst, const int (&c)[8])
{
    unsigned int w[6];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const unsigned int a = (unsigned int)c[4 * h] & 0xffffffu, b = (unsigned int)c[4 * h + 1] & 0xffffffu,
                           d = (unsigned int)c[4 * h + 2] & 0xffffffu, e = (unsigned int)c[4 * h + 3];
        w[3 * h] = a | (b << 24);
        w[3 * h + 1] = (b >> 8) | (d << 16);
        w[3 * h + 2] = (d >> 16) | (e << 8);
    }
    reinterpret_cast<uint2 *>(dst)[0] = make_uint2(w[0], w[1]);
    reinterpret_cast<uint2 *>(dst)[1] = make_uint2(w[2], w[3]);
    reinterpret_cast<uint2 *>(dst)[2] = make_uint2(w[4], w[5]);
}

// stored code at byte offset `off` (= 3 i) of a packed row: one global_load_dword (SGPR row base + 32-bit lane offset; the
// unaligned access mode of HSA queues) + v_bfe_i32.  The dword of a row's last code ends one byte past the row: inside
// the allocation (rows are followed by another row or by the array's slack).
typedef int __attribute__((aligned(1))) k1_unaligned_int;
__device__ __forceinline__ int k1_code3_at(const unsigned char *row, unsigned int off)
{
    const int v = *reinterpret_cast<const k1_unaligned_int *>(row + off);
    return (int)((unsigned int)v << 8) >> 8;
}

// wave sums of a thread's (s1, s2) -> block sums -> the window's accumulators; 256-thread kernels
__device__ __forceinline__ void k1_block_stats_256(long long s1, unsigned long long s2, StatsPartial *acc)
{
    __shared__ long long red1[4];
    __shared__ unsigned long long red2[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, kWave);
        s2 += __shfl_xor(s2, off, kWave);
    }
    if ((threadIdx.x & 63) == 0) {
        red1[threadIdx.x >> 6] = s1;
        red2[threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        stats_atomic_add(acc, red1[0] + red1[1] + red1[2] + red1[3], red2[0] + red2[1] + red2[2] + red2[3]);
}

// envelope codes + window sums of the windows in the envelope class (k_fm_demod skips those).  Same grid.
__global__ __launch_bounds__(256) void k_k1_envelope(const SWDesc *sw, const unsigned long long *power, int *codes,
                                                     long long code_stride, StatsPartial *acc)
{
    const SWDesc d = sw[blockIdx.y];
    if (!k1_envelope_class(power[blockIdx.y], d.len)) return;
    const gptr16 p = k1_global(d.base);
    int *dst = codes + (size_t)blockIdx.y * code_stride;
    const int i0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 8;
    long long s1 = 0;
    unsigned long long s2 = 0;
    if (i0 < d.len) {
        int c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int e = 0;
            if (i0 + k < d.len) {
                e = k1_envelope_code(p[i0 + k]);
                s1 += e;
                s2 += (unsigned long long)((long long)e * e);
            }
            c[k] = -e;                                                   // stored = -code
        }
        k1_store8(dst + i0, c);
    }
    k1_block_stats_256(s1, s2, &acc[blockIdx.y]);
}

// K1 streaming pass: persistent 1024-thread workgroups (one per CU) keep the DIRECT angle table in LDS (128 KB);
// after loading it the 16 waves of a workgroup run independently, each taking (station-window, 2048-sample piece)
// work items round-robin.  Window sums go straight into per-window integer accumulators with atomic adds: exact,
// hence independent of arrival order.
// WRITE = false: the statistics pre-pass of the fused path -- capture bytes in, three atomics per 65536 samples out.
// WRITE = true: also the stored codes, codes[n_sw][code_stride] int32, code_stride a multiple of 8; PACK3: the same rows
// at 3 bytes per code (k1_store8_packed; row w starts at byte 3 code_stride w of `codes`).
// acc: [n_sw], zeroed before the launch.  power: nullptr, or the windows' power sums (optional gate: windows in the
// envelope class are skipped here).
template <bool WRITE, bool PACK3 = false>
__global__ __launch_bounds__(kDemodThreads) void k_fm_demod(const SWDesc *sw, int n_sw, int pieces_per_window,
                                                            const int *dtable, int *codes, long long code_stride,
                                                            StatsPartial *acc, const unsigned long long *power)
{
    extern __shared__ int dlut[];                // kK1DirectEntries angles (128 KB: one workgroup per CU)
    for (int k = threadIdx.x; k < kK1DirectEntries / 4; k += kDemodThreads)
        reinterpret_cast<int4 *>(dlut)[k] = reinterpret_cast<const int4 *>(dtable)[k];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    // work item of a WORKGROUP = kDemodItem consecutive pieces of one window; at every step its 16 waves take 16
    // adjacent pieces, so the workgroup streams 64 KB of contiguous capture bytes per step (one DRAM-friendly run,
    // like a row of the FFT passes) instead of 16 unrelated 4 KB reads.  Consecutive items belong to DIFFERENT
    // windows, so workgroups that run together add into different accumulators.
    __shared__ long long red1[kDemodThreads / kWave];
    __shared__ unsigned long long red2[kDemodThreads / kWave];
    const int runs_per_window = (pieces_per_window + kDemodItem - 1) / kDemodItem;
    const int n_items = n_sw * runs_per_window;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int w = item % n_sw, run = item / n_sw;
        const SWDesc d = sw[w];
        const int len = d.len;
        if (power && k1_envelope_class(power[w], len)) continue;      // power gate on: k_k1_envelope has this window
        const gptr16 p = k1_global(d.base);
        int *out = WRITE && !PACK3 ? codes + (size_t)w * code_stride : nullptr;
        unsigned char *out3 = PACK3 ? reinterpret_cast<unsigned char *>(codes) + 3 * (size_t)w * (size_t)code_stride : nullptr;
        long long s1 = 0;
        unsigned long long s2 = 0;
        for (int piece = run * kDemodItem + wv; piece < (run + 1) * kDemodItem; piece += kDemodThreads / kWave) {
            const int start = piece * kDemodPiece;
            if (start >= len) break;
            if (start >= 1 && start + kDemodPiece <= len) {
                // whole piece inside the window (all but the first and last pieces): a lane owns samples [i0, i0 + 8) of
                // each 512-sample chunk; all loads are issued before any is used.  Every sample is looked up ONCE: the
                // angle before a lane's chunk is the last angle of the lane to its left (one DPP move); lane 0 takes lane
                // 63 of the previous chunk, and for chunk 0 the one sample before the piece.
                uint4 qs[kDemodChunks];
                const unsigned int before = p[start - 1];
#pragma unroll
                for (int h = 0; h < kDemodChunks; h++) qs[h] = k1_fetch8(p, start + h * 512 + lane * 8);
                int carry = k1_direct_angle(before, dlut);
                int t1 = 0;                      // 32 stored codes of |.| <= 2^23: fits
                // code^2 <= 2^46 and a lane adds 32 of them per piece: below 2^53, so a float64 accumulator is EXACT here
                // (v_cvt_f64_i32 + v_fma_f64: two instructions per sample against four for the 64-bit integer square-add)
                double t2 = 0.0;
#pragma unroll
                for (int h = 0; h < kDemodChunks; h++) {
                    int a[9], c[8];
                    k1_direct_angle2(qs[h].x, dlut, a[1], a[2]);
                    k1_direct_angle2(qs[h].y, dlut, a[3], a[4]);
                    k1_direct_angle2(qs[h].z, dlut, a[5], a[6]);
                    k1_direct_angle2(qs[h].w, dlut, a[7], a[8]);
                    const int left = wave_shift_right1(a[8]);
                    a[0] = lane ? left : carry;
                    carry = __builtin_amdgcn_readlane(a[8], kWave - 1);
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        c[k] = k1_stored_code(a[k + 1], a[k]);
                        t1 -= c[k];
                        const double cd = (double)c[k];
                        t2 = __builtin_fma(cd, cd, t2);
                    }
                    if (PACK3) k1_store8_packed(out3 + 3 * (size_t)(start + h * 512 + lane * 8), c);
                    else if (WRITE) k1_store8(out + start + h * 512 + lane * 8, c);
                }
                s1 += t1;
                s2 += (unsigned long long)t2;
            } else {
                // window head (code_0 := code_1) and tail, sample by sample; samples beyond len carry code 0 in memory and
                // do not enter the sums
#pragma unroll 1
                for (int h = 0; h < kDemodChunks; h++) {
                    const int i0 = start + h * 512 + lane * 8;
                    if (i0 >= len) continue;
                    int c[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int i = i0 + k;
                        int v = 0;
                        if (i < len && len >= 2) {
                            const int ii = i == 0 ? 1 : i;
                            v = k1_stored_code(k1_direct_angle(p[ii], dlut), k1_direct_angle(p[ii - 1], dlut));
                        }
                        c[k] = v;
                        if (i < len) {
                            s1 -= v;
                            s2 += (unsigned long long)((long long)v * v);
                        }
                    }
                    if (PACK3) k1_store8_packed(out3 + 3 * (size_t)i0, c);
                    else if (WRITE) k1_store8(out + i0, c);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off, kWave);
            s2 += __shfl_xor(s2, off, kWave);
        }
        if (lane == 0) {
            red1[wv] = s1;
            red2[wv] = s2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long t1 = 0;
            unsigned long long t2 = 0;
            for (int k = 0; k < kDemodThreads / kWave; k++) { t1 += red1[k]; t2 += red2[k]; }     // t2 < 2^46 * 65536
            stats_atomic_add(&acc[w], t1, t2);
        }
        __syncthreads();
    }
}

// Optional smoothing of the discriminator output (tdoa_params.k1_smooth = W; the prebuilt reference binary runs
// applyLowPassFilter(10) between removeDCBias and normalizeSignal): centred moving average with half-window h = W / 2,
// taps outside the window dropped (processor.go:270-296), in exact integer arithmetic on the phase codes:
//   lp_i = floor((2 S + c) / (2 c)),  S = sum of the c in-range codes code_{i-h} .. code_{i+h}     (round half up)
// The average of a constant is that constant, so LP(y - mean) = LP(y) - mean: smoothing the codes and normalising them
// afterwards with THEIR mean and variance is the binary's order up to the edge samples' share of the mean (O(h / L)).
// A thread makes 8 consecutive outputs; the window sums of the smoothed codes go to `acc` like k_fm_demod's.
// grid (ceil(maxlen / 2048), n_sw), 256 threads.
__global__ __launch_bounds__(256) void k_k1_smooth(const SWDesc *sw, const int *in, int *out, long long code_stride, int h0,
                                                   StatsPartial *acc, const unsigned long long *power)
{
    const int len = sw[blockIdx.y].len;
    // power gate on: the binary smooths its discriminator output only; an envelope window is copied (h = 0)
    const int h = power && k1_envelope_class(power[blockIdx.y], len) ? 0 : h0;
    const int *src = in + (size_t)blockIdx.y * code_stride;
    int *dst = out + (size_t)blockIdx.y * code_stride;
    const int i0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 8;
    long long s1 = 0;
    unsigned long long s2 = 0;
    if (i0 < len) {
        int c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + k;
            int v = 0;
            if (i < len) {
                const int lo = i - h < 0 ? 0 : i - h, hi = i + h >= len ? len - 1 : i + h;
                long long sum = 0;
                for (int j = lo; j <= hi; j++) sum -= (long long)src[j];      // stored = -code
                const long long cnt = hi - lo + 1;
                const long long num = 2 * sum + cnt, den = 2 * cnt;           // floor division, den > 0
                const long long lp = num >= 0 ? num / den : -((-num + den - 1) / den);
                v = (int)-lp;
                s1 += lp;
                s2 += (unsigned long long)(lp * lp);
            }
            c[k] = v;
        }
        k1_store8(dst + i0, c);
    }
    k1_block_stats_256(s1, s2, &acc[blockIdx.y]);
}

// mean and scale of every station-window from its exact sums, in f64:
// S2 as (double)(S2 >> 32) * 2^32 + (double)(S2 & 0xffffffff), var = (S2 - S1^2/L) / L
__global__ void k_fm_stats_final(const SWDesc *sw, const StatsPartial *acc, FmStats *stats, int n_sw)
{
#pragma clang fp contract(off)
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_sw) return;
    const int len = sw[id].len;
    const long long s1 = acc[id].s1;
    // canonical 128-bit value of s2b 2^32 + s2a
    const unsigned long long a = acc[id].s2a, b = acc[id].s2b;
    const unsigned long long lo = a + (b << 32);
    const unsigned long long hi = (b >> 32) + (lo < a ? 1ull : 0ull);
    FmStats out;
    out.s1 = s1;
    out.s2_lo = lo;
    out.s2_hi = hi;
    if (len == 0) {
        out.mean = 0.0f;
        out.scale = 1.0f;
    } else {
        const double dn = (double)len;
        out.mean = (float)((double)s1 / dn);
        const double m2 = ((double)s1 * (double)s1) / dn;
        const double s2d = (double)((hi << 32) | (lo >> 32)) * 4294967296.0 + (double)(lo & 0xffffffffull);
        const double var = (s2d - m2) / dn;
        out.scale = var > 0 ? (float)(1.0 / sqrt(var)) : 1.0f;
    }
    stats[id] = out;
}

// inspection hook: the normalised discriminator output of window 0 (from materialised codes)
__global__ void k_fm_dump(const SWDesc *sw, const int *codes, const FmStats *stats, float *out)
{
    const FmStats st = stats[0];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < sw[0].len) out[i] = k1_normalise(codes[i], st.mean, st.scale);
}

}  // namespace tdoa
