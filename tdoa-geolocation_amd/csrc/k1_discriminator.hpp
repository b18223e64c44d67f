// k1_discriminator.hpp -- K1: u8 IQ -> phase-difference FM discriminator -> 16-bit phase
// code, plus the exact window statistics (mean / unit-power scale) mode B normalises with.
//
// Replaces (reference file:line): the u8 -> complex64 conversion of processor.go:195-201 and,
// for the north-star pipeline, the instantaneous-frequency demodulation that exists only in
// the prebuilt processor binary:
//   phase_i = arg(x_i * conj(x_{i-1})) = wrap(arg(x_i) - arg(x_{i-1})),  phase_0 := phase_1
//   code_i  = int16(rint(phase_i * 32768/pi))
// One streaming pass (k_fm_demod) reads the capture bytes once, writes 2 bytes of code per
// sample and accumulates sum(code), sum(code^2) as exact integers; the FFT pass then reads
// the codes.  The arithmetic is an explicit sequence of correctly rounded f32 operations
// (table reciprocal, FMA Horner, no IEEE division in the loop) so the CPU restatement matches
// it bit for bit.
#pragma once

#include "device_common.hpp"

namespace tdoa {

// Reciprocal table RCP[k] = f32(1 / (2k+1)), k < 128, kept in LDS (random per-lane index).
// Every kernel that evaluates K1 calls k1_init_rcp() once (it contains a barrier).
__device__ __forceinline__ void k1_init_rcp(float *rcp)
{
#pragma clang fp contract(off)
    for (int k = threadIdx.x; k < 128; k += blockDim.x) rcp[k] = 1.0f / (float)(2 * k + 1);
    __syncthreads();
}

// theta = arg(I + iQ), I = 2 b_I - 255, Q = 2 b_Q - 255, for one IQ sample s = b_I | b_Q << 8:
// t = min * RCP[max], degree-7 Horner in t^2 with fused multiply-adds, octant fix-ups.
__device__ __forceinline__ float k1_theta(unsigned int s, const float *rcp)
{
#pragma clang fp contract(off)
    const float fi = __builtin_fmaf(2.0f, (float)(s & 0xffu), -255.0f);
    const float fq = __builtin_fmaf(2.0f, (float)((s >> 8) & 0xffu), -255.0f);
    const float ax = fabsf(fi), ay = fabsf(fq);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * rcp[(unsigned int)mx >> 1];
    const float z = t * t;
    float p = -0x1.31f904p-8f;
    p = __builtin_fmaf(p, z, 0x1.8bf058p-6f);
    p = __builtin_fmaf(p, z, -0x1.e655d6p-5f);
    p = __builtin_fmaf(p, z, 0x1.95c0f4p-4f);
    p = __builtin_fmaf(p, z, -0x1.1f0f46p-3f);
    p = __builtin_fmaf(p, z, 0x1.98f84ep-3f);
    p = __builtin_fmaf(p, z, -0x1.5551bcp-2f);
    p = __builtin_fmaf(p, z, 0x1.fffffcp-1f);
    float a = p * t;
    a = ay > ax ? 1.5707963705062866f - a : a;
    a = (s & 0x80u) ? a : 3.1415927410125732f - a;            // I < 0  <=>  b_I < 128
    return __uint_as_float(__float_as_uint(a) ^ ((~s & 0x8000u) << 16));   // Q < 0: negate
}

// wrapped phase step theta1 - theta0
__device__ __forceinline__ float k1_wrap_diff(float th1, float th0)
{
#pragma clang fp contract(off)
    const float d = th1 - th0;
    // branch-free: add -2pi, +2pi or 0 (the codes derived from d are unaffected by d + 0.0f)
    const float adj = d > 3.1415927410125732f ? -6.2831854820251465f : (d < -3.1415927410125732f ? 6.2831854820251465f : 0.0f);
    return d + adj;
}

// phase -> 16-bit code, pi == 32768 (wraps to -32768)
__device__ __forceinline__ int k1_code(float phase)
{
#pragma clang fp contract(off)
    return (int)(short)__float2int_rn(phase * 10430.3779296875f);
}

__device__ __forceinline__ float k1_normalise(int code, float mean, float scale)
{
#pragma clang fp contract(off)
    float d = (float)code - mean;
    return d * scale;
}

struct StatsPartial {
    long long s1;
    unsigned long long s2;
};

struct FmStats {          // mirrors tdoa_fm_stats
    long long s1;
    unsigned long long s2_lo, s2_hi;
    float mean, scale;
};

constexpr int kStatsChunk = 16384;   // samples per block
constexpr int kStatsThreads = 256;

// nine consecutive IQ samples p[i0-1 .. i0+7] with whatever alignment the window start has
__device__ __forceinline__ void k1_load9(const uint16_t *p, int i0, unsigned int (&s)[9])
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p + i0);
    if ((a & 15u) == 0) {
        const uint4 q = *reinterpret_cast<const uint4 *>(p + i0);
        s[1] = q.x & 0xffffu; s[2] = q.x >> 16; s[3] = q.y & 0xffffu; s[4] = q.y >> 16;
        s[5] = q.z & 0xffffu; s[6] = q.z >> 16; s[7] = q.w & 0xffffu; s[8] = q.w >> 16;
    } else if ((a & 3u) == 0) {
        const unsigned int *w = reinterpret_cast<const unsigned int *>(p + i0);
        const unsigned int q0 = w[0], q1 = w[1], q2 = w[2], q3 = w[3];
        s[1] = q0 & 0xffffu; s[2] = q0 >> 16; s[3] = q1 & 0xffffu; s[4] = q1 >> 16;
        s[5] = q2 & 0xffffu; s[6] = q2 >> 16; s[7] = q3 & 0xffffu; s[8] = q3 >> 16;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k + 1] = p[i0 + k];
    }
    s[0] = p[i0 - 1];
}

// K1 demodulation pass.  grid (chunks, n_sw), chunks * kStatsChunk >= maxlen.
// codes: [n_sw][code_stride] int16, code_stride a multiple of 8 (rows 16-byte aligned).
__global__ __launch_bounds__(kStatsThreads) void k_fm_demod(const SWDesc *sw, short *codes, long long code_stride,
                                                            StatsPartial *partials, int chunks_per_window)
{
    __shared__ float rcp[128];
    k1_init_rcp(rcp);
    const SWDesc d = sw[blockIdx.y];
    // the pointer comes out of a descriptor in memory: tell the compiler it is global, not flat
    typedef const __attribute__((address_space(1))) uint16_t *global_u16;
    const uint16_t *p = (const uint16_t *)(global_u16)(const uint16_t *)d.base;
    short *out = codes + (size_t)blockIdx.y * code_stride;
    const int len = d.len;
    const int start = blockIdx.x * kStatsChunk;
    long long s1 = 0;
    unsigned long long s2 = 0;
    for (int i0 = start + threadIdx.x * 8; i0 < start + kStatsChunk && i0 < len; i0 += kStatsThreads * 8) {
        int c[8];
        if (i0 >= 1 && i0 + 8 <= len) {
            // interior: no per-sample conditions
            unsigned int s[9];
            k1_load9(p, i0, s);
            float th[9];
#pragma unroll
            for (int k = 0; k < 9; k++) th[k] = k1_theta(s[k], rcp);
            int t1 = 0;
            unsigned long long t2 = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                c[k] = k1_code(k1_wrap_diff(th[k + 1], th[k]));
                t1 += c[k];
                t2 += (unsigned int)(c[k] * c[k]);     // <= 2^30 each
            }
            s1 += t1;
            s2 += t2;
        } else {
            // window head (code_0 := code_1) and tail; samples beyond len carry code 0 in memory
            // and do not enter the sums
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k;
                int v = 0;
                if (i < len && len >= 2) {
                    const int ii = i == 0 ? 1 : i;
                    v = k1_code(k1_wrap_diff(k1_theta(p[ii], rcp), k1_theta(p[ii - 1], rcp)));
                }
                c[k] = v;
                if (i < len) {
                    s1 += v;
                    s2 += (unsigned long long)(unsigned int)(v * v);
                }
            }
        }
        uint4 w;
        w.x = (unsigned int)(c[0] & 0xffff) | ((unsigned int)c[1] << 16);
        w.y = (unsigned int)(c[2] & 0xffff) | ((unsigned int)c[3] << 16);
        w.z = (unsigned int)(c[4] & 0xffff) | ((unsigned int)c[5] << 16);
        w.w = (unsigned int)(c[6] & 0xffff) | ((unsigned int)c[7] << 16);
        *reinterpret_cast<uint4 *>(out + i0) = w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, kWave);
        s2 += __shfl_xor(s2, off, kWave);
    }
    __shared__ StatsPartial red[kStatsThreads / kWave];
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) {
        red[wid].s1 = s1;
        red[wid].s2 = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        StatsPartial t = red[0];
        for (int w = 1; w < kStatsThreads / kWave; w++) {
            t.s1 += red[w].s1;
            t.s2 += red[w].s2;
        }
        partials[(size_t)blockIdx.y * chunks_per_window + blockIdx.x] = t;
    }
}

// one wave per station-window folds the partials (exact integers: order-free) and derives
// mean and scale in f64
__global__ __launch_bounds__(64) void k_fm_stats_final(const SWDesc *sw, const StatsPartial *partials,
                                                       int chunks_per_window, FmStats *stats)
{
#pragma clang fp contract(off)
    const int id = blockIdx.x;
    const int len = sw[id].len;
    long long s1 = 0;
    unsigned long long s2 = 0;
    for (int c = threadIdx.x; c < chunks_per_window; c += 64) {
        const StatsPartial t = partials[(size_t)id * chunks_per_window + c];
        s1 += t.s1;
        s2 += t.s2;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, kWave);
        s2 += __shfl_xor(s2, off, kWave);
    }
    if (threadIdx.x != 0) return;
    FmStats out;
    out.s1 = s1;
    out.s2_lo = s2;
    out.s2_hi = 0;
    if (len == 0) {
        out.mean = 0.0f;
        out.scale = 1.0f;
    } else {
        const double dn = (double)len;
        out.mean = (float)((double)s1 / dn);
        const double m2 = ((double)s1 * (double)s1) / dn;
        const double var = ((double)s2 - m2) / dn;
        out.scale = var > 0 ? (float)(1.0 / sqrt(var)) : 1.0f;
    }
    stats[id] = out;
}

// inspection hook: the normalised discriminator output of window 0
__global__ void k_fm_dump(const SWDesc *sw, const short *codes, const FmStats *stats, float *out)
{
    const FmStats st = stats[0];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < sw[0].len) out[i] = k1_normalise((int)codes[i], st.mean, st.scale);
}

}  // namespace tdoa
