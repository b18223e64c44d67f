// k1_discriminator.hpp -- K1: u8 IQ -> phase-difference FM discriminator, plus the
// exact window statistics (mean / unit-power scale) that mode B normalises with.
//
// Replaces (reference file:line): the u8 -> complex64 conversion of
// processor.go:195-201 and, for the north-star pipeline, the instantaneous-
// frequency demodulation that exists only in the prebuilt processor binary
// (SURVEY.md section 8, row K1): phase_i = arg(x_i * conj(x_{i-1})), evaluated as the
// wrapped difference of the per-sample angles arg(x_i).  The arithmetic is an explicit
// sequence of correctly rounded f32 operations (table reciprocal, FMA Horner, no IEEE
// division in the hot loop) so a CPU restatement can match it bit for bit.
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
// The same sequence of correctly rounded f32 operations as the CPU restatement (bit-exact).
__device__ __forceinline__ float k1_theta(unsigned int s, const float *rcp)
{
#pragma clang fp contract(off)
    // I = 2 b_I - 255 and Q = 2 b_Q - 255 as exact floats (byte -> float conversions)
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
    // Q < 0 (b_Q < 128): negate by flipping the sign bit
    return __uint_as_float(__float_as_uint(a) ^ ((~s & 0x8000u) << 16));
}

// wrapped phase step theta1 - theta0
__device__ __forceinline__ float k1_wrap_diff(float th1, float th0)
{
#pragma clang fp contract(off)
    float d = th1 - th0;
    if (d > 3.1415927410125732f) d = d - 6.2831854820251465f;
    else if (d < -3.1415927410125732f) d = d + 6.2831854820251465f;
    return d;
}

// phase of x_cur * conj(x_prev) for two IQ samples given as uint16 (I | Q << 8)
__device__ __forceinline__ float k1_phase(unsigned int cur, unsigned int prev, const float *rcp)
{
    return k1_wrap_diff(k1_theta(cur, rcp), k1_theta(prev, rcp));
}

// phase of sample i of a window of len samples; sample 0 repeats sample 1
__device__ __forceinline__ float k1_window_phase(const uint16_t *p, int i, int len, const float *rcp)
{
    if (len < 2) return 0.0f;
    int ii = i == 0 ? 1 : i;
    return k1_phase(p[ii], p[ii - 1], rcp);
}

__device__ __forceinline__ float k1_normalise(float phase, float mean, float scale)
{
#pragma clang fp contract(off)
    float d = phase - mean;
    return d * scale;
}

struct StatsPartial {
    long long s1;
    unsigned long long s2_lo, s2_hi;
};

constexpr int kStatsChunk = 16384;   // samples per block
constexpr int kStatsThreads = 256;

// grid: (ceil(maxlen / kStatsChunk), n_station_windows)
__global__ __launch_bounds__(kStatsThreads) void k_fm_stats(const SWDesc *sw, StatsPartial *partials,
                                                            int chunks_per_window)
{
    __shared__ float rcp[128];
    k1_init_rcp(rcp);
    const SWDesc d = sw[blockIdx.y];
    const uint16_t *p = reinterpret_cast<const uint16_t *>(d.base);
    const int len = d.len;
    const int start = blockIdx.x * kStatsChunk;
    long long s1 = 0;
    unsigned long long lo = 0, hi = 0;
    for (int i = start + threadIdx.x; i < start + kStatsChunk && i < len; i += kStatsThreads) {
        float ph = k1_window_phase(p, i, len, rcp);
        long long q = (long long)__float2int_rn(ph * 268435456.0f);   // |q| < 2^30
        s1 += q;
        unsigned long long sq = (unsigned long long)(q * q);
        unsigned long long nlo = lo + sq;
        hi += nlo < lo ? 1ull : 0ull;
        lo = nlo;
    }
    // wave reduction (exact integer arithmetic: any order gives the same result)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        long long o1 = __shfl_xor(s1, off, kWave);
        unsigned long long olo = __shfl_xor(lo, off, kWave);
        unsigned long long ohi = __shfl_xor(hi, off, kWave);
        s1 += o1;
        unsigned long long nlo = lo + olo;
        hi += ohi + (nlo < lo ? 1ull : 0ull);
        lo = nlo;
    }
    __shared__ StatsPartial red[kStatsThreads / kWave];
    int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) {
        red[wid].s1 = s1;
        red[wid].s2_lo = lo;
        red[wid].s2_hi = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        StatsPartial t = red[0];
        for (int w = 1; w < kStatsThreads / kWave; w++) {
            t.s1 += red[w].s1;
            unsigned long long nlo = t.s2_lo + red[w].s2_lo;
            t.s2_hi += red[w].s2_hi + (nlo < t.s2_lo ? 1ull : 0ull);
            t.s2_lo = nlo;
        }
        partials[(size_t)blockIdx.y * chunks_per_window + blockIdx.x] = t;
    }
}

struct FmStats {          // mirrors tdoa_fm_stats
    long long s1;
    unsigned long long s2_lo, s2_hi;
    float mean, scale;
};

// one thread per station-window: fold the partials, derive mean and scale in f64
__global__ void k_fm_stats_final(const SWDesc *sw, const StatsPartial *partials, int chunks_per_window,
                                 FmStats *stats, int n_sw)
{
#pragma clang fp contract(off)
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_sw) return;
    int len = sw[id].len;
    int chunks = (len + kStatsChunk - 1) / kStatsChunk;
    long long s1 = 0;
    unsigned long long lo = 0, hi = 0;
    for (int c = 0; c < chunks; c++) {
        StatsPartial t = partials[(size_t)id * chunks_per_window + c];
        s1 += t.s1;
        unsigned long long nlo = lo + t.s2_lo;
        hi += t.s2_hi + (nlo < lo ? 1ull : 0ull);
        lo = nlo;
    }
    FmStats out;
    out.s1 = s1;
    out.s2_lo = lo;
    out.s2_hi = hi;
    if (len == 0) {
        out.mean = 0.0f;
        out.scale = 1.0f;
    } else {
        double dn = (double)len;
        double mean_q = (double)s1 / dn;
        out.mean = (float)(mean_q / 268435456.0);
        double s2d = (double)hi * 18446744073709551616.0 + (double)lo;
        double m2 = ((double)s1 * (double)s1) / dn;
        double var = ((s2d - m2) / dn) / 72057594037927936.0;
        out.scale = var > 0 ? (float)(1.0 / sqrt(var)) : 1.0f;
    }
    stats[id] = out;
}

// inspection hook: write the normalised discriminator output of one window
__global__ void k_fm_dump(const SWDesc *sw, const FmStats *stats, float *out)
{
    __shared__ float rcp[128];
    k1_init_rcp(rcp);
    const SWDesc d = sw[blockIdx.y];
    const uint16_t *p = reinterpret_cast<const uint16_t *>(d.base);
    const FmStats st = stats[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.len) out[i] = k1_normalise(k1_window_phase(p, i, d.len, rcp), st.mean, st.scale);
}

}  // namespace tdoa
