// k1_discriminator.hpp -- K1: u8 IQ -> phase-difference FM discriminator, plus the
// exact window statistics (mean / unit-power scale) that mode B normalises with.
//
// Replaces (reference file:line): the u8 -> complex64 conversion of
// processor.go:195-201 and, for the north-star pipeline, the instantaneous-
// frequency demodulation that exists only in the prebuilt processor binary
// (SURVEY.md section 8, row K1).  The arithmetic is an explicit sequence of
// correctly rounded f32 operations so a CPU restatement can match it bit for bit.
#pragma once

#include "device_common.hpp"

namespace tdoa {

// atan2 for finite (y, x) not both zero: one or two IEEE divisions and a
// degree-9 odd polynomial after reduction to |t| <= tan(pi/8).  No FMA.
__device__ __forceinline__ float k1_atan2(float y, float x)
{
#pragma clang fp contract(off)
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    float t = mn / mx;
    float base = 0.0f;
    if (t > 0.4142135679721832f) {
        base = 0.7853981852531433f;
        t = (t - 1.0f) / (t + 1.0f);
    }
    float z = t * t;
    float p = 8.05374449538e-2f * z;
    p = p - 1.38776856032e-1f;
    p = p * z;
    p = p + 1.99777106478e-1f;
    p = p * z;
    p = p - 3.33329491539e-1f;
    p = p * z;
    p = p * t;
    p = p + t;
    float r = base + p;
    if (ay > ax) r = 1.5707963705062866f - r;
    if (x < 0.0f) r = 3.1415927410125732f - r;
    if (y < 0.0f) r = -r;
    return r;
}

// cur/prev: one IQ sample as uint16 (I | Q << 8).  Phase of x_cur * conj(x_prev)
// with x = (2b - 255) (exact odd integers, never zero).
__device__ __forceinline__ float k1_phase(unsigned int cur, unsigned int prev)
{
    int I1 = 2 * (int)(cur & 0xffu) - 255, Q1 = 2 * (int)((cur >> 8) & 0xffu) - 255;
    int I0 = 2 * (int)(prev & 0xffu) - 255, Q0 = 2 * (int)((prev >> 8) & 0xffu) - 255;
    int re = I1 * I0 + Q1 * Q0;
    int im = Q1 * I0 - I1 * Q0;
    return k1_atan2((float)im, (float)re);
}

// phase of sample i of a window of len samples; sample 0 repeats sample 1
__device__ __forceinline__ float k1_window_phase(const uint16_t *p, int i, int len)
{
    if (len < 2) return 0.0f;
    int ii = i == 0 ? 1 : i;
    return k1_phase(p[ii], p[ii - 1]);
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
    const SWDesc d = sw[blockIdx.y];
    const uint16_t *p = reinterpret_cast<const uint16_t *>(d.base);
    const int len = d.len;
    const int start = blockIdx.x * kStatsChunk;
    long long s1 = 0;
    unsigned long long lo = 0, hi = 0;
    for (int i = start + threadIdx.x; i < start + kStatsChunk && i < len; i += kStatsThreads) {
        float ph = k1_window_phase(p, i, len);
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
    const SWDesc d = sw[blockIdx.y];
    const uint16_t *p = reinterpret_cast<const uint16_t *>(d.base);
    const FmStats st = stats[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.len) out[i] = k1_normalise(k1_window_phase(p, i, d.len), st.mean, st.scale);
}

}  // namespace tdoa
