// k1_discriminator.hpp -- K1: u8 IQ -> phase-difference FM discriminator -> 16-bit phase
// code, plus the exact window statistics (mean / unit-power scale) mode B normalises with.
//
// Replaces (reference file:line): the u8 -> complex64 conversion of processor.go:195-201 and,
// for the north-star pipeline, the instantaneous-frequency demodulation that exists only in
// the prebuilt processor binary:
//   a_i    = rint(arg(x_i) * 32768/pi)          angle code of one IQ sample (depends on its 2 bytes)
//   code_i = int16(a_i - a_{i-1})               = arg(x_i * conj(x_{i-1})), wrapped;  code_0 := code_1
// a_i comes from a 65536-entry int16 table (k_k1_build_table, built once per context with the
// explicit f32 arithmetic of k1_theta, which the CPU restatement repeats bit for bit).  One
// streaming pass (k_fm_demod) reads the capture bytes once, looks the angles up in an LDS copy of
// the table, writes 2 bytes of code per sample and accumulates sum(code), sum(code^2) as exact
// integers; the FFT pass then reads the codes.
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

// angle code of one sample: rint(theta * 32768/pi), |code| < 32768
__device__ __forceinline__ int k1_angle_code(unsigned int s, const float *rcp)
{
#pragma clang fp contract(off)
    return __float2int_rn(k1_theta(s, rcp) * 10430.3779296875f);
}

// Table slot of sample s = b_I | b_Q << 8.  LDS banks are picked by bits 1..5 of a 2-byte index;
// captures vary in b_I AND b_Q over a few codes around 127, and b_Q alone would not change the
// bank (stride 512 B), so b_Q's low bits are folded into the bank bits.
__device__ __forceinline__ unsigned int k1_slot(unsigned int s) { return s ^ (((s >> 8) & 31u) << 1); }

// table[k1_slot(s)] = angle code of the IQ sample s;  grid 256 x 256 threads
__global__ __launch_bounds__(256) void k_k1_build_table(short *table)
{
    __shared__ float rcp[128];
    k1_init_rcp(rcp);
    const unsigned int s = blockIdx.x * 256 + threadIdx.x;
    table[k1_slot(s)] = (short)k1_angle_code(s, rcp);
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

constexpr int kDemodThreads = 1024;
constexpr int kDemodPiece = 1024;    // samples per wave step: 64 lanes x 2 x 8
constexpr int kDemodRun = 16;        // pieces per work item (one pair of atomics per 16384 samples)

// eight consecutive IQ samples p[i0 .. i0+7] with whatever alignment the window start has
__device__ __forceinline__ void k1_load8(const uint16_t *p, int i0, unsigned int (&s)[9])
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
}

// K1 demodulation pass: persistent 1024-thread workgroups (one per CU) keep the 128 KB angle
// table in LDS; after loading it the 16 waves of a workgroup run independently, each taking
// (station-window, 1024-sample piece) work items round-robin.  Window sums go straight into
// per-window integer accumulators with atomic adds: exact, hence independent of arrival order.
// codes: [n_sw][code_stride] int16, code_stride a multiple of 8 (rows 16-byte aligned);
// acc: [n_sw] {s1, s2}, zeroed before the launch.
__global__ __launch_bounds__(kDemodThreads) void k_fm_demod(const SWDesc *sw, int n_sw, int pieces_per_window,
                                                            const short *table, short *codes, long long code_stride,
                                                            StatsPartial *acc)
{
    extern __shared__ short lut[];               // 65536 angle codes
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(table);
        uint4 *dst = reinterpret_cast<uint4 *>(lut);
        for (int k = threadIdx.x; k < 8192; k += kDemodThreads) dst[k] = src[k];
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = blockIdx.x * (kDemodThreads / kWave) + threadIdx.x / kWave;
    const int n_waves = gridDim.x * (kDemodThreads / kWave);
    // work item = kDemodRun consecutive pieces of one window; consecutive items belong to DIFFERENT
    // windows, so the waves that run together add into different accumulators
    const int runs_per_window = (pieces_per_window + kDemodRun - 1) / kDemodRun;
    const int n_items = n_sw * runs_per_window;
    for (int item = wave; item < n_items; item += n_waves) {
        const int w = item % n_sw, run = item / n_sw;
        const SWDesc d = sw[w];
        const int len = d.len;
        // the pointer comes out of a descriptor in memory: tell the compiler it is global, not flat
        typedef const __attribute__((address_space(1))) uint16_t *global_u16;
        const uint16_t *p = (const uint16_t *)(global_u16)(const uint16_t *)d.base;
        short *out = codes + (size_t)w * code_stride;
        long long s1 = 0;
        unsigned long long s2 = 0;
        for (int piece = run * kDemodRun; piece < (run + 1) * kDemodRun; piece++) {
            const int start = piece * kDemodPiece;
            if (start >= len) break;
            // a lane owns samples [ia, ia+8) and [ib, ib+8); both loads are issued before either is used
            const int ia = start + lane * 8, ib = ia + 512;
            const bool fa = ia >= 1 && ia + 8 <= len, fb = ib + 8 <= len;     // interior (fast) pieces
            unsigned int sa[9], sb[9];
            if (fa) { k1_load8(p, ia, sa); sa[0] = p[ia - 1]; }
            if (fb) { k1_load8(p, ib, sb); sb[0] = p[ib - 1]; }
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int i0 = half ? ib : ia;
                const bool fast = half ? fb : fa;
                if (i0 >= len) continue;
                int c[8];
                if (fast) {
                    int a[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) a[k] = lut[k1_slot(half ? sb[k] : sa[k])];
                    int t1 = 0;
                    unsigned long long t2 = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        c[k] = (int)(short)(a[k + 1] - a[k]);
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
                            v = (int)(short)(lut[k1_slot(p[ii])] - lut[k1_slot(p[ii - 1])]);
                        }
                        c[k] = v;
                        if (i < len) {
                            s1 += v;
                            s2 += (unsigned long long)(unsigned int)(v * v);
                        }
                    }
                }
                uint4 wv;
                wv.x = (unsigned int)(c[0] & 0xffff) | ((unsigned int)c[1] << 16);
                wv.y = (unsigned int)(c[2] & 0xffff) | ((unsigned int)c[3] << 16);
                wv.z = (unsigned int)(c[4] & 0xffff) | ((unsigned int)c[5] << 16);
                wv.w = (unsigned int)(c[6] & 0xffff) | ((unsigned int)c[7] << 16);
                *reinterpret_cast<uint4 *>(out + i0) = wv;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off, kWave);
            s2 += __shfl_xor(s2, off, kWave);
        }
        if (lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&acc[w].s1), (unsigned long long)s1);   // two's complement
            atomicAdd(&acc[w].s2, s2);
        }
    }
}

// mean and scale of every station-window from its exact sums, in f64
__global__ void k_fm_stats_final(const SWDesc *sw, const StatsPartial *acc, FmStats *stats, int n_sw)
{
#pragma clang fp contract(off)
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_sw) return;
    const int len = sw[id].len;
    const long long s1 = acc[id].s1;
    const unsigned long long s2 = acc[id].s2;
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
