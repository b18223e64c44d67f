// k1_discriminator.hpp -- K1: u8 IQ -> phase-difference FM discriminator -> 16-bit phase
// code, plus the exact window statistics (mean / unit-power scale) mode B normalises with.
//
// Replaces (reference file:line): the u8 -> complex64 conversion of processor.go:195-201 and,
// for the north-star pipeline, the instantaneous-frequency demodulation that exists only in
// the prebuilt processor binary: p = x_i conj(x_{i-1}),
// y_i = atan2(Im p, Re p) in (-pi, +pi], y_0 := y_1.  Quantised to units of pi/32768:
//   a_i    = angle code of one IQ sample (depends on its 2 bytes only; 65536-entry int16 table)
//   w      = int16 wrap of (a_i - a_{i-1})        (the two's-complement wrap is the phase circle)
//   code_i = w, except when the two angle codes are exactly opposite (w = -32768):
//            +32768 if Im p >= 0 (an exactly reversed sample is +pi, as atan2(+0, negative) is), else -32767
// so -32767 <= code_i <= 32768 is within one step of y_i as a real number; code_0 := code_1.
// The table is built (k_k1_build_table, once per context) so that collinear samples share a code and
// a(-I, -Q) = a(I, Q) -+ 32768 exactly: gcd-reduced direction, first-octant angle by the explicit f32
// arithmetic of k1_octant_angle (which the CPU restatement repeats bit for bit), octant / quadrant
// placement in integers.  Every exactly reversed pair therefore has opposite codes.
// In memory a code is held NEGATED as an int16 (stored = -code in [-32768, 32767]).
// One streaming pass (k_fm_demod) reads the capture bytes once, looks the angles up in an LDS copy of
// the table, writes 2 bytes of code per sample and accumulates sum(code), sum(code^2) as exact
// integers; the FFT pass then reads the codes.
#pragma once

#include "device_common.hpp"

namespace tdoa {

// Reciprocal table RCP[k] = f32(1 / (2k+1)), k < 128, kept in LDS (random per-lane index).
__device__ __forceinline__ void k1_init_rcp(float *rcp)
{
#pragma clang fp contract(off)
    for (int k = threadIdx.x; k < 128; k += blockDim.x) rcp[k] = 1.0f / (float)(2 * k + 1);
    __syncthreads();
}

// atan(mn / mx) for odd 0 < mn <= mx <= 255: t = mn * RCP[mx], degree-7 Horner in t^2 with fused multiply-adds
__device__ __forceinline__ float k1_octant_angle(int mn, int mx, const float *rcp)
{
#pragma clang fp contract(off)
    const float t = (float)mn * rcp[mx >> 1];
    const float z = t * t;
    float p = -0x1.31f904p-8f;
    p = __builtin_fmaf(p, z, 0x1.8bf058p-6f);
    p = __builtin_fmaf(p, z, -0x1.e655d6p-5f);
    p = __builtin_fmaf(p, z, 0x1.95c0f4p-4f);
    p = __builtin_fmaf(p, z, -0x1.1f0f46p-3f);
    p = __builtin_fmaf(p, z, 0x1.98f84ep-3f);
    p = __builtin_fmaf(p, z, -0x1.5551bcp-2f);
    p = __builtin_fmaf(p, z, 0x1.fffffcp-1f);
    return p * t;
}

// angle code of the IQ sample s = b_I | b_Q << 8 (I = 2 b_I - 255, Q = 2 b_Q - 255), |code| <= 32768 - 41
__device__ __forceinline__ int k1_angle_code(unsigned int s, const float *rcp)
{
#pragma clang fp contract(off)
    const int I = 2 * (int)(s & 0xffu) - 255, Q = 2 * (int)((s >> 8) & 0xffu) - 255;
    int ax = I < 0 ? -I : I, ay = Q < 0 ? -Q : Q;
    int g = ax, h = ay;
    while (h) { const int t = g % h; g = h; h = t; }      // gcd of two odd numbers (odd, >= 1)
    ax /= g;
    ay /= g;
    const int mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    int c = __float2int_rn(k1_octant_angle(mn, mx, rcp) * 10430.3779296875f);   // f32(32768/pi)
    if (ay > ax) c = 16384 - c;
    if (I < 0) c = 32768 - c;
    if (Q < 0) c = -c;
    return c;
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

// stored (negated) phase code of sample `cur` after sample `prev` from their angle codes; the rare
// exactly-opposite case reads the sign of Im p = Q I' - I Q' off the bytes
__device__ __forceinline__ int k1_stored_code(int a_cur, int a_prev, unsigned int cur, unsigned int prev)
{
    int st = (int)(short)(a_prev - a_cur);
    if (st == -32768) {      // code +32768 unless Im p < 0
        const int I = 2 * (int)(cur & 0xffu) - 255, Q = 2 * (int)((cur >> 8) & 0xffu) - 255;
        const int Ip = 2 * (int)(prev & 0xffu) - 255, Qp = 2 * (int)((prev >> 8) & 0xffu) - 255;
        if (Q * Ip - I * Qp < 0) st = 32767;
    }
    return st;
}

// normalised discriminator sample from a STORED code: (float(code) - mean) * scale with code = -stored
__device__ __forceinline__ float k1_normalise(int stored, float mean, float scale)
{
#pragma clang fp contract(off)
    float d = -(float)stored - mean;
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
constexpr int kDemodChunks = 4;      // 8-sample chunks a lane owns per wave step: 4 x 16-byte loads in flight per lane
constexpr int kDemodPiece = 512 * kDemodChunks;   // samples per wave step
constexpr int kDemodItem = 32;       // pieces per workgroup item: 2 steps of 16 waves (one pair of atomics per 65536 samples)

// Capture bytes are read through pointers in the GLOBAL address space: a pointer that comes out of a descriptor in
// memory is generic to the compiler, and generic (flat) loads are ordered against LDS traffic -- every one of them was
// followed by s_waitcnt vmcnt(0) lgkmcnt(0), i.e. one load in flight per wave.
typedef const __attribute__((address_space(1))) uint16_t *gptr16;

__device__ __forceinline__ gptr16 k1_global(const uint8_t *base)
{
    return (gptr16)(const uint16_t *)base;
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
// Envelope code = round-half-up(90 sqrt(m)) = (isqrt(32400 m) + 1) >> 1, m <= 130050 (32400 m < 2^32, code <= 32456):
// an int16 like the phase codes, held negated like them, same statistics and normalisation downstream.
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
    const unsigned int x = 32400u * k1_sample_power(s);
    unsigned int r = (unsigned int)__builtin_sqrtf((float)x);                 // within a few units of isqrt(x)
#pragma unroll
    for (int k = 0; k < 3; k++)
        if ((unsigned long long)r * r > x) r--;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if ((unsigned long long)(r + 1) * (r + 1) <= x) r++;
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

// envelope codes + window sums of the windows in the envelope class (k_fm_demod skips those).  Same grid.
__global__ __launch_bounds__(256) void k_k1_envelope(const SWDesc *sw, const unsigned long long *power, short *codes,
                                                     long long code_stride, StatsPartial *acc)
{
    __shared__ long long red1[4];
    __shared__ unsigned long long red2[4];
    const SWDesc d = sw[blockIdx.y];
    if (!k1_envelope_class(power[blockIdx.y], d.len)) return;
    const gptr16 p = k1_global(d.base);
    short *dst = codes + (size_t)blockIdx.y * code_stride;
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
        uint4 wv;
        wv.x = (unsigned int)(c[0] & 0xffff) | ((unsigned int)c[1] << 16);
        wv.y = (unsigned int)(c[2] & 0xffff) | ((unsigned int)c[3] << 16);
        wv.z = (unsigned int)(c[4] & 0xffff) | ((unsigned int)c[5] << 16);
        wv.w = (unsigned int)(c[6] & 0xffff) | ((unsigned int)c[7] << 16);
        *reinterpret_cast<uint4 *>(dst + i0) = wv;
    }
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
    if (threadIdx.x == 0) {
        const long long t1 = red1[0] + red1[1] + red1[2] + red1[3];
        const unsigned long long t2 = red2[0] + red2[1] + red2[2] + red2[3];
        if (t1 | (long long)t2) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&acc[blockIdx.y].s1), (unsigned long long)t1);
            atomicAdd(&acc[blockIdx.y].s2, t2);
        }
    }
}

// K1 demodulation pass: persistent 1024-thread workgroups (one per CU) keep the 128 KB angle
// table in LDS; after loading it the 16 waves of a workgroup run independently, each taking
// (station-window, 1024-sample piece) work items round-robin.  Window sums go straight into
// per-window integer accumulators with atomic adds: exact, hence independent of arrival order.
// codes: [n_sw][code_stride] int16, code_stride a multiple of 8 (rows 16-byte aligned);
// acc: [n_sw] {s1, s2}, zeroed before the launch.  power: nullptr, or the windows' power sums (optional gate: windows
// in the envelope class are skipped here).
__global__ __launch_bounds__(kDemodThreads) void k_fm_demod(const SWDesc *sw, int n_sw, int pieces_per_window,
                                                            const short *table, short *codes, long long code_stride,
                                                            StatsPartial *acc, const unsigned long long *power)
{
    extern __shared__ short lut[];               // 65536 angle codes
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(table);
        uint4 *dst = reinterpret_cast<uint4 *>(lut);
        for (int k = threadIdx.x; k < 8192; k += kDemodThreads) dst[k] = src[k];
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    // work item of a WORKGROUP = kDemodItem consecutive pieces of one window; at every step its 16 waves take 16
    // adjacent pieces, so the workgroup streams 32 KB of contiguous capture bytes per step (one DRAM-friendly run,
    // like a row of the FFT passes) instead of 16 unrelated 2 KB reads.  Consecutive items belong to DIFFERENT
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
        short *out = codes + (size_t)w * code_stride;
        long long s1 = 0;
        unsigned long long s2 = 0;
        for (int piece = run * kDemodItem + wv; piece < (run + 1) * kDemodItem; piece += kDemodThreads / kWave) {
            const int start = piece * kDemodPiece;
            if (start >= len) break;
            // a lane owns samples [i0, i0+8) of each 512-sample chunk; all loads are issued before any is used
            uint4 qs[kDemodChunks];
            unsigned int prev[kDemodChunks];
            bool fastc[kDemodChunks];
            bool interior = true;
#pragma unroll
            for (int h = 0; h < kDemodChunks; h++) {
                const int i0 = start + h * 512 + lane * 8;
                fastc[h] = i0 >= 1 && i0 + 8 <= len;                              // interior (fast) chunks
                interior = interior && fastc[h];
                qs[h] = make_uint4(0, 0, 0, 0);
                prev[h] = 0;
            }
            // Exactly opposite angle codes mean +pi unless Im p < 0, and Im p < 0 needs two NON-collinear samples less
            // than one code step away from a reversal: |x_i||x_{i-1}| > 2 * 32768/pi = 20861 in units of (2b - 255).
            // A small-amplitude capture (simulator.go: +-3 LSB) is full of exact reversals but can never get there: if
            // every byte of the piece is within [96, 159], |x|^2 <= 2 * 63^2 = 7938.  One wave-uniform test per piece
            // (2048 samples) then switches the per-sample sign check off altogether.
            bool check_sign = true;
            if (__all(interior)) {
                // whole piece inside the window: the sample before a lane's chunk is the last sample of the lane to
                // its left (lane 0: lane 63 of the previous chunk; chunk 0: one broadcast load) -- no 2-byte gathers
                const unsigned int before = p[start - 1];
#pragma unroll
                for (int h = 0; h < kDemodChunks; h++) qs[h] = k1_fetch8(p, start + h * 512 + lane * 8);
                unsigned int far = 0;            // a byte b is in [96, 159] iff the top three bits of b ^ 0x80 are equal
#pragma unroll
                for (int h = 0; h < kDemodChunks; h++) {
                    const unsigned int left = __shfl_up(qs[h].w >> 16, 1, kWave);
                    const unsigned int wrap = h ? __shfl(qs[h ? h - 1 : 0].w >> 16, kWave - 1, kWave) : before;
                    prev[h] = lane ? left : wrap;
                    const unsigned int y0 = qs[h].x ^ 0x80808080u, y1 = qs[h].y ^ 0x80808080u, y2 = qs[h].z ^ 0x80808080u,
                                       y3 = qs[h].w ^ 0x80808080u;
                    far |= (y0 ^ (y0 << 1)) | (y1 ^ (y1 << 1)) | (y2 ^ (y2 << 1)) | (y3 ^ (y3 << 1));
                }
                {
                    const unsigned int yb = (before | (before << 16)) ^ 0x80808080u;
                    far |= yb ^ (yb << 1);
                }
                check_sign = __any((far & 0xC0C0C0C0u) != 0);
            } else {
#pragma unroll
                for (int h = 0; h < kDemodChunks; h++) {
                    const int i0 = start + h * 512 + lane * 8;
                    if (fastc[h]) { qs[h] = k1_fetch8(p, i0); prev[h] = p[i0 - 1]; }
                }
            }
#pragma unroll
            for (int half = 0; half < kDemodChunks; half++) {
                const int i0 = start + half * 512 + lane * 8;
                const bool fast = fastc[half];
                if (i0 >= len) continue;
                int c[8];
                if (fast) {
                    unsigned int sm[9];
                    k1_unpack8(qs[half], sm);
                    sm[0] = prev[half];
                    int a[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) a[k] = lut[k1_slot(sm[k])];
                    int t1 = 0;
                    unsigned long long t2 = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) c[k] = (int)(short)(a[k] - a[k + 1]);      // stored = -code
                    if (check_sign) {            // wave-uniform: some byte of the piece is far from the centre
                        bool opposite = false;
#pragma unroll
                        for (int k = 0; k < 8; k++) opposite = opposite || c[k] == -32768;
                        if (__any(opposite)) {
#pragma unroll
                            for (int k = 0; k < 8; k++) c[k] = k1_stored_code(a[k + 1], a[k], sm[k + 1], sm[k]);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        t1 -= c[k];
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
                            const unsigned int sc = p[ii], sp = p[ii - 1];
                            v = k1_stored_code(lut[k1_slot(sc)], lut[k1_slot(sp)], sc, sp);
                        }
                        c[k] = v;
                        if (i < len) {
                            s1 -= v;
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
            red1[wv] = s1;
            red2[wv] = s2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long t1 = 0;
            unsigned long long t2 = 0;
            for (int k = 0; k < kDemodThreads / kWave; k++) { t1 += red1[k]; t2 += red2[k]; }
            atomicAdd(reinterpret_cast<unsigned long long *>(&acc[w].s1), (unsigned long long)t1);   // two's complement
            atomicAdd(&acc[w].s2, t2);
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
__global__ __launch_bounds__(256) void k_k1_smooth(const SWDesc *sw, const short *in, short *out, long long code_stride, int h0,
                                                   StatsPartial *acc, const unsigned long long *power)
{
    __shared__ long long red1[4];
    __shared__ unsigned long long red2[4];
    const int len = sw[blockIdx.y].len;
    // power gate on: the binary smooths its discriminator output only; an envelope window is copied (h = 0)
    const int h = power && k1_envelope_class(power[blockIdx.y], len) ? 0 : h0;
    const short *src = in + (size_t)blockIdx.y * code_stride;
    short *dst = out + (size_t)blockIdx.y * code_stride;
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
                int sum = 0;
                for (int j = lo; j <= hi; j++) sum -= (int)src[j];            // stored = -code
                const int cnt = hi - lo + 1;
                const int num = 2 * sum + cnt, den = 2 * cnt;                 // floor division, den > 0
                const int lp = num >= 0 ? num / den : -((-num + den - 1) / den);
                v = -lp;
                s1 += lp;
                s2 += (unsigned long long)((long long)lp * lp);
            }
            c[k] = v;
        }
        uint4 wv;
        wv.x = (unsigned int)(c[0] & 0xffff) | ((unsigned int)c[1] << 16);
        wv.y = (unsigned int)(c[2] & 0xffff) | ((unsigned int)c[3] << 16);
        wv.z = (unsigned int)(c[4] & 0xffff) | ((unsigned int)c[5] << 16);
        wv.w = (unsigned int)(c[6] & 0xffff) | ((unsigned int)c[7] << 16);
        *reinterpret_cast<uint4 *>(dst + i0) = wv;        // rows are 16-byte aligned and padded to a multiple of 8
    }
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
    if (threadIdx.x == 0) {
        const long long t1 = red1[0] + red1[1] + red1[2] + red1[3];
        const unsigned long long t2 = red2[0] + red2[1] + red2[2] + red2[3];
        if (t1 | (long long)t2) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&acc[blockIdx.y].s1), (unsigned long long)t1);
            atomicAdd(&acc[blockIdx.y].s2, t2);
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
