// fft_radix16.hpp -- the hot-size kernels: register-resident radix-16 Stockham stages with
// one LDS exchange between stages, for rows of 4096 = 16^3 points and columns of 256 = 16^2.
// Same math and layouts as fft_stockham.hpp (which stays as the any-size fallback); these
// replace it when the plan is N1 = 4096 (and N2 = 256 for the forward column pass).
//
// Per thread: 16 complex values in VGPRs, a 16-point DFT as 4x4 radix-4 butterflies, twiddles
// w^r built from one sincospi per stage by a depth-4 product tree.  LDS images are padded by one
// element per 16 so that the stride-16 writes of the first stage do not pile onto one bank.
#pragma once

#include "device_common.hpp"
#include "fft_stockham.hpp"
#include "k1_discriminator.hpp"

namespace tdoa {

// 16-point DFT in registers.  Input natural order; output X[k] is left in v[oreg(k)].
__device__ __forceinline__ constexpr int oreg(int k) { return (k >> 2) | ((k & 3) << 2); }

template <bool INV>
__device__ __forceinline__ void fft16(float2 (&v)[16])
{
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) bfly4<INV>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
    // v[n1 + 4*k2] *= W16^(n1*k2)   (forward: e^{-2 pi i m/16}; inverse: conjugate)
    auto tw = [](float2 a, float c, float s) {
        return INV ? make_float2(a.x * c - a.y * s, a.x * s + a.y * c) : make_float2(a.x * c + a.y * s, a.y * c - a.x * s);
    };
    v[1 + 4] = tw(v[1 + 4], c1, s1);      // m = 1
    v[1 + 8] = tw(v[1 + 8], h, h);        // m = 2
    v[1 + 12] = tw(v[1 + 12], s1, c1);    // m = 3
    v[2 + 4] = tw(v[2 + 4], h, h);        // m = 2
    v[2 + 8] = tw(v[2 + 8], 0.0f, 1.0f);  // m = 4
    v[2 + 12] = tw(v[2 + 12], -h, h);     // m = 6
    v[3 + 4] = tw(v[3 + 4], s1, c1);      // m = 3
    v[3 + 8] = tw(v[3 + 8], -h, h);       // m = 6
    v[3 + 12] = tw(v[3 + 12], -c1, -s1);  // m = 9
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) bfly4<INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
}

// v[r] *= w^r, r = 1..15, powers by a depth-4 product tree
__device__ __forceinline__ void mul_powers16(float2 (&v)[16], float2 w)
{
    float2 w2 = cmul(w, w), w3 = cmul(w2, w), w4 = cmul(w2, w2);
    float2 w5 = cmul(w4, w), w6 = cmul(w4, w2), w7 = cmul(w4, w3), w8 = cmul(w4, w4);
    v[1] = cmul(v[1], w);
    v[2] = cmul(v[2], w2);
    v[3] = cmul(v[3], w3);
    v[4] = cmul(v[4], w4);
    v[5] = cmul(v[5], w5);
    v[6] = cmul(v[6], w6);
    v[7] = cmul(v[7], w7);
    v[8] = cmul(v[8], w8);
    v[9] = cmul(v[9], cmul(w8, w));
    v[10] = cmul(v[10], cmul(w8, w2));
    v[11] = cmul(v[11], cmul(w8, w3));
    v[12] = cmul(v[12], cmul(w8, w4));
    v[13] = cmul(v[13], cmul(w8, w5));
    v[14] = cmul(v[14], cmul(w8, w6));
    v[15] = cmul(v[15], cmul(w8, w7));
}

// same, but applied to the outputs X[k] (which live in v[oreg(k)]) with a separate base factor:
// X[k] *= base * step^k
__device__ __forceinline__ void mul_base_step16(float2 (&v)[16], float2 base, float2 step)
{
    float2 s2 = cmul(step, step), s3 = cmul(s2, step), s4 = cmul(s2, s2);
    float2 s5 = cmul(s4, step), s6 = cmul(s4, s2), s7 = cmul(s4, s3), s8 = cmul(s4, s4);
    float2 p[16];
    p[0] = base;
    p[1] = cmul(base, step);
    p[2] = cmul(base, s2);
    p[3] = cmul(base, s3);
    p[4] = cmul(base, s4);
    p[5] = cmul(base, s5);
    p[6] = cmul(base, s6);
    p[7] = cmul(base, s7);
    float2 b8 = cmul(base, s8);
    p[8] = b8;
    p[9] = cmul(b8, step);
    p[10] = cmul(b8, s2);
    p[11] = cmul(b8, s3);
    p[12] = cmul(b8, s4);
    p[13] = cmul(b8, s5);
    p[14] = cmul(b8, s6);
    p[15] = cmul(b8, s7);
#pragma unroll
    for (int k = 0; k < 16; k++) v[oreg(k)] = cmul(v[oreg(k)], p[k]);
}

__device__ __forceinline__ int pad16(int i) { return i + (i >> 4); }
constexpr int kRowLds = 4096 + 256;   // padded float2 per 4096-point row

// stages 2 and 3 of a 4096-point row FFT whose stage-1 outputs X[k] (thread j, in v[oreg(k)])
// are still in registers.  On return thread j holds the final outputs Y[j + 256 k] in v[oreg(k)].
template <bool INV>
__device__ __forceinline__ void row4096_finish(float2 (&v)[16], float2 *lds, int j)
{
#pragma unroll
    for (int k = 0; k < 16; k++) lds[pad16(16 * j + k)] = v[oreg(k)];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[pad16(j + 256 * r)];
    mul_powers16(v, unit_root((float)(j & 15), 2.0f / 256.0f, INV));
    fft16<INV>(v);
    __syncthreads();
    {
        const int d = ((j >> 4) << 8) + (j & 15);
#pragma unroll
        for (int k = 0; k < 16; k++) lds[pad16(d + 16 * k)] = v[oreg(k)];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[pad16(j + 256 * r)];
    mul_powers16(v, unit_root((float)j, 2.0f / 4096.0f, INV));
    fft16<INV>(v);
}

// ---------------------------------------------------------------------------
// forward row pass, N1 = 4096, in place.  grid (N2, n_sw), 256 threads, static LDS 34 KB
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fwd_row4096(float2 *TZ, FftPlan pl)
{
    __shared__ float2 lds[kRowLds];
    float2 *row = TZ + (size_t)blockIdx.y * pl.Nc + (size_t)blockIdx.x * 4096;
    const int j = threadIdx.x;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = row[j + 256 * r];
    fft16<false>(v);
    row4096_finish<false>(v, lds, j);
#pragma unroll
    for (int k = 0; k < 16; k++) row[j + 256 * k] = v[oreg(k)];
}

// ---------------------------------------------------------------------------
// forward column pass fused with K1, N2 = 256, 32 columns per workgroup; windows 4-byte aligned, even length.
// grid (N1/32, n_sw), 512 threads (c = t & 31 column, j = t >> 5 item), LDS 64 KB
// ---------------------------------------------------------------------------
// Branch-free element fetch for windows of EVEN length >= 2 whose start is 4-byte aligned:
// the two loads are issued unconditionally (addresses clamped into the window) so that all 32
// loads of a thread are in flight together; out-of-window elements are zeroed afterwards.
struct K1Raw {
    unsigned int cur, prev;
    bool valid, first;
};

__device__ __forceinline__ K1Raw k1_fetch(const uint8_t *base, long long m, int len)
{
    K1Raw r;
    r.valid = 2 * m < len;
    const long long mm = r.valid ? m : 0;
    r.first = mm == 0;
    r.cur = *reinterpret_cast<const unsigned int *>(base + 4 * mm);            // samples 2m, 2m+1
    r.prev = *reinterpret_cast<const uint16_t *>(base + 4 * mm - (r.first ? 0 : 2));   // sample 2m-1
    return r;
}

__device__ __forceinline__ float2 k1_finish(const K1Raw r, float mean, float scale, const float *rcp)
{
    const float th0 = k1_theta(r.cur & 0xffffu, rcp), th1 = k1_theta(r.cur >> 16, rcp);
    const float thp = k1_theta(r.prev, rcp);
    const float v1 = k1_normalise(k1_wrap_diff(th1, th0), mean, scale);
    float v0 = k1_normalise(k1_wrap_diff(th0, thp), mean, scale);
    v0 = r.first ? v1 : v0;                                  // phase_0 := phase_1
    return r.valid ? make_float2(v0, v1) : make_float2(0.0f, 0.0f);
}

__global__ __launch_bounds__(512) void k_fwd_col256_u8(const SWDesc *sw, const FmStats *stats, float2 *T, FftPlan pl)
{
    extern __shared__ float2 lds[];   // [256][32]
    __shared__ float rcp[128];
    k1_init_rcp(rcp);
    const SWDesc d = sw[blockIdx.y];
    const int len = d.len;
    const float mean = stats[blockIdx.y].mean, scale = stats[blockIdx.y].scale;
    const int c = threadIdx.x & 31, j = threadIdx.x >> 5;     // column, item (0..15)
    const int n1 = (blockIdx.x << 5) + c;
    const int N1 = pl.N1;
    float2 v[16];
    {
        K1Raw raw[16];
#pragma unroll
        for (int r = 0; r < 16; r++) raw[r] = k1_fetch(d.base, (long long)(j + 16 * r) * N1 + n1, len);
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = k1_finish(raw[r], mean, scale, rcp);
    }
    fft16<false>(v);
#pragma unroll
    for (int k = 0; k < 16; k++) lds[((16 * j + k) << 5) + c] = v[oreg(k)];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[((j + 16 * r) << 5) + c];
    mul_powers16(v, unit_root((float)j, 2.0f / 256.0f, false));
    fft16<false>(v);
    // Y[k2 = j + 16k] *= W_Nc^(n1*k2) = W^(n1*j) * (W^(16*n1))^k
    float2 *out = T + (size_t)blockIdx.y * pl.Nc;
    const float inv2 = 2.0f / (float)pl.Nc;
    const long long e0 = ((long long)n1 * j) & (pl.Nc - 1);
    const long long e1 = ((long long)n1 * 16) & (pl.Nc - 1);
    mul_base_step16(v, unit_root((float)e0, inv2, false), unit_root((float)e1, inv2, false));
#pragma unroll
    for (int k = 0; k < 16; k++) out[(size_t)(j + 16 * k) * N1 + n1] = v[oreg(k)];
}

// ---------------------------------------------------------------------------
// inverse row pass with K3 fused, N1 = 4096: one workgroup owns rows a and N2 - a (a >= 1).
// Thread t builds Q[a][t + 256 r] and its mirror Q[N2-a][4095 - t - 256 r] from the same four
// spectrum values, so stage 1 of row a (item t) and of row N2-a (item 255 - t) need no exchange.
// grid (N2/2 - 1, n_pw), 256 threads, dynamic LDS 2*kRowLds*8 = 68 KB.  Row pair a = 0 is left to k_inv_row_pair.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_inv_row_pair4096(const PWDesc *pw, const float2 *Z, float2 *V, FftPlan pl)
{
    extern __shared__ float2 lds[];   // 2 * kRowLds
    const PWDesc d = pw[blockIdx.y];
    const int N2 = pl.N2;
    const int a = blockIdx.x + 1, b = N2 - a;
    const float2 *ZaA = Z + (size_t)d.sw_a * pl.Nc + (size_t)a * 4096;
    const float2 *ZaB = Z + (size_t)d.sw_a * pl.Nc + (size_t)b * 4096;
    const float2 *ZbA = Z + (size_t)d.sw_b * pl.Nc + (size_t)a * 4096;
    const float2 *ZbB = Z + (size_t)d.sw_b * pl.Nc + (size_t)b * 4096;
    const int t = threadIdx.x;
    float2 va[16], vb[16];
    {
        // w(k) = W_N^k, k = (t + 256 r) N2 + a  =>  w = w0 * W_32^r
        const long long k0 = (long long)t * N2 + a;
        const float2 w0 = unit_root((float)k0, 1.0f / (float)pl.Nc, false);
        const float2 st = make_float2(0.98078528040323043f, -0.19509032201612825f);   // e^{-2 pi i/32}
        float2 w = w0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int k1 = t + 256 * r;
            float2 q, qm;
            pair_q(ZaA[k1], ZaB[4095 - k1], ZbA[k1], ZbB[4095 - k1], w, q, qm);
            va[r] = q;
            vb[15 - r] = qm;
            if ((r & 3) == 3) {
                // re-anchor every 4 steps to keep the running product short
                const float x = (float)(k0 + (long long)(r + 1) * 256 * N2);
                w = unit_root(x, 1.0f / (float)pl.Nc, false);
            } else {
                w = cmul(w, st);
            }
        }
    }
    fft16<true>(va);
    fft16<true>(vb);
    // row a: item t; row b: item 255 - t
    float2 *la = lds, *lb = lds + kRowLds;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        la[pad16(16 * t + k)] = va[oreg(k)];
        lb[pad16(16 * (255 - t) + k)] = vb[oreg(k)];
    }
    __syncthreads();
    const int j = t;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        va[r] = la[pad16(j + 256 * r)];
        vb[r] = lb[pad16(j + 256 * r)];
    }
    {
        const float2 w = unit_root((float)(j & 15), 2.0f / 256.0f, true);
        mul_powers16(va, w);
        mul_powers16(vb, w);
    }
    fft16<true>(va);
    fft16<true>(vb);
    __syncthreads();
    {
        const int dd = ((j >> 4) << 8) + (j & 15);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            la[pad16(dd + 16 * k)] = va[oreg(k)];
            lb[pad16(dd + 16 * k)] = vb[oreg(k)];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) {
        va[r] = la[pad16(j + 256 * r)];
        vb[r] = lb[pad16(j + 256 * r)];
    }
    {
        const float2 w = unit_root((float)j, 2.0f / 4096.0f, true);
        mul_powers16(va, w);
        mul_powers16(vb, w);
    }
    fft16<true>(va);
    fft16<true>(vb);
    // V[k2][n1] = y[n1] * W_Nc^(-n1 k2), n1 = j + 256 k
    float2 *out = V + (size_t)blockIdx.y * pl.Nc;
    const float inv2 = 2.0f / (float)pl.Nc;
    {
        const long long e0 = ((long long)j * a) & (pl.Nc - 1), e1 = ((long long)256 * a) & (pl.Nc - 1);
        mul_base_step16(va, unit_root((float)e0, inv2, true), unit_root((float)e1, inv2, true));
        const long long f0 = ((long long)j * b) & (pl.Nc - 1), f1 = ((long long)256 * b) & (pl.Nc - 1);
        mul_base_step16(vb, unit_root((float)f0, inv2, true), unit_root((float)f1, inv2, true));
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        out[(size_t)a * 4096 + j + 256 * k] = va[oreg(k)];
        out[(size_t)b * 4096 + j + 256 * k] = vb[oreg(k)];
    }
}

// ---------------------------------------------------------------------------
// pruned inverse column pass + K5.  Only the outputs n2 that can hold a searched lag are
// evaluated, as direct DFT sums over k2 (lags 2m, 2m+1 with m = n2*N1 + n1; |lag| <= max):
//   n2 in [0, NP)  (non-negative lags)  and  n2 in [N2 - NN, N2)  (negative lags), NP + NN <= 8.
// grid (N1/32, n_pw), 256 threads: c = t & 31 (column), g = t >> 5 (row group, rows g, g+8, ...)
// ---------------------------------------------------------------------------
constexpr int kPruneMax = 8;

__global__ __launch_bounds__(256) void k_inv_col_pruned(const float2 *V, unsigned long long *keys, const PWDesc *pw,
                                                       FftPlan pl, int lag_lo, int lag_hi, int np, int nn,
                                                       float *lag_dump, float dump_scale)
{
    __shared__ float2 wtab[512];                   // e^{+2 pi i k/N2}, N2 <= 512
    __shared__ float2 part[8][kPruneMax][32];
    __shared__ unsigned long long red[4];
    const int N2 = pl.N2, N1 = pl.N1;
    for (int k = threadIdx.x; k < N2; k += 256) wtab[k] = unit_root((float)k, 2.0f / (float)N2, true);
    __syncthreads();
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int n1 = (blockIdx.x << 5) + c;
    const float2 *in = V + (size_t)blockIdx.y * pl.Nc + n1;
    float2 acc[kPruneMax];
#pragma unroll
    for (int o = 0; o < kPruneMax; o++) acc[o] = make_float2(0.0f, 0.0f);
    const int nout = np + nn;
    for (int kb = g; kb < N2; kb += 64) {          // 8 rows per trip, all loads issued first
        float2 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k2 = kb + 8 * u;
            x[u] = k2 < N2 ? in[(size_t)k2 * N1] : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k2 = kb + 8 * u;
#pragma unroll
            for (int o = 0; o < kPruneMax; o++) {
                if (o < nout) {
                    // output n2 = o (o < np) or N2 - nn + (o - np); factor e^{+2 pi i n2 k2 / N2}
                    const int n2 = o < np ? o : N2 - nn + (o - np);
                    const float2 w = wtab[(n2 * k2) & (N2 - 1)];
                    acc[o].x += x[u].x * w.x - x[u].y * w.y;
                    acc[o].y += x[u].x * w.y + x[u].y * w.x;
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < kPruneMax; o++) part[g][o][c] = acc[o];
    __syncthreads();
    unsigned long long best = 0;
    if (threadIdx.x < 32 * nout) {
        const int o = threadIdx.x >> 5;             // wave-uniform for 32-lane halves
        float2 s = part[0][o][c];
#pragma unroll
        for (int gg = 1; gg < 8; gg++) {
            s.x += part[gg][o][c].x;
            s.y += part[gg][o][c].y;
        }
        const int n2 = o < np ? o : N2 - nn + (o - np);
        const long long m = (long long)n2 * N1 + n1;
        long long d0 = 2 * m;
        if (d0 >= pl.Nc) d0 -= 2 * pl.Nc;
        const long long d1 = d0 + 1;
        if (d0 >= lag_lo && d0 <= lag_hi && s.x == s.x) {
            const unsigned long long k = peak_key(s.x, (int)d0);
            best = k > best ? k : best;
            if (lag_dump) lag_dump[d0 - lag_lo] = s.x * dump_scale;
        }
        if (d1 >= lag_lo && d1 <= lag_hi && s.y == s.y) {
            const unsigned long long k = peak_key(s.y, (int)d1);
            best = k > best ? k : best;
            if (lag_dump) lag_dump[d1 - lag_lo] = s.y * dump_scale;
        }
    }
    best = wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long bb = red[0];
        for (int w = 1; w < 4; w++) bb = red[w] > bb ? red[w] : bb;
        if (bb) atomicMax(&keys[pw[blockIdx.y].out_index], bb);
    }
}

// ---------------------------------------------------------------------------
// K1 statistics, vectorised: each thread takes 8 consecutive samples from one aligned 16-byte
// load.  The thread grid is shifted back by a = (base mod 16)/2 samples so that every load is
// 16-byte aligned whatever the window start; partial sums are exact integers, so it does not
// matter which block accounts for a sample.  grid (chunks, n_sw) with chunks*kStatsChunk >= len + 7.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kStatsThreads) void k_fm_stats_vec(const SWDesc *sw, StatsPartial *partials,
                                                                int chunks_per_window)
{
    __shared__ float rcp[128];
    k1_init_rcp(rcp);
    const SWDesc d = sw[blockIdx.y];
    const uint16_t *p = reinterpret_cast<const uint16_t *>(d.base);
    const int len = d.len;
    const int shift = (int)(((uintptr_t)d.base & 15u) >> 1);
    const int start = blockIdx.x * kStatsChunk - shift;
    long long s1 = 0;
    unsigned long long lo = 0, hi = 0;
    for (int i0 = start + threadIdx.x * 8; i0 < start + kStatsChunk && i0 < len; i0 += kStatsThreads * 8) {
        long long ls1 = 0;
        unsigned long long ls2 = 0;                 // 8 squares < 2^60 each: no overflow
        if (i0 >= 1 && i0 + 8 <= len) {
            // interior: one aligned 16-byte load + the previous sample, no per-sample conditions
            const uint4 q = *reinterpret_cast<const uint4 *>(p + i0);
            const unsigned int prev = p[i0 - 1];
            float th[9];
            th[0] = k1_theta(prev, rcp);
            th[1] = k1_theta(q.x & 0xffffu, rcp);
            th[2] = k1_theta(q.x >> 16, rcp);
            th[3] = k1_theta(q.y & 0xffffu, rcp);
            th[4] = k1_theta(q.y >> 16, rcp);
            th[5] = k1_theta(q.z & 0xffffu, rcp);
            th[6] = k1_theta(q.z >> 16, rcp);
            th[7] = k1_theta(q.w & 0xffffu, rcp);
            th[8] = k1_theta(q.w >> 16, rcp);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int qi = __float2int_rn(k1_wrap_diff(th[k + 1], th[k]) * 268435456.0f);
                ls1 += qi;
                const unsigned int aq = (unsigned int)(qi < 0 ? -qi : qi);
                ls2 += (unsigned long long)aq * aq;
            }
        } else {
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k;
                if (i >= 0 && i < len) {
                    const int qi = __float2int_rn(k1_window_phase(p, i, len, rcp) * 268435456.0f);
                    ls1 += qi;
                    const unsigned int aq = (unsigned int)(qi < 0 ? -qi : qi);
                    ls2 += (unsigned long long)aq * aq;
                }
            }
        }
        s1 += ls1;
        const unsigned long long nlo = lo + ls2;
        hi += nlo < lo ? 1ull : 0ull;
        lo = nlo;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long o1 = __shfl_xor(s1, off, kWave);
        const unsigned long long olo = __shfl_xor(lo, off, kWave);
        const unsigned long long ohi = __shfl_xor(hi, off, kWave);
        s1 += o1;
        const unsigned long long nlo = lo + olo;
        hi += ohi + (nlo < lo ? 1ull : 0ull);
        lo = nlo;
    }
    __shared__ StatsPartial red[kStatsThreads / kWave];
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
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
            const unsigned long long nlo = t.s2_lo + red[w].s2_lo;
            t.s2_hi += red[w].s2_hi + (nlo < t.s2_lo ? 1ull : 0ull);
            t.s2_lo = nlo;
        }
        partials[(size_t)blockIdx.y * chunks_per_window + blockIdx.x] = t;
    }
}

// fold the partials with one wave per station-window (exact integers: order-free)
__global__ __launch_bounds__(64) void k_fm_stats_final_wave(const SWDesc *sw, const StatsPartial *partials,
                                                            int chunks_per_window, FmStats *stats)
{
#pragma clang fp contract(off)
    const int id = blockIdx.x;
    const int len = sw[id].len;
    const int chunks = chunks_per_window;          // every block wrote its (possibly empty) partial
    long long s1 = 0;
    unsigned long long lo = 0, hi = 0;
    for (int c = threadIdx.x; c < chunks; c += 64) {
        const StatsPartial t = partials[(size_t)id * chunks_per_window + c];
        s1 += t.s1;
        const unsigned long long nlo = lo + t.s2_lo;
        hi += t.s2_hi + (nlo < lo ? 1ull : 0ull);
        lo = nlo;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long o1 = __shfl_xor(s1, off, kWave);
        const unsigned long long olo = __shfl_xor(lo, off, kWave);
        const unsigned long long ohi = __shfl_xor(hi, off, kWave);
        s1 += o1;
        const unsigned long long nlo = lo + olo;
        hi += ohi + (nlo < lo ? 1ull : 0ull);
        lo = nlo;
    }
    if (threadIdx.x != 0) return;
    FmStats out;
    out.s1 = s1;
    out.s2_lo = lo;
    out.s2_hi = hi;
    if (len == 0) {
        out.mean = 0.0f;
        out.scale = 1.0f;
    } else {
        const double dn = (double)len;
        const double mean_q = (double)s1 / dn;
        out.mean = (float)(mean_q / 268435456.0);
        const double s2d = (double)hi * 18446744073709551616.0 + (double)lo;
        const double m2 = ((double)s1 * (double)s1) / dn;
        const double var = ((s2d - m2) / dn) / 72057594037927936.0;
        out.scale = var > 0 ? (float)(1.0 / sqrt(var)) : 1.0f;
    }
    stats[id] = out;
}

}  // namespace tdoa
