// exact_reference.hpp -- mode A: the reference's EXECUTED chain on the GPU, with the
// accumulation order of the Go source wherever that order is observable.
//
//   processor.go:195-201  u8 -> complex64, true f32 division           k_ex_u8_to_c64
//   processor.go:322-333  calculateSignalPower (f32 products, f64 sum) k_ex_power_*
//   processor.go:299-319  removeDCBias (SEQUENTIAL f32 complex sum)    k_ex_seq_sum, k_ex_sub_const
//   processor.go:270-296  applyLowPassFilter (fresh sequential f32 sum
//                         per output, ascending j, edge-truncated)      k_ex_moving_average
//   processor.go:384-394  applyHighPassFilter = x - lowpass(x)          (epilogue of the above)
//   processor.go:412-434  applyNotchFilter = x - band * 0.8             k_ex_notch_combine
//   processor.go:336-351  normalizeSignal                                k_ex_scale
//   processor.go:646-736  timeDomainCorrelation                          k_ex_tdc_blocks, k_ex_tdc_finalize
//   simple_corr.go:83-160 simpleCorrelate (f32 accumulators)             k_ex_simple_corr
//   fast_analyzer.go:229-253 fastDFT (complex128, table twiddles)        k_ex_fast_dft
//
// Go/amd64 lowering that the kernels mirror: no FMA (contract off), complex64 * complex64
// evaluated in f64 and rounded once, complex64 / complex(real, 0) evaluated by
// runtime.complex128div in f64 and rounded once.
#pragma once

#include "device_common.hpp"

namespace tdoa {

// complex64 / complex(float32(c), 0) exactly as runtime.complex128div does it
__device__ __forceinline__ float2 go_div_real(float2 n, float c)
{
#pragma clang fp contract(off)
    double mr = (double)c, mi = 0.0;
    double ratio = mi / mr;
    double denom = mr + ratio * mi;
    double e = ((double)n.x + (double)n.y * ratio) / denom;
    double f = ((double)n.y - (double)n.x * ratio) / denom;
    return make_float2((float)e, (float)f);
}

// complex64 * complex64 as the Go compiler lowers it (f64 arithmetic, one rounding)
__device__ __forceinline__ float2 go_mul(float2 a, float2 b)
{
#pragma clang fp contract(off)
    double r = (double)a.x * (double)b.x - (double)a.y * (double)b.y;
    double i = (double)a.x * (double)b.y + (double)a.y * (double)b.x;
    return make_float2((float)r, (float)i);
}

__global__ void k_ex_u8_to_c64(const uint8_t *raw, size_t n, float2 *out)
{
#pragma clang fp contract(off)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float iv = ((float)raw[2 * i] - 127.5f) / 127.5f;
    float qv = ((float)raw[2 * i + 1] - 127.5f) / 127.5f;
    out[i] = make_float2(iv, qv);
}

// ---- power: f32 re*re+im*im widened to f64; fixed-shape tree (order differs from the
// sequential Go loop only by f64 reassociation, ~1e-16 relative) ------------------------
constexpr int kPowChunk = 8192;

__global__ __launch_bounds__(256) void k_ex_power_partial(const float2 *sig, size_t n, double *partials)
{
#pragma clang fp contract(off)
    size_t start = (size_t)blockIdx.x * kPowChunk;
    double acc = 0.0;
    for (size_t i = start + threadIdx.x; i < start + kPowChunk && i < n; i += 256) {
        float2 v = sig[i];
        float p = v.x * v.x + v.y * v.y;
        acc += (double)p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// single block: sum the partials, power = total / n
__global__ __launch_bounds__(256) void k_ex_power_final(const double *partials, int count, size_t n, double *power)
{
#pragma clang fp contract(off)
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) acc += partials[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *power = n ? ((red[0] + red[1]) + (red[2] + red[3])) / (double)n : 0.0;
}

// ---- removeDCBias: the f32 running sum is order-dependent at the 1e-5 level for 2e6
// samples, so it is reproduced as ONE sequential chain per component: lane 0 adds the real
// parts, lane 1 the imaginary parts (one v_add_f32 per sample for both), all 64 lanes stage the
// next 1024 samples through registers into split re/im LDS planes while the chain runs.
// The chain is latency-bound at one dependent add per sample; LDS reads are batched 32 ahead.
// Samples beyond n are +0.0: x + (+0.0) is exact and a running sum that starts at +0.0 can
// never be -0.0, so padding equals stopping.  One wave: program order replaces barriers.
// blockIdx.x selects one of several independent signals (sig/n/mean_out arrays of descriptors).
struct SeqSumJob {
    const float2 *sig;
    size_t n;
    float2 *mean_out;
};

__global__ __launch_bounds__(64) void k_ex_seq_sum(SeqSumJob j0, SeqSumJob j1)
{
#pragma clang fp contract(off)
    __shared__ float plane[2][2][1024];          // [buffer][re | im][sample]
    const SeqSumJob job = blockIdx.x ? j1 : j0;
    const float2 *sig = job.sig;
    const size_t n = job.n;
    const int lane = threadIdx.x;
    float acc = 0.0f;
    const size_t nchunks = (n + 1023) / 1024;
    float2 regs[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const size_t idx = (size_t)q * 64 + lane;
        regs[q] = idx < n ? sig[idx] : make_float2(0.0f, 0.0f);
    }
    for (size_t c = 0; c < nchunks; c++) {
        const int b = (int)(c & 1);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            plane[b][0][q * 64 + lane] = regs[q].x;
            plane[b][1][q * 64 + lane] = regs[q].y;
        }
        if (c + 1 < nchunks) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const size_t idx = (c + 1) * 1024 + (size_t)q * 64 + lane;
                regs[q] = idx < n ? sig[idx] : make_float2(0.0f, 0.0f);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the plane writes above have landed
        if (lane < 2) {
            const float4 *p4 = reinterpret_cast<const float4 *>(plane[b][lane]);
            for (int k = 0; k < 256; k += 8) {   // 32 samples per trip: 8 LDS reads issued together
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = p4[k + u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    acc = acc + v[u].x;
                    acc = acc + v[u].y;
                    acc = acc + v[u].z;
                    acc = acc + v[u].w;
                }
            }
        }
    }
    const float sr = __shfl(acc, 0, kWave), si = __shfl(acc, 1, kWave);
    if (lane == 0) *job.mean_out = n ? go_div_real(make_float2(sr, si), (float)n) : make_float2(0.0f, 0.0f);
}

__global__ void k_ex_sub_const(const float2 *in, size_t n, const float2 *c, float2 *out)
{
#pragma clang fp contract(off)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 m = *c, v = in[i];
    out[i] = make_float2(v.x - m.x, v.y - m.y);
}

// ---- applyLowPassFilter: every output is its own sequential f32 sum over ascending j.
// A thread owns R consecutive outputs and walks j once; each loaded tap is added to every
// accumulator whose window contains it, so each accumulator still sees ascending j.
// Taps outside [0, n) are +0.0 in LDS: adding +0.0 is exact and the accumulators can
// never be -0.0 (they start at +0.0), so the result equals skipping those taps.
// LDS index i -> i + i/8 (one pad per 8 complex): stride 9 between lanes, conflict-free.
constexpr int kMaR = 8;
constexpr int kMaThreads = 256;
constexpr int kMaTile = kMaR * kMaThreads;   // outputs per block

__device__ __forceinline__ int ma_pad(int i) { return i + (i >> 3); }

// mode 0: out = MA(in);  mode 1: out = in - MA(in)  (applyHighPassFilter)
__global__ __launch_bounds__(kMaThreads) void k_ex_moving_average(const float2 *in, long long n, int half,
                                                                  float2 *out, int mode)
{
#pragma clang fp contract(off)
    extern __shared__ float2 lds[];
    const long long tile0 = (long long)blockIdx.x * kMaTile;
    const int span = kMaTile + 2 * half;
    for (int k = threadIdx.x; k < span; k += kMaThreads) {
        long long j = tile0 - half + k;
        lds[ma_pad(k)] = (j >= 0 && j < n) ? in[j] : make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    const int base = threadIdx.x * kMaR;   // local index of the first tap of output 0
    float ar[kMaR], ai[kMaR];
#pragma unroll
    for (int r = 0; r < kMaR; r++) ar[r] = ai[r] = 0.0f;
    const int w = 2 * half;                // last tap offset of each window
    const int total = w + kMaR;            // taps this thread walks
    if (w >= kMaR - 1) {
        // prologue: tap k feeds accumulators 0..k
#pragma unroll
        for (int k = 0; k < kMaR - 1; k++) {
            float2 x = lds[ma_pad(base + k)];
#pragma unroll
            for (int r = 0; r < kMaR; r++)
                if (r <= k) { ar[r] = ar[r] + x.x; ai[r] = ai[r] + x.y; }
        }
        // main: every accumulator takes the tap
        for (int k = kMaR - 1; k <= w; k++) {
            float2 x = lds[ma_pad(base + k)];
#pragma unroll
            for (int r = 0; r < kMaR; r++) { ar[r] = ar[r] + x.x; ai[r] = ai[r] + x.y; }
        }
        // epilogue: tap w+1+e feeds accumulators e+1..R-1
#pragma unroll
        for (int e = 0; e < kMaR - 1; e++) {
            float2 x = lds[ma_pad(base + w + 1 + e)];
#pragma unroll
            for (int r = 0; r < kMaR; r++)
                if (r > e) { ar[r] = ar[r] + x.x; ai[r] = ai[r] + x.y; }
        }
    } else {
        for (int k = 0; k < total; k++) {
            float2 x = lds[ma_pad(base + k)];
#pragma unroll
            for (int r = 0; r < kMaR; r++)
                if (k >= r && k - r <= w) { ar[r] = ar[r] + x.x; ai[r] = ai[r] + x.y; }
        }
    }
#pragma unroll
    for (int r = 0; r < kMaR; r++) {
        long long i = tile0 + base + r;
        if (i < n) {
            long long j0 = i - half < 0 ? 0 : i - half;
            long long j1 = i + half > n - 1 ? n - 1 : i + half;
            float2 y = go_div_real(make_float2(ar[r], ai[r]), (float)(int)(j1 - j0 + 1));
            if (mode == 1) {
                float2 x = lds[ma_pad(base + r + half)];
                y = make_float2(x.x - y.x, x.y - y.y);
            }
            out[i] = y;
        }
    }
}

// applyNotchFilter tail: out = x - band * complex64(0.8)
__global__ void k_ex_notch_combine(const float2 *x, const float2 *band, size_t n, float2 *out)
{
#pragma clang fp contract(off)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 p = go_mul(band[i], make_float2(0.8f, 0.0f));
    float2 v = x[i];
    out[i] = make_float2(v.x - p.x, v.y - p.y);
}

// normalizeSignal: scale = float32(1/sqrt(power)); unchanged when power <= 0
__global__ void k_ex_scale(const float2 *in, size_t n, const double *power, float2 *out)
{
#pragma clang fp contract(off)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p = *power;
    float2 v = in[i];
    if (p > 0) {
        float s = (float)(1.0 / sqrt(p));
        v = make_float2(v.x * s, v.y * s);
    }
    out[i] = v;
}

// ---- timeDomainCorrelation --------------------------------------------------------------
// blockcorr[b][d] = (sum_{i in block b} f64(f32(tr*sr + ti*si))) / block, the sum taken
// sequentially in i exactly as processor.go:700-709.  One thread per lag; a workgroup
// covers 256 consecutive lags of one template block, staging both operands in LDS.
constexpr int kTdcLags = 256;

__global__ __launch_bounds__(kTdcLags) void k_ex_tdc_blocks(const float2 *tpl, const float2 *sig, long long sig_len,
                                                           int block, int n_lags, double *blockcorr)
{
#pragma clang fp contract(off)
    extern __shared__ float2 lds[];          // [block] template, then [block + kTdcLags] signal
    float2 *lt = lds, *ls = lds + block;
    const long long b0 = (long long)blockIdx.y * block;
    const int d0 = blockIdx.x * kTdcLags;
    for (int k = threadIdx.x; k < block; k += kTdcLags) lt[k] = tpl[b0 + k];
    for (int k = threadIdx.x; k < block + kTdcLags; k += kTdcLags) {
        long long j = b0 + d0 + k;
        ls[k] = j < sig_len ? sig[j] : make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    const int d = d0 + threadIdx.x;
    if (d >= n_lags) return;
    double acc = 0.0;
    for (int i = 0; i < block; i++) {
        float2 t = lt[i], s = ls[i + threadIdx.x];
        float p = t.x * s.x + t.y * s.y;
        acc += (double)p;
    }
    blockcorr[(size_t)blockIdx.y * n_lags + d] = acc / (double)block;
}

// few lags (the reference's own equal-length call pattern evaluates only lag 0): one wave
// per (lag, block); the in-block sum is a fixed tree instead of the sequential chain.
__global__ __launch_bounds__(256) void k_ex_tdc_blocks_few(const float2 *tpl, const float2 *sig, int block,
                                                          int n_lags, int n_blocks, double *blockcorr)
{
#pragma clang fp contract(off)
    const int wave = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wave >= n_blocks * n_lags) return;
    const int b = wave / n_lags, d = wave % n_lags;
    const long long b0 = (long long)b * block;
    double acc = 0.0;
    for (int i = lane; i < block; i += 64) {
        float2 t = tpl[b0 + i], s = sig[b0 + i + d];
        float p = t.x * s.x + t.y * s.y;
        acc += (double)p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if (lane == 0) blockcorr[(size_t)b * n_lags + d] = acc / (double)block;
}

// per lag: sequential sum over blocks, /numBlocks, * sqrt(numBlocks*block) (processor.go:714-720),
// then first-max |corr| over ascending lag (processor.go:722-725)
__global__ __launch_bounds__(256) void k_ex_tdc_finalize(const double *blockcorr, int n_blocks, int n_lags, int block,
                                                        double *corr_out)
{
#pragma clang fp contract(off)
    int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= n_lags) return;
    double c = 0.0;
    for (int b = 0; b < n_blocks; b++) c += blockcorr[(size_t)b * n_lags + d];
    if (n_blocks > 0) {
        c /= (double)n_blocks;
        c *= sqrt((double)(n_blocks * block));
    }
    corr_out[d] = c;
}

struct ExPeak {
    int delay;
    int pad;
    double corr;
};

// single block: first strictly-greater |corr| in ascending lag order == (max |corr|, lowest lag)
__global__ __launch_bounds__(256) void k_ex_first_max(const double *corr, int n_lags, ExPeak *out)
{
    double best = 0.0;
    int best_d = 0x7fffffff;
    for (int d = threadIdx.x; d < n_lags; d += 256) {
        double a = fabs(corr[d]);
        if (a > best) {            // strictly greater; NaN never wins
            best = a;
            best_d = d;
        }
    }
    __shared__ double sb[256];
    __shared__ int sd[256];
    sb[threadIdx.x] = best;
    sd[threadIdx.x] = best_d;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            double ob = sb[threadIdx.x + s];
            int od = sd[threadIdx.x + s];
            if (ob > sb[threadIdx.x] || (ob == sb[threadIdx.x] && od < sd[threadIdx.x])) {
                sb[threadIdx.x] = ob;
                sd[threadIdx.x] = od;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ExPeak p;
        if (sb[0] > 0.0 && sd[0] != 0x7fffffff) {
            p.delay = sd[0];
            p.corr = corr[sd[0]];
        } else {
            p.delay = 0;
            p.corr = 0.0;
        }
        p.pad = 0;
        *out = p;
    }
}

// ---- simpleCorrelate: per lag, sequential f32 dot product and signal power -----------------
struct SimpleLag {
    float corr, spow;
};

__global__ __launch_bounds__(256) void k_ex_simple_corr(const float2 *tpl, long long tl, const float2 *sig,
                                                       long long sl, int n_lags, SimpleLag *out)
{
#pragma clang fp contract(off)
    __shared__ float2 lt[512];
    __shared__ float2 ls[512 + 256];
    const int d0 = blockIdx.x * 256;
    const int d = d0 + threadIdx.x;
    float c = 0.0f, sp = 0.0f;
    for (long long i0 = 0; i0 < tl; i0 += 512) {
        __syncthreads();
        for (int k = threadIdx.x; k < 512; k += 256) lt[k] = (i0 + k) < tl ? tpl[i0 + k] : make_float2(0.0f, 0.0f);
        for (int k = threadIdx.x; k < 512 + 256; k += 256) {
            long long j = i0 + d0 + k;
            ls[k] = j < sl ? sig[j] : make_float2(0.0f, 0.0f);
        }
        __syncthreads();
        int m = (int)((tl - i0) < 512 ? (tl - i0) : 512);
        for (int i = 0; i < m; i++) {
            if (i0 + i + d >= sl) break;           // simple_corr.go:128-130
            float2 t = lt[i], s = ls[i + threadIdx.x];
            c = c + (t.x * s.x + t.y * s.y);
            sp = sp + (s.x * s.x + s.y * s.y);
        }
    }
    if (d < n_lags) {
        out[d].corr = c;
        out[d].spow = sp;
    }
}

// sequential f32 template power (simple_corr.go:115-118): one lane, LDS-staged like k_ex_seq_sum
__global__ __launch_bounds__(64) void k_ex_seq_power_f32(const float2 *sig, size_t n, float *out)
{
#pragma clang fp contract(off)
    __shared__ float2 buf[1024];
    float acc = 0.0f;
    for (size_t c0 = 0; c0 < n; c0 += 1024) {
        __syncthreads();
        for (int k = threadIdx.x; k < 1024; k += 64) buf[k] = (c0 + k) < n ? sig[c0 + k] : make_float2(0.0f, 0.0f);
        __syncthreads();
        if (threadIdx.x == 0) {
            int m = (int)((n - c0) < 1024 ? (n - c0) : 1024);
            for (int k = 0; k < m; k++) acc = acc + (buf[k].x * buf[k].x + buf[k].y * buf[k].y);
        }
    }
    if (threadIdx.x == 0) *out = acc;
}

// ---- fastDFT: complex128, dft[k] = sum_i x[i] * tw[(k*i) % n], sequential in i -------------
__global__ __launch_bounds__(256) void k_ex_fast_dft(const double2 *x, const double2 *tw, int n, double2 *out)
{
#pragma clang fp contract(off)
    int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double sr = 0.0, si = 0.0;
    int idx = 0;
    for (int i = 0; i < n; i++) {
        double2 a = x[i], b = tw[idx];
        double pr = a.x * b.x - a.y * b.y;
        double pi = a.x * b.y + a.y * b.x;
        sr = sr + pr;
        si = si + pi;
        idx += k;
        if (idx >= n) idx -= n;
    }
    out[k] = make_double2(sr, si);
}

}  // namespace tdoa
