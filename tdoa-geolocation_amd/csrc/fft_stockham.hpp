// fft_stockham.hpp -- K2/K4: LDS-staged Stockham FFT building blocks and the four
// passes of the packed-real cross-correlation (forward column, forward row,
// inverse row with the fused conj-multiply K3, inverse column with fused argmax K5).
//
// A real window y[0..N) is packed as z[m] = y[2m] + i*y[2m+1], m < Nc = N/2, and
// transformed with a four-step complex FFT, Nc = N1 * N2, input index
// m = n2*N1 + n1, spectrum index k = k1*N2 + k2:
//   forward column pass : Y[k2][n1] = sum_n2 z[n2*N1+n1] W_N2^(n2 k2), times W_Nc^(n1 k2)
//   forward row pass    : Zs[k2][k1] = sum_n1 T[k2][n1] W_N1^(n1 k1)       (= Z[k1*N2+k2])
// The spectrum stays in this transposed [k2][k1] layout: the pointwise product
// does not care, and the inverse consumes it directly (no transpose kernel):
//   inverse row pass    : Q from (Za, Zb) at k and Nc-k, row IFFT over k1, times W_Nc^(-n1 k2)
//   inverse column pass : IFFT over k2, q[m] = N*4*(r[2m] + i r[2m+1]), argmax over lags
//
// Reference evidence for the conventions: forward sign e^{-2 pi i kj/n}, unnormalised
// (processor.go:528); padding N = nextPow2(L + maxLag) (processor.go:563); cross power
// conj(template)*signal so that lag > 0 means the signal is delayed (processor.go:700-705).
#pragma once

#include "device_common.hpp"
#include "k1_discriminator.hpp"
#include "k1_single_look.hpp"

namespace tdoa {

template <bool INV>
__device__ __forceinline__ void bfly4(float2 &a0, float2 &a1, float2 &a2, float2 &a3)
{
    float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
    float2 t3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);   // d * (+i | -i)
    a0 = cadd(t0, t2);
    a1 = cadd(t1, t3);
    a2 = csub(t0, t2);
    a3 = csub(t1, t3);
}

// Generic in-LDS Stockham autosort FFT of NV interleaved vectors of length M
// (element e of vector v at [e*NV + v]); radix-4 rounds plus one radix-2 round
// when log2(M) is odd.  Ping-pongs between b0 and b1; returns the buffer that
// holds the result.  All threads of the block must call it.
template <bool INV>
__device__ float2 *lds_fft(float2 *b0, float2 *b1, int M, int logM, int NV, int logNV)
{
    float2 *src = b0, *dst = b1;
    int Ns = 1, rem = logM;
    while (rem > 0) {
        if (rem >= 2) {
            const int q = M >> 2;
            const int items = q << logNV;
            const float inv2 = 2.0f / (float)(Ns * 4);
            for (int it = threadIdx.x; it < items; it += blockDim.x) {
                int v = it & (NV - 1), j = it >> logNV;
                int jm = j & (Ns - 1);
                float2 x0 = src[((j) << logNV) + v];
                float2 x1 = src[((j + q) << logNV) + v];
                float2 x2 = src[((j + 2 * q) << logNV) + v];
                float2 x3 = src[((j + 3 * q) << logNV) + v];
                if (Ns > 1) {
                    float2 w1 = unit_root((float)jm, inv2, INV);
                    float2 w2 = cmul(w1, w1);
                    float2 w3 = cmul(w2, w1);
                    x1 = cmul(x1, w1);
                    x2 = cmul(x2, w2);
                    x3 = cmul(x3, w3);
                }
                bfly4<INV>(x0, x1, x2, x3);
                int d = ((j - jm) << 2) + jm;
                dst[((d) << logNV) + v] = x0;
                dst[((d + Ns) << logNV) + v] = x1;
                dst[((d + 2 * Ns) << logNV) + v] = x2;
                dst[((d + 3 * Ns) << logNV) + v] = x3;
            }
            Ns <<= 2;
            rem -= 2;
        } else {
            const int q = M >> 1;
            const int items = q << logNV;
            const float inv2 = 2.0f / (float)(Ns * 2);
            for (int it = threadIdx.x; it < items; it += blockDim.x) {
                int v = it & (NV - 1), j = it >> logNV;
                int jm = j & (Ns - 1);
                float2 x0 = src[((j) << logNV) + v];
                float2 x1 = src[((j + q) << logNV) + v];
                if (Ns > 1) x1 = cmul(x1, unit_root((float)jm, inv2, INV));
                int d = ((j - jm) << 1) + jm;
                dst[((d) << logNV) + v] = cadd(x0, x1);
                dst[((d + Ns) << logNV) + v] = csub(x0, x1);
            }
            Ns <<= 1;
            rem -= 1;
        }
        __syncthreads();
        float2 *t = src;
        src = dst;
        dst = t;
    }
    return src;
}

struct FftPlan {
    int N1, N2, logN1, logN2;   // Nc = N1*N2
    int C, logC;                // columns per tile in the column passes
    long long Nc;
    // TZ layout of the radix-16 kernels: row k2 of a station-window starts at k2 N1 + (k2 >> 8) zpad, station-windows are Zs
    // apart.  zpad > 0 only for the two-sweep column pass (N2 = 2048, 4096), whose second sweep combines rows that would
    // otherwise lie exactly 256 N1 elements = 8 MB apart (k_fwd_col_finish); everywhere else zpad = 0 and Zs = Nc.
    int zpad;
    long long Zs;
    // round 5: N2 = odd x 2^k with odd = 1 (every plan up to round 4), or 5: the 4096 x 2560 plan of ten-second windows
    // (N = 5 x 2^22 instead of 2^25: the 20 020 000 samples + lags of BASELINE config 3 in 20 971 520 points instead of
    // 33 554 432) and its small plan 4096 x 160.  Where odd != 1 the twiddles whose denominator holds N2 -- W_N2, W_Nc, W_2Nc --
    // come from unit_root_any, and index wraps by N2 or Nc are no longer masks.
    int odd;
};

// ---------------------------------------------------------------------------
// element m of the packed window from MATERIALISED phase codes (int32, stored negated: k1_discriminator.hpp):
// codes (2m, 2m+1) -> normalised float2, zero beyond len.  Rows are 32-byte aligned.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int2 code_fetch(const int *codes, long long m, int len)
{
    return reinterpret_cast<const int2 *>(codes)[2 * m < len ? m : 0];
}

// (two steps, so that a thread can issue all its loads before it converts the first one)
__device__ __forceinline__ float2 code_convert(int2 w, long long m, int len, float mean, float scale)
{
    const long long i0 = 2 * m;
    const float v0 = k1_normalise(w.x, mean, scale);
    const float v1 = k1_normalise(w.y, mean, scale);
    return make_float2(i0 < len ? v0 : 0.0f, i0 + 1 < len ? v1 : 0.0f);
}

__device__ __forceinline__ float2 code_element(const int *codes, long long m, int len, float mean, float scale)
{
    return code_convert(code_fetch(codes, m, len), m, len, mean, scale);
}

// ---------------------------------------------------------------------------
// forward column pass: phase codes -> normalise -> pack -> length-N2 FFT down C adjacent
// columns -> twiddle -> T[k2][n1]
// grid: (N1 / C, n_station_windows), dynamic LDS: 2 * N2 * C * 8 bytes
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fwd_col_c16(const SWDesc *sw, const int *codes, long long code_stride,
                                                     const FmStats *stats, float2 *T, FftPlan pl)
{
    extern __shared__ float2 lds[];
    const int tile = pl.N2 << pl.logC;
    const int len = sw[blockIdx.y].len;
    const int *row = codes + (size_t)blockIdx.y * code_stride;
    const float mean = stats[blockIdx.y].mean, scale = stats[blockIdx.y].scale;
    const int c0 = blockIdx.x << pl.logC;
    for (int e = threadIdx.x; e < tile; e += blockDim.x) {
        int c = e & (pl.C - 1), n2 = e >> pl.logC;
        lds[e] = code_element(row, (long long)n2 * pl.N1 + c0 + c, len, mean, scale);
    }
    __syncthreads();
    float2 *r = lds_fft<false>(lds, lds + tile, pl.N2, pl.logN2, pl.C, pl.logC);
    float2 *out = T + (size_t)blockIdx.y * pl.Nc;
    const float inv2 = 2.0f / (float)pl.Nc;
    for (int e = threadIdx.x; e < tile; e += blockDim.x) {
        int c = e & (pl.C - 1), k2 = e >> pl.logC;
        int n1 = c0 + c;
        long long ex = ((long long)n1 * k2) & (pl.Nc - 1);
        float2 w = unit_root((float)ex, inv2, false);
        out[(size_t)k2 * pl.N1 + n1] = cmul(r[e], w);
    }
}

// ---------------------------------------------------------------------------
// forward row pass: length-N1 FFT of each contiguous row, in place (T -> Zs)
// grid: (N2, n_station_windows), dynamic LDS: 2 * N1 * 8 bytes
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fwd_row(float2 *TZ, FftPlan pl)
{
    extern __shared__ float2 lds[];
    float2 *row = TZ + (size_t)blockIdx.y * pl.Nc + (size_t)blockIdx.x * pl.N1;
    for (int e = threadIdx.x; e < pl.N1; e += blockDim.x) lds[e] = row[e];
    __syncthreads();
    float2 *r = lds_fft<false>(lds, lds + pl.N1, pl.N1, pl.logN1, 1, 0);
    for (int e = threadIdx.x; e < pl.N1; e += blockDim.x) row[e] = r[e];
}

// ---------------------------------------------------------------------------
// K3 fused into the inverse row pass (packed-real mode).
// With E2 = z + conj(zm), O2 = -i (z - conj(zm)) (twice the even/odd spectra),
// A+- = E2a +- w O2a, B+- = E2b +- w O2b, w = W_N^k:
//   G = conj(A+) B+ (= 4 G[k]),  H = conj(A-) B- (= 4 conj(G[Nc-k]))
//   Q[k]    = (G + H) + i (G - H) conj(w)
//   Q[Nc-k] = conj(G + H) + i conj((G - H) conj(w))
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pair_q(float2 za, float2 zam, float2 zb, float2 zbm, float2 w,
                                       float2 &q, float2 &qm)
{
    float2 e2a = make_float2(za.x + zam.x, za.y - zam.y);
    float2 da = make_float2(za.x - zam.x, za.y + zam.y);     // z - conj(zm)
    float2 o2a = make_float2(da.y, -da.x);                    // -i * da
    float2 e2b = make_float2(zb.x + zbm.x, zb.y - zbm.y);
    float2 db = make_float2(zb.x - zbm.x, zb.y + zbm.y);
    float2 o2b = make_float2(db.y, -db.x);
    float2 woa = cmul(w, o2a), wob = cmul(w, o2b);
    float2 ap = cadd(e2a, woa), am = csub(e2a, woa);
    float2 bp = cadd(e2b, wob), bm = csub(e2b, wob);
    float2 g = cmulc(ap, bp), h = cmulc(am, bm);
    float2 qe = cadd(g, h);
    float2 qo = cmul(csub(g, h), cconj(w));
    q = make_float2(qe.x - qo.y, qe.y + qo.x);               // qe + i qo
    qm = make_float2(qe.x + qo.y, -qe.y + qo.x);             // conj(qe) + i conj(qo)
}

// The same in packed-f32 instructions with their operand modifiers (op_sel picks the half of a register pair per result
// lane, neg_lo / neg_hi negate per lane): a complex product is two instructions and a multiplication by +-i or a
// conjugation costs nothing, where the compiler's version of the same arithmetic spent about 100 instructions per
// call, a quarter of them moves that build swapped pairs (22 here).  scripts/microbench/pk_complex.hip checks the
// modifier semantics on the device.  Used by the decimated path (pair_u_pk in k_pair_decimate16, unpack_u_pk in
// k_fwd_row4096_unpack); k_inv_row_pair4096 ran 17 % SLOWER with the packed form of pair_q (the asm statements pin the
// schedule between its 64 loads), so the row kernels keep pair_q.
typedef float v2f __attribute__((ext_vector_type(2)));
#define TDOA_PK2(op, d, a, b, mods) asm(op " %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b))
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b)      // a * b
{
    v2f t, r;
    TDOA_PK2("v_pk_mul_f32", t, a, b, "op_sel:[0,0] op_sel_hi:[0,1]");                       // (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;                                                                               // (- a.y b.y, + a.y b.x) added
}
__device__ __forceinline__ v2f pk_cmulc(v2f a, v2f b)     // conj(a) * b
{
    v2f t, r;
    TDOA_PK2("v_pk_mul_f32", t, a, b, "op_sel:[0,0] op_sel_hi:[0,1]");
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;                                                                               // (+ a.y b.y, - a.y b.x) added
}
// The pair's half of K3 when the stations' spectra arrive unpacked (U of k_fwd_row4096_unpack, fft_radix16.hpp):
// G = conj(Ua[k]) Ub[k], conj(H) = conj(Ua[Nc-k]) Ub[Nc-k], then as above: 10 instructions.
// dc: this is bin 0, whose U holds the two real numbers (A+[0], A-[0]): G = Ua.x Ub.x, H = Ua.y Ub.y.
__device__ __forceinline__ void pair_u_pk(float2 ua_, float2 uam_, float2 ub_, float2 ubm_, float2 w_, bool dc,
                                          float2 &q_, float2 &qm_)
{
    const v2f ua = {ua_.x, ua_.y}, uam = {uam_.x, uam_.y}, ub = {ub_.x, ub_.y}, ubm = {ubm_.x, ubm_.y}, w = {w_.x, w_.y};
    v2f qe, gh, q, qm;
    v2f g = pk_cmulc(ua, ub), hc = pk_cmulc(uam, ubm);
    if (dc) {
        g = v2f{ua_.x * ub_.x, 0.0f};
        hc = v2f{ua_.y * ub_.y, 0.0f};
    }
    TDOA_PK2("v_pk_add_f32", qe, g, hc, "neg_hi:[0,1]");                                     // G + H,  H = conj(hc)
    TDOA_PK2("v_pk_add_f32", gh, g, hc, "neg_lo:[0,1]");                                     // G - H
    const v2f qo = pk_cmulc(w, gh);                                                          // (G - H) conj(w)
    TDOA_PK2("v_pk_add_f32", q, qe, qo, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]");        // qe + i qo
    TDOA_PK2("v_pk_add_f32", qm, qe, qo, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]");       // conj(qe) + i conj(qo)
    q_ = make_float2(q.x, q.y);
    qm_ = make_float2(qm.x, qm.y);
}
// the station's half: U = (z + conj(zm)) - i w (z - conj(zm)), five instructions
__device__ __forceinline__ float2 unpack_u_pk(float2 z_, float2 zm_, float2 w_)
{
    const v2f z = {z_.x, z_.y}, zm = {zm_.x, zm_.y}, w = {w_.x, w_.y};
    v2f e2, d, u;
    TDOA_PK2("v_pk_add_f32", e2, z, zm, "neg_hi:[0,1]");                                     // z + conj(zm)
    TDOA_PK2("v_pk_add_f32", d, z, zm, "neg_lo:[0,1]");                                      // z - conj(zm)
    const v2f wd = pk_cmul(w, d);
    TDOA_PK2("v_pk_add_f32", u, e2, wd, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]");        // E2 - i (w d)
    return make_float2(u.x, u.y);
}
#undef TDOA_PK2

// grid: (N2 / 2, n_pair_windows), dynamic LDS: 2 * (2*N1) * 8 bytes
// block a == 0 owns the two self-mirrored rows (0 and N2/2); block a > 0 owns the
// mutually mirrored rows (a, N2 - a).
__global__ __launch_bounds__(256) void k_inv_row_pair(const PWDesc *pw, const float2 *Z, float2 *V, FftPlan pl)
{
    extern __shared__ float2 lds[];
    const PWDesc d = pw[blockIdx.y];
    const float2 *Za = Z + (size_t)d.sw_a * pl.Nc;
    const float2 *Zb = Z + (size_t)d.sw_b * pl.Nc;
    const int N1 = pl.N1, N2 = pl.N2;
    const int a = blockIdx.x;
    const int rowA = a == 0 ? 0 : a;
    const int rowB = a == 0 ? (N2 >> 1) : N2 - a;
    const float inv2N = 1.0f / (float)pl.Nc;      // 2 / N with N = 2 Nc
    if (a == 0) {
        for (int e = threadIdx.x; e < 2 * N1; e += blockDim.x) {
            int v = e & 1, k1 = e >> 1;
            int row = v ? rowB : rowA;
            int mk1 = v ? (N1 - 1 - k1) : ((N1 - k1) & (N1 - 1));
            size_t i = (size_t)row * N1 + k1, im = (size_t)row * N1 + mk1;
            long long k = (long long)k1 * N2 + row;
            float2 w = unit_root((float)k, inv2N, false);
            float2 q, qm;
            pair_q(Za[i], Za[im], Zb[i], Zb[im], w, q, qm);
            lds[e] = q;
        }
    } else {
        for (int k1 = threadIdx.x; k1 < N1; k1 += blockDim.x) {
            int mk1 = N1 - 1 - k1;
            size_t i = (size_t)rowA * N1 + k1, im = (size_t)rowB * N1 + mk1;
            long long k = (long long)k1 * N2 + rowA;
            float2 w = unit_root((float)k, inv2N, false);
            float2 q, qm;
            pair_q(Za[i], Za[im], Zb[i], Zb[im], w, q, qm);
            lds[2 * k1] = q;
            lds[2 * mk1 + 1] = qm;
        }
    }
    __syncthreads();
    float2 *r = lds_fft<true>(lds, lds + 2 * N1, N1, pl.logN1, 2, 1);
    float2 *out = V + (size_t)blockIdx.y * pl.Nc;
    const float inv2 = 2.0f / (float)pl.Nc;
    for (int e = threadIdx.x; e < 2 * N1; e += blockDim.x) {
        int v = e & 1, n1 = e >> 1;
        int k2 = v ? rowB : rowA;
        long long ex = ((long long)n1 * k2) & (pl.Nc - 1);
        out[(size_t)k2 * N1 + n1] = cmul(r[e], unit_root((float)ex, inv2, true));
    }
}

// ---------------------------------------------------------------------------
// inverse column pass + K5 argmax.  After the length-N2 IFFT down a column,
// element (n2, n1) is q[m], m = n2*N1 + n1.
// lags 2m (real part) and 2m+1 (imag part), minus N when >= N/2.
// Candidates with lag_lo <= lag <= lag_hi enter a 64-bit atomicMax key.
// grid: (N1 / C, n_pair_windows), dynamic LDS: 2 * N2 * C * 8 bytes
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_inv_col_peak(const float2 *V, unsigned long long *keys, const PWDesc *pw,
                                                      FftPlan pl, int lag_lo, int lag_hi, float *lag_dump,
                                                      float dump_scale)
{
    extern __shared__ float2 lds[];
    const int tile = pl.N2 << pl.logC;
    const float2 *in = V + (size_t)blockIdx.y * pl.Nc;
    const int c0 = blockIdx.x << pl.logC;
    for (int e = threadIdx.x; e < tile; e += blockDim.x) {
        int c = e & (pl.C - 1), k2 = e >> pl.logC;
        lds[e] = in[(size_t)k2 * pl.N1 + c0 + c];
    }
    __syncthreads();
    float2 *r = lds_fft<true>(lds, lds + tile, pl.N2, pl.logN2, pl.C, pl.logC);
    unsigned long long best = 0;
    const long long Nc = pl.Nc;
    for (int e = threadIdx.x; e < tile; e += blockDim.x) {
        int c = e & (pl.C - 1), n2 = e >> pl.logC;
        long long m = (long long)n2 * pl.N1 + c0 + c;
        float2 v = r[e];
        long long d0 = 2 * m;
        if (d0 >= Nc) d0 -= 2 * Nc;
        long long d1 = d0 + 1;
        if (d0 >= lag_lo && d0 <= lag_hi && v.x == v.x) {
            unsigned long long k = peak_key(v.x, (int)d0);
            best = k > best ? k : best;
            if (lag_dump) lag_dump[d0 - lag_lo] = v.x * dump_scale;
        }
        if (d1 >= lag_lo && d1 <= lag_hi && v.y == v.y) {
            unsigned long long k = peak_key(v.y, (int)d1);
            best = k > best ? k : best;
            if (lag_dump) lag_dump[d1 - lag_lo] = v.y * dump_scale;
        }
    }
    best = wave_max_u64(best);
    __shared__ unsigned long long red[4];
    int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) red[wid] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = red[0];
        for (int w = 1; w < (int)(blockDim.x / kWave); w++) b = red[w] > b ? red[w] : b;
        if (b) atomicMax(&keys[pw[blockIdx.y].out_index], b);
    }
}

struct PeakOut {      // mirrors tdoa_peak
    int32_t lag;
    float abs_corr;
    double corr;
};

// decode keys -> peaks; scale = 1 / (4 N sqrt(len_a)); slot_gain (single-look K1, k1_single_look.hpp): the pair-window's
// g_t g_s, published by the K5 kernel that built the key (nullptr: the values were normalised before the transforms)
__global__ void k_decode_peaks(const unsigned long long *keys, const double *scales, PeakOut *out, int n,
                               const double *slot_gain = nullptr)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    unsigned long long k = keys[id];
    unsigned int mag = (unsigned int)(k >> 32), low = (unsigned int)k;
    PeakOut p;
    if (k == 0 || mag == 0) {
        p.lag = 0;
        p.abs_corr = 0.0f;
        p.corr = 0.0;
    } else {
        unsigned int rank = 0x7fffffffu - (low >> 1);
        int lag = rank == 0 ? 0 : ((rank & 1u) ? (int)((rank + 1u) >> 1) : -(int)(rank >> 1));
        double v = (double)__uint_as_float(mag) * scales[id];
        if (slot_gain) v *= slot_gain[id];
        if (low & 1u) v = -v;
        p.lag = lag;
        p.corr = v;
        p.abs_corr = (float)fabs(v);
    }
    out[id] = p;
}

// ---------------------------------------------------------------------------
// (f)-4 sub-sample refinement.  For the peak lag d of every pair-window the three correlation
// values c[d-1], c[d], c[d+1] are re-evaluated as direct column sums over V (V[k2][n1] is the
// inverse row pass including its twiddle in every plan, so element m = n2*N1 + n1 of the packed
// result is sum_k2 V[k2][n1] e^{+2 pi i n2 k2/N2}; lag 2m is its real part, 2m+1 its imaginary
// part).  3*N2 scattered reads per unit -- nothing next to the passes that produced V.
// grid n_pw, 64 threads; raw[3*slot + q] unscaled like the keys.
// ---------------------------------------------------------------------------
// gain (decimated inverse): V is then the row-pass output of the small plan and value (lag l) is multiplied by
// gain[|floor(l / 2)|], the window of the decimation divided out.
// oc (single-look K1): the same additive term the K5 kernel gave the candidates.
__global__ __launch_bounds__(64) void k_refine_peaks(const float2 *V, const unsigned long long *keys,
                                                    const PWDesc *pw, FftPlan pl, float *raw, const float *gain = nullptr,
                                                    OnceCorr oc = OnceCorr{})
{
    const int slot = pw[blockIdx.x].out_index;
    const unsigned long long k = keys[slot];
    if (k == 0 || (unsigned int)(k >> 32) == 0) {
        if (threadIdx.x < 3) raw[3 * (size_t)slot + threadIdx.x] = 0.0f;
        return;
    }
    const unsigned int rank = 0x7fffffffu - ((unsigned int)k >> 1);
    const int lag = rank == 0 ? 0 : ((rank & 1u) ? (int)((rank + 1u) >> 1) : -(int)(rank >> 1));
    const float2 *in = V + (size_t)blockIdx.x * pl.Nc;
    const float inv2 = 2.0f / (float)pl.N2;
    for (int q = 0; q < 3; q++) {
        long long l = (long long)lag - 1 + q;
        if (l < 0) l += 2 * pl.Nc;
        const long long m = l >> 1;
        const int n2 = (int)(m >> pl.logN1), n1 = (int)(m & (pl.N1 - 1));
        float acc = 0.0f;
        for (int k2 = threadIdx.x; k2 < pl.N2; k2 += kWave) {
            const float2 x = in[(size_t)k2 * pl.N1 + n1];
            const float2 w = pl.odd == 1 ? unit_root((float)((n2 * k2) & (pl.N2 - 1)), inv2, true)
                                         : unit_root_any((float)((n2 * k2) % pl.N2), 0.25f * (float)pl.N2, 4.0f / (float)pl.N2, true);
            acc += (l & 1) ? x.x * w.y + x.y * w.x : x.x * w.x - x.y * w.y;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
        if (threadIdx.x == 0) {
            const long long ms = ((long long)lag - 1 + q) >> 1;        // signed packed index (arithmetic shift = floor)
            float val = gain ? acc * gain[ms < 0 ? -ms : ms] : acc;
            if (oc.fin) val += once_correction(oc, once_pair(oc, pw[blockIdx.x]), (long long)lag - 1 + q);
            raw[3 * (size_t)slot + q] = val;
        }
    }
}

struct FineOut {      // mirrors tdoa_fine_peak
    double delay;
    float frac;
    float y[3];
    int32_t plausible;
    int32_t reserved;
};

// raw neighbours -> parabola vertex (f64), delay = lag + frac, plausibility gate
__global__ void k_decode_fine(const unsigned long long *keys, const double *scales, const float *raw, FineOut *out,
                              double gate, int n, const double *slot_gain = nullptr)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const unsigned long long k = keys[id];
    FineOut f;
    f.delay = 0.0;
    f.frac = 0.0f;
    f.y[0] = f.y[1] = f.y[2] = 0.0f;
    f.reserved = 0;
    int lag = 0;
    if (k != 0 && (unsigned int)(k >> 32) != 0) {
        const unsigned int low = (unsigned int)k;
        const unsigned int rank = 0x7fffffffu - (low >> 1);
        lag = rank == 0 ? 0 : ((rank & 1u) ? (int)((rank + 1u) >> 1) : -(int)(rank >> 1));
        const double sc0 = slot_gain ? scales[id] * slot_gain[id] : scales[id];
        const double sc = (low & 1u) ? -sc0 : sc0;
        const double ym = (double)raw[3 * (size_t)id] * sc, y0 = (double)raw[3 * (size_t)id + 1] * sc,
                     yp = (double)raw[3 * (size_t)id + 2] * sc;
        const double den = ym - 2.0 * y0 + yp;
        double fr = 0.0;
        if (den < 0.0) {
            fr = 0.5 * (ym - yp) / den;
            fr = fr > 0.5 ? 0.5 : (fr < -0.5 ? -0.5 : fr);
        }
        f.frac = (float)fr;
        f.delay = (double)lag + fr;
        f.y[0] = (float)ym; f.y[1] = (float)y0; f.y[2] = (float)yp;
    }
    f.plausible = fabs(f.delay) <= gate ? 1 : 0;
    out[id] = f;
}

}  // namespace tdoa
