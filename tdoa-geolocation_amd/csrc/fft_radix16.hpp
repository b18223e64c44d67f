// fft_radix16.hpp -- the hot-size kernels: register-resident radix-16 Stockham stages with
// one LDS exchange between stages, for rows of 4096 = 16^3 points and columns of 256 = 16^2.
// Same math and layouts as fft_stockham.hpp (which stays as the any-size fallback); these
// replace it when the plan is N1 = 4096 (and N2 = 256 for the forward column pass).
//
// Per thread: 16 complex values in VGPRs, a 16-point DFT as 4x4 radix-4 butterflies, twiddles
// w^r built from one sincospi per stage by a depth-4 product tree.  LDS images are padded by one
// element per 16 so that the stride-16 writes of the first stage do not pile onto one bank.
#pragma once

#include <initializer_list>
#include <type_traits>
#include <utility>

#include "device_common.hpp"
#include "fft_stockham.hpp"
#include "k1_single_look.hpp"
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
    auto tw = [](float2 a, float c, float s) { return cmul(a, make_float2(c, INV ? s : -s)); };      // a e^{-+i phi}: two packed ops
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

// v[r] *= w^r, r = 1..15.  w^r = g_q * w^(r mod 4) with g_q = w^(4q): products of depth <= 4,
// and only w, w^2, w^3 and the current g_q are live at any time.
__device__ __forceinline__ void mul_powers16(float2 (&v)[16], float2 w)
{
    const float2 w2 = cmul(w, w), w3 = cmul(w2, w), w4 = cmul(w2, w2);
    v[1] = cmul(v[1], w);
    v[2] = cmul(v[2], w2);
    v[3] = cmul(v[3], w3);
    float2 g = w4;
#pragma unroll
    for (int q = 1; q < 4; q++) {
        v[4 * q] = cmul(v[4 * q], g);
        v[4 * q + 1] = cmul(v[4 * q + 1], cmul(g, w));
        v[4 * q + 2] = cmul(v[4 * q + 2], cmul(g, w2));
        v[4 * q + 3] = cmul(v[4 * q + 3], cmul(g, w3));
        if (q == 1) g = cmul(w4, w4);
        if (q == 2) g = cmul(g, w4);
    }
}

// the same for the outputs X[k] (which live in v[oreg(k)]) with a separate base factor:
// X[k] *= base * step^k
__device__ __forceinline__ void mul_base_step16(float2 (&v)[16], float2 base, float2 step)
{
    const float2 s2 = cmul(step, step), s3 = cmul(s2, step), s4 = cmul(s2, s2);
    float2 g = base;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        v[oreg(4 * q)] = cmul(v[oreg(4 * q)], g);
        v[oreg(4 * q + 1)] = cmul(v[oreg(4 * q + 1)], cmul(g, step));
        v[oreg(4 * q + 2)] = cmul(v[oreg(4 * q + 2)], cmul(g, s2));
        v[oreg(4 * q + 3)] = cmul(v[oreg(4 * q + 3)], cmul(g, s3));
        g = cmul(g, s4);
    }
}

// the same on values in natural order: v[r] *= base * step^r
__device__ __forceinline__ void mul_base_step16_nat(float2 (&v)[16], float2 base, float2 step)
{
    const float2 s2 = cmul(step, step), s3 = cmul(s2, step), s4 = cmul(s2, s2);
    float2 g = base;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        v[4 * q] = cmul(v[4 * q], g);
        v[4 * q + 1] = cmul(v[4 * q + 1], cmul(g, step));
        v[4 * q + 2] = cmul(v[4 * q + 2], cmul(g, s2));
        v[4 * q + 3] = cmul(v[4 * q + 3], cmul(g, s3));
        g = cmul(g, s4);
    }
}

// The four-step twiddle W_Nc^(n1 k2) applied by the ROW pass to its inputs (thread j holds columns n1 = j + 256 r of row
// k2): W^(k2 j) * (W^(256 k2))^r.  The fused column kernel k_fwd_col256_k1<false> leaves it out: that kernel is the one
// short of issue slots, the row passes are HBM-bound with half of theirs free.
__device__ __forceinline__ void row_pre_twiddle(float2 (&v)[16], int k2, int j, const FftPlan &pl)
{
    const float inv2 = 2.0f / (float)pl.Nc;
    const int e0 = (k2 * j) & (int)(pl.Nc - 1), e1 = (k2 * 256) & (int)(pl.Nc - 1);      // k2 j < 2^20
    mul_base_step16_nat(v, unit_root((float)e0, inv2, false), unit_root((float)e1, inv2, false));
}

// Two-sweep column pass, first sweep: sub-transform a of G leaves Y_a[kb = j + 16 k] W_(256 G)^(a kb) = W^(a j) (W^(16 a))^k
// (outputs in v[oreg(k)]).  N2 = 256 G; G = 10 (pl.odd = 5) takes its roots from unit_root_any.
__device__ __forceinline__ void sub_twiddle16(float2 (&v)[16], int a, int j, const FftPlan &pl)
{
    if (pl.odd == 1) {
        const float invg = 2.0f / (float)pl.N2;
        mul_base_step16(v, unit_root((float)(a * j), invg, false), unit_root((float)(16 * a), invg, false));
    } else {
        const float qd = 0.25f * (float)pl.N2, iq = 4.0f / (float)pl.N2;      // a j, 16 a < N2
        mul_base_step16(v, unit_root_any((float)(a * j), qd, iq, false), unit_root_any((float)(16 * a), qd, iq, false));
    }
}

__device__ __forceinline__ int pad16(int i) { return i + (i >> 4); }

// a fetched code pair rides in the registers of the float2 it will become
__device__ __forceinline__ float2 code_park(int2 w) { return make_float2(__int_as_float(w.x), __int_as_float(w.y)); }
__device__ __forceinline__ int2 code_unpark(float2 v) { return make_int2(__float_as_int(v.x), __float_as_int(v.y)); }

// A value the compiler must treat as unknown.  The stage twiddles of a row transform are the same for every row a
// workgroup handles; when a kernel transforms several rows one after the other, common-subexpression elimination
// would keep all 2 x 15 twiddle powers alive from the first row to the last (60 VGPRs) instead of rebuilding them
// from one root (~25 complex multiplies).  Laundering the root per call keeps the register file for data.
__device__ __forceinline__ float2 opaque(float2 w)
{
    asm volatile("" : "+v"(w.x), "+v"(w.y));
    return w;
}
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
    mul_powers16(v, opaque(unit_root((float)(j & 15), 2.0f / 256.0f, INV)));
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
    mul_powers16(v, opaque(unit_root((float)j, 2.0f / 4096.0f, INV)));
    fft16<INV>(v);
}

// ---------------------------------------------------------------------------
// forward row pass, N1 = 4096, in place.  grid (N2, n_sw), 256 threads, static LDS 34 KB
// ---------------------------------------------------------------------------
// pre_tw: the column pass left the four-step twiddle to this kernel (row_pre_twiddle).
__global__ __launch_bounds__(256) void k_fwd_row4096(float2 *TZ, FftPlan pl, bool pre_tw)
{
    __shared__ float2 lds[kRowLds];
    const int k2 = blockIdx.x;
    float2 *row = TZ + (size_t)blockIdx.y * pl.Zs + (size_t)k2 * 4096 + (size_t)(k2 >> 8) * pl.zpad;
    const int j = threadIdx.x;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = load_nt(row + j + 256 * r);      // the column pass's output: read once
    if (pre_tw) row_pre_twiddle(v, k2, j, pl);
    fft16<false>(v);
    row4096_finish<false>(v, lds, j);
#pragma unroll
    for (int k = 0; k < 16; k++) row[j + 256 * k] = v[oreg(k)];
}

// ---------------------------------------------------------------------------
// forward row pass for the decimated inverse: rows k2 = a and N2 - a of one station-window together (a = 0: the two
// self-mirrored rows 0 and N2/2), and the HALF OF K3 THAT BELONGS TO THE STATION done here, once, instead of in
// k_pair_decimate16 once per pair the station is in.  With z = Z[k], zm = Z[Nc - k] (k = k2 + N2 k1, the mirror of (k2, k1)
// is (N2 - k2, 4095 - k1); in row 0: (0, 4096 - k1)) and w = W_N^k, N = 2 Nc:
//   U[k] = (z + conj(zm)) - i w (z - conj(zm))        = A+ of pair_q (fft_stockham.hpp), twice the real window's spectrum
// and A-[k] = conj(U[Nc - k]) for every k != 0, so U is all the pair step needs:  G = conj(Ua[k]) Ub[k],
// H = Ua[Nc-k] conj(Ub[Nc-k]).  Bin 0 pairs with itself and carries two real numbers: U[0] := (A+[0], A-[0]) = 2 (Re z + Im z,
// Re z - Im z), the DC and Nyquist terms.
// This pass is HBM-bound with more than half of its issue slots free; the pair step is not.
// Output: only the tiled layout k_pair_decimate16 streams (element (k2, k1) at [k1 / COLS][k2][k1 % COLS], COLS = 4096/N2);
// TZ keeps the column-pass output.
// grid (N2/2, n_sw), 512 threads (t >> 8: which row of the pair), dynamic LDS 2 x 34 KB.
// ---------------------------------------------------------------------------
// ROWMAJOR (4096 x 4096 plan, dec_stream.hpp): U goes back to the rows it came from, `tiled` = TZ -- a workgroup has both of
// its rows in registers before it stores, and no other workgroup touches them.
// The spectra leave the row pass with NON-TEMPORAL stores (round 4): they are 2.5 GB per cfg2 step, read again only by the pair
// step, long after they have left every cache; stored plainly they pushed the NEXT kernel's first reads out of L2 / the Infinity
// Cache on their way: cfg2 2.61-2.63 -> 2.54-2.58 ms per step (the pair step 0.657 -> 0.620 ms, the row pass itself 0.88-0.90 ->
// 0.86-0.90), cfg3 179.9 -> 177.1, cfg4 10.03-10.05 -> 9.96-10.00, same box.  (The column kernels' stores are a different case: non-temporal
// they ran 12 % slower, store_at above.)  TDOA_ROW_PLAIN_STORES rebuilds the old form.
#ifndef TDOA_ROW_PLAIN_STORES
#define TDOA_ROW_STORE(p, val) store_nt(p, val)
#else
#define TDOA_ROW_STORE(p, val) (*(p) = (val))
#endif
template <bool ROWMAJOR = false>
__global__ __launch_bounds__(512) void k_fwd_row4096_unpack(const float2 *TZ, FftPlan pl, float2 *tiled, bool pre_tw, int tile_cols)
{
    extern __shared__ float2 lds2[];                             // [2][kRowLds]
    const int a = blockIdx.x, g = threadIdx.x >> 8, j = threadIdx.x & 255;
    const int k2 = a == 0 ? (g ? pl.N2 >> 1 : 0) : (g ? pl.N2 - a : a);
    float2 *lds = lds2 + g * kRowLds;
    const float2 *row = TZ + (size_t)blockIdx.y * pl.Zs + (size_t)k2 * 4096 + (size_t)(k2 >> 8) * pl.zpad;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = load_nt(row + j + 256 * r);      // the column pass's output: read once
    if (pre_tw) row_pre_twiddle(v, k2, j, pl);
    fft16<false>(v);
    row4096_finish<false>(v, lds, j);                            // Y[k1 = j + 256 k] in v[oreg(k)]
    __syncthreads();                                             // everybody has read its last stage inputs
#pragma unroll
    for (int k = 0; k < 16; k++) lds[pad16(j + 256 * k)] = v[oreg(k)];
    __syncthreads();
    // the mirror elements: the other row of the pair at 4095 - k1; the self-mirrored rows (a = 0) pair inside themselves,
    // row 0 at 4096 - k1.  k1 = j + 256 k: pad16(4095 - j - 256 k) = pad16(4095 - j) - 272 k, one base per thread.
    // (row 0, j = 0, k = 0 reads one element past its image -- the other image's first -- and does not use it: bin 0)
    const bool row0 = a == 0 && g == 0;
    const float2 *mirror = lds2 + (a == 0 ? g : g ^ 1) * kRowLds + pad16(row0 ? 4096 - j : 4095 - j);
    // w = W_N^(k2 + N2 k1) = W_N^(k2 + N2 j) * W_32^k   (N2 * 256 / N = 1 / 32)
    // (k2 + N2 j < 2^20 N2/4096; pl.odd != 1: the denominator N = 2 Nc is 5 x 2^k -- unit_root_any with N/4 = Nc/2)
    const float2 wb = pl.odd == 1 ? unit_root((float)(k2 + pl.N2 * j), 1.0f / (float)pl.Nc, false)
                                  : unit_root_any((float)(k2 + pl.N2 * j), 0.5f * (float)pl.Nc, 2.0f / (float)pl.Nc, false);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const float2 z = v[oreg(k)];
        const float2 zm = mirror[-272 * k];
        constexpr float kPi = 3.14159265358979323846f;
        const float2 w = cmul(wb, make_float2(__builtin_cosf(kPi * (float)k / 16.0f), -__builtin_sinf(kPi * (float)k / 16.0f)));
        float2 u = unpack_u_pk(z, zm, w);
        if (k == 0 && row0 && j == 0) u = make_float2(2.0f * (z.x + z.y), 2.0f * (z.x - z.y));
        v[oreg(k)] = u;
    }
    if (ROWMAJOR) {
        float2 *out = tiled + (size_t)blockIdx.y * pl.Zs + (size_t)k2 * 4096 + (size_t)(k2 >> 8) * pl.zpad + j;
#pragma unroll
        for (int k = 0; k < 16; k++) TDOA_ROW_STORE(out + 256 * k, v[oreg(k)]);
    } else {
        // tiles of COLS columns x N2 rows: column k1 = j + 256 k -> tile k1 / COLS.  COLS = 4096 / N2 (4096 elements per tile:
        // k_pair_decimate16) or tile_cols = 64 (the staged column walk's blocks, dec_staged.hpp) -- a divisor of 256 either way,
        // so the sixteen elements of a thread are 256 / COLS tiles = 256 N2 elements apart
        const int cols = tile_cols ? tile_cols : 4096 / pl.N2;
        float2 *out = tiled + (size_t)blockIdx.y * pl.Nc + (size_t)(j / cols) * ((size_t)pl.N2 * cols) + (size_t)k2 * cols + (j % cols);
#pragma unroll
        for (int k = 0; k < 16; k++) TDOA_ROW_STORE(out + (size_t)256 * pl.N2 * k, v[oreg(k)]);
    }
}

// ---------------------------------------------------------------------------
// forward column pass, N2 = 256, 32 columns per workgroup: phase codes -> normalise -> pack ->
// two radix-16 stages down the columns -> twiddle -> T[k2][n1].
// grid (N1/32, n_sw), 512 threads (c = t & 31 column, j = t >> 5 item), dynamic LDS 64 KB
//
// SUB = true is the first half of a column pass of N2 = 256 G points (G = 16: 10 s windows, N = 2^25; G = 8),
// decimated in time by G: blockIdx.z = a transforms the rows n2 = a (mod G), applies W_(256G)^(a kb)
// and leaves Y_a[kb] in row a*256 + kb; k_fwd_col_finish<G> then runs the G-point transforms over a.
// ---------------------------------------------------------------------------
template <bool SUB>
__global__ __launch_bounds__(512) void k_fwd_col256_c16(const SWDesc *sw, const int *codes, long long code_stride,
                                                        const FmStats *stats, float2 *T, FftPlan pl)
{
    const int G = SUB ? pl.N2 >> 8 : 1, a = SUB ? (int)blockIdx.z : 0;
    extern __shared__ float2 lds[];   // [256][32]
    const int len = sw[blockIdx.y].len;
    const int *row = codes + (size_t)blockIdx.y * code_stride;
    const float mean = stats[blockIdx.y].mean, scale = stats[blockIdx.y].scale;
    const int c = threadIdx.x & 31, j = threadIdx.x >> 5;     // column, item (0..15)
    const int n1 = (blockIdx.x << 5) + c;
    const int N1 = pl.N1;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++)       // all 16 loads first (two codes = 8 bytes each, parked in the value's own registers)
        v[r] = code_park(code_fetch(row, (long long)(a + G * (j + 16 * r)) * N1 + n1, len));
#pragma unroll
    for (int r = 0; r < 16; r++)
        v[r] = code_convert(code_unpark(v[r]), (long long)(a + G * (j + 16 * r)) * N1 + n1, len, mean, scale);
    fft16<false>(v);
#pragma unroll
    for (int k = 0; k < 16; k++) lds[((16 * j + k) << 5) + c] = v[oreg(k)];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[((j + 16 * r) << 5) + c];
    mul_powers16(v, unit_root((float)j, 2.0f / 256.0f, false));
    fft16<false>(v);
    float2 *out = T + (size_t)blockIdx.y * pl.Zs;
    if (SUB) {
        // Y_a[kb = j + 16k] *= W_(256G)^(a kb) = W^(a j) * (W^(16 a))^k
        if (a) sub_twiddle16(v, a, j, pl);
#pragma unroll
        for (int k = 0; k < 16; k++) out[(size_t)(a * 256 + j + 16 * k) * N1 + (size_t)a * pl.zpad + n1] = v[oreg(k)];
    } else {
        // Y[k2 = j + 16k] *= W_Nc^(n1*k2) = W^(n1*j) * (W^(16*n1))^k
        const float inv2 = 2.0f / (float)pl.Nc;
        const long long e0 = ((long long)n1 * j) & (pl.Nc - 1);
        const long long e1 = ((long long)n1 * 16) & (pl.Nc - 1);
        mul_base_step16(v, unit_root((float)e0, inv2, false), unit_root((float)e1, inv2, false));
#pragma unroll
        for (int k = 0; k < 16; k++) out[(size_t)(j + 16 * k) * N1 + n1] = v[oreg(k)];
    }
}

// ---------------------------------------------------------------------------
// K1 fused into the forward column pass: the same transform as k_fwd_col256_c16, but the workgroup reads the CAPTURE
// BYTES and evaluates the discriminator itself (k1_discriminator.hpp) -- no code array is written or read (-8 B per
// sample of HBM traffic against materialised 24-bit codes, -4 B against the 16-bit codes of rounds 1-2), and the
// samples are the float discriminator to float32 resolution.  The window statistics come from the k_fm_demod<false>
// pre-pass.
//
// Element m of the packed window (samples 2m, 2m+1) needs the angle codes of samples 2m-1, 2m, 2m+1.  A thread loads
// the dword of its two samples and looks both up; the angle of sample 2m-1 is the second angle of the lane to its
// LEFT: one DPP move instead of a third load and lookup.  Only the first column of the tile has no left neighbour:
// those "boundary" samples of a wave are loaded and looked up by its first lanes in one go.
// LDS: the angle table (64 KB quadrant table) sits next to ONE float plane [256][64] (64 KB) through which the exchange
// between the two radix-16 stages goes twice, real parts first, then imaginary parts.  The workgroup is persistent --
// one per CU, tiles dealt by col_k1_order -- so the table is loaded once per workgroup, not once per tile.
// ---------------------------------------------------------------------------

// store at a 32-bit unsigned byte offset from a wave-uniform base (global_store v_off, v[data], s[base]): one offset
// register per store instead of a 64-bit address pair (a window's transform is at most 2^27 bytes)
__device__ __forceinline__ void store_at(float2 *base, unsigned int byte_off, float2 v)
{
    *reinterpret_cast<float2 *>(reinterpret_cast<char *>(base) + byte_off) = v;      // (non-temporal stores here: 12 % slower)
}

// One element from the angle codes (a0, a1) of the dword its thread fetched at sample index min(2m, len - 2) (never beyond
// the window) and the angle code `ap` of sample 2m - 1.  Returns the normalised pair; `a1` = angle of the element's LAST valid sample (what
// the lane to the right needs).  head: m may be 0 (code_0 := code_1).  len >= 2.
template <bool SCALED = false, bool ONCE = false>
__device__ __forceinline__ float2 k1_element_from(int a0, int a1, int ap, int i0, int len, float mean, float scale, bool head,
                                                  double *t1 = nullptr, double *t2 = nullptr)
{
    if (len & 1) {                       // wave-uniform: only then can a last element hold ONE sample (2m + 1 = len);
        if (i0 + 1 == len) {             // its dword was fetched one sample early: (2m - 1, 2m)
            ap = a0;
            a0 = a1;
        }
    }
    const int st1 = SCALED ? k1_stored_code_scaled(a1, a0) : k1_stored_code(a1, a0);
    const int st0 = head && i0 == 0 ? st1 : SCALED ? k1_stored_code_scaled(a0, ap) : k1_stored_code(a0, ap);
    // (ONCE: mean = m0, scale = s0 -- k1_normalise_fma(stored, -scale, -mean * scale), see there)
    const float v0 = ONCE ? k1_normalise_fma(st0, -scale, -mean * scale) : k1_normalise(st0, mean, scale);
    const float v1 = ONCE ? k1_normalise_fma(st1, -scale, -mean * scale) : k1_normalise(st1, mean, scale);
    if (ONCE) {                          // single-look K1: the window sums of exactly the samples that are transformed
        if (i0 < len) col_once_accumulate(st0, *t1, *t2);
        if (i0 + 1 < len) col_once_accumulate(st1, *t1, *t2);
    }
    return make_float2(i0 < len ? v0 : 0.0f, i0 + 1 < len ? v1 : 0.0f);
}

// The workgroup: ONE of 1024 threads per CU (four waves per SIMD), c = t & 63 column, j = t >> 6 item.  A wave is one item j
// with 64 adjacent elements of every row r: the left neighbour's angle comes by one whole-wave DPP shift and only lane 0
// has none -- the 16 boundary samples of a wave (one per r) are looked up by its lanes 0..15 and enter the shift as its
// `old` operand (v_readlane + v_mov: no permute, no select).
// Memory order inside a trip: the capture bytes of the NEXT tile are asked for right after the lookups of the current one
// (a whole transform ahead of their use), and the 16 stores of the PREVIOUS tile leave one by one between the lookups.
// The memory counter of a wave retires in order: with the stores at the end of a trip and the loads at the top of the next,
// the first lookup waited for every store before it to be acknowledged (and 256 store instructions left the CU at once).
// grid (n_cu), 1024 threads, dynamic LDS 64 KB (table) + 64 KB (plane [256][64]).
constexpr size_t kColK1Lds = kK1QuadrantBytes + sizeof(float) * 256 * 64;

// Which tile a workgroup takes as its seq-th: workgroups go to the XCDs round-robin (seq % 8, the grid is a multiple of 8).
// A window may start on any 2-byte boundary, in which case the 256-byte row pieces of adjacent column blocks share a cache
// line at either end (and the boundary sample always is the last one of the block before): dealt out in sequence,
// adjacent blocks would always meet in DIFFERENT L2s.  Instead XCD x takes the blocks [x nbx/8, (x + 1) nbx/8) of a row
// group, its 32 workgroups neighbouring blocks at the same time.  (Line-aligned windows -- cfg2 -- are indifferent to
// the order.)  nbx = 64 here; a bijection of [0, n_tiles) for any grid.
__device__ __forceinline__ int col_k1_order(int seq, int nbx)
{
    const int per = nbx >> 3, x = seq & 7, q = seq >> 3;
    return (q / per) * nbx + x * per + q % per;
}

struct ColK1Tile {        // what a trip needs to know about its tile (all wave-uniform)
    int bx, a, w, len;
    gptr16 p;
};

template <bool SUB>
__device__ __forceinline__ ColK1Tile col_k1_tile(const SWDesc *__restrict__ sw, int tile, int nbx, int G)
{
    ColK1Tile t;
    t.bx = tile % nbx;
    const int wa = tile / nbx;
    t.a = SUB ? wa % G : 0;
    t.w = SUB ? wa / G : wa;
    const SWDesc d = sw[t.w];
    t.len = d.len;
    t.p = k1_global(d.base);
    return t;
}

// the 16 dwords of a thread (rows j + 16 r of column n1) and the boundary sample its lane looks up (row rb of item jb)
template <int LOGW>
__device__ __forceinline__ void col_k1_fetch(const ColK1Tile &t, int G, int N1, int j, int c, int jb, int rb, unsigned int (&raw)[16],
                                             unsigned int &sb)
{
    const int n1 = (t.bx << LOGW) + c, last = t.len - 2;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i0 = 2 * ((t.a + G * (j + 16 * r)) * N1 + n1);        // sample index 2 (row N1 + n1) < 2^25 (Nc <= 2^24)
        raw[r] = k1_fetch2(t.p, i0 < last ? i0 : last);
    }
    const int ib = 2 * ((t.a + G * (jb + 16 * rb)) * N1 + (t.bx << LOGW)) - 1;      // the sample before that row of the tile,
    sb = t.p[ib >= 0 && ib < t.len ? ib : 0];                                        // if the window has one
}

// rows of a tile whose table reads are in flight together (2 reads per row): 4 -> 8 reads; TDOA_COL_BATCH=8 for an A/B
#ifndef TDOA_COL_BATCH
#define TDOA_COL_BATCH 4
#endif
constexpr int kColBatch = TDOA_COL_BATCH;
// row kColBatch g + Q of a tile through `f` (g is a loop variable of a fully unrolled loop: the row index has to be a
// compile-time constant for f, so every g is spelled out)
template <typename F, int Q>
__device__ __forceinline__ void batch_rows_general(F &f, int g, std::integral_constant<int, Q>)
{
    if (g == 0) f(std::integral_constant<int, Q>{});
    if constexpr (kColBatch < 16) if (g == 1) f(std::integral_constant<int, kColBatch + Q>{});
    if constexpr (kColBatch < 8) if (g == 2) f(std::integral_constant<int, 2 * kColBatch + Q>{});
    if constexpr (kColBatch < 8) if (g == 3) f(std::integral_constant<int, 3 * kColBatch + Q>{});
    if constexpr (kColBatch < 4) {
        if (g == 4) f(std::integral_constant<int, 4 * kColBatch + Q>{});
        if (g == 5) f(std::integral_constant<int, 5 * kColBatch + Q>{});
        if (g == 6) f(std::integral_constant<int, 6 * kColBatch + Q>{});
        if (g == 7) f(std::integral_constant<int, 7 * kColBatch + Q>{});
    }
}
static_assert(kColBatch == 2 || kColBatch == 4 || kColBatch == 8 || kColBatch == 16, "rows per batch");

template <typename F, int... Q>
__device__ __forceinline__ void batch_general(F &f, int g, std::integer_sequence<int, Q...>)
{
    (void)std::initializer_list<int>{(batch_rows_general(f, g, std::integral_constant<int, Q>{}), 0)...};
}

// ONCE (k1_single_look.hpp): `stats` holds the estimates (m0, s0) of k_once_estimate, and every wave leaves the exact sums
// of its stored codes of a tile, one record per row of 16 lanes, in once_tiles[(16 tile + wave) 4 + row].
// (Round 4 also built this kernel with the whole 128 KB half-plane table in LDS -- 5.5 instructions per look-up instead of
// 13 -- next to an exchange plane of half the tile: bit-identical results, 2 % SLOWER on cfg2 / cfg4 / cfg3, because the
// half plane doubles the exchange's barriers (8 per tile) and LDS instructions.  Commit f53a7d8)
template <bool SUB, bool ONCE = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fwd_col256_k1(const SWDesc *__restrict__ sw, const int *__restrict__ table, const FmStats *__restrict__ stats, float2 *__restrict__ T,
                                                       FftPlan pl, int n_sw, OnceTile *__restrict__ once_tiles)
{
    constexpr int LOGW = 6, W = 1 << LOGW;                      // 64 columns per tile
    constexpr int kTableEntries = kK1QuadrantEntries;
    extern __shared__ int lds_k1[];                             // the table at offset 0 (the offset IS the address), then the plane
    int *lut = lds_k1;
    float *plane = reinterpret_cast<float *>(lds_k1 + kTableEntries);           // [256][W]
    k1_assert_lds0(lut);
    const int G = SUB ? pl.N2 >> 8 : 1;
    const int N1 = pl.N1, nbx = N1 >> LOGW;
    const int n_tiles = n_sw * G * nbx;
    // the first tile's bytes, ahead of the table load
    unsigned int raw_next[16], sb_next = 0;
    if ((int)blockIdx.x < n_tiles) {
        const int tid = threadIdx.x, lane = tid & 63, j = tid >> LOGW;
        col_k1_fetch<LOGW>(col_k1_tile<SUB>(sw, col_k1_order(blockIdx.x, nbx), nbx, G), G, N1, j, tid & (W - 1),
                           j, lane & 15, raw_next, sb_next);
    }
    for (int k = threadIdx.x; k < kTableEntries / 4; k += blockDim.x)
        reinterpret_cast<int4 *>(lut)[k] = reinterpret_cast<const int4 *>(table)[k];
    __syncthreads();
    const float2 wj = unit_root((float)(threadIdx.x >> LOGW), 2.0f / 256.0f, false);      // W_256^j: the thread's item never changes
    float2 v[16];                                                // the previous tile's outputs until they are stored (below)
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = make_float2(0.0f, 0.0f);
    int prev = -1;
    for (int seq = blockIdx.x; seq < n_tiles; seq += gridDim.x) {
        const int tile = col_k1_order(seq, nbx);
        // (laundered per tile: the row / plane / output addresses derived from the thread index are rebuilt inside the
        // trip -- hoisted out of this loop as ~80 invariants they spilled 112 VGPRs)
        const int tid = opaque_i((int)threadIdx.x);
        const int c = tid & (W - 1), j = tid >> LOGW;          // column, item (0..15)
        const ColK1Tile t = col_k1_tile<SUB>(sw, tile, nbx, G);
        const int a = t.a, w = t.w, len = t.len;
        // (the quadrant table's angle codes are scaled by 256 -- so are the mean and, inversely, the scale: exact)
        const float mean = stats[w].mean * 256.0f, scale = stats[w].scale * 0.00390625f;
        const float nscale = -scale, noff = -mean * scale;         // (ONCE: both products are exact)
        // where the previous tile's outputs go: output k of thread j is row (a 256 +) j + 16 k, column n1; it sits in v[oreg(k)]
        const ColK1Tile tp = col_k1_tile<SUB>(sw, prev >= 0 ? prev : tile, nbx, G);
        float2 *outp = T + (size_t)tp.w * pl.Zs + (SUB ? (size_t)tp.a * ((size_t)256 * N1 + pl.zpad) : 0);
        const unsigned int offp = 8u * (unsigned)(j * N1 + (tp.bx << LOGW) + c);
        {
            unsigned int raw[16];
#pragma unroll
            for (int r = 0; r < 16; r++) raw[r] = raw_next[r];
            // boundary samples: lane L took row r = L & 15 of the wave's item
            const int ab = k1_angle_quadrant<false, true>(k1_index_bytes(sb_next), ~sb_next, lut);
            // rows are classified per wave (it holds item j of every r): entirely inside the window -- no bounds selects,
            // the common case --, entirely beyond it -- zero padding, nothing to look up (40 % of the rows of a 10 s window
            // in N = 2^25) --, or general
            const int jw = __builtin_amdgcn_readfirstlane(j);
            double t1 = 0.0, t2 = 0.0;                            // ONCE: exact sums of this thread's stored codes
            // one row r of the wave: the general form (any position of the row against the window)
            auto row_general = [&](auto r_c) {
                constexpr int r = decltype(r_c)::value;
                const int i_first = 2 * ((a + G * (jw + 16 * r)) * N1 + (t.bx << LOGW));
                const int i_end = i_first + 2 * W;                                                   // one past the wave's last sample of this r
                if (__builtin_expect(i_first >= len, 0)) {
                    v[r] = make_float2(0.0f, 0.0f);
                    return;
                }
                int a0, a1;
                k1_angle2_quadrant<true>(raw[r], lut, a0, a1);
                // the angle of sample 2m - 1 is the left lane's second angle; lane 0 keeps `old` = the boundary sample's
                const int bnd = __builtin_amdgcn_readlane(ab, r);
                const int ap = __builtin_amdgcn_update_dpp(bnd, a1, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                if (i_first > 0 && i_end <= len) {
                    const int st0 = k1_stored_code_scaled(a0, ap), st1 = k1_stored_code_scaled(a1, a0);
                    if (ONCE) {
                        col_once_accumulate(st0, t1, t2);
                        col_once_accumulate(st1, t1, t2);
                    }
                    v[r] = ONCE ? make_float2(k1_normalise_fma(st0, nscale, noff), k1_normalise_fma(st1, nscale, noff))
                                : make_float2(k1_normalise(st0, mean, scale), k1_normalise(st1, mean, scale));
                } else {
                    v[r] = k1_element_from<true, ONCE>(a0, a1, ap, i_first + 2 * c, len, mean, scale, r == 0, &t1, &t2);
                }
            };
            // Rows are taken FOUR at a time.  The previous tile's outputs in v[4g .. 4g+3] leave just before the registers are
            // needed again: the 16 stores of a thread go out a few per ~150 instructions of discriminator work instead of as a
            // burst of 256 store instructions from the sixteen waves of the CU at the end of a trip.  Unconditional: on a
            // workgroup's first trip v[] is zero and goes to the place of THIS tile, which the same thread overwrites with the
            // real values one trip later (one wave's stores to an address stay in order).
            // When the four rows lie inside the window (all but the window's first row and its tail) their EIGHT table reads
            // are issued back to back and only then placed: one read followed by its placement (the form above) waits out the
            // LDS latency for every sample while the CU's sixteen waves all queue at the LDS at the same time.
#pragma unroll
            for (int g = 0; g < 16 / kColBatch; g++) {
#pragma unroll
                for (int q = 0; q < kColBatch; q++) store_at(outp + (size_t)(16 * oreg(kColBatch * g + q)) * N1, offp, v[kColBatch * g + q]);
                const int i_first0 = 2 * ((a + G * (jw + 16 * (kColBatch * g))) * N1 + (t.bx << LOGW));
                const int i_end3 = 2 * ((a + G * (jw + 16 * (kColBatch * g + kColBatch - 1))) * N1 + (t.bx << LOGW)) + 2 * W;
                if (__builtin_expect(i_first0 > 0 && i_end3 <= len, 1)) {
                    unsigned int neg[kColBatch];
                    int c0[kColBatch], c1[kColBatch];
#pragma unroll
                    for (int q = 0; q < kColBatch; q++) {
                        const unsigned int x = k1_index_bytes(raw[kColBatch * g + q]);
                        neg[q] = ~raw[kColBatch * g + q];
                        c0[q] = k1_table_read<true>(lut, k1_quadrant_offset<false>(x));
                        c1[q] = k1_table_read<true>(lut, k1_quadrant_offset<true>(x));
                    }
                    __builtin_amdgcn_sched_barrier(0);                  // (the eight reads stay ahead of their placements)
#pragma unroll
                    for (int q = 0; q < kColBatch; q++) {
                        const int r = kColBatch * g + q;
                        const int a0 = k1_quadrant_place<false>(c0[q], neg[q]), a1 = k1_quadrant_place<true>(c1[q], neg[q]);
                        const int bnd = __builtin_amdgcn_readlane(ab, r);
                        const int ap = __builtin_amdgcn_update_dpp(bnd, a1, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                        const int st0 = k1_stored_code_scaled(a0, ap), st1 = k1_stored_code_scaled(a1, a0);
                        if (ONCE) {
                            col_once_accumulate(st0, t1, t2);
                            col_once_accumulate(st1, t1, t2);
                        }
                        v[r] = ONCE ? make_float2(k1_normalise_fma(st0, nscale, noff), k1_normalise_fma(st1, nscale, noff))
                                    : make_float2(k1_normalise(st0, mean, scale), k1_normalise(st1, mean, scale));
                    }
                } else {
                    batch_general(row_general, g, std::make_integer_sequence<int, kColBatch>{});      // the batch's rows, one by one
                }
            }
            if (ONCE) col_once_wave_record(t1, t2, once_tiles + (size_t)tile * kOnceWavesPerTile + (size_t)(tid >> 6) * kOnceRecordsPerWave);
        }
        // the capture bytes of the next tile: asked for now, used a whole transform later
        __builtin_amdgcn_sched_barrier(0);
        if (seq + (int)gridDim.x < n_tiles) {
            const int tid2 = opaque_i((int)threadIdx.x), lane2 = tid2 & 63, j2 = tid2 >> LOGW;
            col_k1_fetch<LOGW>(col_k1_tile<SUB>(sw, col_k1_order(seq + gridDim.x, nbx), nbx, G), G, N1, j2, tid2 & (W - 1),
                               j2, lane2 & 15, raw_next, sb_next);
        }
        __builtin_amdgcn_sched_barrier(0);
        fft16<false>(v);
        // exchange through one float plane: real parts, then imaginary parts
#pragma unroll
        for (int k = 0; k < 16; k++) plane[((16 * j + k) << LOGW) + c] = v[oreg(k)].x;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) v[r].x = plane[((j + 16 * r) << LOGW) + c];      // .y still holds stage-1 outputs (oreg order)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) plane[((16 * j + k) << LOGW) + c] = v[oreg(k)].y;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) v[r].y = plane[((j + 16 * r) << LOGW) + c];
        __syncthreads();                                         // the next tile writes the plane again
        mul_powers16(v, opaque(wj));                             // (laundered: its 15 powers are rebuilt per tile, not kept)
        fft16<false>(v);
        if (SUB) {
            // Y_a[kb = j + 16k] *= W_(256G)^(a kb) = W^(a j) * (W^(16 a))^k
            if (a) sub_twiddle16(v, a, j, pl);
        }
        // (one-sweep plans: the four-step twiddle W_Nc^(n1 k2) is applied by the row pass to its inputs, row_pre_twiddle)
        prev = tile;
    }
    if (prev >= 0) {                                             // the last tile of this workgroup
        const int tid = opaque_i((int)threadIdx.x);
        const ColK1Tile tp = col_k1_tile<SUB>(sw, prev, nbx, G);
        float2 *outp = T + (size_t)tp.w * pl.Zs + (SUB ? (size_t)tp.a * ((size_t)256 * N1 + pl.zpad) : 0);
        const unsigned int offp = 8u * (unsigned)((tid >> LOGW) * N1 + (tp.bx << LOGW) + (tid & (W - 1)));
#pragma unroll
        for (int r = 0; r < 16; r++) store_at(outp + (size_t)(16 * oreg(r)) * N1, offp, v[r]);
    }
}

// 5-point DFT, forward (e^{-2 pi i n k / 5}), natural order in and out: 4 real factors, the classic sums and differences
__device__ __forceinline__ void dft5(float2 &x0, float2 &x1, float2 &x2, float2 &x3, float2 &x4)
{
    constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;      // cos(2 pi/5), cos(4 pi/5)
    constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;       // sin(2 pi/5), sin(4 pi/5)
    const float2 t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
    const cplx_v m1 = cv(x0) + c1 * cv(t1) + c2 * cv(t2), m2 = cv(x0) + c2 * cv(t1) + c1 * cv(t2);
    const cplx_v r1 = s1 * cv(t3) + s2 * cv(t4), r2 = s2 * cv(t3) - s1 * cv(t4);
    x0 = cf(cv(x0) + cv(t1) + cv(t2));
    // X1 = m1 - i r1, X4 = m1 + i r1, X2 = m2 - i r2, X3 = m2 + i r2   (-i (a, b) = (b, -a))
    x1 = make_float2(m1.x + r1.y, m1.y - r1.x);
    x4 = make_float2(m1.x - r1.y, m1.y + r1.x);
    x2 = make_float2(m2.x + r2.y, m2.y - r2.x);
    x3 = make_float2(m2.x - r2.y, m2.y + r2.x);
}

// 10-point DFT, forward, natural order in and out: two 5-point DFTs over the even and the odd inputs, then
// X[k] = E[k] + W_10^k O[k], X[k + 5] = E[k] - W_10^k O[k]
__device__ __forceinline__ void dft10(float2 (&v)[10])
{
    dft5(v[0], v[2], v[4], v[6], v[8]);
    dft5(v[1], v[3], v[5], v[7], v[9]);
    constexpr float wr[5] = {1.0f, 0.80901699437494742f, 0.30901699437494742f, -0.30901699437494742f, -0.80901699437494742f};
    constexpr float wi[5] = {0.0f, -0.58778525229247313f, -0.95105651629515357f, -0.95105651629515357f, -0.58778525229247313f};
    float2 x[10];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const float2 e = v[2 * k], o = k ? cmul(v[2 * k + 1], make_float2(wr[k], wi[k])) : v[1];
        x[k] = cadd(e, o);
        x[k + 5] = csub(e, o);
    }
#pragma unroll
    for (int k = 0; k < 10; k++) v[k] = x[k];
}

// 12-point DFT, forward, natural order in and out: n = 3 a + b, k = c + 4 d -- four-point transforms over a for each b,
// twiddle W_12^(b c), three-point transforms over b for each c
__device__ __forceinline__ void dft12(float2 (&v)[12])
{
    float2 y[3][4];
#pragma unroll
    for (int b = 0; b < 3; b++) {
        y[b][0] = v[b]; y[b][1] = v[3 + b]; y[b][2] = v[6 + b]; y[b][3] = v[9 + b];
        bfly4<false>(y[b][0], y[b][1], y[b][2], y[b][3]);            // Y_b[c], c = 0..3
    }
    // W_12^m = exp(-2 pi i m / 12): m = 1: (sqrt3/2, -1/2)  2: (1/2, -sqrt3/2)  3: (0, -1)  4: (-1/2, -sqrt3/2)  6: (-1, 0)
    constexpr float h = 0.5f, r = 0.86602540378443865f;
    y[1][1] = cmul(y[1][1], make_float2(r, -h));                     // b c = 1
    y[1][2] = cmul(y[1][2], make_float2(h, -r));                     // 2
    y[1][3] = make_float2(y[1][3].y, -y[1][3].x);                    // 3: times -i
    y[2][1] = cmul(y[2][1], make_float2(h, -r));                     // 2
    y[2][2] = cmul(y[2][2], make_float2(-h, -r));                    // 4
    y[2][3] = make_float2(-y[2][3].x, -y[2][3].y);                   // 6: times -1
#pragma unroll
    for (int c = 0; c < 4; c++) {
        // three-point DFT over b: X[c + 4 d], d = 0, 1, 2;  w = exp(-2 pi i / 3) = (-1/2, -sqrt3/2)
        const float2 x0 = y[0][c], t = cadd(y[1][c], y[2][c]), u = csub(y[1][c], y[2][c]);
        const float2 m = make_float2(x0.x - h * t.x, x0.y - h * t.y), su = make_float2(r * u.x, r * u.y);
        v[c] = cadd(x0, t);
        v[c + 4] = make_float2(m.x + su.y, m.y - su.x);              // m - i s u
        v[c + 8] = make_float2(m.x - su.y, m.y + su.x);              // m + i s u
    }
}

// second half, G = 16 (N2 = 4096), 8 (N2 = 2048), 10 (N2 = 2560) or 12 (N2 = 3072; both round 5): X[kb + 256 ka] = sum_a W_G^(a ka) Y_a[kb] in
// registers (G = 8 runs the 16-point butterfly on inputs spread to the even slots: W_16^(2a k) = W_8^(a k); G = 10 a
// 10-point DFT as 2 x 5), then the four-step twiddle W_Nc^(n1 k2) = W^(n1 kb) * (W^(256 n1))^ka; in place (a thread rewrites
// the rows it read).
// grid (N1/512, 256, n_sw), 256 threads = 512 adjacent columns.
template <int G>
__global__ __launch_bounds__(256) void k_fwd_col_finish(float2 *T, FftPlan pl)
{
    static_assert(G == 8 || G == 16 || G == 10 || G == 12, "two-sweep column pass: N2 = 2048, 2560, 3072 or 4096");
    // a thread takes TWO adjacent columns (one 16-byte access per row): a workgroup moves 4 KB runs of each of its 16 rows
    const int n1 = ((blockIdx.x << 8) + threadIdx.x) * 2, kb = blockIdx.y;
    float2 *base = T + (size_t)blockIdx.z * pl.Zs + (size_t)kb * pl.N1 + n1;
    const size_t stride = (size_t)256 * pl.N1 + pl.zpad;      // (zpad = 0: 27 % slower -- every row of the sum on one channel)
    typedef float f4v __attribute__((ext_vector_type(4)));
    if constexpr (G == 10 || G == 12) {
        float2 v[G], u[G];
#pragma unroll
        for (int a = 0; a < G; a++) {
            const f4v x = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(base + a * stride));      // read once
            v[a] = make_float2(x.x, x.y);
            u[a] = make_float2(x.z, x.w);
        }
        if constexpr (G == 10) {
            dft10(v);
            dft10(u);
        } else {
            dft12(v);
            dft12(u);
        }
        // W_Nc^(n1 (kb + 256 ka)), Nc = 4096 x 2560 or x 3072: n1 kb, 256 n1 < 2^20 -- no wrap; the denominator is 5 x 2^21 / 3 x 2^22
        const float qd = 0.25f * (float)pl.Nc, iq = 4.0f / (float)pl.Nc;
        // base * step^ka, ka = 0..G-1, by products of depth <= 5 (like mul_base_step16): step^2, ^4, ^8 by squaring
        auto twiddle10 = [&](float2 (&x)[G], int col) {
            const float2 g = unit_root_any((float)(col * kb), qd, iq, false), s1 = unit_root_any((float)(col * 256), qd, iq, false);
            const float2 s2 = cmul(s1, s1), s4 = cmul(s2, s2), s8 = cmul(s4, s4), g4 = cmul(g, s4), g8 = cmul(g, s8);
            const float2 s3 = cmul(s2, s1);
            x[0] = cmul(x[0], g);
            x[1] = cmul(x[1], cmul(g, s1));
            x[2] = cmul(x[2], cmul(g, s2));
            x[3] = cmul(x[3], cmul(g, s3));
            x[4] = cmul(x[4], g4);
            x[5] = cmul(x[5], cmul(g4, s1));
            x[6] = cmul(x[6], cmul(g4, s2));
            x[7] = cmul(x[7], cmul(g4, s3));
            x[8] = cmul(x[8], g8);
            x[9] = cmul(x[9], cmul(g8, s1));
            if constexpr (G == 12) {
                x[10] = cmul(x[10], cmul(g8, s2));
                x[11] = cmul(x[11], cmul(g8, s3));
            }
        };
        twiddle10(v, n1);
        twiddle10(u, n1 + 1);
#pragma unroll
        for (int ka = 0; ka < G; ka++) {
            f4v y;
            y.x = v[ka].x; y.y = v[ka].y; y.z = u[ka].x; y.w = u[ka].y;
            __builtin_nontemporal_store(y, reinterpret_cast<f4v *>(base + ka * stride));
        }
    } else {
        float2 v[16], u[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = u[r] = make_float2(0.0f, 0.0f);
#pragma unroll
        for (int a = 0; a < G; a++) {
            const f4v x = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(base + a * stride));      // read once
            v[(16 / G) * a] = make_float2(x.x, x.y);
            u[(16 / G) * a] = make_float2(x.z, x.w);
        }
        fft16<false>(v);
        fft16<false>(u);
        const float inv2 = 2.0f / (float)pl.Nc;
        {
            const long long e0 = ((long long)n1 * kb) & (pl.Nc - 1);
            const long long e1 = ((long long)n1 * 256) & (pl.Nc - 1);
            mul_base_step16(v, unit_root((float)e0, inv2, false), unit_root((float)e1, inv2, false));
        }
        {
            const long long e0 = ((long long)(n1 + 1) * kb) & (pl.Nc - 1);
            const long long e1 = ((long long)(n1 + 1) * 256) & (pl.Nc - 1);
            mul_base_step16(u, unit_root((float)e0, inv2, false), unit_root((float)e1, inv2, false));
        }
#pragma unroll
        for (int ka = 0; ka < G; ka++) {
            f4v y;
            y.x = v[oreg(ka)].x; y.y = v[oreg(ka)].y; y.z = u[oreg(ka)].x; y.w = u[oreg(ka)].y;
            __builtin_nontemporal_store(y, reinterpret_cast<f4v *>(base + ka * stride));
        }
    }
}

// ---------------------------------------------------------------------------
// forward column pass, N2 = 256 F (F = 2: 1 s windows at 4 Msps, N = 2^22; F = 4: N = 2^23).
// Decimation in time by F: thread group `par` runs the 256-point transform of the rows
// n2 = par (mod F) exactly as above in its own LDS image; a last radix-F butterfly across the
// images, X[k + 256 q] = sum_p W_F^(qp) W_(256F)^(kp) Y_p[k], is done by group q.
// 32/F columns per workgroup: grid (N1 F/32, n_sw), 512 threads, dynamic LDS 64 KB.
// ---------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(512) void k_fwd_colx_c16(const SWDesc *sw, const int *codes, long long code_stride,
                                                      const FmStats *stats, float2 *T, FftPlan pl)
{
    static_assert(F == 2 || F == 4, "last stage is radix 2 or 4");
    constexpr int C = 32 / F, LOGC = F == 2 ? 4 : 3;
    extern __shared__ float2 lds[];   // [F][256][C]
    const int len = sw[blockIdx.y].len;
    const int *row = codes + (size_t)blockIdx.y * code_stride;
    const float mean = stats[blockIdx.y].mean, scale = stats[blockIdx.y].scale;
    const int c = threadIdx.x & (C - 1), j = (threadIdx.x >> LOGC) & 15, par = threadIdx.x >> (LOGC + 4);
    const int n1 = blockIdx.x * C + c;
    const int N1 = pl.N1;
    float2 *img = lds + par * 256 * C;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = code_park(code_fetch(row, (long long)(F * (j + 16 * r) + par) * N1 + n1, len));
#pragma unroll
    for (int r = 0; r < 16; r++)
        v[r] = code_convert(code_unpark(v[r]), (long long)(F * (j + 16 * r) + par) * N1 + n1, len, mean, scale);
    fft16<false>(v);
#pragma unroll
    for (int k = 0; k < 16; k++) img[(16 * j + k) * C + c] = v[oreg(k)];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = img[(j + 16 * r) * C + c];
    mul_powers16(v, unit_root((float)j, 2.0f / 256.0f, false));
    fft16<false>(v);
    // Y_par[k = j + 16 kk] *= W_(256F)^(k par)
    if (par)      // uniform per wave
        mul_base_step16(v, unit_root((float)(j * par), 2.0f / (256.0f * F), false),
                        unit_root((float)(16 * par), 2.0f / (256.0f * F), false));
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) img[(j + 16 * k) * C + c] = v[oreg(k)];
    __syncthreads();
    const int q = par;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int idx = (j + 16 * k) * C + c;
        const float2 a0 = lds[idx], a1 = lds[256 * C + idx];
        if (F == 2) {
            v[oreg(k)] = q ? csub(a0, a1) : cadd(a0, a1);
        } else {
            const float2 a2 = lds[2 * 256 * C + idx], a3 = lds[3 * 256 * C + idx];
            const float2 s02 = (q & 1) ? csub(a0, a2) : cadd(a0, a2);
            const float2 e13 = (q & 1) ? csub(a1, a3) : cadd(a1, a3);
            // W_4^q = (-i)^q on the odd pair: q = 0, 2: +-(a1 + a3);  q = 1: -i (a1 - a3);  q = 3: +i (a1 - a3)
            const float2 o = (q & 1) ? make_float2(e13.y, -e13.x) : e13;
            v[oreg(k)] = (q & 2) ? csub(s02, o) : cadd(s02, o);
        }
    }
    // X[k2 = j + 16 k + 256 q] *= W_Nc^(n1 k2) = W^(n1 (j + 256 q)) * (W^(16 n1))^k
    float2 *out = T + (size_t)blockIdx.y * pl.Nc;
    const float inv2 = 2.0f / (float)pl.Nc;
    const long long e0 = ((long long)n1 * (j + 256 * q)) & (pl.Nc - 1);
    const long long e1 = ((long long)n1 * 16) & (pl.Nc - 1);
    mul_base_step16(v, unit_root((float)e0, inv2, false), unit_root((float)e1, inv2, false));
#pragma unroll
    for (int k = 0; k < 16; k++) out[(size_t)(j + 16 * k + 256 * q) * N1 + n1] = v[oreg(k)];
}

// ---------------------------------------------------------------------------
// K1 fused into the forward column pass for N2 = 512 (1 s windows at 4 Msps, N = 2^22: BASELINE config 5): k_fwd_colx_c16<2>
// with the capture bytes as its input, built like k_fwd_col256_k1 -- one persistent 1024-thread workgroup per CU, the
// quadrant table of scaled angle codes at LDS address 0, next tile's bytes prefetched, previous tile's stores spread over
// the lookups, four-step twiddle left to the row pass.  32 columns per workgroup; thread (c = t & 31, j = (t >> 5) & 15,
// par = t >> 9) transforms the rows n2 = 2 (j + 16 r) + par; a wave holds two consecutive items j of one parity (one per
// half-wave), so the left-lane angle sharing runs inside a half-wave and the 32 boundary samples of a wave (16 rows x 2
// items) are looked up by its lanes 0..31 and handed out by a lane permute.
// grid (n_cu), 1024 threads, dynamic LDS 64 KB (table) + 64 KB (planes [2][256][32]).
// ---------------------------------------------------------------------------
constexpr size_t kCol512Lds = kK1QuadrantBytes + sizeof(float) * 2 * 256 * 32;

struct Col512Tile {
    int bx, w, len;
    gptr16 p;
};

__device__ __forceinline__ Col512Tile col512_tile(const SWDesc *__restrict__ sw, int tile, int nbx)
{
    Col512Tile t;
    t.bx = tile % nbx;
    t.w = tile / nbx;
    const SWDesc d = sw[t.w];
    t.len = d.len;
    t.p = k1_global(d.base);
    return t;
}

__device__ __forceinline__ void col512_fetch(const Col512Tile &t, int N1, int tid, unsigned int (&raw)[16], unsigned int &sb)
{
    const int lane = tid & 63, c = tid & 31, j = (tid >> 5) & 15, par = tid >> 9;
    const int n1 = t.bx * 32 + c, last = t.len - 2;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i0 = 2 * ((2 * (j + 16 * r) + par) * N1 + n1);
        raw[r] = k1_fetch2(t.p, i0 < last ? i0 : last);
    }
    // boundary samples (first column of the tile): lane L < 32 takes row r = L >> 1 of the wave's item (j & ~1) + (L & 1)
    const int jb = (j & ~1) + (lane & 1), rb = (lane & 31) >> 1;
    const int ib = 2 * ((2 * (jb + 16 * rb) + par) * N1 + t.bx * 32) - 1;
    sb = t.p[ib >= 0 && ib < t.len ? ib : 0];
}

template <bool ONCE = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fwd_col512_k1(const SWDesc *__restrict__ sw, const int *__restrict__ qtable, const FmStats *__restrict__ stats,
                                                       float2 *__restrict__ T, FftPlan pl, int n_sw, OnceTile *__restrict__ once_tiles)
{
    constexpr int F = 2, C = 32, LOGC = 5;
    extern __shared__ int lds_k1[];                             // the table at offset 0 (the offset IS the address), then the planes
    int *lut = lds_k1;
    float *plane = reinterpret_cast<float *>(lds_k1 + kK1QuadrantEntries);      // [F][256][C]
    k1_assert_lds0(lut);
    const int N1 = pl.N1, nbx = N1 / C;
    const int n_tiles = n_sw * nbx;
    unsigned int raw_next[16], sb_next = 0;
    if ((int)blockIdx.x < n_tiles) col512_fetch(col512_tile(sw, col_k1_order(blockIdx.x, nbx), nbx), N1, threadIdx.x, raw_next, sb_next);
    for (int k = threadIdx.x; k < kK1QuadrantEntries; k += blockDim.x) lut[k] = qtable[k];
    __builtin_amdgcn_s_waitcnt(0x0f70);                          // vmcnt(0): the loop is entered with nothing pending
    __syncthreads();
    // thread constants: the item's stage twiddle W_256^j and the parity twiddle W_512^j (odd rows only)
    const float2 wj = unit_root((float)((threadIdx.x >> LOGC) & 15), 2.0f / 256.0f, false);
    const float2 wp = unit_root((float)((threadIdx.x >> LOGC) & 15), 2.0f / (256.0f * F), false);
    float2 v[16];                                                // the previous tile's outputs until they are stored
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = make_float2(0.0f, 0.0f);
    int prev = -1;
    for (int seq = blockIdx.x; seq < n_tiles; seq += gridDim.x) {
        const int tile = col_k1_order(seq, nbx);
        const int tid = opaque_i((int)threadIdx.x);              // (laundered per tile: see k_fwd_col256_k1)
        const int lane = tid & 63;
        const int c = tid & (C - 1), j = (tid >> LOGC) & 15, par = tid >> (LOGC + 4);
        const Col512Tile t = col512_tile(sw, tile, nbx);
        const int w = t.w, len = t.len;
        const float mean = stats[w].mean * 256.0f, scale = stats[w].scale * 0.00390625f;      // scaled codes: k_fwd_col256_k1
        const float nscale = -scale, noff = -mean * scale;
        float *img = plane + par * 256 * C;
        // where the previous tile's outputs go: output k of thread (j, par) is row j + 16 k + 256 par, column n1, in v[oreg(k)]
        const Col512Tile tp = col512_tile(sw, prev >= 0 ? prev : tile, nbx);
        float2 *outp = T + (size_t)tp.w * pl.Zs;
        const unsigned int offp = 8u * (unsigned)((j + 256 * par) * N1 + tp.bx * C + c);
        {
            unsigned int raw[16];
#pragma unroll
            for (int r = 0; r < 16; r++) raw[r] = raw_next[r];
            const int ab = k1_angle_quadrant<false, true>(k1_index_bytes(sb_next), ~sb_next, lut);
            const int jw = __builtin_amdgcn_readfirstlane(j & ~1), pw = __builtin_amdgcn_readfirstlane(par);
            double t1 = 0.0, t2 = 0.0;                            // ONCE: exact sums of this thread's stored codes
            auto row_general = [&](auto r_c) {
                constexpr int r = decltype(r_c)::value;
                const int i_first = 2 * ((F * (jw + 16 * r) + pw) * N1 + t.bx * C);
                const int i_end = 2 * ((F * (jw + 1 + 16 * r) + pw) * N1 + t.bx * C + C);      // one past the wave's last sample of this r
                if (__builtin_expect(i_first >= len, 0)) {
                    v[r] = make_float2(0.0f, 0.0f);
                    return;
                }
                int a0, a1;
                k1_angle2_quadrant<true>(raw[r], lut, a0, a1);
                const int left = wave_shift_right1(a1);
                const int bnd = __shfl(ab, 2 * r + (lane >> LOGC), kWave);
                const int ap = c ? left : bnd;
                if (i_first > 0 && i_end <= len) {
                    const int st0 = k1_stored_code_scaled(a0, ap), st1 = k1_stored_code_scaled(a1, a0);
                    if (ONCE) {
                        col_once_accumulate(st0, t1, t2);
                        col_once_accumulate(st1, t1, t2);
                    }
                    v[r] = ONCE ? make_float2(k1_normalise_fma(st0, nscale, noff), k1_normalise_fma(st1, nscale, noff))
                                : make_float2(k1_normalise(st0, mean, scale), k1_normalise(st1, mean, scale));
                } else {
                    v[r] = k1_element_from<true, ONCE>(a0, a1, ap, i_first + 2 * (F * (j - jw) * N1 + c), len, mean, scale, r == 0,
                                                       &t1, &t2);
                }
            };
            // rows four at a time, their eight table reads in flight together (see k_fwd_col256_k1)
#pragma unroll
            for (int g = 0; g < 16 / kColBatch; g++) {
#pragma unroll
                for (int q = 0; q < kColBatch; q++) store_at(outp + (size_t)(16 * oreg(kColBatch * g + q)) * N1, offp, v[kColBatch * g + q]);      // (unconditional: see k_fwd_col256_k1)
                const int i_first0 = 2 * ((F * (jw + 16 * (kColBatch * g)) + pw) * N1 + t.bx * C);
                const int i_end3 = 2 * ((F * (jw + 1 + 16 * (kColBatch * g + kColBatch - 1)) + pw) * N1 + t.bx * C + C);
                if (__builtin_expect(i_first0 > 0 && i_end3 <= len, 1)) {
                    unsigned int neg[kColBatch];
                    int c0[kColBatch], c1[kColBatch];
#pragma unroll
                    for (int q = 0; q < kColBatch; q++) {
                        const unsigned int x = k1_index_bytes(raw[kColBatch * g + q]);
                        neg[q] = ~raw[kColBatch * g + q];
                        c0[q] = k1_table_read<true>(lut, k1_quadrant_offset<false>(x));
                        c1[q] = k1_table_read<true>(lut, k1_quadrant_offset<true>(x));
                    }
                    __builtin_amdgcn_sched_barrier(0);                  // (the eight reads stay ahead of their placements)
#pragma unroll
                    for (int q = 0; q < kColBatch; q++) {
                        const int r = kColBatch * g + q;
                        const int a0 = k1_quadrant_place<false>(c0[q], neg[q]), a1 = k1_quadrant_place<true>(c1[q], neg[q]);
                        const int left = wave_shift_right1(a1);
                        const int bnd = __shfl(ab, 2 * r + (lane >> LOGC), kWave);
                        const int ap = c ? left : bnd;
                        const int st0 = k1_stored_code_scaled(a0, ap), st1 = k1_stored_code_scaled(a1, a0);
                        if (ONCE) {
                            col_once_accumulate(st0, t1, t2);
                            col_once_accumulate(st1, t1, t2);
                        }
                        v[r] = ONCE ? make_float2(k1_normalise_fma(st0, nscale, noff), k1_normalise_fma(st1, nscale, noff))
                                    : make_float2(k1_normalise(st0, mean, scale), k1_normalise(st1, mean, scale));
                    }
                } else {
                    batch_general(row_general, g, std::make_integer_sequence<int, kColBatch>{});      // the batch's rows, one by one
                }
            }
            if (ONCE) col_once_wave_record(t1, t2, once_tiles + (size_t)tile * kOnceWavesPerTile + (size_t)(tid >> 6) * kOnceRecordsPerWave);
        }
        // the capture bytes of the next tile: asked for now, used a whole transform later
        __builtin_amdgcn_sched_barrier(0);
        if (seq + (int)gridDim.x < n_tiles)
            col512_fetch(col512_tile(sw, col_k1_order(seq + gridDim.x, nbx), nbx), N1, opaque_i((int)threadIdx.x), raw_next, sb_next);
        __builtin_amdgcn_sched_barrier(0);
        fft16<false>(v);
        // first exchange (inside the image of this parity), real parts then imaginary parts
#pragma unroll
        for (int k = 0; k < 16; k++) img[(16 * j + k) * C + c] = v[oreg(k)].x;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) v[r].x = img[(j + 16 * r) * C + c];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) img[(16 * j + k) * C + c] = v[oreg(k)].y;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) v[r].y = img[(j + 16 * r) * C + c];
        __syncthreads();
        mul_powers16(v, opaque(wj));
        fft16<false>(v);
        // Y_par[k = j + 16 kk] *= W_512^(k par) = W_512^j * (W_32)^kk
        if (par)      // uniform per wave
            mul_base_step16(v, opaque(wp), make_float2(0.98078528040323043f, -0.19509032201612825f));
        // second exchange + the last radix-2 butterfly across the two images: X[k + 256 q] = Y_0[k] +- Y_1[k], q = par
#pragma unroll
        for (int k = 0; k < 16; k++) img[(j + 16 * k) * C + c] = v[oreg(k)].x;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int idx = (j + 16 * k) * C + c;
            const float e0 = plane[idx], e1 = plane[256 * C + idx];
            v[oreg(k)].x = par ? e0 - e1 : e0 + e1;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) img[(j + 16 * k) * C + c] = v[oreg(k)].y;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int idx = (j + 16 * k) * C + c;
            const float e0 = plane[idx], e1 = plane[256 * C + idx];
            v[oreg(k)].y = par ? e0 - e1 : e0 + e1;
        }
        __syncthreads();                                         // the next tile writes the planes again
        // (the four-step twiddle W_Nc^(n1 k2) is applied by the row pass to its inputs, row_pre_twiddle)
        prev = tile;
    }
    if (prev >= 0) {                                             // the last tile of this workgroup
        const int tid = opaque_i((int)threadIdx.x);
        const Col512Tile tp = col512_tile(sw, prev, nbx);
        float2 *outp = T + (size_t)tp.w * pl.Zs;
        const unsigned int offp = 8u * (unsigned)((((tid >> LOGC) & 15) + 256 * (tid >> (LOGC + 4))) * N1 + tp.bx * C + (tid & (C - 1)));
#pragma unroll
        for (int r = 0; r < 16; r++) store_at(outp + (size_t)(16 * oreg(r)) * N1, offp, v[r]);
    }
}

// ---------------------------------------------------------------------------
// forward column pass for short columns, N2 = 16 F (F = 1, 2, 4, 8: windows of 0.06 .. 0.5 s at 2 Msps,
// N = 2^17 .. 2^20).  Decimation in time by F: thread (column c, part p) transforms the 16 rows
// n2 = p (mod F) in registers, applies W_(16F)^(k p), and after one LDS exchange computes the outputs
// X[k + 16 q], q = p, as the F-point DFT across the parts.  256/F columns per workgroup:
// grid (N1 F/256, n_sw), 256 threads, static LDS 32 KB.
// ---------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(256) void k_fwd_col16x_c16(const SWDesc *sw, const int *codes, long long code_stride,
                                                        const FmStats *stats, float2 *T, FftPlan pl)
{
    static_assert(F == 1 || F == 2 || F == 4 || F == 8, "N2 = 16, 32, 64 or 128");
    constexpr int C = 256 / F;
    __shared__ float2 img[F == 1 ? 1 : F * 16 * C];   // [part][k][column]
    const int len = sw[blockIdx.y].len;
    const int *row = codes + (size_t)blockIdx.y * code_stride;
    const float mean = stats[blockIdx.y].mean, scale = stats[blockIdx.y].scale;
    const int c = threadIdx.x % C, p = threadIdx.x / C;
    const int n1 = blockIdx.x * C + c;
    const int N1 = pl.N1;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = code_park(code_fetch(row, (long long)(F * r + p) * N1 + n1, len));
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = code_convert(code_unpark(v[r]), (long long)(F * r + p) * N1 + n1, len, mean, scale);
    fft16<false>(v);
    const int q = p;
    if constexpr (F > 1) {
        // Y_p[k] *= W_(16F)^(k p); part 0 needs no factor
        if (p) mul_base_step16(v, make_float2(1.0f, 0.0f), unit_root((float)p, 2.0f / (16.0f * F), false));
#pragma unroll
        for (int k = 0; k < 16; k++) img[(p * 16 + k) * C + c] = v[oreg(k)];
        __syncthreads();
        // W_F^m for m = 0 .. F-1 are entries (8/F) m of the eighth roots of unity e^{-2 pi i m/8}
        constexpr float h = 0.70710678118654752f;
        const float2 w8[8] = {{1.0f, 0.0f}, {h, -h}, {0.0f, -1.0f}, {-h, -h}, {-1.0f, 0.0f}, {-h, h}, {0.0f, 1.0f}, {h, h}};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            float2 acc = img[k * C + c];
#pragma unroll
            for (int pp = 1; pp < F; pp++) acc = cadd(acc, cmul(img[(pp * 16 + k) * C + c], w8[((q * pp) % F) * (8 / F)]));
            v[oreg(k)] = acc;
        }
    }
    // X[k2 = k + 16 q] *= W_Nc^(n1 k2) = W^(16 q n1) * (W^(n1))^k
    float2 *out = T + (size_t)blockIdx.y * pl.Nc;
    const float inv2 = 2.0f / (float)pl.Nc;
    const long long e0 = ((long long)n1 * 16 * q) & (pl.Nc - 1);
    mul_base_step16(v, unit_root((float)e0, inv2, false), unit_root((float)n1, inv2, false));
#pragma unroll
    for (int k = 0; k < 16; k++) out[(size_t)(k + 16 * q) * N1 + n1] = v[oreg(k)];
}

// ---------------------------------------------------------------------------
// second half of the inverse row pair kernels: the two rows' Q values are in registers (va: row a as stage-1
// item t; vb: row b as stage-1 item 255 - t, or t for the self-mirrored pair); three radix-16 stages through two
// LDS images, the four-step twiddle, then either the two V rows or (FK > 0) the row pair's share of the short-lag
// column sums.  All threads of the workgroup call it; LDS must be free to overwrite on entry.
// ---------------------------------------------------------------------------
template <bool SELF, int FK>
__device__ __forceinline__ void inv_row_pair_tail(float2 (&va)[16], float2 (&vb)[16], float2 *lds, const int t, const int a,
                                                  const int b, const int pw_index, float2 *V, const FftPlan &pl)
{
    constexpr bool self = SELF;
    const int N2 = pl.N2;
    const int item_b = self ? t : 255 - t;   // which item of row b this thread's stage-1 butterfly is
    fft16<true>(va);
    fft16<true>(vb);
    // Both rows go through stages 2 and 3 together, each in its own LDS image.  (Sending them through
    // one 34 KB image one after the other would fit three workgroups per CU, but the compiler then
    // needs > 170 VGPRs or spills: measured 2.4 ms against 1.6 ms for this form.)
    float2 *la = lds, *lb = lds + kRowLds;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        la[pad16(16 * t + k)] = va[oreg(k)];
        lb[pad16(16 * item_b + k)] = vb[oreg(k)];
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
    float2 *out = V + (size_t)pw_index * pl.Nc;
    const float inv2 = 2.0f / (float)pl.Nc;
    {
        const long long e0 = ((long long)j * a) & (pl.Nc - 1), e1 = ((long long)256 * a) & (pl.Nc - 1);
        mul_base_step16(va, unit_root((float)e0, inv2, true), unit_root((float)e1, inv2, true));
        const long long f0 = ((long long)j * b) & (pl.Nc - 1), f1 = ((long long)256 * b) & (pl.Nc - 1);
        mul_base_step16(vb, unit_root((float)f0, inv2, true), unit_root((float)f1, inv2, true));
    }
    if constexpr (FK == 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            out[(size_t)a * 4096 + j + 256 * k] = va[oreg(k)];
            out[(size_t)b * 4096 + j + 256 * k] = vb[oreg(k)];
        }
    } else {
        const float2 ca = unit_root((float)a, 2.0f / (float)N2, false);     // conj(w_a)
        const float2 cb = unit_root((float)b, 2.0f / (float)N2, false);
        float2 *part = out + (size_t)a * (2 * 256 * FK);                    // row pair a (0 for the SELF instance)
#pragma unroll
        for (int k = 0; k < FK; k++) {
            part[k * 256 + j] = cadd(va[oreg(k)], vb[oreg(k)]);
            part[256 * FK + k * 256 + j] = cadd(cmul(va[oreg(16 - FK + k)], ca), cmul(vb[oreg(16 - FK + k)], cb));
        }
    }
}

// ---------------------------------------------------------------------------
// inverse row pass with K3 fused, N1 = 4096: one workgroup owns rows a and N2 - a (a >= 1).
// Thread t builds Q[a][t + 256 r] and its mirror Q[N2-a][4095 - t - 256 r] from the same four
// spectrum values, so stage 1 of row a (item t) and of row N2-a (item 255 - t) need no exchange.
// grid (N2/2 - 1, n_pw) [+ (1, n_pw) for the SELF instance], 256 threads, dynamic LDS 68 KB.
// ---------------------------------------------------------------------------
//
// FK > 0 is the short-lag form (|lag| < 512 FK - 1): only the elements m = n2 N1 + n1 with n2 = 0, n1 < 256 FK
// and n2 = N2 - 1, n1 >= 4096 - 256 FK can hold a searched lag, and they sit in registers k < FK and k >= 16 - FK
// of every thread.  Instead of the two V rows (64 KB) the workgroup writes its share of those column sums,
//   P0[k][j] = V[a][n1] + V[b][n1]                          (n2 = 0,      n1 = j + 256 k)
//   P1[k][j] = V[a][n1] conj(w_a) + V[b][n1] conj(w_b)      (n2 = N2 - 1, n1 = 4096 - 256 FK + j + 256 k)
// with w_r = e^{2 pi i r/N2}: part[pw][row pair][P0 | P1][256 FK] float2 (4 FK KB); k_fused_reduce adds the
// N2/2 shares in a fixed order.  V is never written or read: the inverse side moves 16 N bytes less.
// (amdgpu_waves_per_eu(2, 2): the 68 KB of dynamic LDS already limit a CU to two workgroups = two waves per
// SIMD; told so, the scheduler keeps ~20 loads in flight instead of squeezing registers for a third wave it
// cannot have -- the short-lag instances otherwise dropped to two loads in flight, 1.41 ms against 1.25 ms.)
template <bool SELF, int FK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SELF ? 1 : 2, SELF ? 8 : 2))) void k_inv_row_pair4096(const PWDesc *pw, const float2 *Z, float2 *V, FftPlan pl, int group_pairs, int n_pw)
{
    extern __shared__ float2 lds[];   // 2 * kRowLds
    const int N2 = pl.N2;
    // SELF: the two self-mirrored rows (0, N2/2), grid (1, n_pw); else the mirrored pair (a, N2 - a),
    // a = blockIdx.x + 1, grid (N2/2 - 1, n_pw).  (Two instantiations: the self-mirrored loader needs
    // twice the loads and would cost the paired kernel its occupancy.)
    // An XCD-aware 1-D remap that puts the station pairs sharing a spectrum row on one XCD measured 4-8 % slower
    // than this plain grid while the kernel was still latency-bound, and exactly the same (1.257 ms both ways) once
    // it was bandwidth-bound: the plain grid stays.
    // group_pairs > 0 (many pairs per window: 8 stations -> 28, 16 -> 120): 1-D grid, XCD-aware.  The group_pairs
    // pair-windows of one window all read the same S station rows a and N2 - a; workgroups are dealt round-robin over
    // the 8 XCDs (b and b + 8 share an L2), so the workgroups of one (window, row pair) group are given consecutive
    // slots of ONE XCD: a station row then comes from HBM once and from that XCD's L2 for its other S - 2 pairs.
    // (With 3 stations a row has two readers and on cfg2 -- 8 MB spectra, the whole window waits in the Infinity Cache -- the
    // remap measured neutral: the plain 2-D grid stays for P <= S unless a window's spectra exceed 64 MB, cfg3.)
    constexpr bool self = SELF;
    int pw_index = blockIdx.y, a = self ? 0 : blockIdx.x + 1;
    if (!self && group_pairs > 0) {
        const int RP = N2 / 2 - 1;
        const unsigned int L = blockIdx.x, xcd = L & 7u, slot = L >> 3;
        const unsigned int g = slot / (unsigned int)group_pairs, p = slot % (unsigned int)group_pairs;
        const unsigned int G = g * 8u + xcd;                       // (window, row pair) group
        const unsigned int w = G / (unsigned int)RP;
        if (w * (unsigned int)group_pairs >= (unsigned int)n_pw) return;      // padding of the last round of 8 groups
        a = (int)(G % (unsigned int)RP) + 1;
        pw_index = (int)(w * (unsigned int)group_pairs + p);
    }
    const PWDesc d = pw[pw_index];
    const int b = self ? N2 / 2 : N2 - a;
    const float2 *ZaA = Z + (size_t)d.sw_a * pl.Zs + (size_t)a * 4096 + (size_t)(a >> 8) * pl.zpad;
    const float2 *ZaB = Z + (size_t)d.sw_a * pl.Zs + (size_t)b * 4096 + (size_t)(b >> 8) * pl.zpad;
    const float2 *ZbA = Z + (size_t)d.sw_b * pl.Zs + (size_t)a * 4096 + (size_t)(a >> 8) * pl.zpad;
    const float2 *ZbB = Z + (size_t)d.sw_b * pl.Zs + (size_t)b * 4096 + (size_t)(b >> 8) * pl.zpad;
    const int t = threadIdx.x;
    const float invNc = 1.0f / (float)pl.Nc;
    float2 va[16], vb[16];
    if constexpr (self) {
        // each row mirrors onto itself: row 0 by k1 -> (4096 - k1) mod 4096, row N2/2 by k1 -> 4095 - k1;
        // every thread builds only its own Q values (the mirror operand is re-read)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int k1 = t + 256 * r;
            float2 q, qm;
            const int m0 = (4096 - k1) & 4095;
            pair_q(ZaA[k1], ZaA[m0], ZbA[k1], ZbA[m0], unit_root((float)((long long)k1 * N2), invNc, false), q, qm);
            va[r] = q;
            const int m1 = 4095 - k1;
            pair_q(ZaB[k1], ZaB[m1], ZbB[k1], ZbB[m1], unit_root((float)((long long)k1 * N2 + b), invNc, false), q, qm);
            vb[r] = q;
        }
    } else {
        // w(k) = W_N^k, k = (t + 256 r) N2 + a  =>  w = w0 * W_32^r
        const long long k0 = (long long)t * N2 + a;
        const float2 st = make_float2(0.98078528040323043f, -0.19509032201612825f);   // e^{-2 pi i/32}
        float2 w = unit_root((float)k0, invNc, false);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int k1 = t + 256 * r;
            float2 q, qm;
            pair_q(ZaA[k1], ZaB[4095 - k1], ZbA[k1], ZbB[4095 - k1], w, q, qm);
            va[r] = q;
            vb[15 - r] = qm;
            if ((r & 3) == 3)   // re-anchor every 4 steps to keep the running product short
                w = unit_root((float)(k0 + (long long)(r + 1) * 256 * N2), invNc, false);
            else
                w = cmul(w, st);
        }
    }
    inv_row_pair_tail<SELF, FK>(va, vb, lds, t, a, b, pw_index, V, pl);
}

// short-lag form, second half: element sums over the N2/2 row-pair shares in a fixed order (four interleaved
// partial sums, then ((s0 + s1) + (s2 + s3))), lag filter and K5.
// part = V + pw_index * Nc: [RP][2][256 FK] float2, followed by the lag array lags[1024 FK] (float),
// lags[li] = c[li - 512 FK] unscaled, kept for the sub-sample refinement.
// grid (2 FK, n_pw): blockIdx.x = side * FK + k; 256 threads.
template <int FK>
__global__ __launch_bounds__(256) void k_fused_reduce(float2 *V, unsigned long long *keys, const PWDesc *pw, FftPlan pl,
                                                      int lag_lo, int lag_hi, float *lag_dump, float dump_scale)
{
    __shared__ unsigned long long red[4];
    const int RP = pl.N2 / 2, j = threadIdx.x;
    const int side = blockIdx.x / FK, k = blockIdx.x % FK;
    float2 *part = V + (size_t)blockIdx.y * pl.Nc;
    float *lags = reinterpret_cast<float *>(part + (size_t)RP * 2 * 256 * FK);
    const float2 *src = part + side * 256 * FK + k * 256 + j;
    const size_t stride = (size_t)2 * 256 * FK;
    float2 s4[4] = {make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f)};
    int rp = 0;
    for (; rp + 4 <= RP; rp += 4) {
        float2 x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = src[(size_t)(rp + u) * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) s4[u] = cadd(s4[u], x[u]);
    }
    for (int u = 0; rp < RP; rp++, u++) s4[u] = cadd(s4[u], src[(size_t)rp * stride]);
    const float2 acc = cadd(cadd(s4[0], s4[1]), cadd(s4[2], s4[3]));
    const int m = side == 0 ? 256 * k + j : -256 * FK + 256 * k + j;    // element index, negative side wraps
    const float vals[2] = {acc.x, acc.y};
    unsigned long long best = 0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int d = 2 * m + q;
        lags[d + 512 * FK] = vals[q];
        if (d >= lag_lo && d <= lag_hi && vals[q] == vals[q]) {
            const unsigned long long key = peak_key(vals[q], d);
            best = key > best ? key : best;
            if (lag_dump) lag_dump[d - lag_lo] = vals[q] * dump_scale;
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

// refinement neighbours c[lag-1], c[lag], c[lag+1] from the lag array k_fused_reduce left behind
// (the host only takes the short-lag form when lag_hi + 1 and lag_lo - 1 are inside that array)
template <int FK>
__global__ void k_refine_fused(const float2 *V, const unsigned long long *keys, const PWDesc *pw, FftPlan pl, int n_pw,
                               float *raw)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per pair-window
    if (id >= n_pw) return;
    const int slot = pw[id].out_index;
    const unsigned long long k = keys[slot];
    float r[3] = {0.0f, 0.0f, 0.0f};
    if (k != 0 && (unsigned int)(k >> 32) != 0) {
        const unsigned int rank = 0x7fffffffu - ((unsigned int)k >> 1);
        const int lag = rank == 0 ? 0 : ((rank & 1u) ? (int)((rank + 1u) >> 1) : -(int)(rank >> 1));
        const float *lags = reinterpret_cast<const float *>(V + (size_t)id * pl.Nc + (size_t)(pl.N2 / 2) * 2 * 256 * FK);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int li = lag - 1 + q + 512 * FK;
            r[q] = li >= 0 && li < 1024 * FK ? lags[li] : 0.0f;
        }
    }
#pragma unroll
    for (int q = 0; q < 3; q++) raw[3 * (size_t)slot + q] = r[q];
}

// ---------------------------------------------------------------------------
// pruned inverse column pass + K5.  Only the outputs n2 that can hold a searched lag are
// evaluated, as direct DFT sums over k2 (lags 2m, 2m+1 with m = n2*N1 + n1; |lag| <= max):
//   n2 in [0, NP)  (non-negative lags)  and  n2 in [N2 - NN, N2)  (negative lags), NP + NN <= 8.
// ---------------------------------------------------------------------------
constexpr int kPruneMax = 8;

// grid (N1/128, n_pw), 256 threads: cp = t & 63 (column PAIR: n1 = 128*bx + 2cp, +1), g = t >> 6
// (row group: rows g, g+4, ...); 16-byte loads -> 1 KB contiguous per row, 8 rows in flight per thread
// oc (single-look K1, k1_single_look.hpp): the residual-mean terms added to every candidate before the argmax
__global__ __launch_bounds__(256) void k_inv_col_pruned_any(const float2 *V, unsigned long long *keys, const PWDesc *pw,
                                                       FftPlan pl, int lag_lo, int lag_hi, int np, int nn,
                                                       float *lag_dump, float dump_scale, OnceCorr oc)
{
    extern __shared__ float2 wtab[];               // e^{+2 pi i k/N2}, N2 entries (dynamic LDS)
    __shared__ float4 part[4][kPruneMax][64];
    __shared__ unsigned long long red[4];
    const int N2 = pl.N2, N1 = pl.N1;
    for (int k = threadIdx.x; k < N2; k += 256) wtab[k] = unit_root((float)k, 2.0f / (float)N2, true);
    __syncthreads();
    const int cp = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n1 = (blockIdx.x << 7) + 2 * cp;
    const PWDesc pwd = pw[blockIdx.y];
    if (oc.fin && blockIdx.x == 0 && threadIdx.x == 0) once_publish_gain(oc, pwd);
    OncePair op{};
    if (oc.fin) op = once_pair(oc, pwd);
    const float4 *in = reinterpret_cast<const float4 *>(V + (size_t)blockIdx.y * pl.Nc + n1);
    const size_t row_stride = (size_t)N1 / 2;      // in float4 units
    float4 acc[kPruneMax];
#pragma unroll
    for (int o = 0; o < kPruneMax; o++) acc[o] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int nout = np + nn;
    for (int kb = g; kb < N2; kb += 32) {          // 8 rows per trip, all loads issued first
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k2 = kb + 4 * u;
            x[u] = k2 < N2 ? in[(size_t)k2 * row_stride] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k2 = kb + 4 * u;
#pragma unroll
            for (int o = 0; o < kPruneMax; o++) {
                if (o < nout) {
                    // output n2 = o (o < np) or N2 - nn + (o - np); factor e^{+2 pi i n2 k2 / N2}
                    const int n2 = o < np ? o : N2 - nn + (o - np);
                    const float2 w = wtab[(n2 * k2) & (N2 - 1)];
                    acc[o].x += x[u].x * w.x - x[u].y * w.y;
                    acc[o].y += x[u].x * w.y + x[u].y * w.x;
                    acc[o].z += x[u].z * w.x - x[u].w * w.y;
                    acc[o].w += x[u].z * w.y + x[u].w * w.x;
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < kPruneMax; o++) part[g][o][cp] = acc[o];
    __syncthreads();
    unsigned long long best = 0;
    for (int e = threadIdx.x; e < 64 * nout; e += 256) {
        const int o = e >> 6, c = e & 63;
        float4 s = part[0][o][c];
#pragma unroll
        for (int gg = 1; gg < 4; gg++) {
            const float4 q = part[gg][o][c];
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
        const int n2 = o < np ? o : N2 - nn + (o - np);
        float vals[4] = {s.x, s.y, s.z, s.w};   // lags 2m .. 2m+3 with m = n2*N1 + n1
        long long d = 2 * ((long long)n2 * N1 + (blockIdx.x << 7) + 2 * c);
        if (d >= pl.Nc) d -= 2 * pl.Nc;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const long long dq = d + q;
            if (oc.fin) vals[q] += once_correction(oc, op, dq);
            if (dq >= lag_lo && dq <= lag_hi && vals[q] == vals[q]) {
                const unsigned long long k = peak_key(vals[q], (int)dq);
                best = k > best ? k : best;
                if (lag_dump) lag_dump[dq - lag_lo] = vals[q] * dump_scale;
            }
        }
    }
    best = wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long bb = red[0];
        for (int w = 1; w < 4; w++) bb = red[w] > bb ? red[w] : bb;
        if (bb) atomicMax(&keys[pwd.out_index], bb);
    }
}

// The same kernel for a compile-time output set (NP outputs 0..NP-1, NN outputs N2-NN..N2-1): the
// factor e^{2 pi i n2 k2/N2} is a power of w = e^{2 pi i k2/N2} (one table read per row, the powers
// by multiplication, conjugates for the negative side), which removes the per-(row, output) index
// arithmetic and LDS reads of the generic form (6150 -> ~3000 VALU instructions per thread at 3+3).
template <int NP, int NN>
__global__ __launch_bounds__(256) void k_inv_col_pruned(const float2 *V, unsigned long long *keys, const PWDesc *pw,
                                                       FftPlan pl, int lag_lo, int lag_hi, float *lag_dump,
                                                       float dump_scale, OnceCorr oc)
{
    constexpr int NOUT = NP + NN;
    constexpr int NPW = (NP - 1 > NN ? NP - 1 : NN) + 1;     // powers w^0 .. w^(NPW-1)
    extern __shared__ float2 wtab[];               // e^{+2 pi i k/N2}, N2 entries (dynamic LDS)
    __shared__ float4 part[4][NOUT][64];
    __shared__ unsigned long long red[4];
    const int N2 = pl.N2, N1 = pl.N1;
    for (int k = threadIdx.x; k < N2; k += 256) wtab[k] = unit_root((float)k, 2.0f / (float)N2, true);
    __syncthreads();
    const int cp = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n1 = (blockIdx.x << 7) + 2 * cp;
    const PWDesc pwd = pw[blockIdx.y];
    if (oc.fin && blockIdx.x == 0 && threadIdx.x == 0) once_publish_gain(oc, pwd);
    OncePair op{};
    if (oc.fin) op = once_pair(oc, pwd);
    const float4 *in = reinterpret_cast<const float4 *>(V + (size_t)blockIdx.y * pl.Nc + n1);
    const size_t row_stride = (size_t)N1 / 2;      // in float4 units
    float4 acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; o++) acc[o] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // software pipeline: the 8 loads of trip i+1 are in flight while trip i is accumulated
    float4 x[8], xn[8];
#pragma unroll
    for (int u = 0; u < 8; u++) x[u] = in[(size_t)(g + 4 * u) * row_stride];        // N2 % 32 == 0 (host checks)
#pragma unroll 1
    for (int kb = g; kb < N2; kb += 32) {
        const int kn = kb + 32 < N2 ? kb + 32 : kb;                                   // last trip: harmless re-read
#pragma unroll
        for (int u = 0; u < 8; u++) xn[u] = in[(size_t)(kn + 4 * u) * row_stride];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float2 wp[NPW];
            wp[0] = make_float2(1.0f, 0.0f);
            if (NPW > 1) wp[1] = wtab[kb + 4 * u];
#pragma unroll
            for (int q = 2; q < NPW; q++) wp[q] = cmul(wp[q - 1], wp[1]);
#pragma unroll
            for (int o = 0; o < NOUT; o++) {
                if (o == 0 && NP > 0) {
                    acc[0].x += x[u].x; acc[0].y += x[u].y; acc[0].z += x[u].z; acc[0].w += x[u].w;
                } else {
                    const float2 w = o < NP ? wp[o] : cconj(wp[NN - (o - NP)]);
                    acc[o].x += x[u].x * w.x - x[u].y * w.y;
                    acc[o].y += x[u].x * w.y + x[u].y * w.x;
                    acc[o].z += x[u].z * w.x - x[u].w * w.y;
                    acc[o].w += x[u].z * w.y + x[u].w * w.x;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = xn[u];
    }
#pragma unroll
    for (int o = 0; o < NOUT; o++) part[g][o][cp] = acc[o];
    __syncthreads();
    unsigned long long best = 0;
    for (int e = threadIdx.x; e < 64 * NOUT; e += 256) {
        const int o = e >> 6, c = e & 63;
        float4 s = part[0][o][c];
#pragma unroll
        for (int gg = 1; gg < 4; gg++) {
            const float4 q = part[gg][o][c];
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
        const int n2 = o < NP ? o : N2 - NN + (o - NP);
        float vals[4] = {s.x, s.y, s.z, s.w};   // lags 2m .. 2m+3 with m = n2*N1 + n1
        long long d = 2 * ((long long)n2 * N1 + (blockIdx.x << 7) + 2 * c);
        if (d >= pl.Nc) d -= 2 * pl.Nc;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const long long dq = d + q;
            if (oc.fin) vals[q] += once_correction(oc, op, dq);
            if (dq >= lag_lo && dq <= lag_hi && vals[q] == vals[q]) {
                const unsigned long long k = peak_key(vals[q], (int)dq);
                best = k > best ? k : best;
                if (lag_dump) lag_dump[dq - lag_lo] = vals[q] * dump_scale;
            }
        }
    }
    best = wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long bb = red[0];
        for (int w = 1; w < 4; w++) bb = red[w] > bb ? red[w] : bb;
        if (bb) atomicMax(&keys[pwd.out_index], bb);
    }
}

}  // namespace tdoa
