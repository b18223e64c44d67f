// dec_stream.hpp -- the pair step of the decimated inverse as a COLUMN WALK (round 4): K3 + the 16:1 decimating FIR of
// fft_radix8.hpp (k_pair_decimate16's filter, same taps, same outputs) with one thread per spectrum column.
//
// Consecutive bins k = k2 + N2 k1 run down the columns of the [k2][k1] spectrum, so G[j] = sum_t h[t] Q[16 j + t] is a
// stencil DOWN the columns.  k_pair_decimate16 gathers 4096 consecutive bins (16 or 8 columns) into an LDS image and
// runs the FIR there with per-lane taps; on the 4096 x 4096 plan (ten-second windows, BASELINE config 3) such a tile is
// one column -- 8-byte pieces of 4096 rows -- and the plan had no decimated inverse at all.  Here a thread owns a column
// and walks its rows; what a wave reads is contiguous along k1 like every other row pass, and there is no LDS image, no
// barrier, no quad sums:
//   * The forward row pass leaves the UNPACKED spectra U (the station's half of K3, k_fwd_row4096_unpack<true>) in place,
//     row-major.  K3's other half needs U[k] and U[Nc - k] of both stations, (k2, k1) and (N2 - k2, 4095 - k1): one pair of
//     loads per station yields Q[k] AND Q[Nc - k] (pair_u_pk).  Thread k1 < 2048 therefore walks column k1 downwards
//     (rows 0 .. N2 - 1) and, with the same values, column km = 4095 - k1 upwards (rows N2 - 1 .. 1); the two rows 0 pair
//     inside row 0 and are evaluated on their own, before and after the loop.  Every bin is evaluated exactly once.
//   * A row r = 16 g + p feeds the kDecSteps outputs i = g + C - s with the tap of (phase p, step s): the same twelve numbers
//     for every lane (three ds_read_b128 at a wave-uniform address; the tile kernel's lanes differ in the phase).  Twelve
//     accumulators per walk in registers, shifted by one every 16 rows.  By the filter's symmetry the upward walk uses the
//     same twelve taps as the downward one.
//   * Output i (0 .. N2/16 - 1) of column c is G[(N2/16) c + i], stored as the small plan's row i: coalesced.  The six
//     outputs next to either end of a column also need bins of the neighbouring column; a walk leaves what ITS bins add to
//     them in X[pw][12][4096] (slot 6 + i': output i' = 0..5 of the NEXT column; slot i' + 6, i' = -6..-1: output N2/16 + i'
//     of the PREVIOUS one) and k_inv_rows_plain_r8 adds slot rows of the neighbouring columns when it loads G -- coalesced
//     like G itself.  No thread ever waits for another.
// grid (32 column blocks of 64, ceil(n_pw / kDecWavesPerWg)), 64 kDecWavesPerWg threads, 1 KB of LDS (the taps).
#pragma once

#include "fft_radix8.hpp"

namespace tdoa {

#ifndef TDOA_DEC_STREAM_BATCH
#define TDOA_DEC_STREAM_BATCH 2
#endif
#ifndef TDOA_DEC_STREAM_WAVES
#define TDOA_DEC_STREAM_WAVES 3
#endif
constexpr int kDecStreamBatch = TDOA_DEC_STREAM_BATCH; // rows fetched ahead, per buffer (two buffers)
constexpr int kDecShareRows = 2 * kDecEdge;            // X: rows per pair-window (each 4096 columns)
#ifndef TDOA_DEC_WAVES_PER_WG
#define TDOA_DEC_WAVES_PER_WG 4
#endif
constexpr int kDecWavesPerWg = TDOA_DEC_WAVES_PER_WG;  // pair-windows per workgroup (one wave each, the same 64 columns)

// The walk's register stencil is written for 8 or 12 steps per phase (whole float4 of taps, mac_all): a measurement build
// with another filter length (TDOA_DEC_STEPS=14: the 140 dB filter of rounds 2-3) has the tile form only -- the library
// then runs without the column walk (ctx->dec_cols off: no decimated inverse on the two-sweep plans).
#if TDOA_DEC_STEPS == 8 || TDOA_DEC_STEPS == 12
#define TDOA_HAVE_DEC_COLS 1
// taps: the tile kernel's table [16 phases][16 steps] (256 floats), then rot[16] = W_N^p as float2 (N = 2 Nc)
// N2 is the template argument itself (round 5: 2560 = 5 x 512 next to the powers of two); where it is not a power of two
// the row rotations W_N^k (N = 2 Nc = 5 x 2^k) come from unit_root_any.
template <int N2_>
__global__ __launch_bounds__(64 * kDecWavesPerWg) __attribute__((amdgpu_waves_per_eu(TDOA_DEC_STREAM_WAVES, TDOA_DEC_STREAM_WAVES))) void k_pair_decimate_cols(const PWDesc *pw, const float2 *U, float2 *G, float2 *X, FftPlan pl,
                                                              const float *__restrict__ taps, int n_pw)
{
    constexpr int N2 = N2_, N1 = 4096, C = kDecCentre, S = kDecSteps, B = kDecStreamBatch;
    constexpr bool POW2 = (N2 & (N2 - 1)) == 0;
    constexpr int NG = N2 / 16;                                     // groups of 16 rows = outputs per column
    // A workgroup = kDecWavesPerWg waves, every one of them the SAME 64 columns (and their 64 partners) of a DIFFERENT
    // pair-window: wave w of workgroup (cb, q) walks pair-window kDecWavesPerWg q + w.  Consecutive pair-windows are a
    // window's pairs in the order (0,1), (0,2), ... -- they mostly share the template station, so the waves of a workgroup ask
    // for the same template rows within a few hundred cycles of each other: one of them brings a row in, the others find it
    // in the CU's L1 / the XCD's L2.  (One 256-column block of ONE pair-window per workgroup pulled 24 GB per cfg4 step through
    // the fabric for 5.6 GB of spectra: walks that share a station started whenever a slot came free, tens of microseconds
    // apart against an L2 turnover of ~6.)
    constexpr int W = kDecWavesPerWg;                               // (grid.x: the 32 column blocks of 64 of the left half)
    static_assert(NG >= 2 * C + 2 && N2 % (2 * B) == 0 && 16 % (2 * B) == 0, "ring and loop geometry");
    // LDS: the taps [phase][step]; phase 16 = phase 0 with its steps reversed (the upward walk's row of phase 0); then the 16
    // row rotations
    __shared__ __attribute__((aligned(16))) float ltaps[17 * S];
    __shared__ __attribute__((aligned(16))) float2 lrot[16];
    const int t = threadIdx.x;
    for (int e = t; e < 17 * S; e += 64 * W) {
        const int p = e / S, s = e % S;
        ltaps[e] = p < 16 ? taps[16 * p + s] : taps[S - 1 - s];
    }
    if (t < 16) lrot[t] = reinterpret_cast<const float2 *>(taps + 256)[t];
    __syncthreads();
    const unsigned int cbu = blockIdx.x, pwu = (unsigned int)blockIdx.y * W + (unsigned int)__builtin_amdgcn_readfirstlane(t >> 6);
    if (pwu >= (unsigned int)n_pw) return;
    const int k1 = (int)cbu * 64 + (t & 63), km = N1 - 1 - k1;
    const PWDesc d = pw[pwu];
    const float2 *Ua = U + (size_t)d.sw_a * pl.Zs, *Ub = U + (size_t)d.sw_b * pl.Zs;
    const int zpad = pl.zpad;
    auto row_at = [&](const float2 *base, int k2) { return base + (size_t)k2 * N1 + (size_t)(k2 >> 8) * zpad; };
    const float invNc = 1.0f / (float)pl.Nc;
    // W_N^num, N = 2 Nc, num = k2 + N2 k1 < Nc <= 2^24 (exact as a float)
    auto w_n = [&](float num) {
        if constexpr (POW2) return unit_root(num, invNc, false);
        else return unit_root_any(num, 0.5f * (float)pl.Nc, 2.0f * invNc, false);
    };
    const size_t rc = (size_t)(pl.Nc / kDecD);
    float2 *g_top = G + (size_t)pwu * rc + k1, *g_bot = G + (size_t)pwu * rc + km;
    float2 *x_top = X + (size_t)pwu * kDecShareRows * N1 + k1, *x_bot = X + (size_t)pwu * kDecShareRows * N1 + km;

    float2 at[S], ab[S];        // at[s]: output g + C - s of column k1;  ab[u]: output gb - (C - 1) + u of column km
#pragma unroll
    for (int s = 0; s < S; s++) at[s] = ab[s] = make_float2(0.0f, 0.0f);

    // the taps of a phase (the same for every lane): three ds_read_b128 at a wave-uniform address.  A complex value times a
    // real tap is ONE v_pk_fma_f32 whose tap operand is a register PAIR read through op_sel -- the low half for both lanes
    // (tap 2 j) or the high half (tap 2 j + 1) -- so the twelve taps stay the six pairs they were loaded as (written in C the
    // compiler copied every tap into both halves of a pair of its own first: 24 v_mov and 24 registers per row).
    typedef float v2f __attribute__((ext_vector_type(2)));
    struct TapRow { v2f h2[S / 2]; };
    auto tap_row = [&](int p) {
        static_assert(S % 4 == 0, "whole float4 of taps per phase");
        const float4 *tp = reinterpret_cast<const float4 *>(ltaps + S * p);
        TapRow r;
#pragma unroll
        for (int s = 0; s < S / 4; s++) {
            const float4 v = tp[s];
            r.h2[2 * s] = v2f{v.x, v.y};
            r.h2[2 * s + 1] = v2f{v.z, v.w};
        }
        return r;
    };
    auto mac = [](float2 &acc, const TapRow &tr, auto s_c, float2 q_) {       // acc += tap[s] q
        constexpr int s = decltype(s_c)::value;
        v2f a = {acc.x, acc.y};
        const v2f q = {q_.x, q_.y};
        if (s & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a) : "v"(q), "v"(tr.h2[s / 2]));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(a) : "v"(q), "v"(tr.h2[s / 2]));
        acc = make_float2(a.x, a.y);
    };
    auto mac_all = [&](float2 (&acc)[S], const TapRow &tr, float2 q) {
        mac(acc[0], tr, std::integral_constant<int, 0>{}, q);
        mac(acc[1], tr, std::integral_constant<int, 1>{}, q);
        mac(acc[2], tr, std::integral_constant<int, 2>{}, q);
        mac(acc[3], tr, std::integral_constant<int, 3>{}, q);
        mac(acc[4], tr, std::integral_constant<int, 4>{}, q);
        mac(acc[5], tr, std::integral_constant<int, 5>{}, q);
        mac(acc[6], tr, std::integral_constant<int, 6>{}, q);
        mac(acc[7], tr, std::integral_constant<int, 7>{}, q);
        if constexpr (S > 8) {
            mac(acc[8], tr, std::integral_constant<int, 8>{}, q);
            mac(acc[9], tr, std::integral_constant<int, 9>{}, q);
            mac(acc[10], tr, std::integral_constant<int, 10>{}, q);
            mac(acc[11], tr, std::integral_constant<int, 11>{}, q);
        }
        static_assert(S == 8 || S == 12, "steps per phase");
    };
    // row 0 of a column pairs inside row 0: (0, c) with (0, 4096 - c); bin 0 carries (A+[0], A-[0]).  For column k1 it is
    // the first row of the downward walk (phase 0 of group 0: output C - s in at[s]).
    {
        const float2 *ra = row_at(Ua, 0), *rb = row_at(Ub, 0);
        const int kp = (N1 - k1) & (N1 - 1);
        const TapRow tr = tap_row(0);
        float2 q, qm;
        pair_u_pk(ra[k1], ra[kp], rb[k1], rb[kp], w_n((float)k1 * (float)N2), k1 == 0, q, qm);
        mac_all(at, tr, q);
    }
    x_top[0] = make_float2(0.0f, 0.0f);          // the output 6 places before a column gets nothing from it (|t| >= 96)
    x_bot[0] = make_float2(0.0f, 0.0f);

    // rows (k2, k1) of both stations and their partners (N2 - k2, 4095 - k1), through four per-lane pointers that move one
    // row per fetch (+ the plan's padding after every 256 rows).  The partner of row 0 is not a row of the walk: its slot
    // reads row 0 again (the value is dropped) and the pointers then jump to row N2 - 1.
    // (wave-uniform row pointers + a 32-bit lane offset: global_load with an SGPR base, no vector instruction per address)
    const float2 *rta = row_at(Ua, 0), *rtb = row_at(Ub, 0), *rma = rta, *rmb = rtb;
    const unsigned int off_t = 8u * (unsigned int)k1, off_m = 8u * (unsigned int)km;
    auto at_off = [](const float2 *base, unsigned int byte_off) {
        return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(base) + byte_off);
    };
    auto fetch_row = [&](float2 (&dst)[4], int k2) {                // row k2, then on to row k2 + 1
        // (the two lane offsets pass through an empty asm per fetch: hoisted out of the loop as 64-bit values they cost a
        // v_lshl_add_u64 per load; seen as 32-bit values next to the load they become its offset operand -- global_load v, v_off, s[base])
        unsigned int ot = off_t, om = off_m;
        asm volatile("" : "+v"(ot), "+v"(om));
        dst[0] = at_off(rta, ot);
        dst[1] = at_off(rma, om);
        dst[2] = at_off(rtb, ot);
        dst[3] = at_off(rmb, om);
        long long dt = N1 + ((k2 & 255) == 255 ? zpad : 0);         // row k2 + 1 starts a block of 256 rows
        long long dm = -(long long)N1 - ((k2 & 255) == 0 ? zpad : 0);   // row N2 - k2 ends one (going down)
        if (k2 == 0) dm = (long long)(N2 - 1) * N1 + (long long)((N2 - 1) >> 8) * zpad;
        if (k2 >= N2 - 1) dt = dm = 0;                              // (the prefetch past the last row re-reads it)
        rta += dt;
        rtb += dt;
        rma += dm;
        rmb += dm;
    };
    // One row: K3 (pair_u_pk) on the four values, then the twelve multiply-adds of either walk.  The phase is a RUN-TIME
    // value: the loop below is not unrolled over a group's 16 phases -- unrolled, every form of the tap fetch (scalar loads,
    // LDS reads) was hoisted to the top of the group by the compiler and the 192 values spilled, SGPRs into vector lanes (880
    // v_readlane / v_writelane per group), VGPRs into scratch.
    //   downward walk: row (g, p) -> output g + C - s in at[s], tap (p, s).
    //   upward walk: row N2 - 16 g - p of column km.  p > 0: phase 16 - p of its group gb = NG - 1 - g; slot u holds output
    //   gb - (C - 1) + u, step S - 1 - u, and tap(16 - p, S - 1 - u) = h[-(16 (u - C) + p)] = tap(p, u) by the filter's
    //   symmetry: the SAME twelve.  p = 0: phase 0 of group NG - g, the last row of that group: tap(0, S - 1 - u), then the
    //   group's finished output NG - g + C leaves and the slots move up.
    auto bottom_leaves = [&](int i) {                               // ab[S - 1] is output i of column km, complete as far as km's bins go
        if (i >= NG) x_bot[(size_t)(kDecEdge + i - NG) * N1] = ab[S - 1];
        else if (i >= 0) g_bot[(size_t)i * N1] = ab[S - 1];
        else x_bot[(size_t)(kDecEdge + i) * N1] = ab[S - 1];
#pragma unroll
        for (int u = S - 1; u > 0; u--) ab[u] = ab[u - 1];
        ab[0] = make_float2(0.0f, 0.0f);
    };
    auto top_leaves = [&](int i) {
        if (i < 0) x_top[(size_t)(kDecEdge + i) * N1] = at[S - 1];
        else if (i < NG) g_top[(size_t)i * N1] = at[S - 1];
        else x_top[(size_t)(kDecEdge + i - NG) * N1] = at[S - 1];
#pragma unroll
        for (int s = S - 1; s > 0; s--) at[s] = at[s - 1];
        at[0] = make_float2(0.0f, 0.0f);
    };
    float2 wg = make_float2(1.0f, 0.0f);
    auto row = [&](const float2 (&v)[4], int k2, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;            // the only row of an iteration whose phase can be 0
        const int p = k2 & 15, g = k2 >> 4;
        const TapRow tr = tap_row(p);
        float2 q, qm;
        pair_u_pk(v[0], v[1], v[2], v[3], cmul(wg, lrot[p]), false, q, qm);
        if (FIRST && k2 == 0) q = qm = make_float2(0.0f, 0.0f);     // row 0: done above
        mac_all(at, tr, q);
        if (!FIRST) {
            mac_all(ab, tr, qm);
        } else {
            const TapRow tb = tap_row(p ? p : 16);                  // (phase 0: the reversed row)
            mac_all(ab, tb, qm);
            if (p == 0 && g > 0) bottom_leaves(NG - g + C);
        }
    };
    float2 bufa[B][4], bufb[B][4];
#pragma unroll
    for (int r = 0; r < B; r++) fetch_row(bufa[r], r);
#pragma unroll 1
    for (int k2 = 0; k2 < N2; k2 += 2 * B) {
        if ((k2 & 15) == 0) wg = w_n((float)k2 + (float)k1 * (float)N2);     // W_N^(16 g + N2 k1): < 2^24, exact
#pragma unroll
        for (int r = 0; r < B; r++) fetch_row(bufb[r], k2 + B + r);
        row(bufa[0], k2, std::true_type{});
#pragma unroll
        for (int r = 1; r < B; r++) row(bufa[r], k2 + r, std::false_type{});
#pragma unroll
        for (int r = 0; r < B; r++) fetch_row(bufa[r], k2 + 2 * B + r);
#pragma unroll
        for (int r = 0; r < B; r++) row(bufb[r], k2 + B + r, std::false_type{});
        if ((k2 & 15) == 16 - 2 * B) top_leaves((k2 >> 4) - (S - 1 - C));       // group g of the downward walk is complete: output g - 5
    }
    // the downward walk is through: at[s] holds output NG + C - 1 - s as far as column k1 goes -- NG - 5 .. NG - 1 are its
    // own, NG .. NG + 5 the next column's first six
#pragma unroll
    for (int n = 0; n < S - 1; n++) top_leaves(NG - (S - 1 - C) + n);
    // row 0 of column km: pairs with (0, 4096 - km) = (0, k1 + 1); phase 0 of group 0, the upward walk's last row
    {
        const float2 *ra = row_at(Ua, 0), *rb = row_at(Ub, 0);
        const TapRow tr = tap_row(16);
        float2 q, qm;
        pair_u_pk(ra[km], ra[k1 + 1], rb[km], rb[k1 + 1], w_n((float)km * (float)N2), false, q, qm);
        mac_all(ab, tr, q);
    }
    // ab[u] holds output u - (C - 1): C .. 0 are column km's own, -1 .. -5 the previous column's last five
#pragma unroll
    for (int n = 0; n < S; n++) bottom_leaves(C - n);
}
#else
#define TDOA_HAVE_DEC_COLS 0
#endif

}  // namespace tdoa
