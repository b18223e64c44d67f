// dec_stream.hpp -- the decimated inverse (fft_radix8.hpp, k_pair_decimate16) for the 4096 x 4096 plan: ten-second
// windows, N = 2^25 (BASELINE config 3).
//
// On that plan consecutive bins k = k2 + 4096 k1 run down a whole COLUMN of the [k2][k1] spectrum, so the 4096-bin tiles
// k_pair_decimate16 works on would be 8-byte pieces of 4096 different rows.  The FIR G[j] = sum_t h[t] Q[16 j + t] is a
// stencil DOWN the columns instead, and this kernel runs it as one: a thread owns a column and walks the rows; what it
// reads is contiguous along k1 like every other row pass.
//   * The forward row pass leaves the UNPACKED spectra U (the station's half of K3, k_fwd_row4096_unpack<true>) in place,
//     row-major; K3's other half needs U[k] and U[Nc - k] of both stations: (k2, k1) and (4096 - k2, 4095 - k1).  One pair
//     of loads per station therefore yields Q[k] AND Q[Nc - k] (pair_u_pk): thread k1 walks column k1 downwards from row 0
//     and, with the same values, column 4095 - k1 upwards from row 4095.  Each walk goes 96 rows past the middle (the
//     stencil's reach), so a column's first 128 outputs come from the thread that walks it downwards and the last 128
//     from the one that walks it upwards: 4.7 % of the rows are read by both.
//   * A row r = 16 g + p feeds the kDecSteps outputs i = g + C - s with the tap of (phase p, step s) -- wave-uniform, so the
//     taps are one LDS address per wave (the tile kernel's lanes differ in the phase).  Twelve
//     accumulators per walk in registers, shifted by one every 16 rows; no LDS, no barrier.
//   * Output i of column c is G[256 c + i], stored where the small plan's row pass wants it ([i][c]: coalesced).  The
//     outputs next to a column's ends need bins of the neighbouring columns: the walk leaves its share of them in
//     E[pw][column][2 kDecEdge] exactly as the tile kernel does (k_inv_rows_plain_r8 adds them).
// grid: 16 column blocks x n_pw (or the XCD-grouped 1-D form of k_pair_decimate16), 256 threads, 1.2 KB of LDS (the taps).
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
static_assert(16 % (2 * kDecStreamBatch) == 0, "a group of 16 rows is a whole number of buffer pairs");

// taps: the tile kernel's table [16 phases][16 steps] (256 floats), then rot[16] = W_N^p as float2 (N = 2 Nc)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TDOA_DEC_STREAM_WAVES, TDOA_DEC_STREAM_WAVES))) void k_pair_decimate_stream(const PWDesc *pw, const float2 *U, float2 *G, float2 *E, FftPlan pl,
                                                              const float *__restrict__ taps, int group_pairs, int n_pw)
{
    constexpr int N2 = 4096, N1 = 4096, C = kDecCentre, S = kDecSteps, B = kDecStreamBatch;
    constexpr int kGroups = (N2 / 2 + 16 * C) / 16;                 // 134 groups of 16 rows: rows 0 .. 2143
    constexpr int kColBlocks = N1 / 256;
    unsigned int cbu = blockIdx.x, pwu = blockIdx.y;
    if (group_pairs > 0) {                                          // same XCD grouping as k_pair_decimate16
        const unsigned int L = blockIdx.x, xcd = L & 7u, slot = L >> 3;
        const unsigned int g = slot / (unsigned int)group_pairs, p = slot % (unsigned int)group_pairs;
        const unsigned int Gi = g * 8u + xcd, w = Gi / kColBlocks;
        if (w * (unsigned int)group_pairs >= (unsigned int)n_pw) return;
        cbu = Gi % kColBlocks;
        pwu = w * (unsigned int)group_pairs + p;
    }
    __shared__ __attribute__((aligned(16))) float ltaps[256 + 32];
    const int t = threadIdx.x, k1 = (int)cbu * 256 + t, km = N1 - 1 - k1;
    ltaps[t] = taps[t];
    if (t < 32) ltaps[256 + t] = taps[256 + t];
    __syncthreads();
    const PWDesc d = pw[pwu];
    const float2 *Ua = U + (size_t)d.sw_a * pl.Zs, *Ub = U + (size_t)d.sw_b * pl.Zs;
    const int zpad = pl.zpad;
    auto row_at = [&](const float2 *base, int k2) { return base + (size_t)k2 * N1 + (size_t)(k2 >> 8) * zpad; };
    const float invNc = 1.0f / (float)pl.Nc;

    float2 at[S], ab[S];        // at[s]: output g + C - s of column k1;  ab[u]: output gb - (C - 1) + u of column km
#pragma unroll
    for (int s = 0; s < S; s++) at[s] = ab[s] = make_float2(0.0f, 0.0f);

    // row 0 pairs inside itself: (0, k1) with (0, 4096 - k1); bin 0 carries (A+[0], A-[0]).  Only the downward walk has it.
    {
        const float2 *ra = row_at(Ua, 0), *rb = row_at(Ub, 0);
        const int kp = (N1 - k1) & (N1 - 1);
        float2 q, qm;
        pair_u_pk(ra[k1], ra[kp], rb[k1], rb[kp], unit_root((float)k1 * (float)N2, invNc, false), k1 == 0, q, qm);
#pragma unroll
        for (int s = 0; s < S; s++) {
            const float h = taps[s];
            at[s].x += h * q.x;
            at[s].y += h * q.y;
        }
    }
    const size_t rc = (size_t)(pl.Nc / kDecD);
    float2 *g_top = G + (size_t)pwu * rc + k1, *g_bot = G + (size_t)pwu * rc + km;
    float2 *e_top = E + ((size_t)pwu * N1 + k1) * (2 * kDecEdge), *e_bot = E + ((size_t)pwu * N1 + km) * (2 * kDecEdge) + kDecEdge;
    e_top[0] = make_float2(0.0f, 0.0f);          // the output 6 places before a column gets nothing from it (|t| >= 96)

    // rows (k2, k1) of both stations and their partners (4096 - k2, 4095 - k1), through four per-lane pointers that move one
    // row per fetch (+ the plan's padding after every 256 rows).  The partner of row 0 is not a row of the walk: its slot
    // reads row 0 again (the value is dropped) and the pointers then jump to row 4095.
    const float2 *pta = row_at(Ua, 0) + k1, *ptb = row_at(Ub, 0) + k1, *pma = row_at(Ua, 0) + km, *pmb = row_at(Ub, 0) + km;
    auto fetch_row = [&](float2 (&dst)[4], int k2) {                // row k2, then on to row k2 + 1
        dst[0] = *pta;
        dst[1] = *pma;
        dst[2] = *ptb;
        dst[3] = *pmb;
        const long long dt = N1 + ((k2 & 255) == 255 ? zpad : 0);   // row k2 + 1 starts a block of 256 rows
        long long dm = -(long long)N1 - ((k2 & 255) == 0 ? zpad : 0);   // row 4096 - k2 ends one (going down)
        if (k2 == 0) dm = (long long)(N2 - 1) * N1 + (long long)((N2 - 1) >> 8) * zpad;
        pta += dt;
        ptb += dt;
        pma += dm;
        pmb += dm;
    };
    // One row: K3 (pair_u_pk) on the four values, then the twelve multiply-adds of either walk.  The taps of a row are the
    // same for every lane: three ds_read_b128 at a wave-uniform address that depends on the row's phase -- a run-time value
    // here, because the loop below is NOT unrolled over a group's 16 phases: unrolled, every form of the tap fetch (scalar
    // loads, LDS reads) was hoisted to the top of the group by the compiler and the 192 values spilled -- SGPRs into vector
    // lanes (880 v_readlane / v_writelane per group), VGPRs into scratch.
    //   h[s] = tap of (phase p, step s): downward walk, row (g, p) -> output g + C - s in at[s].
    //   upward walk: row 4096 - 16 g - p of column km.  p > 0: phase 16 - p of its group gb = 255 - g; slot u holds output
    //   gb - (C - 1) + u, step S - 1 - u, and tap(16 - p, S - 1 - u) = h[-(16 (u - C) + p)] = tap(p, u) by the filter's
    //   symmetry: the SAME twelve.  p = 0: phase 0 of group 256 - g, the last row of that group: tap(0, S - 1 - u), then the
    //   group's finished output 262 - g leaves and the slots move up.
    float2 wg = make_float2(1.0f, 0.0f);
    auto row = [&](const float2 (&v)[4], int k2, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;            // the only row of an iteration whose phase can be 0
        const int p = k2 & 15, g = k2 >> 4;
        const float4 *tp = reinterpret_cast<const float4 *>(ltaps + 16 * p);
        const float4 h0 = tp[0], h1 = tp[1], h2 = tp[2];
        const float h[12] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w};
        static_assert(S <= 12, "three float4 of taps per phase");
        const float2 rot = *reinterpret_cast<const float2 *>(ltaps + 256 + 2 * p);
        float2 q, qm;
        pair_u_pk(v[0], v[1], v[2], v[3], cmul(wg, rot), false, q, qm);
        if (FIRST && k2 == 0) q = qm = make_float2(0.0f, 0.0f);     // row 0: done above
#pragma unroll
        for (int s = 0; s < S; s++) {
            at[s].x += h[s] * q.x;
            at[s].y += h[s] * q.y;
        }
        if (!FIRST || p != 0) {
#pragma unroll
            for (int u = 0; u < S; u++) {
                ab[u].x += h[u] * qm.x;
                ab[u].y += h[u] * qm.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < S; u++) {
                ab[u].x += h[S - 1 - u] * qm.x;
                ab[u].y += h[S - 1 - u] * qm.y;
            }
            if (g > 0) {
                const int i = 256 + C - g;
                if (i >= 256) e_bot[i - 256] = ab[S - 1];
                else g_bot[(size_t)i * N1] = ab[S - 1];
#pragma unroll
                for (int u = S - 1; u > 0; u--) ab[u] = ab[u - 1];
                ab[0] = make_float2(0.0f, 0.0f);
            }
        }
    };
    float2 bufa[B][4], bufb[B][4];
#pragma unroll
    for (int r = 0; r < B; r++) fetch_row(bufa[r], r);
#pragma unroll 1
    for (int k2 = 0; k2 < 16 * kGroups; k2 += 2 * B) {
        if ((k2 & 15) == 0) wg = unit_root((float)k2 + (float)k1 * (float)N2, invNc, false);     // W_N^(16 g + 4096 k1): < 2^24, exact
#pragma unroll
        for (int r = 0; r < B; r++) fetch_row(bufb[r], k2 + B + r);
        row(bufa[0], k2, std::true_type{});
#pragma unroll
        for (int r = 1; r < B; r++) row(bufa[r], k2 + r, std::false_type{});
#pragma unroll
        for (int r = 0; r < B; r++) fetch_row(bufa[r], k2 + 2 * B + r);      // (past the last group: rows inside the spectrum, never used)
#pragma unroll
        for (int r = 0; r < B; r++) row(bufb[r], k2 + B + r, std::false_type{});
        if ((k2 & 15) == 16 - 2 * B) {                              // group g of the downward walk is complete: output g - 5
            const int i = (k2 >> 4) - (S - 1 - C);
            if (i < 0) e_top[kDecEdge + i] = at[S - 1];
            else if (i < 128) g_top[(size_t)i * N1] = at[S - 1];
#pragma unroll
            for (int s = S - 1; s > 0; s--) at[s] = at[s - 1];
            at[0] = make_float2(0.0f, 0.0f);
        }
    }
    // output 128 of column km: its last contributing row would be phase 0 of group 122, whose tap (t = -96) is zero
    g_bot[(size_t)128 * N1] = ab[S - 1];
}

}  // namespace tdoa
