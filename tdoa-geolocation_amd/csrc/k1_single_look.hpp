// k1_single_look.hpp -- K1 with every capture byte read ONCE (round 4): no statistics pre-pass.
//
// The fused column kernels (fft_radix16.hpp) used to need the window's mean and scale BEFORE they could transform
// v_i = (code_i - mean) scale, which is what k_fm_demod<false> was for: a second look-up of every sample (9 % of a cfg2
// step's HBM traffic, 12.7 % of its time).  Correlation is bilinear, so the exact mean and scale can come afterwards:
//   * k_once_estimate looks at 4096 samples of every station-window (256 evenly spaced runs of 16) and fixes an INTEGER
//     estimate m0 of the mean code and a power of two s0 ~ 1/sigma; k_once_edges leaves the prefix sums of the first and
//     the last K samples of  w_i = (code_i - m0) s0  (head[k] = sum_{i<k} w_i, tail[k] = sum_{i>=L-k} w_i, k <= K = the
//     largest |lag| searched);
//   * the column kernels transform w (same code path as before: their `mean` is m0, their `scale` s0) and add up the
//     EXACT integer window sums S1 = sum code, S2 = sum code^2 of the samples they look up anyway (float64 accumulators
//     that cannot round: see col_once_accumulate), one record per tile -- no atomics, no zeroing;
//   * k_once_final adds the tile records in integers and evaluates mean and scale with the formulas of k_fm_stats_final
//     (the statistics are bit-identical to the pre-pass's), plus eps = (mean - m0) s0, the mean of w, and g = scale / s0;
//   * with v = g (w - eps), W = sum_i w_i (EXACT: s0 (S1 - L m0) from the integer window sum) and windows of equal length L
//     the correlation at lag d is
//       sum_i v^t_i v^s_{i+d} = g_t g_s [ C_w(d) - eps_s W_t - eps_t W_s + eps_t eps_s (L - |d|) + eps_s X_t(|d|) + eps_t Y_s(|d|) ],
//     (X, Y) = (tail_t, head_s) for d >= 0 and (head_t, tail_s) for d < 0  (once_correction; derivation in DESIGN.md
//     section 3): the K5 kernels add the bracket's correction terms to every candidate before it enters the argmax, and
//     k_decode_peaks multiplies the winner by g_t g_s next to the slot's scale.  eps is what the pre-pass path subtracts --
//     (float32(mean) - m0) s0, the ROUNDED mean -- so W is not L eps: it differs by L s0 (float32(mean) - mean), first order
//     in the mean's rounding, which a carrier with a large frequency offset (|mean| >> sigma) makes visible (round 5; rounds
//     up to 4 used W = L eps).
// Nothing here is approximate: m0 and s0 only decide how large the removed terms are (|eps| ~ 1/64 on noise-like codes).
// Applies when every pair-window has two windows of the same length (always true for tdoa_process; pair calls with
// n1 = n2), K < L/2, and the peak is picked by k_small_col_peak or the pruned column kernels; every other case keeps the
// pre-pass.  TDOA_NO_K1_ONCE=1 / TDOA_DEBUG_NO_K1_ONCE: pre-pass everywhere.
#pragma once

#include "k1_discriminator.hpp"

namespace tdoa {

constexpr int kOnceRuns = 256;         // sampled runs per window (k_once_estimate: one per thread)
constexpr int kOnceRun = 16;           // samples per run
constexpr int kOncePieceLog = 11;      // k_once_edges: a wave takes 2048 consecutive positions of a region
constexpr int kOncePiece = 1 << kOncePieceLog;
constexpr int kOnceMaxPieces = 16;     // K + 1 <= 16 x 2048 (host checks)

struct OnceFin {           // per station-window, written by k_once_final
    double eps;            // (float32(mean) - m0) s0: what the pre-pass path subtracts from w = (code - m0) s0
    double gain;           // true scale / s0
    double wsum;           // W = sum of w over the window = s0 (S1 - L m0), exact
    double pad;
};

struct OnceCorr {          // what a K5 kernel needs to correct its candidates (by value; fin == nullptr: no correction)
    const float *edges;    // [n_sw][2][k1]: running sums of w over the first K (+ 1) and over the last K (+ 1) samples
    const OnceFin *fin;    // [n_sw]
    double *slot_gain;     // [slots]: g_t g_s of the pair-window, published for k_decode_peaks / k_decode_fine
    int k1;                // entries per array (K + 1 rounded up to a multiple of 4)
    int k_max;             // K
    float raw_per_unit;    // raw units of the kernel's values per unit of sum w w  (4 N for every inverse here)
};

// A pair-window's constants, set up once per thread.  With head(k) = H[k], tail(k) = T[K] - T[K - k] (H, T the running
// sums of the head and of the tail region) the term of lag d is, for either sign of d,
//   t0 + a |d| + bx E[off_x - d] + cy E[off_y + d]          (E = the edges buffer)
//   d >= 0:  off_x = T_t + K, bx = -b, off_y = H_s,     cy = +c, t0 = z + b T_t[K]
//   d <  0:  off_x = H_t,     bx = +b, off_y = T_s + K, cy = -c, t0 = z + c T_s[K]
// a = -eps_t eps_s raw, b = eps_s raw, c = eps_t raw, z = -(a L + b W_t + c W_s) (float64, rounded once): two loads and four
// arithmetic instructions per candidate.
// The term is small against what it is added to (|eps| ~ 1/60: a few hundred units where the noise floor of C_w is
// ~1400 and its peak >= 7000), so float32 evaluates it to ~1e-7 of ITSELF.
struct OnceSide {
    int off_x, off_y;
    float t0, bx, cy;
};
struct OncePair {
    OnceSide pos, neg;
    float a;
};

__device__ __forceinline__ OncePair once_pair(const OnceCorr &oc, const PWDesc &p)
{
    OncePair r;
    const double et = oc.fin[p.sw_a].eps, es = oc.fin[p.sw_b].eps, raw = (double)oc.raw_per_unit;
    const float b = (float)(es * raw), c = (float)(et * raw);
    r.a = (float)(-et * es * raw);
    const int head_t = p.sw_a * 2 * oc.k1, tail_t = head_t + oc.k1, head_s = p.sw_b * 2 * oc.k1, tail_s = head_s + oc.k1;
    // the lag-independent part: eps_t eps_s L - eps_s W_t - eps_t W_s  (= -eps_t eps_s L when W = L eps)
    const float al = (float)(raw * (et * es * (double)p.len_a - es * oc.fin[p.sw_a].wsum - et * oc.fin[p.sw_b].wsum));
    r.pos.off_x = tail_t + oc.k_max;
    r.pos.bx = -b;
    r.pos.off_y = head_s;
    r.pos.cy = c;
    r.pos.t0 = __builtin_fmaf(b, oc.edges[tail_t + oc.k_max], al);
    r.neg.off_x = head_t;
    r.neg.bx = b;
    r.neg.off_y = tail_s + oc.k_max;
    r.neg.cy = -c;
    r.neg.t0 = __builtin_fmaf(c, oc.edges[tail_s + oc.k_max], al);
    return r;
}

// the additive term of a candidate at lag d on side sd (pos for d >= 0, neg for d < 0), in the kernel's raw units.
// Safe for ANY d of that sign (it is clamped to +-K: the value is then unused): a kernel evaluates the terms of all its
// candidates first -- every load in flight at once -- and only then filters the lags.
__device__ __forceinline__ float once_term(const float *edges, const OnceSide &sd, float a, int d, int k_max)
{
    const int dc = d < -k_max ? -k_max : d > k_max ? k_max : d;      // v_med3_i32
    const float xr = edges[sd.off_x - dc], yr = edges[sd.off_y + dc];
    return __builtin_fmaf(sd.cy, yr, __builtin_fmaf(sd.bx, xr, __builtin_fmaf(a, __builtin_fabsf((float)dc), sd.t0)));
}

__device__ __forceinline__ float once_correction(const OnceCorr &oc, const OncePair &r, long long d)
{
    const int di = (int)(d < -oc.k_max ? -oc.k_max : d > oc.k_max ? oc.k_max : d);
    return once_term(oc.edges, di >= 0 ? r.pos : r.neg, r.a, di, oc.k_max);
}

// the block that handles a pair-window's first columns publishes g_t g_s for the decode kernels
__device__ __forceinline__ void once_publish_gain(const OnceCorr &oc, const PWDesc &p)
{
    oc.slot_gain[p.out_index] = oc.fin[p.sw_a].gain * oc.fin[p.sw_b].gain;
}

// ---- column-kernel side ---------------------------------------------------------------------------------------------
// A thread adds the stored codes it forms (st = -256 code, |st| <= 2^31) into two float64 accumulators per tile:
// t1 = sum st (|.| <= 32 x 2^31: exact) and t2 = sum st^2 = 2^16 sum code^2 (code^2 < 2^46 has 46 significant bits, 32 of
// them < 2^51: exact).  v_cvt_f64_i32 + v_add_f64 + v_fma_f64: three instructions per sample, none of them rounds.
__device__ __forceinline__ void col_once_accumulate(int st, double &t1, double &t2)
{
    const double c = (double)st;
    t1 += c;
    t2 = __builtin_fma(c, c, t2);
}

// x + (x of another lane), both halves moved by DPP; lanes outside ROW_MASK add 0
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_add_f64(double x)
{
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return x + __hiloint2double(hi2, lo2);
}

// sum over each ROW of 16 lanes of a wave, valid in the row's last lane (15, 31, 47, 63): four steps inside the rows.  Exact as
// long as every row's sum is (every caller splits its values so that it is).
__device__ __forceinline__ double row_sum_f64_lane15(double x)
{
    x = dpp_add_f64<0xB1>(x);                // quad_perm [1,0,3,2]
    x = dpp_add_f64<0x4E>(x);                // quad_perm [2,3,0,1]
    x = dpp_add_f64<0x141>(x);               // row_half_mirror
    x = dpp_add_f64<0x140>(x);               // row_mirror
    return x;
}

// A wave's records for one tile: ONE PER ROW OF 16 LANES, three float64 that hold exact integers.  A lane's t2 = 2^16 q with
// q < 2^51 an integer, and the lanes of a row would exceed 2^53, so q is split first: hi = floor(q / 2^26) < 2^25, lo = q - 2^26 hi
// < 2^26; over 16 lanes the parts stay below 2^29 and 2^30.  t1 = -256 x (a lane's sum of codes), |sum over a row| <= 16 x 2^36.
// The rows' last lanes write the four records straight to memory in one store instruction -- no LDS, no barrier, nothing for
// another wave to wait for; k_once_final adds a window's records up.  (Rounds 4-5 reduced over the whole wave first -- two more
// DPP steps, row_bcast:15 and :31, on each of the three values: 18 vector instructions per tile and wave in a kernel that is
// short of exactly those -- and wrote one record per wave; four records per wave are 40 MB per cfg2 step instead of 10.)
constexpr int kOnceRecordsPerWave = 4;
constexpr int kOnceWavesPerTile = 16 * kOnceRecordsPerWave;      // records per tile (sixteen waves)
struct OnceTile {
    double s1;             // sum st over a 16-lane row's part of the tile     (= -256 x sum code)
    double q_hi, q_lo;     // sum of floor(q / 2^26) and of q mod 2^26, q = sum code^2 of a lane
    double pad;
};

__device__ __forceinline__ void col_once_split(double t2, double &q_hi, double &q_lo)
{
    const double q = t2 * 0x1p-16;                       // exact
    q_hi = __builtin_floor(q * 0x1p-26);
    q_lo = __builtin_fma(q_hi, -0x1p26, q);              // exact
}

// rec: the wave's first record (kOnceRecordsPerWave consecutive ones)
__device__ __forceinline__ void col_once_wave_record(double t1, double t2, OnceTile *rec)
{
    double hi, lo;
    col_once_split(t2, hi, lo);
    t1 = row_sum_f64_lane15(t1);
    hi = row_sum_f64_lane15(hi);
    lo = row_sum_f64_lane15(lo);
    if ((threadIdx.x & 15) == 15) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 *o = reinterpret_cast<d2 *>(rec + ((threadIdx.x & 63) >> 4));
        o[0] = d2{t1, hi};
        o[1] = d2{lo, 0.0};
    }
}

// ---- k_once_estimate ------------------------------------------------------------------------------------------------
// (m0, s0) of every station-window from 256 evenly spaced runs of 16 samples: one 256-thread workgroup per window, a
// thread per run, angle look-ups in the direct half-plane table in global memory (17 independent gathers per thread, L2).
// m0 = the nearest integer to the sampled mean code, s0 = 2^-e with 2^e within sqrt(2) of the sampled sigma.
// Writes stats[w].mean = m0, stats[w].scale = s0 (what the column kernels normalise with).  len >= 64.
__global__ __launch_bounds__(256) void k_once_estimate(const SWDesc *sw, const int *dtable, FmStats *stats)
{
    __shared__ long long red[2][4];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const SWDesc d = sw[w];
    const int len = d.len;
    const gptr16 p = k1_global(d.base);
    const long long start = 1 + (long long)tid * (long long)(len - 1 - kOnceRun) / (kOnceRuns - 1);      // samples [start - 1, start + 16)
    unsigned int raw[9];
#pragma unroll
    for (int k = 0; k < 9; k++) raw[k] = k < 8 ? k1_fetch2(p, start - 1 + 2 * k) : (unsigned int)p[start + 15];
    int ang[18];
#pragma unroll
    for (int k = 0; k < 9; k++) k1_direct_angle2(raw[k] | (k == 8 ? 0x80000000u : 0u), dtable, ang[2 * k], ang[2 * k + 1]);
    long long a1 = 0, a2 = 0;
#pragma unroll
    for (int k = 1; k <= kOnceRun; k++) {
        const int c = -k1_stored_code(ang[k], ang[k - 1]);
        a1 += c;
        a2 += (long long)c * c;                        // < 2^46 x 16
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a1 += __shfl_xor(a1, off, kWave);
        a2 += __shfl_xor(a2, off, kWave);
    }
    if (lane == 0) { red[0][wv] = a1; red[1][wv] = a2; }
    __syncthreads();
    if (tid == 0) {
        const long long sum1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const long long sum2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];      // < 2^46 x 4096
        const double n_s = (double)(kOnceRuns * kOnceRun);
        const double mean = (double)sum1 / n_s;
        const double var = (double)sum2 / n_s - mean * mean;
        int e = 0;
        if (var > 1.0) {                                  // s0 = 2^-e with 2^e within a factor sqrt(2) of sigma
            const int ex = (int)((__double_as_longlong(var) >> 52) & 0x7ff) - 1023;      // floor(log2(var))
            e = (ex + 1) >> 1;
            e = e < 0 ? 0 : e > 24 ? 24 : e;
        }
        FmStats out;
        out.s1 = 0;
        out.s2_lo = 0;
        out.s2_hi = 0;
        out.mean = (float)(int)llrint(mean);               // |m0| <= 2^23: exact
        out.scale = __int_as_float((127 - e) << 23);
        stats[w] = out;
    }
}

// ---- k_once_edges ---------------------------------------------------------------------------------------------------
// Running sums of w = (code - m0) s0 over the head region (samples 0 .. K) and the tail region (samples len - K .. len) of
// every station-window, in the streaming style of k_fm_demod: persistent 1024-thread workgroups keep the direct angle
// table in LDS (128 KB).  A workgroup takes one REGION at a time; wave q < pieces takes its q-th PIECE of 2048 consecutive
// positions: a lane owns 32 consecutive samples (four 16-byte loads, issued before any is used), looks each sample up
// once (the angle before its run is the last angle of the lane to its left), adds its 32 values of (code - m0) up, a scan
// over the lanes' totals gives every position its running sum inside the piece, the pieces' totals meet in LDS, and
//   edges[w][side][i] = s0 * sum of (code - m0) over the region's positions [0, i)     (float32, i = 0 .. K)
// goes out.  Position K of a region has no code (it only closes the last running sum).  items = n_sw x 2.
__global__ __launch_bounds__(kDemodThreads) void k_once_edges(const SWDesc *sw, int n_sw, const int *dtable, const FmStats *stats,
                                                              float *edges, int k_max, int k1, int pieces)
{
    extern __shared__ int dlut[];                // kK1DirectEntries angles (128 KB: one workgroup per CU)
    __shared__ double piece_tot[kDemodThreads / kWave];
    for (int k = threadIdx.x; k < kK1DirectEntries / 4; k += kDemodThreads)
        reinterpret_cast<int4 *>(dlut)[k] = reinterpret_cast<const int4 *>(dtable)[k];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), piece = threadIdx.x / kWave;
    for (int item = blockIdx.x; item < 2 * n_sw; item += gridDim.x) {
        const int side = item & 1, w = item >> 1;
        const SWDesc d = sw[w];
        const int len = d.len;
        const gptr16 p = k1_global(d.base);
        const int m0 = (int)stats[w].mean;
        const double s0 = (double)stats[w].scale;
        const int region = side ? len - k_max : 0;          // first sample of the region
        const int pos0 = piece * kOncePiece + lane * 32;     // the lane's first position
        const int i0 = region + pos0;                        // ... and sample
        int c[32];
        int tot = 0;
        double incl = 0.0;
        if (piece < pieces) {
            // A lane's 32 positions are either all codes of the region (pos0 + 32 <= K: the vector path -- every lane of
            // every piece but the one that holds position K), all beyond it (zeros), or -- one lane per region -- straddle K:
            // sample by sample, never reading beyond the window.  The window's first sample has no predecessor:
            // code_0 := code_1.
            const bool whole = pos0 + 32 <= k_max, partial = !whole && pos0 < k_max;
            uint4 qs[4] = {};
            if (whole) {
#pragma unroll
                for (int h = 0; h < 4; h++) qs[h] = k1_fetch8(p, i0 + 8 * h);
            }
            const int first = region + piece * kOncePiece;         // the piece's first sample
            const unsigned int before = p[first > 0 ? first - 1 : 0];
            int a[33];
#pragma unroll
            for (int h = 0; h < 4; h++) {
                k1_direct_angle2(qs[h].x, dlut, a[8 * h + 1], a[8 * h + 2]);
                k1_direct_angle2(qs[h].y, dlut, a[8 * h + 3], a[8 * h + 4]);
                k1_direct_angle2(qs[h].z, dlut, a[8 * h + 5], a[8 * h + 6]);
                k1_direct_angle2(qs[h].w, dlut, a[8 * h + 7], a[8 * h + 8]);
            }
            const int left = wave_shift_right1(a[32]);             // (a lane left of a `whole` lane is whole)
            a[0] = lane ? left : k1_direct_angle(before, dlut);
#pragma unroll
            for (int k = 0; k < 32; k++) c[k] = whole ? -k1_stored_code(a[k + 1], a[k]) - m0 : 0;
            if (i0 == 0 && whole) c[0] = c[1];                     // code_0 := code_1
            if (__builtin_amdgcn_ballot_w64(partial)) {
                if (partial) {
#pragma unroll 1
                    for (int k = 0; k < 32; k++) {
                        const int i = i0 + k;
                        int v = 0;
                        if (pos0 + k < k_max && i < len) {
                            const int ii = i == 0 ? 1 : i;
                            v = -k1_stored_code(k1_direct_angle(p[ii], dlut), k1_direct_angle(p[ii - 1], dlut)) - m0;
                        }
                        c[k] = v;
                    }
                }
            }
            // running sums inside the lane (|code - m0| < 2^24, 32 of them: int32), then over the lanes (exact in float64)
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const int v = c[k];
                c[k] = tot;                                    // exclusive
                tot += v;
            }
            incl = (double)tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double t = __shfl_up(incl, off, kWave);
                if (lane >= off) incl += t;
            }
        }
        if (lane == 63) piece_tot[piece] = incl;               // (0 for the waves without a piece)
        __syncthreads();
        if (piece < pieces) {
            double base = incl - (double)tot;
            for (int q = 0; q < piece; q++) base += piece_tot[q];
            float *out = edges + ((size_t)w * 2 + side) * k1 + pos0;
            if (pos0 + 32 <= k1) {
#pragma unroll
                for (int h = 0; h < 8; h++) {
                    float4 o;
                    o.x = (float)((base + (double)c[4 * h]) * s0);
                    o.y = (float)((base + (double)c[4 * h + 1]) * s0);
                    o.z = (float)((base + (double)c[4 * h + 2]) * s0);
                    o.w = (float)((base + (double)c[4 * h + 3]) * s0);
                    reinterpret_cast<float4 *>(out)[h] = o;
                }
            } else {
                for (int k = 0; k < 32; k++)
                    if (pos0 + k < k1) out[k] = (float)((base + (double)c[k]) * s0);
            }
        }
        __syncthreads();                                       // piece_tot is rewritten by the next item
    }
}

// ---- k_once_final ---------------------------------------------------------------------------------------------------
// One 256-thread workgroup per station-window adds its tiles' records (integers; any order gives the same bits) and evaluates the
// statistics with the expressions of k_fm_stats_final, then eps and g against the (m0, s0) the column kernel used.
// tiles: the records of tile (w, a, bx) at index (((w G + a) nbx + bx) 16 + wave) 4 + row of 16 lanes, i.e. 64 x tiles consecutive records
// per window (tiles_per_sw counts records here).
__global__ __launch_bounds__(256) void k_once_final(const SWDesc *sw, const OnceTile *tiles, int tiles_per_sw, FmStats *stats,
                                                    OnceFin *fin, int n_sw)
{
#pragma clang fp contract(off)
    __shared__ long long red1[4];
    __shared__ unsigned long long red2[4], red3[4];
    const int w = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (w >= n_sw) return;
    const OnceTile *t = tiles + (size_t)w * tiles_per_sw;
    long long s1 = 0;
    unsigned long long hi = 0, lo = 0;
    // records as 16-byte pairs, four in flight per thread (a ten-second window has 40 960 of them: read one double at a time by
    // one workgroup they cost 0.4 ms per cfg3 step)
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 *t2 = reinterpret_cast<const d2 *>(t);
    int k = threadIdx.x;
    for (; k + 768 < tiles_per_sw; k += 1024) {
        d2 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a[u] = __builtin_nontemporal_load(t2 + 2 * (size_t)(k + 256 * u));
            b[u] = __builtin_nontemporal_load(t2 + 2 * (size_t)(k + 256 * u) + 1);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s1 += (long long)a[u].x;
            hi += (unsigned long long)a[u].y;
            lo += (unsigned long long)b[u].x;
        }
    }
    for (; k < tiles_per_sw; k += 256) {
        s1 += (long long)t[k].s1;
        hi += (unsigned long long)t[k].q_hi;
        lo += (unsigned long long)t[k].q_lo;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, kWave);
        hi += __shfl_xor(hi, off, kWave);
        lo += __shfl_xor(lo, off, kWave);
    }
    if (lane == 0) { red1[wv] = s1; red2[wv] = hi; red3[wv] = lo; }
    __syncthreads();
    if (threadIdx.x) return;
    s1 = red1[0] + red1[1] + red1[2] + red1[3];
    hi = red2[0] + red2[1] + red2[2] + red2[3];
    lo = red3[0] + red3[1] + red3[2] + red3[3];
    // S1 = -(sum st) / 256; S2 = hi 2^26 + lo as a 128-bit integer
    const long long S1 = -(s1 / 256);
    const unsigned long long h_lo = hi << 26, h_hi = hi >> 38;
    const unsigned long long s2_lo = h_lo + lo;
    const unsigned long long s2_hi = h_hi + (s2_lo < h_lo ? 1ull : 0ull);
    const int len = sw[w].len;
    const double m0 = (double)stats[w].mean, s0 = (double)stats[w].scale;      // what k_once_edges chose
    FmStats out;
    out.s1 = S1;
    out.s2_lo = s2_lo;
    out.s2_hi = s2_hi;
    OnceFin f;
    if (len == 0) {
        out.mean = 0.0f;
        out.scale = 1.0f;
        f.eps = 0.0;
        f.gain = 1.0;
        f.wsum = 0.0;
    } else {
        const double dn = (double)len;
        out.mean = (float)((double)S1 / dn);
        const double m2 = ((double)S1 * (double)S1) / dn;
        const double s2d = (double)((s2_hi << 32) | (s2_lo >> 32)) * 4294967296.0 + (double)(s2_lo & 0xffffffffull);
        const double var = (s2d - m2) / dn;
        out.scale = var > 0 ? (float)(1.0 / sqrt(var)) : 1.0f;
        // the pre-pass path transforms (f32(code) - mean) scale with the float32 mean and scale: the same two numbers here
        f.eps = ((double)out.mean - m0) * s0;
        f.gain = (double)out.scale / s0;
        f.wsum = ((double)S1 - dn * m0) * s0;                 // |S1|, L |m0| < 2^48: exact
    }
    f.pad = 0.0;
    stats[w] = out;
    fin[w] = f;
}

}  // namespace tdoa
