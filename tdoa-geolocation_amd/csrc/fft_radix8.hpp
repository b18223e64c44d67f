// fft_radix8.hpp -- 4096-point row transforms on 512 threads: 8 complex values per thread, radix-8 Stockham stages
// 8 x 8 x 8 x 8 with three LDS exchanges.  Same math and layouts as fft_radix16.hpp; what changes is the register
// footprint: a VALU-heavy kernel that holds several rows at once (the segment form's frames, the decimated inverse)
// fits 128 VGPRs on 512 threads and runs four waves per SIMD, where 256 threads x 16 values get two (one wave alone
// gets a vector instruction every 4 cycles, MI355X_MICROARCH.md).
//
// LDS images are 4096 float2 (32 KB) with an XOR swizzle instead of padding: element i lives at
//   swz(i) = i ^ ((i >> 5) & 7) ^ (((i >> 6) & 3) << 3)
// which only permutes the 32 elements of an aligned 256-byte line.  Every access pattern of the four stages then
// touches 32 distinct 8-byte bank pairs per half-wave: consecutive elements (the stage reads, j + 512 r), stride 8
// (stage-1 writes, 8 j + k), runs of 8 at stride 64 (stage-2 writes) and runs of 64 (stage-3 writes).
#pragma once

#include <type_traits>

#include "device_common.hpp"
#include "fft_stockham.hpp"
#include "fft_radix16.hpp"

namespace tdoa {

__device__ __forceinline__ constexpr int oreg8(int k) { return ((k & 1) << 2) | (k >> 1); }

// 8-point DFT in registers: natural order in, X[k] left in v[oreg8(k)].
// n = 4 n1 + n2, k = k1 + 2 k2:  radix-2 over n1, twiddle W8^(n2 k1), radix-4 over n2.
template <bool INV>
__device__ __forceinline__ void fft8(float2 (&v)[8])
{
    constexpr float h = 0.70710678118654752f;
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
        const float2 s = cadd(v[n2], v[n2 + 4]), d = csub(v[n2], v[n2 + 4]);
        v[n2] = s;
        v[n2 + 4] = d;
    }
    // v[4 + n2] *= W8^n2   (forward e^{-2 pi i n2/8}; inverse: conjugate)
    {
        v[5] = cmul(v[5], make_float2(h, INV ? h : -h));      // W8^1 = h (1 -+ i)
        const float2 b = v[6];                                 // W8^2 = -+ i
        v[6] = INV ? make_float2(-b.y, b.x) : make_float2(b.y, -b.x);
        v[7] = cmul(v[7], make_float2(-h, INV ? h : -h));     // W8^3 = h (-1 -+ i)
    }
    bfly4<INV>(v[0], v[1], v[2], v[3]);
    bfly4<INV>(v[4], v[5], v[6], v[7]);
}

// v[r] *= w^r, r = 1..7
__device__ __forceinline__ void mul_powers8(float2 (&v)[8], float2 w)
{
    const float2 w2 = cmul(w, w), w3 = cmul(w2, w), w4 = cmul(w2, w2);
    v[1] = cmul(v[1], w);
    v[2] = cmul(v[2], w2);
    v[3] = cmul(v[3], w3);
    v[4] = cmul(v[4], w4);
    v[5] = cmul(v[5], cmul(w4, w));
    v[6] = cmul(v[6], cmul(w4, w2));
    v[7] = cmul(v[7], cmul(w4, w3));
}

// X[k] (in v[oreg8(k)]) *= base * step^k
__device__ __forceinline__ void mul_base_step8(float2 (&v)[8], float2 base, float2 step)
{
    const float2 s2 = cmul(step, step), s3 = cmul(s2, step), s4 = cmul(s2, s2);
    const float2 g = cmul(base, s4);
    v[oreg8(0)] = cmul(v[oreg8(0)], base);
    v[oreg8(1)] = cmul(v[oreg8(1)], cmul(base, step));
    v[oreg8(2)] = cmul(v[oreg8(2)], cmul(base, s2));
    v[oreg8(3)] = cmul(v[oreg8(3)], cmul(base, s3));
    v[oreg8(4)] = cmul(v[oreg8(4)], g);
    v[oreg8(5)] = cmul(v[oreg8(5)], cmul(g, step));
    v[oreg8(6)] = cmul(v[oreg8(6)], cmul(g, s2));
    v[oreg8(7)] = cmul(v[oreg8(7)], cmul(g, s3));
}

__device__ __forceinline__ int swz(int i) { return i ^ ((i >> 5) & 7) ^ (((i >> 6) & 3) << 3); }

// Left alone, the backend pairs the 8-byte LDS accesses of a stage into ds_read2st64_b64 / ds_write2st64_b64, which the
// LDS serves at half the rate of plain b64 (MI355X_MICROARCH.md, LDS table) and with a banking the swizzle was not
// built for.  The subtarget feature only exists in the device pass.
#if defined(__HIP_DEVICE_COMPILE__)
#define TDOA_PLAIN_DS_OPS , target("no-load-store-opt")
#else
#define TDOA_PLAIN_DS_OPS
#endif

// stored phase code at a 32-bit unsigned byte offset from a uniform row pointer (global_load_dword v, v_off, s[base])
__device__ __forceinline__ int code_at(const int *row, unsigned byte_off)
{
    return *reinterpret_cast<const int *>(reinterpret_cast<const char *>(row) + byte_off);
}

// a station-window's normalisation in the segment kernels: two instructions per code (v_cvt_f32_i32 + v_fma_f32)
struct SegNorm {
    float ns, ns_unit, off;
    __device__ __forceinline__ SegNorm(const FmStats &st, float unit)
    {
#pragma clang fp contract(off)
        ns = -st.scale;
        ns_unit = ns * unit;
        off = -(st.mean * st.scale);
    }
    __device__ __forceinline__ float fast(int raw) const { return __builtin_fmaf((float)raw, ns_unit, off); }     // raw in units of `unit`
    __device__ __forceinline__ float plain(int stored) const { return __builtin_fmaf((float)stored, ns, off); }
};

// a segment kernel's view of a code row: int32 codes (4 bytes) or packed ones (3 bytes, k1_store8_packed)
template <bool PACK3>
struct SegCodes {
    static constexpr unsigned int kBytes = PACK3 ? 3u : 4u;
    const unsigned char *row;
    __device__ __forceinline__ SegCodes(const int *codes, int sw_index, long long code_stride)
        : row(reinterpret_cast<const unsigned char *>(codes) + (size_t)kBytes * (size_t)sw_index * (size_t)code_stride) {}
    // code at a 32-bit unsigned BYTE offset (= kBytes x sample index) from the uniform row pointer
    __device__ __forceinline__ int at(unsigned int byte_off) const
    {
        return PACK3 ? k1_code3_at(row, byte_off) : code_at(reinterpret_cast<const int *>(row), byte_off);
    }
    __device__ __forceinline__ int operator[](int i) const { return at(kBytes * (unsigned int)i); }
    // The fast path of a frame: the four lanes of a quad want four CONSECUTIVE codes whose first index is a multiple of 4 --
    // on packed rows three aligned dwords W0 W1 W2.  Lane q loads W[min(q, 2)], takes its left neighbour's dword by DPP
    // (lane 0: its own) and one v_perm_b32 puts the three bytes of its code (bytes 3 q .. 3 q + 2 of W2:W1:W0) above a zero
    // byte: 256 x code, sign in place -- the factor goes into the window's scale (unit()).  One aligned, coalesced load + two
    // vector instructions per code.  (A lane-private unaligned dword load at 3 i, the obvious form, made the quad kernel
    // 13 % slower than int32 rows; DPP + v_alignbit + v_bfe 8 % slower: the kernel is short of issue slots, not of bytes.)
    //   lane_off(t) = 12 (t >> 2) + 4 min(t & 3, 2)  [+ 3 x first sample of the frame, + 1536 per row of 512 positions]
    //   lane_sel(t): v_perm selector, bytes 0-3 = the neighbour's dword, 4-7 = the lane's own, 0x0c = zero
    static constexpr float kUnit = PACK3 ? 1.0f / 256.0f : 1.0f;        // what quad_at's value counts in
    __device__ static __forceinline__ unsigned int lane_off(int t)
    {
        return PACK3 ? 12u * (unsigned int)(t >> 2) + 4u * (unsigned int)((t & 3) < 2 ? (t & 3) : 2) : 4u * (unsigned int)t;
    }
    __device__ static __forceinline__ unsigned int lane_sel(int t)
    {
        const int q = t & 3;
        return q == 0 ? 0x0201000cu : q == 1 ? 0x0504030cu : q == 2 ? 0x0403020cu : 0x0302010cu;
    }
    __device__ __forceinline__ int quad_at(unsigned int off, unsigned int sel) const
    {
        const int w = code_at(reinterpret_cast<const int *>(row), off);
        if (!PACK3) return w;
        const int left = __builtin_amdgcn_mov_dpp(w, 0x90, 0xf, 0xf, true);          // quad_perm [0, 0, 1, 2]
        return (int)__builtin_amdgcn_perm((unsigned int)w, (unsigned int)left, sel);
    }
};

constexpr int kRow8Lds = 4096;       // float2 per row image (no padding)

// Stages 2..4 of TWO 4096-point row transforms side by side (row x through image la, row y through lb), whose
// stage-1 outputs X[k] are still in registers (in v[oreg8(k)]): x belongs to stage-1 item jx, y to item jy (the
// caller may have run item 511 - t for the mirrored row); from stage 2 on thread j = threadIdx.x owns item j of
// both rows.  On return thread j holds Y[j + 512 k] of each row in [oreg8(k)].  512 threads; the images must be
// free to overwrite on entry; they are free again after the caller's next barrier.
// w2, w3, w4: the stage roots e^{-+2 pi i (j & 7)/64}, e^{-+2 pi i (j & 63)/512}, e^{-+2 pi i j/4096} of thread j
template <bool INV>
__device__ __forceinline__ void rows2_r8_finish_w(float2 (&x)[8], float2 (&y)[8], float2 *la, float2 *lb, const int j,
                                                  const int jx, const int jy, const float2 w2, const float2 w3,
                                                  const float2 w4)
{
    {
        // element 8 jj + k: (i >> 5) & 7 = (jj >> 2) & 7, (i >> 6) & 3 = (jj >> 3) & 3
        const int sx = ((jx >> 2) & 7) ^ (((jx >> 3) & 3) << 3), sy = ((jy >> 2) & 7) ^ (((jy >> 3) & 3) << 3);
        // every store address of a stage is one base XOR a compile-time constant.  The bases are laundered so that the
        // eight addresses are rebuilt where they are used (one v_xor each): hoisted out of a caller's loop as 3 x 8
        // invariants they would be spilled and reloaded from scratch on every trip.
        const int bx = opaque_i(((8 * jx) ^ (sx & 0x18)) | (sx & 7)), by = opaque_i(((8 * jy) ^ (sy & 0x18)) | (sy & 7));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            la[bx ^ k] = x[oreg8(k)];
            lb[by ^ k] = y[oreg8(k)];
        }
    }
    __syncthreads();
    const int rb = swz(j);              // element j + 512 r lives at swz(j) + 512 r
#pragma unroll
    for (int r = 0; r < 8; r++) {
        x[r] = la[rb + 512 * r];
        y[r] = lb[rb + 512 * r];
    }
    mul_powers8(x, w2);
    mul_powers8(y, w2);
    fft8<INV>(x);
    fft8<INV>(y);
    __syncthreads();
    {
        // swz(d + 8 k) = swz(d) ^ (k << 3) ^ (k >> 2) for d = 64 (j >> 3) + (j & 7)
        const int d = opaque_i(swz(((j >> 3) << 6) + (j & 7)));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int p = d ^ ((k << 3) ^ (k >> 2));
            la[p] = x[oreg8(k)];
            lb[p] = y[oreg8(k)];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        x[r] = la[rb + 512 * r];
        y[r] = lb[rb + 512 * r];
    }
    mul_powers8(x, w3);
    mul_powers8(y, w3);
    fft8<INV>(x);
    fft8<INV>(y);
    __syncthreads();
    {
        // swz(d + 64 k) = swz(d) ^ (k << 6) ^ ((k & 3) << 1) ^ ((k & 3) << 3) for d = 512 (j >> 6) + (j & 63)
        const int d = opaque_i(swz(((j >> 6) << 9) + (j & 63)));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int p = d ^ ((k << 6) ^ ((k & 3) << 1) ^ ((k & 3) << 3));
            la[p] = x[oreg8(k)];
            lb[p] = y[oreg8(k)];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        x[r] = la[rb + 512 * r];
        y[r] = lb[rb + 512 * r];
    }
    mul_powers8(x, w4);
    mul_powers8(y, w4);
    fft8<INV>(x);
    fft8<INV>(y);
}

// the same with the roots rebuilt on every call (opaque: see fft_radix16.hpp -- a kernel that transforms several row
// pairs must not keep 3 x 7 twiddle powers alive between the calls)
template <bool INV>
__device__ __forceinline__ void rows2_r8_finish(float2 (&x)[8], float2 (&y)[8], float2 *la, float2 *lb, const int j,
                                                const int jx, const int jy)
{
    rows2_r8_finish_w<INV>(x, y, la, lb, j, jx, jy, opaque(unit_root((float)(j & 7), 2.0f / 64.0f, INV)),
                           opaque(unit_root((float)(j & 63), 2.0f / 512.0f, INV)),
                           opaque(unit_root((float)j, 2.0f / 4096.0f, INV)));
}

// two inverse rows a, b (outputs of rows2_r8_finish<true>) -> four-step twiddle -> V[k2][n1]
__device__ __forceinline__ void inv_rows_twiddle_store(float2 (&va)[8], float2 (&vb)[8], const int t, const int a, const int b,
                                                       float2 *out, const FftPlan &pl)
{
    // V[k2][n1] = y[n1] * W_Nc^(-n1 k2), n1 = t + 512 k
    if (pl.odd == 1) {
        const float inv2 = 2.0f / (float)pl.Nc;
        const long long e0 = ((long long)t * a) & (pl.Nc - 1), e1 = ((long long)512 * a) & (pl.Nc - 1);
        mul_base_step8(va, unit_root((float)e0, inv2, true), unit_root((float)e1, inv2, true));
        const long long f0 = ((long long)t * b) & (pl.Nc - 1), f1 = ((long long)512 * b) & (pl.Nc - 1);
        mul_base_step8(vb, unit_root((float)f0, inv2, true), unit_root((float)f1, inv2, true));
    } else {      // the small plan of a 5 x 2^k transform (4096 x 160): t a, 512 a < 512 N2 < Nc -- no wrap
        const float qd = 0.25f * (float)pl.Nc, iq = 4.0f / (float)pl.Nc;
        mul_base_step8(va, unit_root_any((float)(t * a), qd, iq, true), unit_root_any((float)(512 * a), qd, iq, true));
        mul_base_step8(vb, unit_root_any((float)(t * b), qd, iq, true), unit_root_any((float)(512 * b), qd, iq, true));
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        out[(size_t)a * 4096 + t + 512 * k] = va[oreg8(k)];
        out[(size_t)b * 4096 + t + 512 * k] = vb[oreg8(k)];
    }
}

// (Rounds 1-2 also kept three opt-in pair-step variants here -- k_inv_row_pair_r8 (512 threads x 8 values),
// k_pair_rows_fused_r8 (forward rows inside the pair kernel) and k_rows_tri_fused (all six row transforms of a
// three-station window's row pair in one 1024-thread workgroup).  Each was parity-green and each lost to the two-kernel
// form on every BASELINE configuration (DESIGN.md section 3 keeps the measurements and the counters); they were
// removed in round 3 -- last present in commit 3ec148d.)

// ---------------------------------------------------------------------------
// Decimated inverse (N2 = 256 or 512; search ranges of a few ten thousand lags in a transform of millions).
// Only |lag| <= max_lag of the 2 Nc inverse outputs are wanted: ~1 % for the reference's 20 000 lags in N = 2^21.  The
// general form still runs the whole inverse (row pass: Q -> V, 8 Nc bytes written and read back per pair-window).
// Multiplying the lag sequence q[m] by a smooth window w[m] that vanishes outside |m| < R - M lets the SPECTRUM be
// decimated: G[j] = sum_t h[t] Q[16 j + t] (h = the transform of w: a Kaiser-windowed sinc, 213 real taps for 140 dB),
// and the R = Nc/16-point inverse of G is q[m] w[m] for |m| <= M, exactly up to the stop-band leakage of h (aliases of
// lags beyond R - M, measured 1e-7 of the peak on noise-level simulator peaks, 3e-10 on FM).  w[m] is known (the
// host evaluates it from the rounded taps) and is divided out.  So K3 is followed by a 13-tap-per-bin FIR instead of
// a 4096-point row transform, V shrinks 16 times, and the pair step is one streaming read of the two spectra.
//
// k = k2 + N2 k1: consecutive bins run down a column, so a workgroup takes a tile of all N2 rows x 4096/N2 columns
// (4096 consecutive bins; the forward row pass k_fwd_row4096_unpack writes the stations' UNPACKED spectra U tile by tile
// for this kernel -- the station's half of K3 is done there --, so a tile is one contiguous 32 KB run instead of N2 row
// pieces) and, because K3 needs U[k] and U[Nc - k] together, the mirrored tile:
//   tile A: bins [4096 bx - 112, 4096 (bx + 1) + 112), tile B: bins Nc - (those), both with their halos.
// Q of both tiles goes to LDS phase-major (bin o of a tile at [o & 15][o >> 4]) so that thread i of the FIR reads
// element i + const of one phase for every tap: conflict-free.  256 outputs per tile, one per thread.
// G is written as the [N2'][4096] four-step layout of the R-point inverse (j = j2 + N2' j1, N2' = N2/16).
// (A variant that kept the template's tiles in registers for up to four pair-windows sharing it read a third fewer
// bytes and ran 8 - 30 % slower: one more barrier pair per pair-window and 128 VGPRs with spills.)
// grid (N2/2, n_pw), 512 threads, dynamic LDS 2 x 16 x 296 x 8 B = 74 KB (+ 1 KB of taps).
// (Round 3 also measured a persistent form that prefetches the next item's spectrum tiles behind the FIR -- neutral on 3
// pairs, 4 % slower on the 4096 x 512 plan -- and a form whose FIR is wave-uniform in the phase, with scalar taps and the
// phase groups added through LDS: 1.35 ms against 0.79; neither is kept.  DESIGN.md section 3.)
// ---------------------------------------------------------------------------
constexpr int kDecD = 16;
// Taps per phase.  Rounds 2-3 ran 14 steps (213 taps, 140 dB on cfg2's transition band); 12 steps hold T <= 95, i.e. a
// Kaiser design of 126 dB there (the host takes what the band allows, up to 140 dB: tdoa_mi355x.hip ensure_decimation):
// the alias leakage on noise-level peaks goes from 1.5e-7 to 7e-7 of the peak (float64 restatement, tests/
// test_mode_b_anchors.py) for a seventh fewer multiply-adds and LDS reads.  TDOA_DEC_STEPS=14 rebuilds the old filter.
#ifndef TDOA_DEC_STEPS
#define TDOA_DEC_STEPS 12
#endif
constexpr int kDecSteps = TDOA_DEC_STEPS;                   // a tap t = 16 (s - C) + p: phase p = 0..15, step s = 0..kDecSteps-1
constexpr int kDecCentre = kDecSteps / 2;                   // C
constexpr int kDecTmax = 16 * kDecCentre - 1;               // |t| <= T <= kDecTmax
constexpr int kDecEdge = kDecCentre;                        // outputs on either side of a tile boundary that the other tile's bins reach
static_assert(kDecSteps % 2 == 0 && kDecSteps >= 4 && kDecSteps <= 14, "FIR geometry: 8 zero slots per side, 16 taps per phase in LDS");
// LDS image of a tile (4096 consecutive bins, no halo): bin o lives at [o & 15][dec_slot((o >> 4) + 8)],
// dec_slot(i) = i + ((i + 8) >> 4).  8 zero slots on either side stand for the neighbouring tiles (their share of an
// edge output is added by THEIR workgroup, see below); one pad slot per 16 (= per column of the tile) puts the 16 columns
// a wave stores side by side (256 bins = 16 slots apart) into 16 different bank pairs instead of 2.  Pitch 296 = 8 (mod 32):
// the four phase groups of a reading half-wave then fill the remaining bank pairs (17 g + 8 pq covers 0..31 once).
constexpr int kDecPitch = 296;                              // >= dec_slot(271) + 1 = 289
__device__ __forceinline__ constexpr int dec_slot(int i) { return i + ((i + 8) >> 4); }

// Tile A = bins [4096 bx, 4096 (bx + 1)), tile B = bins [Nc - 4096 (bx + 1), Nc - 4096 bx): bin o >= 1 of A and bin
// 4096 - o of B are partners (k, Nc - k); the partners of A[0] and B[0] lie in the neighbouring workgroups' tiles and
// are only read.  An output whose 2T+1 taps cross a tile boundary gets the far side's share from the workgroup that owns
// those bins: every tile also evaluates the kDecEdge + kDecEdge outputs just outside it over its own bins and leaves them in
// E[pw][tile][2 kDecEdge] (first half: the last outputs of the previous tile, second half: the first ones of the next);
// k_inv_rows_plain_r8 adds them when it loads G.  No halo is fetched: the halo bins of a [tile][k2][col] layout are 8-byte
// pieces of 224 different lines per tile side and spectrum, which nearly doubled the bytes this kernel pulled in.
template <int LOGN2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4) TDOA_PLAIN_DS_OPS)) void k_pair_decimate16(const PWDesc *pw, const float2 *Z, float2 *G, float2 *E, FftPlan pl,
                                                         const float *taps, int small_n2, int group_pairs, int n_pw, float2 rot)
{
    // group_pairs = 0: grid (N2/2, n_pw), workgroup (bx, pw).  group_pairs = P > 0 (every window of the batch carries the same P
    // pair-windows): 1-D grid; the P pair-windows of one (window, tile pair) all read the same station tiles, so they are given
    // consecutive slots of ONE XCD (workgroups go to the XCDs round-robin: L and L + 8 share an L2) -- a station's tile then
    // comes from memory for its first pair and from that XCD's L2 for the others (k_inv_row_pair4096 has the same scheme).
    unsigned int bxu = blockIdx.x, pwu = blockIdx.y;
    if (group_pairs > 0) {
        const unsigned int L = blockIdx.x, xcd = L & 7u, slot = L >> 3;
        const unsigned int g = slot / (unsigned int)group_pairs, p = slot % (unsigned int)group_pairs;
        const unsigned int Gi = g * 8u + xcd, tiles = (unsigned int)(pl.N2 >> 1);
        const unsigned int w = Gi / tiles;
        if (w * (unsigned int)group_pairs >= (unsigned int)n_pw) return;      // padding of the last round of 8 groups
        bxu = Gi % tiles;
        pwu = w * (unsigned int)group_pairs + p;
    }
    const unsigned int bx_ = bxu, pw_ = pwu;
    constexpr int N2 = 1 << LOGN2, COLS = 4096 >> LOGN2;     // a tile: all N2 rows x COLS columns = 4096 consecutive bins
    extern __shared__ float2 lds[];                          // [2][16][kDecPitch]
    // the FIR taps [phase][16] in LDS: a lane's phases p = pq + 4 m differ from its neighbours', so the taps cannot be scalar
    // operands, and fetched from global memory inside the FIR (rounds 1-2: 56 per-lane loads per thread, each a round trip
    // to L2 between the multiply-adds) they cost this kernel a fifth of its time: 0.96 -> 0.79 ms on cfg2
    __shared__ float ltaps[256];
    float2 *qa = lds, *qb = lds + 16 * kDecPitch;
    const int t = threadIdx.x;
    if (t < 256) ltaps[t] = taps[t];
    const PWDesc d = pw[pw_];
    const float2 *Za = Z + (size_t)d.sw_a * pl.Nc, *Zb = Z + (size_t)d.sw_b * pl.Nc;
    const long long mask = pl.Nc - 1;
    const long long kA0 = 4096ll * bx_;               // first bin of tile A
    const float invNc = 1.0f / (float)pl.Nc;
    // spectra in COLS-column tiles (k_fwd_row4096_unpack): element (k2, k1) at [k1 / COLS][k2][k1 % COLS]
    auto coords = [&](long long k, unsigned int &at, unsigned int &atm) {
        const int k2 = (int)(k & (N2 - 1)), k1 = (int)(k >> LOGN2);
        const int pr = (N2 - k2) & (N2 - 1), pc = ((k2 == 0 ? 4096 : 4095) - k1) & 4095;
        at = (unsigned int)(k1 / COLS) * 4096u + (unsigned int)k2 * COLS + (unsigned int)(k1 % COLS);
        atm = (unsigned int)(pc / COLS) * 4096u + (unsigned int)pr * COLS + (unsigned int)(pc % COLS);
    };
    auto slot_of = [](int o) { return (o & 15) * kDecPitch + dec_slot((o >> 4) + 8); };
    // zero slots on either side of both images: 16 phases x (8 + 8) x 2 tiles = 512 entries, one per thread
    {
        const int img = t >> 8, p = (t >> 4) & 15, z = t & 15;
        (img ? qb : qa)[p * kDecPitch + dec_slot(z < 8 ? z : 256 + z)] = make_float2(0.0f, 0.0f);
    }
    // B[0]: its partner is A[0] of the next workgroup's tile.  One thread (of the last wave) fetches the four values
    // up front, next to everybody's main loads, and forms Q after its main work
    const bool has_b0 = t == 511;
    const long long kb0 = (pl.Nc - 4096ll * ((long long)bx_ + 1)) & mask;
    float2 b0[4] = {};
    if (has_b0) {
        unsigned int at, atm;
        coords(kb0, at, atm);
        b0[0] = Za[at]; b0[1] = Za[atm]; b0[2] = Zb[at]; b0[3] = Zb[atm];
    }
    {
        // a tile's 4096 elements, 8 per thread, x = t + 512 it in tile order (k2 = x / COLS, column x % COLS: one
        // contiguous 4 KB run per trip); all 32 loads of a thread are issued first.  From trip 1 on everything is linear in
        // the trip (k2 >= 512 / COLS > 0: the partner's row just counts down, no wrap): own element + 512 it, partner
        // - 512 (it - 1); for N2 = 256 also the image slots (+- 2 per trip, inside one column of the image: constant pad).
        constexpr int DK = 512 / COLS;                             // bins between a thread's consecutive elements
        static_assert(DK % 16 == 0, "a thread's elements share their phase");
        float2 za[8], zam[8], zb[8], zbm[8];
        const int o0 = t / COLS + N2 * (t % COLS);                 // x -> bin offset k2 + N2 col
        unsigned int at0, atm0, at1, atm1;
        coords(kA0 + o0, at0, atm0);
        coords(kA0 + o0 + DK, at1, atm1);
        // (32-bit byte offsets from the two wave-uniform bases instead of 32 64-bit addresses)
        auto at = [](const float2 *base, unsigned int byte_off) {
            return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(base) + byte_off);
        };
        za[0] = at(Za, 8u * at0); zam[0] = at(Za, 8u * atm0); zb[0] = at(Zb, 8u * at0); zbm[0] = at(Zb, 8u * atm0);
#pragma unroll
        for (int it = 1; it < 8; it++) {
            za[it] = at(Za, 8u * at1 + 4096u * (it - 1)); zam[it] = at(Za, 8u * atm1 - 4096u * (it - 1));
            zb[it] = at(Zb, 8u * at1 + 4096u * (it - 1)); zbm[it] = at(Zb, 8u * atm1 - 4096u * (it - 1));
        }
        // w(k) = W_N^k; a thread's bins are DK apart: one root, then a fixed rotation (rot = W_N^DK, the same for every
        // thread of every workgroup: a kernel argument from the host)
        float2 w = unit_root((float)(kA0 + o0), invNc, false);
        const int sa1 = slot_of(o0 + DK), sb1 = slot_of(4096 - o0 - DK);
        {
            float2 q, qm;
            pair_u_pk(za[0], zam[0], zb[0], zbm[0], w, kA0 + o0 == 0, q, qm);
            qa[slot_of(o0)] = q;
            if (o0) qb[slot_of(4096 - o0)] = qm;                   // the partner of A[0] is B[0] of the workgroup before
            w = cmul(w, rot);
        }
#pragma unroll
        for (int it = 1; it < 8; it++) {
            float2 q, qm;
            pair_u_pk(za[it], zam[it], zb[it], zbm[it], w, false, q, qm);
            if constexpr (LOGN2 == 8) {                            // N2 = 256: a thread's slots stay inside one column of the image
                qa[sa1 + (DK / 16) * (it - 1)] = q;
                qb[sb1 - (DK / 16) * (it - 1)] = qm;
            } else {
                // N2 = 512 (COLS = 8, DK = 64): the phase stays and the slot index moves by 4 per trip, across the pad slot
                // that follows every 16 slots.  Where it crosses is known at compile time: with o0 = (t >> 3) + 512 (t & 7)
                // the slot index of A is = t >> 7 (0..3) mod 16 once the pad's 8 is added, so (ia + 8) >> 4 steps up exactly
                // at trip 4; B counts down from a slot that is 12..15 mod 16 -- one step down at trip 4 -- except for the
                // threads t < 8 (0 mod 16: one step down from trip 1 on, a second one from trip 5 on).  Two bases and
                // compile-time offsets instead of rebuilding the pad term per store (5 instructions x 14 stores).
                static_assert(LOGN2 == 9 && COLS == 8 && DK == 64, "slot walk of the 4096 x 512 plan");
                const int z = t < 8 ? 1 : 0;
                qa[sa1 + 4 * (it - 1) + (it >= 4 ? 1 : 0)] = q;
                qb[(it == 4 ? sb1 + z : sb1) - 4 * (it - 1) - (it >= 4 ? 1 : 0)] = qm;
            }
            w = cmul(w, rot);
        }
    }
    if (has_b0) {
        float2 q, qm;
        pair_u_pk(b0[0], b0[1], b0[2], b0[3], unit_root((float)kb0, invNc, false), false, q, qm);
        qb[slot_of(0)] = q;
    }
    __syncthreads();
    // FIR + decimation.  Output i of a tile is sum_t h[t] Q[16 i + t] = sum_{p, s} tab[p][s] img[p][i + s - C]
    // (t = 16 (s - C) + p, C = kDecCentre; the host lays the taps out as taps[p][16], s = 0..kDecSteps-1, zero where |t| > T).
    // A thread takes FOUR consecutive outputs and FOUR phases p = pq + 4 m: image slot 4 g + s' (counted from the 8 zero
    // slots) serves output o = 0..3 with step s = s' - o - (8 - C), so kDecSteps + 3 LDS reads per phase feed 4 kDecSteps multiply-adds
    // (one read per output and tap would make the kernel LDS-bound: 13 reads of every bin).  Waves 0..3: tile A, 4..7:
    // tile B; wave w takes the output groups g = 4 g_l + (w & 3), lane = 4 g_l + pq: physical slot of 4 g + s' is
    // 17 g_l + [4 wq + s' + ((4 wq + s' + 8) >> 4)], the bracket a compile-time offset once the wave's wq is fixed (four
    // copies of the loop, wave-uniform switch).
    const int tile = __builtin_amdgcn_readfirstlane(t >> 8), wq = __builtin_amdgcn_readfirstlane((t >> 6) & 3);
    const int pq = t & 3, gl = (t & 63) >> 2;
    const float2 *img = tile ? qb : qa;
    const float2 *src = img + 17 * gl;
    float2 acc[4];
#pragma unroll
    for (int o = 0; o < 4; o++) acc[o] = make_float2(0.0f, 0.0f);
    auto fir = [&](auto wq_c) {
        constexpr int WQ = decltype(wq_c)::value;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int p = pq + 4 * m;
            float h[kDecSteps];
#pragma unroll
            for (int s2 = 0; s2 < kDecSteps; s2++) h[s2] = ltaps[16 * p + s2];
            // two views of the row, the second laundered: adjacent slots then come from pointers the compiler cannot relate,
            // so it cannot pair them into ds_read2_b64 -- two 8-byte accesses per lane at HALF the rate of two ds_read_b64
            // (MI355X_MICROARCH.md, LDS table), and with a banking the padding above was not built for
            int zero = 0;
            asm volatile("" : "+v"(zero));                        // (an offset, not the pointer: it must stay an LDS pointer)
            const float2 *row = src + p * kDecPitch, *row_odd = row + zero;
#pragma unroll
            for (int s1 = 8 - kDecCentre; s1 < 8 - kDecCentre + kDecSteps + 3; s1++) {
                const float2 v = ((s1 & 1) ? row_odd : row)[4 * WQ + s1 + ((4 * WQ + s1 + 8) >> 4)];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int s2 = s1 - o - (8 - kDecCentre);
                    if (s2 >= 0 && s2 < kDecSteps) {
                        acc[o].x += h[s2] * v.x;
                        acc[o].y += h[s2] * v.y;
                    }
                }
            }
        }
    };
    if (wq == 0) fir(std::integral_constant<int, 0>{});
    else if (wq == 1) fir(std::integral_constant<int, 1>{});
    else if (wq == 2) fir(std::integral_constant<int, 2>{});
    else fir(std::integral_constant<int, 3>{});
    // sum over the four phase groups (adjacent lanes); lane pq then writes output 4 g + pq
#pragma unroll
    for (int o = 0; o < 4; o++) {
        acc[o].x = quad_sum(acc[o].x);
        acc[o].y = quad_sum(acc[o].y);
    }
    const float2 mine = pq == 0 ? acc[0] : pq == 1 ? acc[1] : pq == 2 ? acc[2] : acc[3];
    // tile number in bin order: bx for tile A, N2 - 1 - bx for tile B; its outputs are G[256 tn + i]; four-step layout of
    // the small plan: j = j2 + N2' j1 at [j2][j1]
    const int rc = (int)(pl.Nc / kDecD);
    const int tn = tile ? N2 - 1 - (int)bx_ : (int)bx_;
    {
        const int j = 256 * tn + 4 * (4 * gl + wq) + pq;
        G[(size_t)pw_ * (size_t)rc + (size_t)(j & (small_n2 - 1)) * 4096 + (j / small_n2)] = mine;
    }
    // this tile's share of the kDecEdge outputs before it (i = -kDecEdge..-1) and the kDecEdge after it (i = 256..): lane
    // (output, phase), 2 kDecEdge x 16 lanes of the tile's first four waves; sum over the steps whose bins lie inside the tile, then over
    // the phases (16 adjacent lanes)
    {
        const int tl = t & 255, eo = tl >> 4, p = tl & 15;         // eo = 0 .. 2 kDecEdge - 1 (the rest: idle lanes)
        float2 e = make_float2(0.0f, 0.0f);
        if (eo < 2 * kDecEdge) {
            // Of an edge output's kDecSteps steps only those whose bin group lies inside this tile count: for the output k
            // places before the tile (i = -k) the steps s >= C + k -- C - k of them, a run that ends at the last step --, for
            // the output j places after it (i = 256 + j) the steps s <= C - 1 - j.  So a lane walks C steps, not 2 C, and
            // every address is linear in the step: the slots of a run stay inside one 16-slot column of the image (constant
            // pad term).  No branch: a step outside the run reads the run's first slot with a zero tap.
            // (the 16 lanes of an output differ in the phase only, and a phase is 16 banks on: every lane starts its walk at
            // another step -- p >> 2 steps on, cyclically -- so that the lanes of one read fall into different banks)
            const bool left = eo < kDecEdge;
            const int cnt = left ? eo : 2 * kDecEdge - eo;                         // steps of the run (left: C - k with k = C - eo)
            const int s_lo = left ? kDecSteps - eo : 0;                            // its first step
            const int idx_lo = left ? 0 : 256 - kDecCentre + (eo - kDecEdge);      // and that step's bin group: i + s_lo - C
            const float2 *slot0 = img + p * kDecPitch + dec_slot(idx_lo + 8);
            const float *tap0 = ltaps + 16 * p + (cnt ? s_lo : 0);
            int r = p >> 2;
#pragma unroll
            for (int it = 0; it < kDecCentre; it++) {
                const bool on = r < cnt;
                const int rr = on ? r : 0;
                const float hh = on ? tap0[rr] : 0.0f;
                const float2 v = slot0[rr];
                e.x += hh * v.x;
                e.y += hh * v.y;
                r = r + 1 == kDecCentre ? 0 : r + 1;
            }
        }
        e.x = row16_sum(e.x);
        e.y = row16_sum(e.y);
        if (p == 0 && eo < 2 * kDecEdge) E[((size_t)pw_ * N2 + tn) * (2 * kDecEdge) + eo] = e;
    }
}

// inverse rows of the decimated spectrum (no K3, no mirror): rows a = 2 bx, b = a + 1 of G[N2'][4096] -> V'[k2][n1]
// with the four-step twiddle of the small plan.  G[j], j = 256 tile + i, still lacks the neighbouring tiles' shares
// near the tile boundaries (k_pair_decimate16): i < kDecEdge gets E[tile - 1][kDecEdge + i], i >= 256 - kDecEdge gets
// E[tile + 1][i - (256 - kDecEdge)].
// grid (N2'/2, n_pw), 512 threads, dynamic LDS 64 KB.
// by_column (k_pair_decimate_cols, dec_stream.hpp): the shares arrive as X[pw][12][4096] -- row i < kDecEdge of G gets slot
// row kDecEdge + i of the column to its left, row i >= N2' - kDecEdge slot row i - (N2' - kDecEdge) of the column to its
// right (slot row 0 is all zeros and is skipped): one more coalesced row read for 11 of the N2' rows.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4) TDOA_PLAIN_DS_OPS)) void k_inv_rows_plain_r8(const float2 *G, const float2 *E, float2 *V, FftPlan pl, int big_n2, int by_column)
{
    extern __shared__ float2 lds[];   // 2 * kRow8Lds
    float2 *la = lds, *lb = lds + kRow8Lds;
    const int t = threadIdx.x;
    const int a = 2 * blockIdx.x, b = a + 1;
    const float2 *g = G + (size_t)blockIdx.y * pl.Nc;
    const float2 *e = E + (size_t)blockIdx.y * big_n2 * (2 * kDecEdge);
    float2 va[8], vb[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        va[r] = g[(size_t)a * 4096 + t + 512 * r];
        vb[r] = g[(size_t)b * 4096 + t + 512 * r];
    }
    // j = row + N2' (t + 512 r): its place in the tile, i = j & 255, does not depend on r (N2' 512 is a multiple of 256),
    // so a thread either needs an edge share for all eight of its elements of a row or for none (one lane in 16 does)
    auto merge = [&](int row, float2 (&v)[8]) {
        if (by_column) {
            const bool left = row < kDecEdge;
            if ((!left && row < pl.N2 - kDecEdge) || row == pl.N2 - kDecEdge) return;
            const float2 *x = E + ((size_t)blockIdx.y * (2 * kDecEdge) + (size_t)(left ? kDecEdge + row : row - (pl.N2 - kDecEdge))) * 4096;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float2 s = x[(t + 512 * r + (left ? 4095 : 1)) & 4095];
                v[r].x += s.x;
                v[r].y += s.y;
            }
            return;
        }
        const int i = (row + pl.N2 * t) & 255;
        if (i >= kDecEdge && i < 256 - kDecEdge) return;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int tn = (row + pl.N2 * (t + 512 * r)) >> 8;
            const float2 x = i < kDecEdge ? e[(size_t)((tn + big_n2 - 1) & (big_n2 - 1)) * (2 * kDecEdge) + kDecEdge + i]
                                          : e[(size_t)((tn + 1) & (big_n2 - 1)) * (2 * kDecEdge) + (i - (256 - kDecEdge))];
            v[r].x += x.x;
            v[r].y += x.y;
        }
    };
    merge(a, va);
    merge(b, vb);
    fft8<true>(va);
    fft8<true>(vb);
    rows2_r8_finish<true>(va, vb, la, lb, t, t, t);
    inv_rows_twiddle_store(va, vb, t, a, b, V + (size_t)blockIdx.y * pl.Nc, pl);
}

// column pass + K5 of the small plan (N2' = 16 or 32 rows of 4096): one thread per column n1 reads its N2' values
// (coalesced 2 KB runs per row), evaluates the np + nn outputs n2 that can hold a searched lag as direct sums, divides the
// window out (gain[|m|] = 1 / w[m], m = packed lag index) and keeps the best peak key.  Same lag bookkeeping as
// k_inv_col_pruned_any, which spends most of its time setting up for long columns.
// grid (N1 / 256, n_pw), 256 threads.
// oc (single-look K1, k1_single_look.hpp): the residual-mean terms added to every candidate before the argmax.
// NP, NN > 0: the output set is known at compile time (outputs 0 .. NP-1 and N2' - NN .. N2' - 1; the reference's 20 000 lags
// on 4096 x 16 / x 32 small plans are 3 + 3): the factor of row k for output n2 is a power of w_k = e^{2 pi i k / N2'} -- one
// table read per row, the powers by multiplication, conjugates for the negative side -- instead of an index computation
// and an LDS read per (row, output): 1 375 -> ~700 vector instructions per wave in a kernel that issues 85 % of its time
// on the many-pair configurations.  NP = NN = 0: any output set (np, nn at run time).
template <int NP = 0, int NN = 0>
__global__ __launch_bounds__(256) void k_small_col_peak(const float2 *V, unsigned long long *keys, const PWDesc *pw, FftPlan pl,
                                                        int lag_lo, int lag_hi, int np_rt, int nn_rt, float *lag_dump,
                                                        float dump_scale, const float *gain, OnceCorr oc)
{
    constexpr bool FIXED = NP > 0 && NN > 0;
    const int np = FIXED ? NP : np_rt, nn = FIXED ? NN : nn_rt;
    __shared__ float2 wtab[256];                   // e^{+2 pi i k / N2'}  (N2' = 16, 32; 256 / 160 behind the column walk of the 4096 x 4096 / x 2560 plans)
    __shared__ unsigned long long red[4];
    const int N2 = pl.N2, N1 = pl.N1;
    if (threadIdx.x < N2)
        wtab[threadIdx.x] = pl.odd == 1 ? unit_root((float)threadIdx.x, 2.0f / (float)N2, true)
                                        : unit_root_any((float)threadIdx.x, 0.25f * (float)N2, 4.0f / (float)N2, true);
    __syncthreads();
    const int n1 = blockIdx.x * 256 + threadIdx.x;
    const float2 *in = V + (size_t)blockIdx.y * pl.Nc + n1;
    const PWDesc pwd = pw[blockIdx.y];
    if (oc.fin && n1 == 0) once_publish_gain(oc, pwd);
    OncePair op{};
    if (oc.fin) op = once_pair(oc, pwd);
    float2 acc[kPruneMax];
#pragma unroll
    for (int o = 0; o < kPruneMax; o++) acc[o] = make_float2(0.0f, 0.0f);
    const int nout = np + nn;
    // single-look K1: the additive terms of all candidates, asked for before the column sums (unconditional, clamped loads:
    // all in flight together, and back long before they are used)
    float term[kPruneMax][2];
#pragma unroll
    for (int o = 0; o < kPruneMax; o++) {
        term[o][0] = term[o][1] = 0.0f;
        if (oc.fin && o < nout) {
            const int n2 = o < np ? o : N2 - nn + (o - np);
            long long d = 2 * ((long long)n2 * N1 + n1);
            if (d >= pl.Nc) d -= 2 * pl.Nc;
            const int di = (int)(d < -oc.k_max ? -oc.k_max : d > oc.k_max ? oc.k_max - 1 : d);      // (d + 1 stays on d's side)
            const OnceSide &sd = o < np ? op.pos : op.neg;                                          // rows of one sign: uniform
            term[o][0] = once_term(oc.edges, sd, op.a, di, oc.k_max);
            term[o][1] = once_term(oc.edges, sd, op.a, di + 1, oc.k_max);
        }
    }
    for (int k0 = 0; k0 < N2; k0 += 8) {
        float2 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = in[(size_t)(k0 + u) * N1];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (FIXED) {
                constexpr int NPW = (NP - 1 > NN ? NP - 1 : NN) + 1;          // powers w^0 .. w^(NPW-1)
                float2 wp[NPW];
                wp[0] = make_float2(1.0f, 0.0f);
                if (NPW > 1) wp[1] = wtab[k0 + u];
#pragma unroll
                for (int q = 2; q < NPW; q++) wp[q] = cmul(wp[q - 1], wp[1]);
#pragma unroll
                for (int o = 0; o < NP + NN; o++) {
                    if (o == 0) {
                        acc[0] = cadd(acc[0], x[u]);
                    } else {
                        const float2 w = o < NP ? wp[o] : cconj(wp[NN - (o - NP)]);
                        acc[o] = cadd(acc[o], cmul(x[u], w));
                    }
                }
            } else {
#pragma unroll
                for (int o = 0; o < kPruneMax; o++) {
                    if (o < nout) {
                        const int n2 = o < np ? o : N2 - nn + (o - np);
                        const float2 w = wtab[pl.odd == 1 ? (n2 * (k0 + u)) & (N2 - 1) : (n2 * (k0 + u)) % N2];
                        acc[o].x += x[u].x * w.x - x[u].y * w.y;
                        acc[o].y += x[u].x * w.y + x[u].y * w.x;
                    }
                }
            }
        }
    }
    unsigned long long best = 0;
#pragma unroll
    for (int o = 0; o < kPruneMax; o++) {
        if (o < nout) {
            const int n2 = o < np ? o : N2 - nn + (o - np);
            long long d = 2 * ((long long)n2 * N1 + n1);           // lags d (real part) and d + 1 (imaginary part)
            if (d >= pl.Nc) d -= 2 * pl.Nc;
            if (d + 1 >= lag_lo && d <= lag_hi) {
                const long long m = d >> 1;
                const float gg = gain[m < 0 ? -m : m];
                const float vals[2] = {acc[o].x * gg + term[o][0], acc[o].y * gg + term[o][1]};
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const long long dq = d + q;
                    if (dq >= lag_lo && dq <= lag_hi && vals[q] == vals[q]) {
                        const unsigned long long k = peak_key(vals[q], (int)dq);
                        best = k > best ? k : best;
                        if (lag_dump) lag_dump[dq - lag_lo] = vals[q] * dump_scale;
                    }
                }
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

// k_inv_rows_plain_r8 + k_small_col_peak<3, 3> in ONE pass (round 5): V' -- 8 Nc/16 bytes written and read back per
// pair-window, a seventh of BASELINE config 5's step -- never exists.  One workgroup takes a pair-window and runs its N2' / 2 row
// pairs one after the other; the six column outputs that can hold a searched lag (n2 = 0, 1, 2 and N2' - 3 .. N2' - 1: direct
// sums, as in k_small_col_peak) are accumulated in registers across the rows -- thread t keeps them for its eight columns
// n1 = t + 512 k: 6 x 8 complex numbers --, in the rows' order and with the same factors (powers of w_k = e^{2 pi i k / N2'} by
// multiplication), so the candidates are bit for bit those of the two kernels.  The next row pair is asked for before the
// current one is transformed.  Same lag bookkeeping, window gain, residual-mean terms and key as k_small_col_peak.
// Power-of-two small plans (4096 x 16 / x 32) with the 3 + 3 output set; everything else keeps the two kernels.
// grid (n_pw), 512 threads at two waves per SIMD (~200 VGPRs), dynamic LDS 64 KB.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2) TDOA_PLAIN_DS_OPS)) void k_small_rows_col_peak(
    const float2 *G, const float2 *E, unsigned long long *keys, const PWDesc *pw, FftPlan pl, int big_n2, int by_column, int lag_lo,
    int lag_hi, float *lag_dump, float dump_scale, const float *gain, OnceCorr oc)
{
    constexpr int NO = 6;                                           // outputs n2 = 0, 1, 2, N2 - 3, N2 - 2, N2 - 1
    extern __shared__ float2 lds[];   // 2 * kRow8Lds
    __shared__ unsigned long long red[8];
    float2 *la = lds, *lb = lds + kRow8Lds;
    const int t = threadIdx.x, N2 = pl.N2, N1 = pl.N1;
    const float2 *g = G + (size_t)blockIdx.x * pl.Nc;
    const float2 *e = E + (size_t)blockIdx.x * big_n2 * (2 * kDecEdge);
    const PWDesc pwd = pw[blockIdx.x];
    if (oc.fin && t == 0) once_publish_gain(oc, pwd);
    float2 acc[NO][8];
#pragma unroll
    for (int o = 0; o < NO; o++)
#pragma unroll
        for (int k = 0; k < 8; k++) acc[o][k] = make_float2(0.0f, 0.0f);
    // the neighbours' shares of a row near the ends of the big plan's columns (k_inv_rows_plain_r8's merge)
    auto merge = [&](int row, float2 (&v)[8]) {
        if (by_column) {
            const bool left = row < kDecEdge;
            if ((!left && row < N2 - kDecEdge) || row == N2 - kDecEdge) return;
            const float2 *x = E + ((size_t)blockIdx.x * (2 * kDecEdge) + (size_t)(left ? kDecEdge + row : row - (N2 - kDecEdge))) * 4096;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float2 s_ = x[(t + 512 * r + (left ? 4095 : 1)) & 4095];
                v[r].x += s_.x;
                v[r].y += s_.y;
            }
            return;
        }
        const int i = (row + N2 * t) & 255;
        if (i >= kDecEdge && i < 256 - kDecEdge) return;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int tn = (row + N2 * (t + 512 * r)) >> 8;
            const float2 x = i < kDecEdge ? e[(size_t)((tn + big_n2 - 1) & (big_n2 - 1)) * (2 * kDecEdge) + kDecEdge + i]
                                          : e[(size_t)((tn + 1) & (big_n2 - 1)) * (2 * kDecEdge) + (i - (256 - kDecEdge))];
            v[r].x += x.x;
            v[r].y += x.y;
        }
    };
    // acc[o] += x w^n2(o): the factors of row `row` are powers of w = e^{2 pi i row / N2'}, conjugates on the negative side
    auto gather = [&](int row, const float2 (&v)[8]) {
        float2 wp[4];
        wp[1] = unit_root((float)row, 2.0f / (float)N2, true);
        wp[2] = cmul(wp[1], wp[1]);
        wp[3] = cmul(wp[2], wp[1]);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float2 x = v[oreg8(k)];
            acc[0][k] = cadd(acc[0][k], x);
            acc[1][k] = cadd(acc[1][k], cmul(x, wp[1]));
            acc[2][k] = cadd(acc[2][k], cmul(x, wp[2]));
            acc[3][k] = cadd(acc[3][k], cmul(x, cconj(wp[3])));
            acc[4][k] = cadd(acc[4][k], cmul(x, cconj(wp[2])));
            acc[5][k] = cadd(acc[5][k], cmul(x, cconj(wp[1])));
        }
    };
    float2 va[8], vb[8], na[8], nb[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        va[r] = g[t + 512 * r];
        vb[r] = g[(size_t)4096 + t + 512 * r];
    }
    const float inv2 = 2.0f / (float)pl.Nc;
#pragma unroll 1
    for (int a = 0; a < N2; a += 2) {
        const int b = a + 1;
        if (a + 2 < N2) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                na[r] = g[(size_t)(a + 2) * 4096 + t + 512 * r];
                nb[r] = g[(size_t)(a + 3) * 4096 + t + 512 * r];
            }
        }
        merge(a, va);
        merge(b, vb);
        fft8<true>(va);
        fft8<true>(vb);
        rows2_r8_finish<true>(va, vb, la, lb, t, t, t);
        // V'[k2][n1] = y[n1] W_Nc^(-n1 k2), n1 = t + 512 k (inv_rows_twiddle_store, kept in registers)
        const long long e0 = ((long long)t * a) & (pl.Nc - 1), e1 = ((long long)512 * a) & (pl.Nc - 1);
        mul_base_step8(va, unit_root((float)e0, inv2, true), unit_root((float)e1, inv2, true));
        const long long f0 = ((long long)t * b) & (pl.Nc - 1), f1 = ((long long)512 * b) & (pl.Nc - 1);
        mul_base_step8(vb, unit_root((float)f0, inv2, true), unit_root((float)f1, inv2, true));
        gather(a, va);
        gather(b, vb);
        __syncthreads();                                            // the images are free for the next pair
#pragma unroll
        for (int r = 0; r < 8; r++) {
            va[r] = na[r];
            vb[r] = nb[r];
        }
    }
    // K5 over this thread's 6 x 8 candidates (two lags each)
    OncePair op{};
    if (oc.fin) op = once_pair(oc, pwd);
    unsigned long long best = 0;
#pragma unroll
    for (int o = 0; o < NO; o++) {
        const int n2 = o < 3 ? o : N2 - 3 + (o - 3);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int n1 = t + 512 * k;
            long long d = 2 * ((long long)n2 * N1 + n1);           // lags d (real part) and d + 1 (imaginary part)
            if (d >= pl.Nc) d -= 2 * pl.Nc;
            float term0 = 0.0f, term1 = 0.0f;
            if (oc.fin) {
                const int di = (int)(d < -oc.k_max ? -oc.k_max : d > oc.k_max ? oc.k_max - 1 : d);      // (d + 1 stays on d's side)
                const OnceSide &sd = o < 3 ? op.pos : op.neg;
                term0 = once_term(oc.edges, sd, op.a, di, oc.k_max);
                term1 = once_term(oc.edges, sd, op.a, di + 1, oc.k_max);
            }
            if (d + 1 >= lag_lo && d <= lag_hi) {
                const long long m = d >> 1;
                const float gg = gain[m < 0 ? -m : m];
                const float vals[2] = {acc[o][k].x * gg + term0, acc[o][k].y * gg + term1};
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const long long dq = d + q;
                    if (dq >= lag_lo && dq <= lag_hi && vals[q] == vals[q]) {
                        const unsigned long long key = peak_key(vals[q], (int)dq);
                        best = key > best ? key : best;
                        if (lag_dump) lag_dump[dq - lag_lo] = vals[q] * dump_scale;
                    }
                }
            }
        }
    }
    best = wave_max_u64(best);
    if ((t & 63) == 0) red[t >> 6] = best;
    __syncthreads();
    if (t == 0) {
        unsigned long long bb = red[0];
        for (int w = 1; w < 8; w++) bb = red[w] > bb ? red[w] : bb;
        if (bb) atomicMax(&keys[pwd.out_index], bb);
    }
}

// ---------------------------------------------------------------------------
// Segment form (search ranges up to 1024 lags): the whole correlation stays in LDS and registers.
// The deployed geometry bounds |TDOA| by 114 samples (PROJECT_NOTES.md:29-32); a caller who searches a few hundred
// lags instead of the reference's 20 000 (processor.go:633) does not need a 2^21-point transform.  Overlap-save
// over the template: segment k is t[kH, kH + H), H = 4096 - 2P (P = 256 PQ >= max_lag, PQ = 1, 2, 4), placed at [P, P + H) of a
// 4096-point frame next to the signal samples s[kH - P, kH - P + 4096).  One complex transform Z of z = t + i s
// carries both real frames; with A = |Z[k]|^2 and B = Z[k] Z[-k]
//   conj(T[k]) S[k] = Im(B[k]) / 2 - i (A[k] - A[-k]) / 4,
// so a workgroup only ADDS 4 FMAs per bin per segment into 2 x 8 registers and runs one inverse transform at the end.
// No wrap-around reaches the lags |d| <= P.  HBM traffic: the phase codes of both stations, about 2.7 B per sample
// and station (frames overlap by 2P), against 31.4 B per sample of the four-step form.
// grid (n_chunks, n_pw), 512 threads, dynamic LDS 64 KB.  Chunk c of a pair-window takes the segment pairs
// c, c + n_chunks, ...; its 2P + 1 lag sums (d = -P .. P) go to part[pw][c][2P + 8] (floats, at V + pw * Nc),
// k_segments_reduce adds the chunks in a fixed order.
// ---------------------------------------------------------------------------
template <int PQ, bool PACK3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4) TDOA_PLAIN_DS_OPS)) void k_xcorr_segments(const SWDesc *sw, const PWDesc *pw, const int *codes, long long code_stride, const FmStats *stats, float2 *V, FftPlan pl, int n_chunks)
{
    constexpr int P = 256 * PQ, H = 4096 - 2 * P;
    extern __shared__ float2 lds[];   // 2 * kRow8Lds
    float2 *la = lds, *lb = lds + kRow8Lds;
    const int t = threadIdx.x;
    const PWDesc d = pw[blockIdx.y];
    const int len_t = sw[d.sw_a].len, len_s = sw[d.sw_b].len;
    const SegCodes<PACK3> ct(codes, d.sw_a, code_stride), cs(codes, d.sw_b, code_stride);
    // (float(code) - mean) scale as ONE fma on the stored (negated) code: stored x (-scale) + (-mean scale); the fast
    // path's values count in SegCodes::kUnit (packed rows: 256 x code, a power of two that goes into the factor exactly)
    const SegNorm nt(stats[d.sw_a], SegCodes<PACK3>::kUnit), ns(stats[d.sw_b], SegCodes<PACK3>::kUnit);
    const int n_seg = (len_t + H - 1) / H, len_min = len_t < len_s ? len_t : len_s;
    // forward stage roots of this thread, once (6 VGPRs instead of three root evaluations per trip)
    const float2 w2 = unit_root((float)(t & 7), 2.0f / 64.0f, false), w3 = unit_root((float)(t & 63), 2.0f / 512.0f, false),
                 w4 = unit_root((float)t, 2.0f / 4096.0f, false);
    // mirror element (4096 - (t + 512 k)) mod 4096 = ((512 - t) mod 512) + 512 ((7 - k + [t == 0]) mod 8)
    const int mb = swz((512 - t) & 511), mz = t == 0 ? 1 : 0, rb = swz(t);
    float accA[8], accB[8];
#pragma unroll
    for (int k = 0; k < 8; k++) accA[k] = accB[k] = 0.0f;

    for (int s0 = 2 * blockIdx.x; s0 < n_seg; s0 += 2 * n_chunks) {
        float2 x[8], y[8];          // two frames per trip (they share the stage twiddles)
        // sample index of frame position n: i = seg * H + n - P (32-bit: windows are shorter than 2^31 samples);
        // the row pointers are uniform, so every load is base (SGPR) + 32-bit lane offset
        const int i0 = s0 * H - P + t, i1 = i0 + H;
        if (s0 * H - P >= 0 && (s0 + 1) * H - P + 4096 <= len_min) {
            // both frames lie inside both windows (all but the first and last trips): no bounds checks; the template
            // part of a frame is the positions [P, P + H): whole values r of every thread, except that with P = 256
            // the band starts and ends in the middle of r = 0 and r = 7
            constexpr unsigned kB = SegCodes<PACK3>::kBytes;
            // (the lane constants are rebuilt per trip from a laundered thread index: kept across the loop next to the roots and
            //  the mirror offsets they were the 129th register of the packed P = 256 / 512 instances -- 8 bytes of scratch)
            const int tl = opaque_i(t);
            const unsigned bo = kB * (unsigned)(s0 * H - P) + SegCodes<PACK3>::lane_off(tl), sh = SegCodes<PACK3>::lane_sel(tl);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool tpos = t + 512 * r >= P && t + 512 * r < P + H;
                const int a0 = tpos ? ct.quad_at(bo + 512u * kB * r, sh) : 0, b0 = cs.quad_at(bo + 512u * kB * r, sh);
                const int a1 = tpos ? ct.quad_at(bo + kB * H + 512u * kB * r, sh) : 0, b1 = cs.quad_at(bo + kB * H + 512u * kB * r, sh);
                x[r] = make_float2(tpos ? nt.fast(a0) : 0.0f, ns.fast(b0));
                y[r] = make_float2(tpos ? nt.fast(a1) : 0.0f, ns.fast(b1));
            }
        } else {
            const bool odd_ok = s0 + 1 < n_seg;     // the odd segment of the last pair may not exist: then BOTH its
                                                    // parts are zero (a signal-only frame adds nothing to the result,
                                                    // but |S|^2 would go through the accumulators and cancel only to
                                                    // rounding)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool tpos = t + 512 * r >= P && t + 512 * r < P + H;
                {
                    const int i = i0 + 512 * r;
                    const bool in_t = tpos && i < len_t, in_s = i >= 0 && i < len_s;
                    const int qt = in_t ? ct[i] : 0, qs = in_s ? cs[i] : 0;
                    x[r] = make_float2(in_t ? nt.plain(qt) : 0.0f, in_s ? ns.plain(qs) : 0.0f);
                }
                {
                    const int i = i1 + 512 * r;
                    const bool in_t = tpos && i < len_t, in_s = odd_ok && i >= 0 && i < len_s;
                    const int qt = in_t ? ct[i] : 0, qs = in_s ? cs[i] : 0;
                    y[r] = make_float2(in_t ? nt.plain(qt) : 0.0f, in_s ? ns.plain(qs) : 0.0f);
                }
            }
        }
        fft8<false>(x);
        fft8<false>(y);
        // (opaque: keep the three roots, rebuild the 3 x 7 powers -- hoisted out of the loop they would be spilled)
        rows2_r8_finish_w<false>(x, y, la, lb, t, t, t, opaque(w2), opaque(w3), opaque(w4));      // Z[t + 512 k] in [oreg8(k)]
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            la[rb + 512 * k] = x[oreg8(k)];
            lb[rb + 512 * k] = y[oreg8(k)];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int o = 512 * ((7 - k + mz) & 7);
            const float2 zx = x[oreg8(k)], zy = y[oreg8(k)], mx = la[mb + o], my = lb[mb + o];
            accA[k] += zx.x * zx.x + zx.y * zx.y + zy.x * zy.x + zy.y * zy.y;
            accB[k] += zx.x * mx.y + zx.y * mx.x + zy.x * my.y + zy.y * my.x;
        }
        __syncthreads();
    }
    // C[k] = Im B / 2 - i (A[k] - A[-k]) / 4, then one inverse transform: M r[d] = sum_k C[k] e^{+2 pi i k d / M}
    float2 c[8], zero[8];
#pragma unroll
    for (int k = 0; k < 8; k++) la[rb + 512 * k] = make_float2(accA[k], 0.0f);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float am = la[mb + 512 * ((7 - k + mz) & 7)].x;
        c[k] = make_float2(0.5f * accB[k], -0.25f * (accA[k] - am));      // input k of stage-1 item t
        zero[k] = make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    fft8<true>(c);
    rows2_r8_finish<true>(c, zero, la, lb, t, t, t);
    // lags d = n for n <= P and d = n - 4096 for n >= 4096 - P, n = t + 512 k; part index d + P (2P + 1 lags, row
    // pitch 2P + 8)
    float *part = reinterpret_cast<float *>(V + (size_t)blockIdx.y * pl.Nc) + (size_t)blockIdx.x * (2 * P + 8);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int n = t + 512 * k;
        if (512 * k <= P && n <= P) part[P + n] = c[oreg8(k)].x;
        if (512 * (k + 1) > 4096 - P && n >= 4096 - P) part[n - (4096 - P)] = c[oreg8(k)].x;
    }
}

// The same for up to four pair-windows of one window at a time: templates (a, b) against signals (c, d).
// Z1 = FFT(t_a + i t_b) (both masked to [P, P + H)) and Z2 = FFT(s_c + i s_d) (full frames) per segment; with
// p = Z1[k] + conj(Z1[-k]) = 2 T_a[k] and m = Z1[k] - conj(Z1[-k]) = 2 i T_b[k]
//   conj(T_a) Z2 = conj(p) Z2 / 2      -> inverse transform = c_ac + i c_ad
//   conj(T_b) Z2 = i conj(m) Z2 / 2    -> inverse transform = c_bc + i c_bd
// i.e. two forward transforms per segment serve four pair-windows (three for the reference's three stations: a = 0,
// b = c = 1, d = 2, the (1, 1) output is not wanted), against one transform per pair-window above.  Nothing is
// subtracted between accumulators (the pair form's A[k] - A[-k] is gone).
// grid (n_chunks, n_quads), 512 threads, dynamic LDS 64 KB; chunk c takes the segments c, c + n_chunks, ...; output
// layout as above, one part row per wanted pair-window.
template <int PQ, bool PACK3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4) TDOA_PLAIN_DS_OPS)) void k_xcorr_segments_quad(const SWDesc *sw, const QuadDesc *quads, const int *codes, long long code_stride, const FmStats *stats, float2 *V, FftPlan pl, int n_chunks)
{
    constexpr int P = 256 * PQ, H = 4096 - 2 * P;
    extern __shared__ float2 lds[];   // 2 * kRow8Lds
    float2 *la = lds, *lb = lds + kRow8Lds;
    const int t = threadIdx.x;
    const QuadDesc q = quads[blockIdx.y];
    const bool has_b = q.sw_tb >= 0, has_d = q.sw_sd >= 0;
    const int ia = q.sw_ta, ib = has_b ? q.sw_tb : q.sw_ta, ic = q.sw_sc, id = has_d ? q.sw_sd : q.sw_sc;
    const int len_a = sw[ia].len, len_b = has_b ? sw[ib].len : 0, len_c = sw[ic].len, len_d = has_d ? sw[id].len : 0;
    const SegCodes<PACK3> ca(codes, ia, code_stride), cb(codes, ib, code_stride), cc(codes, ic, code_stride), cd(codes, id, code_stride);
    const SegNorm na(stats[ia], SegCodes<PACK3>::kUnit), nb(stats[ib], SegCodes<PACK3>::kUnit), nc(stats[ic], SegCodes<PACK3>::kUnit),
        nd(stats[id], SegCodes<PACK3>::kUnit);
    const int len_t = len_a > len_b ? len_a : len_b, n_seg = (len_t + H - 1) / H;
    int len_min = len_a < len_c ? len_a : len_c;
    len_min = len_min < len_b ? len_min : len_b;
    len_min = len_min < len_d ? len_min : len_d;       // 0 when a slot is empty: every trip takes the checked path
    const float2 w2 = unit_root((float)(t & 7), 2.0f / 64.0f, false), w3 = unit_root((float)(t & 63), 2.0f / 512.0f, false),
                 w4 = unit_root((float)t, 2.0f / 4096.0f, false);
    const int mq = swz(512 - t), rb = swz(t);
    float2 accX[8], accY[8];
#pragma unroll
    for (int k = 0; k < 8; k++) accX[k] = accY[k] = make_float2(0.0f, 0.0f);

    for (int s0 = blockIdx.x; s0 < n_seg; s0 += n_chunks) {
        float2 x[8], y[8];          // x: the two template frames, y: the two signal frames
        const int i0 = s0 * H - P + t;
        if (s0 * H - P >= 0 && s0 * H - P + 4096 <= len_min) {
            // uniform row pointer (SGPR pair) + unsigned 32-bit byte offset of the lane: one offset register serves all
            // four streams, where signed sample indices would cost a 64-bit address pair per stream and position
            constexpr unsigned kB = SegCodes<PACK3>::kBytes;
            const unsigned bo = kB * (unsigned)(s0 * H - P) + SegCodes<PACK3>::lane_off(t), sh = SegCodes<PACK3>::lane_sel(t);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool tpos = t + 512 * r >= P && t + 512 * r < P + H;
                const int a0 = tpos ? ca.quad_at(bo + 512u * kB * r, sh) : 0, b0 = tpos ? cb.quad_at(bo + 512u * kB * r, sh) : 0;
                const int c0 = cc.quad_at(bo + 512u * kB * r, sh), d0 = cd.quad_at(bo + 512u * kB * r, sh);
                x[r] = make_float2(tpos ? na.fast(a0) : 0.0f, tpos ? nb.fast(b0) : 0.0f);
                y[r] = make_float2(nc.fast(c0), nd.fast(d0));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool tpos = t + 512 * r >= P && t + 512 * r < P + H;
                const int i = i0 + 512 * r;
                const bool in_a = tpos && i < len_a, in_b = tpos && i < len_b;
                const bool in_c = i >= 0 && i < len_c, in_d = i >= 0 && i < len_d;
                const int qa = in_a ? ca[i] : 0, qb = in_b ? cb[i] : 0, qc = in_c ? cc[i] : 0, qd = in_d ? cd[i] : 0;
                x[r] = make_float2(in_a ? na.plain(qa) : 0.0f, in_b ? nb.plain(qb) : 0.0f);
                y[r] = make_float2(in_c ? nc.plain(qc) : 0.0f, in_d ? nd.plain(qd) : 0.0f);
            }
        }
        fft8<false>(x);
        fft8<false>(y);
        rows2_r8_finish_w<false>(x, y, la, lb, t, t, t, opaque(w2), opaque(w3), opaque(w4));      // Z[t + 512 k] in [oreg8(k)]
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) la[rb + 512 * k] = x[oreg8(k)];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // mirror of bin t + 512 k: element (512 - t) + 512 (7 - k) -- one base + immediates; thread 0 (whose mirror
            // of bin 0 is bin 0 itself) reads one element past the image for k = 0 (lb[0], in bounds) and ignores it
            const float2 z = x[oreg8(k)], w = y[oreg8(k)], zl = la[mq + 512 * (7 - k)];
            const float2 zm = (k == 0 && t == 0) ? z : zl;
            const float2 p = make_float2(z.x + zm.x, z.y - zm.y), m = make_float2(z.x - zm.x, z.y + zm.y);
            accX[k].x += p.x * w.x + p.y * w.y;
            accX[k].y += p.x * w.y - p.y * w.x;
            accY[k].x += m.x * w.x + m.y * w.y;
            accY[k].y += m.x * w.y - m.y * w.x;
        }
        __syncthreads();
    }
    float2 cx[8], cy[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        cx[k] = make_float2(0.5f * accX[k].x, 0.5f * accX[k].y);          // input k of stage-1 item t
        cy[k] = make_float2(-0.5f * accY[k].y, 0.5f * accY[k].x);
    }
    fft8<true>(cx);
    fft8<true>(cy);
    rows2_r8_finish<true>(cx, cy, la, lb, t, t, t);
    const size_t pitch = 2 * P + 8;
    // (the four outputs spelled out with their pair-window in a scalar: indexing q.pw[] with the loop variable of a loop that
    // `continue`s kept the descriptor in 48 bytes of scratch)
    auto emit = [&](int pwi, auto o_c) {
        constexpr int o = decltype(o_c)::value;
        if (pwi < 0) return;
        float *part = reinterpret_cast<float *>(V + (size_t)pwi * pl.Nc) + (size_t)blockIdx.x * pitch;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int n = t + 512 * k;
            const float2 c2 = o < 2 ? cx[oreg8(k)] : cy[oreg8(k)];
            const float c = (o & 1) ? c2.y : c2.x;
            if (512 * k <= P && n <= P) part[P + n] = c;
            if (512 * (k + 1) > 4096 - P && n >= 4096 - P) part[n - (4096 - P)] = c;
        }
    };
    const int pw0 = q.pw[0], pw1 = q.pw[1], pw2 = q.pw[2], pw3 = q.pw[3];
    emit(pw0, std::integral_constant<int, 0>{});
    emit(pw1, std::integral_constant<int, 1>{});
    emit(pw2, std::integral_constant<int, 2>{});
    emit(pw3, std::integral_constant<int, 3>{});
}

// chunk sums in a fixed order -> lag array (kept for the sub-sample refinement: lags[li] = c[li - P], unscaled like the
// keys, at float offset N2 * P of the pair-window's V row -- behind every chunk sum), lag filter, K5.
// mul = 4 N / 4096 brings the sums to the scale of the four-step form (decode multiplies by 1 / (4 N sqrt(len))).
// grid (2 PQ + 1, n_pw), 256 threads.
template <int PQ>
__global__ __launch_bounds__(256) void k_segments_reduce(float2 *V, unsigned long long *keys, const PWDesc *pw, FftPlan pl,
                                                        int n_chunks, float mul, int lag_lo, int lag_hi, float *lag_dump,
                                                        float dump_scale)
{
    constexpr int P = 256 * PQ;
    __shared__ unsigned long long red[4];
    float *base = reinterpret_cast<float *>(V + (size_t)blockIdx.y * pl.Nc);
    float *lags = reinterpret_cast<float *>(V + (size_t)blockIdx.y * pl.Nc) + (size_t)pl.N2 * P;
    const int li = blockIdx.x * 256 + threadIdx.x;            // lag index d + P, valid up to 2 P
    const bool live = li <= 2 * P;
    float s = 0.0f;
    if (live)
        for (int ch = 0; ch < n_chunks; ch++) s += base[(size_t)ch * (2 * P + 8) + li];
    const float v = s * mul;
    const int dlag = li - P;
    if (live) lags[li] = v;
    unsigned long long best = 0;
    if (live && dlag >= lag_lo && dlag <= lag_hi && v == v) {
        best = peak_key(v, dlag);
        if (lag_dump) lag_dump[dlag - lag_lo] = v * dump_scale;
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

// refinement neighbours c[lag-1], c[lag], c[lag+1] from the lag array k_segments_reduce left behind (2P + 1 entries)
template <int PQ>
__global__ void k_refine_segments(const float2 *V, const unsigned long long *keys, const PWDesc *pw, FftPlan pl, int n_pw,
                                  float *raw)
{
    constexpr int P = 256 * PQ;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per pair-window
    if (id >= n_pw) return;
    const int slot = pw[id].out_index;
    const unsigned long long k = keys[slot];
    float r[3] = {0.0f, 0.0f, 0.0f};
    if (k != 0 && (unsigned int)(k >> 32) != 0) {
        const unsigned int rank = 0x7fffffffu - ((unsigned int)k >> 1);
        const int lag = rank == 0 ? 0 : ((rank & 1u) ? (int)((rank + 1u) >> 1) : -(int)(rank >> 1));
        const float *lags = reinterpret_cast<const float *>(V + (size_t)id * pl.Nc) + (size_t)pl.N2 * P;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int li = lag - 1 + q + P;
            r[q] = li >= 0 && li <= 2 * P ? lags[li] : 0.0f;
        }
    }
#pragma unroll
    for (int q = 0; q < 3; q++) raw[3 * (size_t)slot + q] = r[q];
}

}  // namespace tdoa
