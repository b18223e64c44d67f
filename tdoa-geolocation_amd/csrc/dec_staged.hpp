// dec_staged.hpp -- the column walk of the decimated pair step with the stations' rows STAGED IN LDS (round 5).
//
// k_pair_decimate_cols (dec_stream.hpp) gives every pair-window its own walk: each walk loads the rows of its two stations
// itself, so a window of S stations and P pairs asks for 2 P / S = S - 1 times its spectra and leaves it to the caches to
// serve the repeats.  They do so poorly: the walks that share a station drift apart by more than the L2 holds, and the
// memory-side counters showed 2.1 x (8 stations, BASELINE config 4) and 3.2 x (16 stations, config 5) the compulsory bytes
// coming through the fabric.  Here one workgroup owns 64 columns (and their 64 partner columns) of ONE window
// for a GROUP of its pairs:
//   * a LOADER wave brings the rows (k2, N2 - k2) of every station the group's pairs touch into LDS by LDS-DMA
//     (global_load_lds_dwordx4: one wave-instruction = the 512 bytes of a station's row piece + the 512 bytes of its partner
//     row piece, no VGPR in between), R rows per phase into a ring of two (or more) phases -- a phase's bytes are asked for while
//     the phase before it is read;
//   * every COMPUTE wave walks one pair: the same register stencil as k_pair_decimate_cols (twelve accumulators per walk, K3 on
//     packed pairs), its four operands per row four ds_read_b64, the row's taps through the scalar cache into SGPR pairs;
//   * one raw s_barrier per phase; the loader waits with a counted vmcnt (later phases stay in flight across the barrier).
// (Or no loader wave at all -- the FOLDED form below: sixteen walks, the last waves each bring one station -- where that makes
// fewer workgroups: 13 - 16 stations.)
// Which pairs a workgroup takes comes from a table the host builds once (StgGroup, build_stg_groups in tdoa_mi355x.hip): up to
// eight stations consecutive runs of the window's pair list, more than eight a greedy share-out that keeps every group within
// eight stations -- a station sits in the ring at its RANK among the group's stations, so eight rows per phase fit for any S.
// A station's row piece comes from memory once per GROUP; the groups of one (window, column block) take consecutive slots
// of one XCD and start together, so later groups find rows in that XCD's L2.
// Where the plan leaves room (N2 = 256, 512) the row pass writes the spectra in BLOCKS of 64 columns, [column / 64][k2][column % 64]:
// a loader's pieces of consecutive rows are then consecutive in memory and a station's R rows go out back to back.
// Outputs: exactly k_pair_decimate_cols's -- G[pw][N2/16][4096] and the neighbour shares X[pw][12][4096].
//
// What bounds it (round 5, measurement build -DTDOA_STG_TIMING + scripts/microbench/lds_dma_probe.hip, profiles/r05_staged_walk_*):
// the walks' vector issue.  The counters first read otherwise -- the walks 28 - 37 % of their cycles at the barrier, the loader
// 73 - 82 % of its cycles inside its issue loop -- but (a) with the loader's priority raised (s_setprio 3) its issue share falls
// to 35 - 38 %, it waits 38 - 46 % at the barrier, and the kernel takes the same time: the loader was only being served last, in
// slots the walks left; (b) with the walks idle the loaders alone need half the kernel's time (cfg4 1.55 of 3.0 ms, cfg5 32 of
// 72); (c) a walk's time "at the barrier" is mostly spent while the other walks of its SIMD issue -- the oldest wave runs
// ahead and waits for its neighbours.  What is left: 42 vector instructions per row and pair (24 of them the filter), issue
// share 0.70, fourteen or fifteen walks on four SIMDs.  Nothing on the loader's side changes the time: a second loader wave,
// a branch-free block, the cache-policy bits, the layout of the spectra.
//
// grid: 8 x ceil(windows 32 / 8) x groups workgroups (1-D), 64 (compute waves + 1) threads; dynamic LDS nb x R x slots KB.
#pragma once

#include "dec_stream.hpp"

namespace tdoa {

constexpr int kStgMaxStations = 16;          // station slots in LDS (1 KB per row and station)
constexpr int kStgMaxWaves = 16;
constexpr int kStgLdsBytes = 128 * 1024;     // the ring: phases x rows per phase x stations x 1 KB
constexpr int kStgMaxInFlight = 60;          // LDS-DMA instructions a loader wave leaves outstanding (the counter holds 63)
constexpr int kStgBlockCols = 64;            // the blocked layout of the unpacked spectra: [column / 64][row k2][column % 64]
// one workgroup's share of a window's pairs: `n` of them, by their index in the window's pair list, and the stations they touch
// (bit s = the window's station s); a station's place in the LDS ring is its rank among the set bits
struct StgGroup {
    uint32_t mask;
    int32_t n;
    uint8_t pair[16];
};

#if TDOA_HAVE_DEC_COLS

#ifdef TDOA_STG_TIMING
// measurement build: wave-cycles [0] walks in total, [1] walks at the barrier, [2] loaders in total, [3] loaders waiting for
// their LDS-DMA to land, [4] loaders at the barrier, [5] walks counted, [6] loaders counted, [7] loaders inside issue() (read and
// cleared through tdoa_debug_stg_prof; bench.py prints them with TDOA_STG_PROF=1)
__device__ unsigned long long g_stg_prof[8];
#define TDOA_STG_T(...) __VA_ARGS__
#else
#define TDOA_STG_T(...)
#endif

// one s_waitcnt vmcnt(n) for a run-time n (the field is an immediate): n = the LDS-DMA instructions of the later phases that
// may stay in flight, a multiple of 2 (rows per phase) up to kStgMaxInFlight
__device__ __forceinline__ void stg_wait_vm(int n)
{
    switch (n) {
#define TDOA_VM(n_) case n_: asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory"); break;
        TDOA_VM(0) TDOA_VM(2) TDOA_VM(4) TDOA_VM(6) TDOA_VM(8) TDOA_VM(10) TDOA_VM(12) TDOA_VM(14) TDOA_VM(16) TDOA_VM(18)
        TDOA_VM(20) TDOA_VM(22) TDOA_VM(24) TDOA_VM(26) TDOA_VM(28) TDOA_VM(30) TDOA_VM(32) TDOA_VM(34) TDOA_VM(36) TDOA_VM(38)
        TDOA_VM(40) TDOA_VM(42) TDOA_VM(44) TDOA_VM(46) TDOA_VM(48) TDOA_VM(50) TDOA_VM(52) TDOA_VM(54) TDOA_VM(56) TDOA_VM(58)
        TDOA_VM(60)
#undef TDOA_VM
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// (Round 5 also built the ring WITHOUT barriers -- two LDS counters per slot, `ready` set by the loader and `done` by the walks,
// bounded spins, the walks free to drift nb - 2 phases apart: bit-identical and slower on every geometry, cfg4 3.98-5.78 ms
// against 3.45-3.61, cfg5 124-159 against 89.6 (profiles/r05_staged_walk_flag_ring_sweep.txt): an LDS atomic round trip per
// phase and wave costs more than the barrier it replaces.  Commit 73f2769 holds it.)
// R_: rows (and partner rows) per phase = per barrier; nb: phases in the ring (nb - 1 of them are in flight or being read:
// what hides the memory latency is (nb - 2) R S KB per workgroup)
template <int N2_, int R_>
// (plain ds_read_b64 with immediate row offsets: left to itself the backend pairs the reads of two rows into ds_read2st64_b64,
//  which the LDS serves at half the rate of two plain reads -- MI355X_MICROARCH.md, LDS table; -DTDOA_STG_PAIRED_READS for the A/B)
#ifdef TDOA_STG_PAIRED_READS
#define TDOA_STG_DS_OPS
#else
#define TDOA_STG_DS_OPS TDOA_PLAIN_DS_OPS
#endif
__global__ __launch_bounds__(64 * kStgMaxWaves) __attribute__((amdgpu_waves_per_eu(4, 4) TDOA_STG_DS_OPS)) void k_pair_decimate_staged(const PWDesc *pw, const float2 *U, float2 *G, float2 *X, FftPlan pl,
                                                                             const float *__restrict__ taps, const StgGroup *__restrict__ groups,
                                                                             int n_items, int P, int S, int n_cw, int n_groups, int nb,
                                                                             long long u_stride, int blocked)
{
    constexpr int N2 = N2_, N1 = 4096, C = kDecCentre, SS = kDecSteps, R = R_;
    constexpr int NG = N2 / 16;
    constexpr bool POW2 = (N2 & (N2 - 1)) == 0;
    static_assert(NG >= 2 * C + 2 && 16 % R == 0 && N2 % R == 0 && R % 2 == 0, "ring and loop geometry");
    // the ring: [phase slot][station][row of the phase][fwd | partner][64] float2 -- the row index innermost, so that a walk's
    // four operand addresses change once per PHASE (one v_add each) and the row inside the phase is an immediate offset
    extern __shared__ __attribute__((aligned(16))) unsigned char stage_raw[];
    const int t = threadIdx.x;
    // workgroup -> (window, column block, group of pairs): the groups of an item are consecutive slots of one XCD
    const unsigned int b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
    const int item = (int)(xcd + 8u * (slot / (unsigned int)n_groups)), grp = (int)(slot % (unsigned int)n_groups);
    if (item >= n_items) return;                                   // (the whole workgroup: no barrier is left waiting)
    const int w = item >> 5, cb = item & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const PWDesc *wpw = pw + (size_t)w * P;
    const int sw_base = wpw[0].sw_a;                               // the window's stations are sw_base .. (pair 0 = stations 0, 1)
    const StgGroup *gd = groups + grp;
    const int p_cnt = gd->n;
    const unsigned int mask = gd->mask;                            // stations this group's pairs touch: S or fewer
    const int zpad = pl.zpad;
    // where element (row 0, column c) of a station's spectrum lives: row-major rows (in place, the 4096 x 2048 and larger plans)
    // or blocks of 64 columns, [c / 64][k2][c % 64] (kStgBlockCols; see the loader)
    auto row0_at = [&](const float2 *base, int c) { return blocked ? base + (size_t)(c >> 6) * ((size_t)N2 * 64) + (c & 63) : base + c; };
    constexpr int NP = N2 / R;                                     // phases

    // FOLDED form (no loader wave: blockDim = 64 n_cw; blocked layout, two phases in the ring): the LAST popcount(mask) waves of the
    // workgroup -- idle ones first -- each bring ONE station's rows, at the top of a phase for the next one: 3 instructions per
    // LDS-DMA next to the 336 of a walk's phase.  All sixteen waves can then walk: 28 pairs are 16 + 12 walks (four and three per
    // SIMD) instead of 14 + 14 (two SIMDs a wave short in both).
    const int n_waves = (int)(blockDim.x >> 6);
    const bool folded = n_waves == n_cw;
    const int duty = folded ? n_waves - 1 - wave : -1;                 // this wave's station: the duty-th of the group's (by rank)
    const bool has_duty = folded && duty < __builtin_popcount(mask);
    unsigned long long duty_base = 0;
    if (has_duty) {
        unsigned int m = mask;
        for (int i = 0; i < duty; i++) m &= m - 1;
        duty_base = (unsigned long long)(uintptr_t)(U + (size_t)(sw_base + __builtin_ctz(m)) * (size_t)u_stride);
    }
    typedef __attribute__((address_space(3))) unsigned char *lds_ptr_f;
    const unsigned int lds0_f = (unsigned int)(uintptr_t)((lds_ptr_f)stage_raw);
    auto issue_one = [&](int ph, int buf_) {                           // rows ph R .. ph R + R - 1 of the duty station -> ring slot buf_
        unsigned int keep;
        const unsigned int dst = lds0_f + (unsigned int)(buf_ * S * R + duty * R) * 1024u;
        asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]" : [keep] "=&s"(keep) : [dst] "s"(dst) : "memory");
        if (ph > 0) {                                                  // a station's pieces of consecutive rows: 512 bytes apart
            const int k2 = ph * R, km2 = N2 - k2;
            const unsigned int of = 8u * ((unsigned int)cb * (N2 * 64u) + (unsigned int)k2 * 64u);
            const unsigned int om = 8u * ((unsigned int)(63 - cb) * (N2 * 64u) + (unsigned int)km2 * 64u);
            unsigned int v = 16u * (unsigned int)(lane & 31) + (lane < 32 ? of : om);
            const int dv = lane < 32 ? 512 : -512;
#pragma unroll
            for (int r = 0; r < R; r++)
                asm volatile("global_load_lds_dwordx4 %[v], %[b]\n\tv_add_u32 %[v], %[v], %[dv]\n\ts_add_u32 m0, m0, 0x400"
                             : [v] "+v"(v) : [b] "s"(duty_base), [dv] "v"(dv) : "memory", "scc");
        } else {                                                       // (row 0 pairs with row 0, not with row N2)
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int k2 = r, km2 = k2 ? N2 - k2 : 0;
                const unsigned int of = 8u * ((unsigned int)cb * (N2 * 64u) + (unsigned int)k2 * 64u);
                const unsigned int om = 8u * ((unsigned int)(63 - cb) * (N2 * 64u) + (unsigned int)km2 * 64u);
                const unsigned int v = 16u * (unsigned int)(lane & 31) + (lane < 32 ? of : om);
                asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %[v], %[b]\n\ts_add_u32 m0, m0, 0x400" : : [v] "v"(v), [b] "s"(duty_base) : "memory", "scc");
            }
        }
        asm volatile("s_mov_b32 m0, %[keep]" : : [keep] "s"(keep) : "memory");
    };
    if (has_duty) issue_one(0, 0);                                     // phase 0; phase ph + 1 goes out at the top of phase ph

    if (wave >= n_cw) {
        // ---- loader: rows (k2, N2 - k2) of the stations in `mask`, phase by phase, nb - 1 phases ahead of the readers -----------
        // lanes 0..31: 16 bytes each of row k2, columns [64 cb, 64 cb + 64); lanes 32..63: of the partner row, columns
        // [4032 - 64 cb, 4096 - 64 cb) -- 1 KB per instruction, contiguous in LDS: [row of the phase][station][fwd | partner].
        // The instruction is spelled out (global_load_lds_dwordx4 v_off, s[base]: the station's base in a scalar pair, ONE lane
        // offset per row for all stations): built from per-lane 64-bit pointers through the compiler's builtin a loader wave
        // spent ~20 instructions per LDS-DMA and two of them were slower than the fourteen walks they feed (cfg4 pair step
        // 4.6-5.1 ms against 4.08 for the per-pair walk).
        // loader lw of n_lw takes every n_lw-th of the group's stations (by rank = LDS slot), from the lw-th
        const int lw = wave - n_cw, n_lw = (int)(blockDim.x >> 6) - n_cw;
        unsigned int mine = 0;
        {
            unsigned int m = mask;
            for (int i = 0; m; i++, m &= m - 1)
                if (i % n_lw == lw) mine |= m & (0u - m);
        }
        const int per_phase = R * __builtin_popcount(mine);         // (the counted wait is this wave's own counter)
        const unsigned int lane_off = 8u * (unsigned int)(lane < 32 ? 64 * cb + 2 * lane : (4032 - 64 * cb) + 2 * (lane - 32));
        unsigned long long sbase[kStgMaxStations];
#pragma unroll
        for (int s_ = 0; s_ < kStgMaxStations; s_++) sbase[s_] = (unsigned long long)(uintptr_t)(U + (size_t)(sw_base + s_) * (size_t)u_stride);
        typedef __attribute__((address_space(3))) unsigned char *lds_ptr;
        const unsigned int lds0 = (unsigned int)(uintptr_t)((lds_ptr)stage_raw);      // the ring's LDS byte address
        auto issue = [&](int ph, int buf) {
            // blocked layout, every phase but the first (whose row 0 pairs with row 0, not with row N2): station by station, a station's R
            // rows back to back -- its 512-byte pieces are consecutive in memory (forward: ascending, partner: descending), 4 KB runs;
            // row by row across the stations the same loads took 6 % longer on sixteen stations (cfg5 78.8 -> 73.8 ms, 8 stations: equal)
            if (blocked && ph > 0) {
                const int k2 = ph * R, km2 = N2 - k2;
                const unsigned int of = 8u * ((unsigned int)cb * (N2 * 64u) + (unsigned int)k2 * 64u);
                const unsigned int om = 8u * ((unsigned int)(63 - cb) * (N2 * 64u) + (unsigned int)km2 * 64u);
                const unsigned int voff0 = 16u * (unsigned int)(lane & 31) + (lane < 32 ? of : om);
                const int dv = lane < 32 ? 512 : -512;
                unsigned int dst = lds0 + (unsigned int)(buf * S * R + lw * R) * 1024u;
#pragma unroll
                for (int s_ = 0; s_ < kStgMaxStations; s_++) {
                    if ((mine >> s_) & 1u) {
                        unsigned int keep, v = voff0;
                        asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0" : [keep] "=&s"(keep) : [dst] "s"(dst) : "memory");
#pragma unroll
                        for (int r = 0; r < R; r++)
                            asm volatile("global_load_lds_dwordx4 %[v], %[b]\n\tv_add_u32 %[v], %[v], %[dv]\n\ts_add_u32 m0, m0, 0x400"
                                         : [v] "+v"(v) : [b] "s"(sbase[s_]), [dv] "v"(dv) : "memory", "scc");
                        asm volatile("s_mov_b32 m0, %[keep]" : : [keep] "s"(keep) : "memory");
                        dst += (unsigned int)(n_lw * R) * 1024u;
                    }
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int k2 = ph * R + r, km2 = k2 ? N2 - k2 : 0;
                unsigned int voff;
                if (blocked) {      // block cb, row k2 | block 63 - cb, row km2: 512 contiguous bytes each, the next row right behind
                    const unsigned int of = 8u * ((unsigned int)cb * (N2 * 64u) + (unsigned int)k2 * 64u);
                    const unsigned int om = 8u * ((unsigned int)(63 - cb) * (N2 * 64u) + (unsigned int)km2 * 64u);
                    voff = 16u * (unsigned int)(lane & 31) + (lane < 32 ? of : om);
                } else {
                    const unsigned int of = 8u * ((unsigned int)k2 * N1 + (unsigned int)(k2 >> 8) * (unsigned int)zpad);
                    const unsigned int om = 8u * ((unsigned int)km2 * N1 + (unsigned int)(km2 >> 8) * (unsigned int)zpad);
                    voff = lane_off + (lane < 32 ? of : om);
                }
                const unsigned int dst_row = lds0 + (unsigned int)(buf * S * R + lw * R + r) * 1024u;      // + n_lw R KB per station of this loader
                // One block for the row's stations, laid out by hand: a station that is present FALLS THROUGH to its LDS-DMA
                // (bit test, untaken branch, the load, M0 += the station stride: four issue slots).  Written as sixteen `if`s the
                // compiler moved every present station out of line -- two taken branches, three scalar operations for the LDS
                // address and an M0 save / restore around each load, ~90 cycles per LDS-DMA on a wave that can issue one
                // instruction every four: the loader needed 5.8 k cycles per phase of cfg4 (64 loads) against the 5.4 k its
                // fourteen walks compute in, and twice the walks' time on cfg5 (measurement build -DTDOA_STG_TIMING: the walks
                // spent 32 - 45 % of their cycles at the barrier).  The instructions between an M0 write and the next LDS-DMA
                // (bit test + branch) are the wait state that pair needs.
                unsigned int keep, tmp;
#ifndef TDOA_STG_DMA_MOD
#define TDOA_STG_DMA_MOD ""                  // cache-policy bits of the LDS-DMA (measurement builds: " nt", " sc1", ...)
#endif
#define TDOA_STG_DMA(k_)                                                                                                       \
    "s_bitcmp1_b32 %[mm], " #k_ "\n\ts_cbranch_scc0 .Lstg%=_" #k_ "\n\tglobal_load_lds_dwordx4 %[voff], %[b" #k_ "]" TDOA_STG_DMA_MOD "\n\t" \
    "s_add_u32 m0, m0, %[stride]\n.Lstg%=_" #k_ ":\n\t"
#define TDOA_STG_MORE(k_) "s_lshr_b32 %[tmp], %[mm], " #k_ "\n\ts_cbranch_scc0 .Lstg%=_end\n\t"
                asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\t"
                             TDOA_STG_DMA(0) TDOA_STG_DMA(1) TDOA_STG_DMA(2) TDOA_STG_DMA(3) TDOA_STG_MORE(4)
                             TDOA_STG_DMA(4) TDOA_STG_DMA(5) TDOA_STG_DMA(6) TDOA_STG_DMA(7) TDOA_STG_MORE(8)
                             TDOA_STG_DMA(8) TDOA_STG_DMA(9) TDOA_STG_DMA(10) TDOA_STG_DMA(11) TDOA_STG_MORE(12)
                             TDOA_STG_DMA(12) TDOA_STG_DMA(13) TDOA_STG_DMA(14) TDOA_STG_DMA(15)
                             ".Lstg%=_end:\n\ts_mov_b32 m0, %[keep]"
                             : [keep] "=&s"(keep), [tmp] "=&s"(tmp)
                             : [voff] "v"(voff), [mm] "s"(mine), [dst] "s"(dst_row), [stride] "s"((unsigned int)(n_lw * R) * 1024u),
                               [b0] "s"(sbase[0]), [b1] "s"(sbase[1]), [b2] "s"(sbase[2]), [b3] "s"(sbase[3]), [b4] "s"(sbase[4]),
                               [b5] "s"(sbase[5]), [b6] "s"(sbase[6]), [b7] "s"(sbase[7]), [b8] "s"(sbase[8]), [b9] "s"(sbase[9]),
                               [b10] "s"(sbase[10]), [b11] "s"(sbase[11]), [b12] "s"(sbase[12]), [b13] "s"(sbase[13]),
                               [b14] "s"(sbase[14]), [b15] "s"(sbase[15])
                             : "memory", "scc");
#undef TDOA_STG_DMA
#undef TDOA_STG_MORE
            }
        };
        // phases 0 .. nb - 2 go out at once; at barrier ph the readers are through with phase ph - 1, whose buffer then takes
        // phase ph + nb - 1
        TDOA_STG_T(const unsigned long long t_in = __builtin_readcyclecounter(); unsigned long long t_vm = 0, t_bar = 0, t_iss = 0;)
        for (int ph = 0; ph < nb - 1 && ph < NP; ph++) issue(ph, ph);
        int buf_next = (nb - 1) % nb;
        for (int ph = 0; ph < NP; ph++) {
            const int later = NP - 1 - ph < nb - 2 ? NP - 1 - ph : nb - 2;      // phases issued after ph that may stay in flight
            TDOA_STG_T(const unsigned long long t0 = __builtin_readcyclecounter();)
            stg_wait_vm(later * per_phase);                        // phase ph has landed
            TDOA_STG_T(const unsigned long long t1 = __builtin_readcyclecounter();)
            __builtin_amdgcn_s_barrier();
            TDOA_STG_T(const unsigned long long t2 = __builtin_readcyclecounter(); t_vm += t1 - t0; t_bar += t2 - t1;)
            TDOA_STG_T(const unsigned long long t3 = __builtin_readcyclecounter();)
            if (ph + nb - 1 < NP) issue(ph + nb - 1, buf_next);
            TDOA_STG_T(t_iss += __builtin_readcyclecounter() - t3;)
            buf_next = buf_next + 1 == nb ? 0 : buf_next + 1;
        }
        TDOA_STG_T(if (lane == 0) { atomicAdd(&g_stg_prof[2], __builtin_readcyclecounter() - t_in); atomicAdd(&g_stg_prof[3], t_vm);
                                    atomicAdd(&g_stg_prof[4], t_bar); atomicAdd(&g_stg_prof[6], 1ull); atomicAdd(&g_stg_prof[7], t_iss); })
        return;
    }
    if (wave >= p_cnt) {                                           // a compute wave without a pair (last group): barriers only
        for (int ph = 0; ph < NP; ph++) {
            if (has_duty) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (has_duty && ph + 1 < NP) issue_one(ph + 1, (ph + 1) & 1);
        }
        return;
    }

    // ---- compute: the walk of k_pair_decimate_cols for pair-window pwu, operands from LDS -----------------------------------
    const int pidx = gd->pair[wave];
    const unsigned int pwu = (unsigned int)(w * P + pidx);
    const PWDesc d = wpw[pidx];
    const int sa = __builtin_popcount(mask & ((1u << (d.sw_a - sw_base)) - 1u)), sb = __builtin_popcount(mask & ((1u << (d.sw_b - sw_base)) - 1u));
    const int k1 = cb * 64 + lane, km = N1 - 1 - k1;
    const float2 *Ua = U + (size_t)d.sw_a * (size_t)u_stride, *Ub = U + (size_t)d.sw_b * (size_t)u_stride;
    const float invNc = 1.0f / (float)pl.Nc;
    auto w_n = [&](float num) {
        if constexpr (POW2) return unit_root(num, invNc, false);
        else return unit_root_any(num, 0.5f * (float)pl.Nc, 2.0f * invNc, false);
    };
    const size_t rc = (size_t)(pl.Nc / kDecD);
    float2 *g_top = G + (size_t)pwu * rc + k1, *g_bot = G + (size_t)pwu * rc + km;
    float2 *x_top = X + (size_t)pwu * kDecShareRows * N1 + k1, *x_bot = X + (size_t)pwu * kDecShareRows * N1 + km;

    float2 at[SS], ab[SS];
#pragma unroll
    for (int s = 0; s < SS; s++) at[s] = ab[s] = make_float2(0.0f, 0.0f);
    typedef float v2f __attribute__((ext_vector_type(2)));
    // The taps of a row's phase are the same twelve numbers for every lane AND every wave: they are read through the scalar
    // cache into SGPR pairs (s_load_dwordx8 + x4 at a wave-uniform address, issued next to the rows' LDS reads) and enter the
    // multiply-adds as the scalar operand (v_pk_fma_f32 v, v, s[lo:hi] with op_sel picking the half).  From LDS they were three
    // ds_read_b128 per row and wave -- 3 KB of LDS traffic beside the 2 KB of operands, the LDS pipe at ~95 % of the walks'
    // issue time (cfg4 pair step 3.45 -> 3.08 ms, cfg5 83.7 -> 79.2).  Table: [16 phases][16 steps] floats, rot16 at 256, phase 0
    // with its steps reversed at 288 (ensure_decimation).
    struct TapRow { v2f h2[SS / 2]; };
    const float2 *rot16 = reinterpret_cast<const float2 *>(taps + 256);
    auto tap_row = [&](int p) {
        static_assert(SS % 4 == 0, "whole float4 of taps per phase");
        TapRow r;
        const float4 *tp = reinterpret_cast<const float4 *>(taps + (p < 16 ? 16 * p : 288));
#pragma unroll
        for (int s = 0; s < SS / 4; s++) {
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
        if (s & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a) : "v"(q), "s"(tr.h2[s / 2]));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(a) : "v"(q), "s"(tr.h2[s / 2]));
        acc = make_float2(a.x, a.y);
    };
    auto mac_all = [&](float2 (&acc)[SS], const TapRow &tr, float2 q) {
        mac(acc[0], tr, std::integral_constant<int, 0>{}, q);
        mac(acc[1], tr, std::integral_constant<int, 1>{}, q);
        mac(acc[2], tr, std::integral_constant<int, 2>{}, q);
        mac(acc[3], tr, std::integral_constant<int, 3>{}, q);
        mac(acc[4], tr, std::integral_constant<int, 4>{}, q);
        mac(acc[5], tr, std::integral_constant<int, 5>{}, q);
        mac(acc[6], tr, std::integral_constant<int, 6>{}, q);
        mac(acc[7], tr, std::integral_constant<int, 7>{}, q);
        if constexpr (SS > 8) {
            mac(acc[8], tr, std::integral_constant<int, 8>{}, q);
            mac(acc[9], tr, std::integral_constant<int, 9>{}, q);
            mac(acc[10], tr, std::integral_constant<int, 10>{}, q);
            mac(acc[11], tr, std::integral_constant<int, 11>{}, q);
        }
    };
    // row 0 of column k1 pairs inside row 0 (dec_stream.hpp): straight from memory, once per walk
    {
        const int kp = (N1 - k1) & (N1 - 1);
        const TapRow tr = tap_row(0);
        float2 q, qm;
        pair_u_pk(*row0_at(Ua, k1), *row0_at(Ua, kp), *row0_at(Ub, k1), *row0_at(Ub, kp), w_n((float)k1 * (float)N2), k1 == 0, q, qm);
        mac_all(at, tr, q);
    }
    x_top[0] = make_float2(0.0f, 0.0f);
    x_bot[0] = make_float2(0.0f, 0.0f);
    auto bottom_leaves = [&](int i) {
        if (i >= NG) x_bot[(size_t)(kDecEdge + i - NG) * N1] = ab[SS - 1];
        else if (i >= 0) g_bot[(size_t)i * N1] = ab[SS - 1];
        else x_bot[(size_t)(kDecEdge + i) * N1] = ab[SS - 1];
#pragma unroll
        for (int u = SS - 1; u > 0; u--) ab[u] = ab[u - 1];
        ab[0] = make_float2(0.0f, 0.0f);
    };
    auto top_leaves = [&](int i) {
        if (i < 0) x_top[(size_t)(kDecEdge + i) * N1] = at[SS - 1];
        else if (i < NG) g_top[(size_t)i * N1] = at[SS - 1];
        else x_top[(size_t)(kDecEdge + i - NG) * N1] = at[SS - 1];
#pragma unroll
        for (int s = SS - 1; s > 0; s--) at[s] = at[s - 1];
        at[0] = make_float2(0.0f, 0.0f);
    };
    // LDS byte offsets of this lane's four operands inside a (phase, row) block of S KB: column k1 of the forward piece,
    // column km = 4095 - k1 of the partner piece (its block is stored ascending: lane 63 - l)
    const unsigned int o_af = (unsigned int)(sa * R) * 1024u + 8u * (unsigned int)lane, o_am = (unsigned int)(sa * R) * 1024u + 512u + 8u * (unsigned int)(63 - lane);
    const unsigned int o_bf = (unsigned int)(sb * R) * 1024u + 8u * (unsigned int)lane, o_bm = (unsigned int)(sb * R) * 1024u + 512u + 8u * (unsigned int)(63 - lane);
    auto lds_at = [&](unsigned int off) { return *reinterpret_cast<const float2 *>(stage_raw + off); };
    float2 wg = make_float2(1.0f, 0.0f);
    // tr: the taps of the row's phase; tb (FIRST rows only): the upward walk's taps -- phase 0: the reversed row
    auto row = [&](float2 ua, float2 uam, float2 ub, float2 ubm, int k2, auto first_c, const TapRow &tr, const TapRow &tb) {
        constexpr bool FIRST = decltype(first_c)::value;            // the only row of a phase whose FIR phase can be 0
        const int p = k2 & 15, g = k2 >> 4;
        float2 q, qm;
        pair_u_pk(ua, uam, ub, ubm, cmul(wg, rot16[p]), false, q, qm);
        if (FIRST && k2 == 0) q = qm = make_float2(0.0f, 0.0f);     // row 0: done above
        mac_all(at, tr, q);
        if (!FIRST) {
            mac_all(ab, tr, qm);
        } else {
            mac_all(ab, tb, qm);
            if (p == 0 && g > 0) bottom_leaves(NG - g + C);
        }
    };
    int buf = 0;
    TDOA_STG_T(const unsigned long long t_in = __builtin_readcyclecounter(); unsigned long long t_bar = 0;)
#pragma unroll 1
    for (int ph = 0; ph < NP; ph++) {
        TDOA_STG_T(const unsigned long long t0 = __builtin_readcyclecounter();)
        if (has_duty) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of phase ph has landed
        __builtin_amdgcn_s_barrier();                              // phase ph is in LDS
        TDOA_STG_T(t_bar += __builtin_readcyclecounter() - t0;)
        if (has_duty && ph + 1 < NP) issue_one(ph + 1, (ph + 1) & 1);      // (folded form: nb = 2)
        const unsigned int base = (unsigned int)(buf * R) * (unsigned int)S * 1024u;
        const unsigned int a_af = base + o_af, a_am = base + o_am, a_bf = base + o_bf, a_bm = base + o_bm;      // this phase's operands
        if (++buf == nb) buf = 0;
        const int k2 = ph * R;
        if ((k2 & 15) == 0) wg = w_n((float)k2 + (float)k1 * (float)N2);
        // two rows at a time (their eight reads in flight together; four rows' worth of operands did not fit 128 registers)
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 2) {
            float2 v[2][4];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                v[r][0] = lds_at(a_af + 1024u * (unsigned int)(r0 + r));
                v[r][1] = lds_at(a_am + 1024u * (unsigned int)(r0 + r));
                v[r][2] = lds_at(a_bf + 1024u * (unsigned int)(r0 + r));
                v[r][3] = lds_at(a_bm + 1024u * (unsigned int)(r0 + r));
            }
            const int p0 = (k2 + r0) & 15;
            const TapRow t0 = tap_row(p0), t1 = tap_row(p0 + 1);
            if (r0 == 0) {
                const TapRow tb = tap_row(p0 ? p0 : 16);
                row(v[0][0], v[0][1], v[0][2], v[0][3], k2, std::true_type{}, t0, tb);
            } else {
                row(v[0][0], v[0][1], v[0][2], v[0][3], k2 + r0, std::false_type{}, t0, t0);
            }
            row(v[1][0], v[1][1], v[1][2], v[1][3], k2 + r0 + 1, std::false_type{}, t1, t1);
        }
        if (((k2 + R - 1) & 15) == 15) top_leaves((k2 >> 4) - (SS - 1 - C));
    }
#pragma unroll
    for (int n = 0; n < SS - 1; n++) top_leaves(NG - (SS - 1 - C) + n);
    {
        const TapRow tr = tap_row(16);
        float2 q, qm;
        pair_u_pk(*row0_at(Ua, km), *row0_at(Ua, k1 + 1), *row0_at(Ub, km), *row0_at(Ub, k1 + 1), w_n((float)km * (float)N2), false, q, qm);
        mac_all(ab, tr, q);
    }
#pragma unroll
    for (int n = 0; n < SS; n++) bottom_leaves(C - n);
    TDOA_STG_T(if (lane == 0) { atomicAdd(&g_stg_prof[0], __builtin_readcyclecounter() - t_in); atomicAdd(&g_stg_prof[1], t_bar);
                                atomicAdd(&g_stg_prof[5], 1ull); })
}

#endif  // TDOA_HAVE_DEC_COLS

}  // namespace tdoa
