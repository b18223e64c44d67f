// tdoa_mi355x.hip -- C-ABI implementation (include/tdoa_mi355x.h) for gfx950.
//
// Host side: context, FFT plans, (station, window) x (pair, window) batching,
// HIP stream + event plumbing.  Device side: the kernels in the headers below.
// There is no CPU fallback: without a HIP device every compute entry point
// fails with TDOA_ERR_NO_DEVICE.
#include "../../include/tdoa_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "device_common.hpp"
#include "k1_discriminator.hpp"
#include "k1_single_look.hpp"
#include "fft_stockham.hpp"
#include "fft_radix16.hpp"
#include "fft_radix8.hpp"
#include "dec_stream.hpp"
#include "dec_staged.hpp"
#include "exact_reference.hpp"
#include "synth_capture.hpp"
#include "window_quality.hpp"
#include "host_geodesy.hpp"
#include "host_upload.hpp"
#include "segment_quads.hpp"

using namespace tdoa;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct ProfRec {
    int kernel;
    int e0, e1;          // indices into the context's event pool
    double bytes;
};

}  // namespace

struct tdoa_ctx {
    tdoa_params prm;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;

    struct Capture {
        const uint8_t *dev = nullptr;
        size_t n = 0;
        bool owned = false;
        size_t cap_bytes = 0;               // size of an owned allocation (reused by the next upload if it fits)
    };
    std::vector<Capture> caps;

    DevBuf k1_direct;                       // kK1DirectEntries half-plane angle codes of the streaming K1 kernel
    DevBuf k1_quad;                         // kK1QuadrantEntries first-quadrant angle codes (k_fwd_col256_k1)
    DevBuf sw_desc, pw_desc, partials, stats, codes, codes_lp, k1_power, tz, v, keys, scales, peaks, scratch_a, scratch_b, lagdump;
    DevBuf ex_a, ex_b, ex_c, ex_d, ex_part;

    bool profiling = false;
    unsigned int prof_mask = ~0u;           // scopes that record events (tdoa_profile_select)
    // profiling INSIDE the replayed step graph (tdoa_profile_enable(ctx, 2)): while the step is captured the selected scopes
    // note the capture's last node before their first and after their last kernel; after the capture an event-record NODE
    // goes in at either place (events recorded on a capturing stream are dropped by this ROCm; explicit nodes are timed
    // correctly: scripts/microbench/graph_event_nodes.hip)
    bool graph_prof = false;
    bool capturing = false;
    struct GraphMark { int kernel; double bytes; hipGraphNode_t before, last; hipEvent_t e0, e1; };
    std::vector<GraphMark> graph_marks;
    bool force_generic = false;   // tests: run the any-size kernels even at the hot sizes
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> prof_pool;      // events of the profiling path, reused from call to call
    size_t prof_used = 0;                   // handed out since the last prof_collect
    int prof_last = -1;                     // stop event of the previous scope: the next scope starts there (one event
                                            // between two kernels instead of two)
    double prof_ms[TDOA_K_COUNT] = {0};
    int64_t prof_launches[TDOA_K_COUNT] = {0};
    double prof_bytes[TDOA_K_COUNT] = {0};

    FftPlan plan{};
    int64_t plan_n = 0;

    // whole-step hipGraph of tdoa_process (launch-bound when windows are processed in many groups)
    int n_cu = 256;                         // multiprocessors of this device
    double workspace_limit = 24.0 * 1073741824.0;      // bytes of FFT workspace one launch group may take: a third of the device's memory
    bool use_graph = true;                  // TDOA_NO_GRAPH=1 at tdoa_create time turns the whole-step hipGraph off
    bool short_lag = true;                  // TDOA_NO_SHORT_LAG=1 at tdoa_create time forces the general inverse for short searches
    bool segment_form = true;               // TDOA_NO_SEGMENT_FORM=1: no LDS-resident overlap-save form for short searches
    bool segment_quads = true;              // TDOA_NO_SEGMENT_QUADS=1: segment form one pair-window at a time (no shared station transforms)
    bool memset_nodes = false;              // TDOA_DEBUG_MEMSET_NODES=1 (probe only): zero the step's accumulators with
                                            // hipMemsetAsync nodes instead of k_zero_u64 kernel nodes
    bool dec_cols = true;                   // TDOA_NO_DEC_COLS=1: the tile form of the decimated pair step (k_pair_decimate16; none on 4096 x 4096 plans)
    bool dec_cols_always = false;           // TDOA_DEC_COLS_ALWAYS=1: the column walk wherever the decimated inverse applies (measurements)
    bool dec_staged = true;                 // TDOA_NO_DEC_STAGED=1: the column walk one pair-window per wave from memory (k_pair_decimate_cols), no LDS staging
    int stg_loaders = 0;                    // TDOA_DEC_STAGED_LOADERS=n: loader waves per workgroup of k_pair_decimate_staged (0: the library's choice)
    // the staged walk's share-out of a window's pairs to workgroups, for every station count 2 .. 16 (build_stg_groups)
    struct StgTable { int off = 0, count = 0, slots = 0, max_n = 0; } stg_tab[kStgMaxStations + 1], stg_tab16[kStgMaxStations + 1];      // (..16: the folded form's, sixteen walks per workgroup)
    bool stg_folded_always = false;         // TDOA_STG_FOLDED_ALWAYS=1: ... wherever the blocked layout applies (tests)
    bool stg_folded = true;                 // TDOA_NO_STG_FOLDED=1: always a loader wave next to at most fifteen walks
    DevBuf stg_groups;
    bool stg_ready = false;
    bool small_fused_always = false;        // TDOA_SMALL_FUSED_ALWAYS=1 / tdoa_debug_flags: ... for any number of pair-windows (tests)
    bool small_fused = true;                // TDOA_NO_SMALL_FUSED=1: the small plan of the decimated inverse as two kernels with V' in memory between them
    bool stg_blocks = true;                 // TDOA_NO_STG_BLOCKS=1: the staged walk reads row-major spectra on every plan
    int stg_cw = 0;                         // TDOA_DEC_STAGED_CW=n: at most n walks (compute waves) per workgroup (0: fifteen -- sixteen waves less the loader)
    int stg_rows = 0, stg_bufs = 0;         // TDOA_DEC_STAGED_ROWS=2|4|8, TDOA_DEC_STAGED_BUFS=n: rows per phase, phases in the LDS ring (0: the library's choice)
    bool seg_pack3 = true;                  // TDOA_NO_SEG_PACK3=1: the segment form reads int32 code rows (round 3's layout)
    int seg_chunks_override = 0;            // TDOA_SEG_CHUNKS=n at tdoa_create time: chunk count of the segment form
    int graph_nodes = 0, graph_edges = 0, graph_roots = 0, graph_memsets = 0;      // structure of the captured step (tdoa_debug_graph_info)
    bool fused_k1 = true;                   // TDOA_NO_FUSED_K1=1: K1 always materialises its codes (no discriminator inside the column kernels)
    int zpad = 256;                         // TDOA_ZPAD=n at tdoa_create time: padding (elements) after every 256 rows of a two-sweep plan's TZ:
                                            // 2 KB; measured on cfg3: 0 -> 107 ms column pass, 128 -> 91, 256 -> 87, 512 -> 89
    bool decimate = true;                   // TDOA_NO_DECIMATE=1: general form with the full inverse even where the decimated one applies
    bool pow2_only = false;                 // TDOA_POW2_ONLY=1: transform lengths are powers of two everywhere (no 5 x 2^22 plan for ten-second windows)
    bool k1_once = true;                    // TDOA_NO_K1_ONCE=1: the statistics pre-pass everywhere (no single-look K1, k1_single_look.hpp)
    bool once_active = false;               // the last step (run_fm_batch, or the replayed graph) took the single-look path: decode multiplies by slot_gain
    bool graph_once = false;                // ... of the step the cached graph holds (a pair call on another path in between must not change what a replay reports)
    DevBuf once_edges, once_tiles, once_fin, slot_gain;
    // decimated inverse (k_pair_decimate16): FIR taps and window correction for (Nc, reach); small plan of the R-point inverse
    DevBuf dec_taps, dec_gain;
    long long dec_nc = 0;
    int dec_reach = -1, dec_T = 0;
    bool xcd_rows = true;                   // TDOA_NO_XCD_ROWS=1: plain 2-D grid of the pair kernel even with more pairs than stations
    int xcd_pair_mb = 48;                   // TDOA_XCD_PAIR_MB=n: k_pair_decimate16 groups a window's pair-windows on one XCD when the
                                            // window's spectra exceed n MB (round 4, same-box A/B: cfg4, 8 x 8.4 MB, 11.05 ms grouped
                                            // against 11.20 -- its pair step pulled 27 GB per step through the fabric for 5.6 GB of
                                            // spectra; cfg2, 3 x 8.4 MB: 0.69 ms grouped against 0.66 plain)
    uint64_t alloc_gen = 0;                 // bumped whenever a workspace buffer moves
    std::vector<uint64_t> graph_key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    DevBuf g_sw_desc, g_pw_desc, g_quad_desc, g_scales, g_keys;
    std::map<std::vector<int>, std::vector<StationQuad>> quad_cache;   // owned pair ids of a window -> its quad cover
    DevBuf qual;                            // QualAcc per (window, station)
    StagedUploader uploader;                // pinned staging buffers + copy streams, created on first use
    DevBuf fine_raw, fine;                  // (f)-4 refinement: 3 raw neighbours and tdoa_fine_peak per slot
};

namespace {

static_assert(sizeof(PeakOut) == sizeof(tdoa_peak), "tdoa_peak layout");
static_assert(sizeof(FmStats) == sizeof(tdoa_fm_stats), "tdoa_fm_stats layout");

int fail(tdoa_ctx *ctx, int status, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        char buf[512];
        if (e != hipSuccess)
            snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        else
            snprintf(buf, sizeof(buf), "%s", what);
        ctx->last_error = buf;
    }
    return status;
}

#define HIPCHK(ctx, call)                                              \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) return fail(ctx, TDOA_ERR_HIP, #call, e_); \
    } while (0)

int ensure(tdoa_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return TDOA_OK;
    if (b.p) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, TDOA_ERR_NOMEM, "hipMalloc", e);
    }
    b.cap = want;
    ctx->alloc_gen++;
    return TDOA_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

int ilog2(long long v)
{
    int l = 0;
    while ((1ll << l) < v) l++;
    return l;
}

long long next_pow2(long long n)   // processor.go:502-512
{
    long long p = 1;
    while (p < n) p <<= 1;
    return p;
}

constexpr size_t kLdsCap = 128 * 1024;

// factor Nc = N1 * N2 for the four-step FFT; rows (N1) live whole in LDS
// zpad: elements of padding after every 256 rows of a two-sweep plan's TZ (tdoa_ctx::zpad; 0 for the small plans)
int make_plan(long long n_real, bool packed, FftPlan *pl, int zpad = 0)
{
    long long nc = packed ? n_real / 2 : n_real;
    // 5 x 2^k (round 5): only the two shapes that have kernels -- 4096 x 2560 (N = 5 x 2^22, the column pass as ten 256-point
    // sub-transforms + k_fwd_col_finish<10>, the pair step as a column walk) and the small plan of its decimated inverse,
    // 4096 x 160
    // 3 x 2^k likewise: 4096 x 3072 (N = 3 x 2^23: twelve sub-transforms + k_fwd_col_finish<12>) and its small plan 4096 x 192
    const int odd = (nc == 4096ll * 2560 || nc == 4096ll * 160) ? 5 : (nc == 4096ll * 3072 || nc == 4096ll * 192) ? 3 : 1;
    if (nc < 32 || nc > (1ll << 24) || (odd == 1 && (nc & (nc - 1)))) return TDOA_ERR_UNSUPPORTED;
    long long n1, n2;
    pl->odd = odd;
    if (nc >= 65536) {
        n1 = 4096;
        n2 = nc / n1;
    } else if (nc >= 256) {
        n2 = 16;
        n1 = nc / n2;
    } else {
        n2 = 2;
        n1 = nc / n2;
    }
    long long c = 32;
    while (c > 1 && (c > n1 || (size_t)(2 * n2 * c * 8) > kLdsCap)) c >>= 1;
    pl->N1 = (int)n1;
    pl->N2 = (int)n2;
    pl->logN1 = ilog2(n1);
    pl->logN2 = ilog2(n2);
    pl->C = (int)c;
    pl->logC = ilog2(c);
    pl->Nc = nc;
    pl->zpad = n1 == 4096 && (n2 == 4096 || n2 == 2048 || n2 == 2560 || n2 == 3072) ? zpad : 0;      // two-sweep column pass (fft_stockham.hpp, FftPlan)
    pl->Zs = nc + (long long)(n2 / 256) * pl->zpad;
    return TDOA_OK;
}

template <typename K>
int set_lds(tdoa_ctx *ctx, K kernel, size_t bytes)
{
    if (bytes > 48 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return TDOA_OK;
}

void clear_graph_marks(tdoa_ctx *ctx)
{
    for (auto &m : ctx->graph_marks) {
        if (m.e0) (void)hipEventDestroy(m.e0);
        if (m.e1) (void)hipEventDestroy(m.e1);
    }
    ctx->graph_marks.clear();
}

// next free event of the pool, recorded on the context's stream; -1 on failure
int prof_mark(tdoa_ctx *ctx)
{
    if (ctx->prof_used == ctx->prof_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        ctx->prof_pool.push_back(e);
    }
    const int id = (int)ctx->prof_used++;
    if (hipEventRecord(ctx->prof_pool[id], ctx->stream) != hipSuccess) return -1;
    return id;
}

// Per-kernel timing of the profiling path: consecutive scopes share the event between them (the stop of one is the
// start of the next), so a step costs one event per kernel boundary; the few microseconds between two kernels count
// towards the later one.  Work enqueued outside any scope must reset ctx->prof_last first.
struct ProfScope {
    tdoa_ctx *ctx;
    ProfRec rec{};
    bool on;
    int mark = -1;                           // graph mode: index into ctx->graph_marks
    static hipGraphNode_t capture_tail(tdoa_ctx *c)
    {
        hipStreamCaptureStatus stt;
        const hipGraphNode_t *deps = nullptr;
        size_t nd = 0;
        if (hipStreamGetCaptureInfo_v2(c->stream, &stt, nullptr, nullptr, &deps, &nd) != hipSuccess || nd != 1) return nullptr;
        return deps[0];
    }
    ProfScope(tdoa_ctx *c, int kernel, double bytes) : ctx(c), on(c->profiling)
    {
        if (c->graph_prof && c->capturing && ((c->prof_mask >> kernel) & 1u)) {
            tdoa_ctx::GraphMark m{kernel, bytes, capture_tail(c), nullptr, nullptr, nullptr};
            if (m.before) {
                mark = (int)c->graph_marks.size();
                c->graph_marks.push_back(m);
            }
        }
        if (on && !((c->prof_mask >> kernel) & 1u)) {      // not selected: its launches are unscoped work
            on = false;
            c->prof_last = -1;
        }
        if (!on) return;
        rec.kernel = kernel;
        rec.bytes = bytes;
        rec.e0 = ctx->prof_last >= 0 ? ctx->prof_last : prof_mark(ctx);
        if (rec.e0 < 0) on = false;
    }
    ~ProfScope()
    {
        if (mark >= 0) ctx->graph_marks[mark].last = capture_tail(ctx);
        if (!on) return;
        rec.e1 = prof_mark(ctx);
        ctx->prof_last = rec.e1;
        if (rec.e1 >= 0) ctx->recs.push_back(rec);
    }
};

void prof_collect(tdoa_ctx *ctx)
{
    for (auto &r : ctx->recs) {
        float ms = 0;
        if (hipEventSynchronize(ctx->prof_pool[r.e1]) == hipSuccess &&
            hipEventElapsedTime(&ms, ctx->prof_pool[r.e0], ctx->prof_pool[r.e1]) == hipSuccess) {
            ctx->prof_ms[r.kernel] += ms;
            ctx->prof_launches[r.kernel] += 1;
            ctx->prof_bytes[r.kernel] += r.bytes;
        }
    }
    ctx->recs.clear();
    ctx->prof_used = 0;
    ctx->prof_last = -1;
}

// The K1 angle table (k1_discriminator.hpp): first-octant directions (mn, mx), index mx (mx + 1) / 2 + mn over the
// indices of the odd magnitudes 2 idx + 1; entry = llround(atan2(mn', mx') 2^23 / pi) of the gcd-reduced pair, float64.
// (oracle/tdoa_oracle.c: ob_octant_code states the same expression; tests compare the device's codes with it bit for bit)
void k1_build_table_host(std::vector<int32_t> &tab, std::vector<int32_t> &direct, std::vector<int32_t> &quad)
{
    tab.resize(kK1TableEntries);
    for (int mx = 0; mx < 128; mx++)
        for (int mn = 0; mn <= mx; mn++) {
            int a = 2 * mx + 1, b = 2 * mn + 1;
            int g = a, h = b;
            while (h) { const int t = g % h; g = h; h = t; }
            a /= g;
            b /= g;
            tab[(size_t)mx * (mx + 1) / 2 + mn] = (int32_t)std::llround(std::atan2((double)b, (double)a) * (8388608.0 / M_PI));
        }
    // the direct half-plane table of the streaming kernel: D[b_I | (b_Q & 0x7f) << 8] = a(I, Q) for b_Q >= 128 (Q > 0),
    // placed from the first-octant codes by the integer rules of k1_discriminator.hpp
    direct.resize(kK1DirectEntries);
    for (int bq = 128; bq < 256; bq++)
        for (int bi = 0; bi < 256; bi++) {
            const int ia = bi >= 128 ? bi - 128 : 127 - bi, iq = bq - 128;
            const int mx = std::max(ia, iq), mn = std::min(ia, iq);
            int c = tab[(size_t)mx * (mx + 1) / 2 + mn];
            if (iq > ia) c = (kK1Half >> 1) - c;
            if (bi < 128) c = kK1Half - c;
            direct[(size_t)bi | ((size_t)(bq & 0x7f) << 8)] = c;
        }
    // the first-quadrant table Q[iq][ia] = a(2 ia + 1, 2 iq + 1): the |Q| > |I| reflection done here instead of per sample
    quad.resize(kK1QuadrantEntries);
    for (int iq = 0; iq < 128; iq++)
        for (int ia = 0; ia < 128; ia++) {
            const int mx = std::max(ia, iq), mn = std::min(ia, iq);
            const int c = tab[(size_t)mx * (mx + 1) / 2 + mn];
            quad[(size_t)iq * 128 + ia] = (iq > ia ? (kK1Half >> 1) - c : c) * 256;      // scaled: a full turn = 2^32
        }
}

// zero n_sw window accumulators (a kernel node: the captured step holds kernel nodes only)
void zero_partials(hipStream_t st, StatsPartial *partials, int n_sw, bool memset_node = false)
{
    static_assert(sizeof(StatsPartial) == 32, "four 64-bit words per station-window");
    const size_t words = 4 * (size_t)n_sw;
    if (memset_node) {                       // probe only (TDOA_DEBUG_MEMSET_NODES=1)
        (void)hipMemsetAsync(partials, 0, 8 * words, st);
        return;
    }
    hipLaunchKernelGGL(k_zero_u64, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<unsigned long long *>(partials), words);
}

// K1 for n_sw station-windows: capture bytes -> exact window statistics, and -- when `materialise` -- the 24-bit codes
// (int32) for the consumers that read them from memory; every buffer must have been reserved (no allocation here: the
// caller may be capturing a graph).  Returns the code array downstream reads (nullptr: fused path, the forward column
// kernels evaluate the discriminator themselves).
// Optional steps (tdoa_params): k1_gate -- the prebuilt binary's power gate (windows of mean power <= 0.01 get envelope
// codes instead of phase codes); k1_smooth -- its moving average on the discriminator output.  Both need the codes.
// pack3: the codes go to memory at 3 bytes each (k1_store8_packed; only the segment kernels read that layout, so it is
// never combined with k1_gate / k1_smooth, whose kernels work on int32 rows).
int *launch_k1(tdoa_ctx *ctx, hipStream_t st, const SWDesc *d_sw, int n_sw, int maxlen, int pieces, long long code_stride,
               bool materialise, bool pack3 = false)
{
    auto *partials = static_cast<StatsPartial *>(ctx->partials.p);
    auto *stats = static_cast<FmStats *>(ctx->stats.p);
    auto *codes = static_cast<int *>(ctx->codes.p);
    const auto *table = static_cast<const int *>(ctx->k1_direct.p);
    const dim3 per_chunk((unsigned)((maxlen + 2047) / 2048), n_sw);
    unsigned long long *power = nullptr;
    if (ctx->prm.k1_gate) {
        power = static_cast<unsigned long long *>(ctx->k1_power.p);
        hipLaunchKernelGGL(k_zero_u64, dim3((unsigned)(((size_t)n_sw + 255) / 256)), dim3(256), 0, st, power, (size_t)n_sw);
        hipLaunchKernelGGL(k_k1_power, per_chunk, dim3(256), 0, st, d_sw, power);
    }
    zero_partials(st, partials, n_sw, ctx->memset_nodes);
    const long long items = (long long)((pieces + kDemodItem - 1) / kDemodItem) * n_sw;      // workgroup items
    const int blocks = (int)std::max<long long>(1, std::min<long long>(items, ctx->n_cu));        // one workgroup per CU (128 KB table)
    if (materialise && pack3)
        hipLaunchKernelGGL((k_fm_demod<true, true>), dim3(blocks), dim3(kDemodThreads), kK1DirectBytes, st, d_sw, n_sw, pieces, table,
                           codes, code_stride, partials, power);
    else if (materialise)
        hipLaunchKernelGGL(k_fm_demod<true>, dim3(blocks), dim3(kDemodThreads), kK1DirectBytes, st, d_sw, n_sw, pieces, table,
                           codes, code_stride, partials, power);
    else
        hipLaunchKernelGGL(k_fm_demod<false>, dim3(blocks), dim3(kDemodThreads), kK1DirectBytes, st, d_sw, n_sw, pieces, table,
                           static_cast<int *>(nullptr), code_stride, partials, power);
    if (power) hipLaunchKernelGGL(k_k1_envelope, per_chunk, dim3(256), 0, st, d_sw, power, codes, code_stride, partials);
    if (ctx->prm.k1_smooth > 1) {
        // statistics of the smoothed codes replace those of the raw ones
        auto *lp = static_cast<int *>(ctx->codes_lp.p);
        zero_partials(st, partials, n_sw);
        hipLaunchKernelGGL(k_k1_smooth, per_chunk, dim3(256), 0, st, d_sw, codes, lp, code_stride, ctx->prm.k1_smooth / 2,
                           partials, power);
        codes = lp;
    }
    hipLaunchKernelGGL(k_fm_stats_final, dim3((n_sw + 63) / 64), dim3(64), 0, st, d_sw, partials, stats, n_sw);
    return materialise ? codes : nullptr;
}

// ---- decimated inverse (fft_radix8.hpp, k_pair_decimate16) ----------------------------------------------------------
// applies to the general form on 4096 x 256 plans when the packed search range M = reach/2 + 2 leaves a transition band:
// R = Nc/16 = 65536, pass band |m| <= M, stop band |m| >= R - M
// Filter design: Kaiser-windowed sinc with T taps a side, T = what 140 dB needs on the transition band, at most kDecTmax
// (fft_radix8.hpp: 95 with 12 steps per phase); the attenuation is then what T buys there, A = 8 + 2.285 dw 2T, and the
// form applies from 120 dB on (cfg2 / cfg4: 126 dB, T = 95; cfg5: 140 dB, T = 87).  Alias leakage measured in float64 on
// noise-level simulator.go peaks: ~4 x 10^(-A/20) of the peak (7e-7 at 126 dB; scripts/dec_filter_sweep.py).
constexpr double kDecAttenuationDb = 140.0, kDecMinAttenuationDb = 120.0;
struct DecDesign { bool ok; int T; double att; };
DecDesign decimation_design(const FftPlan &pl, int reach)
{
    const long long M = reach / 2 + 2, R = pl.Nc / kDecD;
    if (R - 2 * M <= 0) return {false, 0, 0.0};
    const double dw = 2.0 * M_PI * (double)(R - 2 * M) / (double)pl.Nc;
    int T = (int)std::ceil((kDecAttenuationDb - 8.0) / (2.285 * dw) / 2.0);
    double att = kDecAttenuationDb;
    if (T > kDecTmax) {
        T = kDecTmax;
        att = 8.0 + 2.285 * dw * 2.0 * T;
    }
    return {att >= kDecMinAttenuationDb, T, att};
}

// two-sweep plans with a decimated inverse: a 4096-bin tile of their spectrum is (less than) one column, so only the column
// walk (dec_stream.hpp) serves them, and the row pass leaves the unpacked spectra in TZ
// (N2 = 2048 -- windows of 4 to 8 s at 2 Msps, N = 2^24 -- joined in round 5: until then that plan ran the full inverse)
bool cols_only_plan(const FftPlan &pl) { return pl.N1 == 4096 && (pl.N2 == 4096 || pl.N2 == 3072 || pl.N2 == 2560 || pl.N2 == 2048); }

bool decimation_applies(const tdoa_ctx *ctx, const FftPlan &pl, int lag_lo, int lag_hi)
{
    if (!ctx->decimate || ctx->force_generic || pl.N1 != 4096 || (pl.N2 != 256 && pl.N2 != 512 && !cols_only_plan(pl))) return false;
    if (cols_only_plan(pl) && !(ctx->dec_cols && TDOA_HAVE_DEC_COLS)) return false;
    {   // the small plan's K5 kernel evaluates the column outputs that can hold a searched lag as direct sums: at most kPruneMax
        // of them (run_fm_batch's `pruned`; 4096 packed lags per output: search ranges up to ~32 000 lags).  choose_fft_size
        // relies on this function alone -- a 5 x 2^k plan has no other inverse to fall back to.
        const long long n_real = 2 * pl.Nc;
        const int np = lag_hi >= 0 ? (int)((lag_hi / 2) / pl.N1) + 1 : 0;
        const int nn = lag_lo < 0 ? pl.N2 - (int)(((n_real + lag_lo) / 2) / pl.N1) : 0;
        if (np + nn > kPruneMax || lag_hi >= pl.Nc || lag_lo <= -pl.Nc) return false;
    }
    const int reach = std::max(lag_hi + 1, -(lag_lo - 1));
    if (reach <= 4095) return false;                       // the short-lag forms take those
    return decimation_design(pl, reach).ok;
}

// modified Bessel function I0 (Kaiser window)
double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 200; k++) {
        term *= (x / (2.0 * k)) * (x / (2.0 * k));
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

// layout of the decimated inverse inside the V workspace (float2 elements): G [n_pw][R], V' [n_pw][R], the tiles' edge
// shares E [n_pw][N2][2 kDecEdge], then the stations' spectra in tiles [n_sw][Nc]
size_t dec_edge_offset(const FftPlan &pl, int n_pw) { return 2 * (size_t)(pl.Nc / kDecD) * (size_t)n_pw; }
// (E: [n_pw][N2][12] for the tile kernel, X: [n_pw][12][4096] for the column walk -- room for the larger)
size_t dec_spectra_offset(const FftPlan &pl, int n_pw)
{
    return dec_edge_offset(pl, n_pw) + (size_t)n_pw * (size_t)std::max(pl.N2, 4096) * (2 * kDecEdge);
}
// stations per window of a UNIFORM batch -- every window carries all the P = S (S - 1) / 2 pairs of its S <= 16 stations,
// station-windows laid out window by window (process_impl's window-major order), so that a workgroup can name a window's
// stations sw_base .. sw_base + S - 1 -- else 0
int uniform_batch_stations(int n_sw, int n_pw, int pairs_per_window)
{
    if (pairs_per_window <= 0 || n_pw % pairs_per_window != 0) return 0;
    const int n_win = n_pw / pairs_per_window;
    if (n_sw % n_win != 0) return 0;
    const int st = n_sw / n_win;
    return st >= 2 && st <= kStgMaxStations && st * (st - 1) / 2 == pairs_per_window ? st : 0;
}
// Which form the decimated pair step takes.  The column walk (dec_stream.hpp, dec_staged.hpp) is the only one on the 4096 x 2048
// and larger plans.  On the others: with the stations' rows staged in LDS it is ahead from three stations on (cfg2, 3 pairs:
// 0.57 ms against 0.60 for the tile form; cfg4: 3.0 against 5.05; cfg5: 72 against 118); one pair-window per wave from memory
// (batches the staged walk does not take) where windows carry more pairs than stations; the tile form otherwise -- it asks
// for a tile's 32 KB at once and a lone pair waits for nothing else.
bool dec_walks_columns(const tdoa_ctx *ctx, const FftPlan &pl, int n_sw, int n_pw, int pairs_per_window)
{
    if (!ctx->dec_cols || !TDOA_HAVE_DEC_COLS) return false;      // (TDOA_DEC_STEPS other than 8 / 12: the walk is not built)
    if (cols_only_plan(pl) || ctx->dec_cols_always) return true;
    if (pairs_per_window <= 0 || n_pw % pairs_per_window != 0 || n_sw <= 0) return false;
    if (ctx->dec_staged && uniform_batch_stations(n_sw, n_pw, pairs_per_window) >= 3) return true;
    return pairs_per_window > n_sw / (n_pw / pairs_per_window);
}

// stations per window when the decimated pair step runs as the LDS-staged column walk (dec_staged.hpp), else 0
int staged_walk_stations(const tdoa_ctx *ctx, const FftPlan &pl, int n_sw, int n_pw, int pairs_per_window)
{
    if (!TDOA_HAVE_DEC_COLS || !ctx->dec_staged || !dec_walks_columns(ctx, pl, n_sw, n_pw, pairs_per_window)) return 0;
    return uniform_batch_stations(n_sw, n_pw, pairs_per_window);
}
// ... and whether its spectra are laid out in blocks of 64 columns (out of place, where the tile form keeps its tiles: the
// plans that have that room; the 4096 x 2048 and larger plans keep their rows in place).  A loader's piece of a row is then
// followed in memory by its piece of the next row -- 4 KB runs per station and phase instead of 512-byte pieces 32 KB apart.
bool staged_walk_blocks(const tdoa_ctx *ctx, const FftPlan &pl, int n_sw, int n_pw, int pairs_per_window)
{
    return ctx->stg_blocks && !cols_only_plan(pl) && staged_walk_stations(ctx, pl, n_sw, n_pw, pairs_per_window) > 0;
}

// taps h[t] = sinc(t/16) * kaiser(t), |t| <= T, rounded to f32; gain[m] = 1 / w[m], w[m] = sum_t h[t] cos(2 pi t m / Nc) / 16
// evaluated from the ROUNDED taps, so the correction is exact for the filter that runs.  No-op when already built.
int ensure_decimation(tdoa_ctx *ctx, const FftPlan &pl, int lag_lo, int lag_hi)
{
    const int reach = std::max(lag_hi + 1, -(lag_lo - 1));
    if (ctx->dec_nc == pl.Nc && ctx->dec_reach == reach) return TDOA_OK;
    const long long M = reach / 2 + 2;
    const DecDesign dd = decimation_design(pl, reach);
    const int T = dd.T;
    const double beta = 0.1102 * (dd.att - 8.7), i0b = bessel_i0(beta);
    std::vector<float> taps(2 * T + 1);
    for (int t = -T; t <= T; t++) {
        const double x = (double)t / kDecD, r = (double)t / T;
        const double sinc = t == 0 ? 1.0 : std::sin(M_PI * x) / (M_PI * x);
        taps[t + T] = (float)(sinc * bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0b);
    }
    std::vector<float> gain(M + 4);
    for (long long m = 0; m < M + 4; m++) {
        double w = 0.0;
        for (int t = -T; t <= T; t++) w += (double)taps[t + T] * std::cos(2.0 * M_PI * (double)t * (double)m / (double)pl.Nc);
        gain[m] = (float)((double)kDecD / w);
    }
    // the kernel's layout: phase p x step s, the tap t = 16 (s - kDecCentre) + p (zero where |t| > T)
    // (then W_N^p, p = 0..15, N = 2 Nc, as float2: the row rotations of k_pair_decimate_cols)
    // (then, at 288: phase 0 with its steps reversed -- the upward walks' row of phase 0)
    std::vector<float> tab(256 + 32 + 16, 0.0f);
    for (int t = -T; t <= T; t++) {
        const int p = ((t % 16) + 16) % 16, sidx = (t - p) / 16 + kDecCentre;
        tab[16 * p + sidx] = taps[t + T];
    }
    for (int s = 0; s < kDecSteps; s++) tab[288 + s] = tab[kDecSteps - 1 - s];
    for (int p = 0; p < 16; p++) {
        const double ang = -M_PI * (double)p / (double)pl.Nc;
        tab[256 + 2 * p] = (float)std::cos(ang);
        tab[256 + 2 * p + 1] = (float)std::sin(ang);
    }
    int rc;
    if ((rc = ensure(ctx, ctx->dec_taps, sizeof(float) * tab.size()))) return rc;
    if ((rc = ensure(ctx, ctx->dec_gain, sizeof(float) * gain.size()))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->dec_taps.p, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dec_gain.p, gain.data(), sizeof(float) * gain.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // host vectors go out of scope
    ctx->dec_nc = pl.Nc;
    ctx->dec_reach = reach;
    ctx->dec_T = T;
    return TDOA_OK;
}

// make every workspace buffer of run_fm_batch large enough (no allocation may happen while a
// stream capture is open)
// segment form (search ranges up to 1024 lags, hot row size): 256-lag quarter count of its frames, 0 = does not apply
int segment_pq(const tdoa_ctx *ctx, const FftPlan &pl, int lag_lo, int lag_hi, int n_pw)
{
    // lags lag_lo - 1 .. lag_hi + 1 (refinement neighbours included) must lie in [-P, P], P = 256 seg_pq
    const int reach = std::max(lag_hi + 1, -(lag_lo - 1));
    const int pq = reach <= 256 ? 1 : reach <= 512 ? 2 : reach <= 1024 ? 4 : 0;
    const bool row16 = pl.N1 == 4096 && !ctx->force_generic;
    return pq && row16 && ctx->short_lag && ctx->segment_form && n_pw > 0 && pl.N2 >= 8 ? pq : 0;
}

// K1 evaluated inside the forward column kernels (no code array): the plans with a k_fwd_col*_k1 kernel, unless a
// consumer needs the codes in memory (segment form, k1_smooth, k1_gate) or a window may be shorter than two samples
bool fused_k1_applies(const tdoa_ctx *ctx, const FftPlan &pl, int lag_lo, int lag_hi, int n_pw, bool allow)
{
    if (!allow || !ctx->fused_k1 || ctx->force_generic || ctx->prm.k1_smooth > 1 || ctx->prm.k1_gate) return false;
    if (pl.N1 != 4096 || !(pl.N2 == 256 || pl.N2 == 512 || pl.N2 == 2048 || pl.N2 == 2560 || pl.N2 == 3072 || pl.N2 == 4096)) return false;
    return segment_pq(ctx, pl, lag_lo, lag_hi, n_pw) == 0;
}

// single-look K1 (k1_single_look.hpp): largest |lag| a K5 kernel or the refinement looks at, entries per edge array,
// tile records per station-window of the fused column kernels
int once_k_max(int lag_lo, int lag_hi) { return std::max(lag_hi + 1, -(lag_lo - 1)); }
int once_k1(int lag_lo, int lag_hi) { return (once_k_max(lag_lo, lag_hi) + 1 + 3) & ~3; }
int once_tiles_per_sw(const FftPlan &pl)      // records per station-window: one per wave and tile (k1_single_look.hpp)
{
    return kOnceWavesPerTile * (pl.N2 == 512 ? pl.N1 / 32 : (pl.N1 / 64) * std::max(1, pl.N2 / 256));
}

// The staged column walk (dec_staged.hpp) gives a workgroup up to `cap` of a window's pairs and stages the rows of every
// station those pairs touch.  Pairs are numbered as process_impl lays them out: (0,1), (0,2), ..., (S-2,S-1).
//  * Up to eight stations: consecutive runs of equal length (28 pairs: 14 + 14) -- every group touches every station anyway.
//  * More: what the loader can bring in is the bound there (the CU's memory pipeline takes ~1 KB of LDS-DMA per 50 - 65 cycles),
//    and sixteen stations per group leave room for four rows per phase only.  Groups are grown greedily around the first pair
//    not yet placed -- the station that adds the most unplaced pairs joins until `cap` pairs or eight stations are reached
//    (the first groups are the 15 pairs of six stations) --, then small leftovers are merged: 16 stations become 9 groups that
//    stage 62 station-rows per row of the window instead of 8 x 16 = 128, each within eight stations: eight rows per phase.
std::vector<StgGroup> build_stg_groups(int S, int cap, bool fill = false)
{
    const int P = S * (S - 1) / 2, M = 8;
    std::vector<std::pair<int, int>> pairs;
    for (int i = 0; i < S; i++)
        for (int j = i + 1; j < S; j++) pairs.emplace_back(i, j);
    auto pidx = [&](int a, int b) { if (a > b) std::swap(a, b); return a * S - a * (a + 1) / 2 + (b - a - 1); };
    std::vector<StgGroup> out;
    if (S <= M) {
        // (fill: full groups first -- sixteen walks are four per SIMD, the remainder of 28 pairs three -- instead of equal runs)
        const int groups = (P + cap - 1) / cap, n = fill ? cap : (P + groups - 1) / groups;
        for (int g = 0; g < groups; g++) {
            StgGroup sg{};
            for (int p = g * n; p < std::min(P, (g + 1) * n); p++) {
                sg.pair[sg.n++] = (uint8_t)p;
                sg.mask |= (1u << pairs[p].first) | (1u << pairs[p].second);
            }
            out.push_back(sg);
        }
        return out;
    }
    std::vector<char> open(P, 1);
    int left = P;
    while (left) {
        int seed = 0;
        while (!open[seed]) seed++;
        std::vector<int> T = {pairs[seed].first, pairs[seed].second};
        auto inside = [&] {
            int c = 0;
            for (size_t x = 0; x < T.size(); x++)
                for (size_t y = x + 1; y < T.size(); y++) c += open[pidx(T[x], T[y])];
            return c;
        };
        while ((int)T.size() < M && inside() < cap) {
            int best = -1, gain = 0;
            for (int v = 0; v < S; v++) {
                if (std::find(T.begin(), T.end(), v) != T.end()) continue;
                int g = 0;
                for (int t : T) g += open[pidx(t, v)];
                if (g > gain) { gain = g; best = v; }
            }
            if (best < 0) break;
            T.push_back(best);
        }
        std::sort(T.begin(), T.end());
        StgGroup sg{};
        for (size_t x = 0; x < T.size(); x++)
            for (size_t y = x + 1; y < T.size(); y++) {
                const int p = pidx(T[x], T[y]);
                if (!open[p] || sg.n >= cap) continue;
                open[p] = 0;
                left--;
                sg.pair[sg.n++] = (uint8_t)p;
                sg.mask |= (1u << T[x]) | (1u << T[y]);
            }
        out.push_back(sg);
    }
    for (bool merged = true; merged;) {          // leftovers: two groups that fit one workgroup and eight stations together
        merged = false;
        for (size_t a = 0; a < out.size() && !merged; a++)
            for (size_t b = a + 1; b < out.size() && !merged; b++)
                if (out[a].n + out[b].n <= cap && __builtin_popcount(out[a].mask | out[b].mask) <= M) {
                    for (int q = 0; q < out[b].n; q++) out[a].pair[out[a].n++] = out[b].pair[q];
                    out[a].mask |= out[b].mask;
                    out.erase(out.begin() + (long)b);
                    merged = true;
                }
    }
    return out;
}

int ensure_stg_groups(tdoa_ctx *ctx)
{
    if (ctx->stg_ready) return TDOA_OK;
    const int n_lw = std::max(1, std::min(ctx->stg_loaders ? ctx->stg_loaders : 1, 4));
    const int cap = ctx->stg_cw > 0 ? std::min(ctx->stg_cw, kStgMaxWaves - n_lw) : kStgMaxWaves - n_lw;
    std::vector<StgGroup> all;
    for (int pass = 0; pass < 2; pass++)
    for (int S = 2; S <= kStgMaxStations; S++) {
        const std::vector<StgGroup> g = pass ? build_stg_groups(S, ctx->stg_cw > 0 ? std::min(ctx->stg_cw + 1, kStgMaxWaves) : kStgMaxWaves, true)
                                             : build_stg_groups(S, cap);
        auto &t = pass ? ctx->stg_tab16[S] : ctx->stg_tab[S];
        t.off = (int)all.size();
        t.count = (int)g.size();
        t.slots = t.max_n = 0;
        for (const StgGroup &x : g) {
            t.slots = std::max(t.slots, __builtin_popcount(x.mask));
            t.max_n = std::max(t.max_n, (int)x.n);
        }
        all.insert(all.end(), g.begin(), g.end());
    }
    int rc;
    if ((rc = ensure(ctx, ctx->stg_groups, sizeof(StgGroup) * all.size()))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stg_groups.p, all.data(), sizeof(StgGroup) * all.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stg_ready = true;
    return TDOA_OK;
}

int reserve_fm_batch(tdoa_ctx *ctx, int n_sw, int maxlen, int n_pw, const FftPlan &pl, int lag_lo, int lag_hi,
                     bool allow_fused_k1)
{
    int rc;
    const long long code_stride = ((long long)maxlen + 15) / 8 * 8;
    if ((rc = ensure(ctx, ctx->partials, sizeof(StatsPartial) * (size_t)n_sw))) return rc;
    if ((rc = ensure(ctx, ctx->stats, sizeof(FmStats) * (size_t)n_sw))) return rc;
    if (!fused_k1_applies(ctx, pl, lag_lo, lag_hi, n_pw, allow_fused_k1)) {
        if ((rc = ensure(ctx, ctx->codes, sizeof(int) * (size_t)code_stride * n_sw))) return rc;
        if (ctx->prm.k1_smooth > 1 && (rc = ensure(ctx, ctx->codes_lp, sizeof(int) * (size_t)code_stride * n_sw))) return rc;
    }
    if (ctx->prm.k1_gate && (rc = ensure(ctx, ctx->k1_power, sizeof(unsigned long long) * (size_t)n_sw))) return rc;
    if (ctx->k1_once && n_pw && fused_k1_applies(ctx, pl, lag_lo, lag_hi, n_pw, allow_fused_k1)) {      // single-look K1
        if ((rc = ensure(ctx, ctx->once_edges, sizeof(float) * 2 * (size_t)once_k1(lag_lo, lag_hi) * n_sw))) return rc;
        if ((rc = ensure(ctx, ctx->once_tiles, sizeof(OnceTile) * (size_t)once_tiles_per_sw(pl) * n_sw))) return rc;
        if ((rc = ensure(ctx, ctx->once_fin, sizeof(OnceFin) * (size_t)n_sw))) return rc;
    }
    if ((rc = ensure(ctx, ctx->tz, sizeof(float2) * (size_t)pl.Zs * n_sw))) return rc;
    if (n_pw && decimation_applies(ctx, pl, lag_lo, lag_hi) && (rc = ensure_decimation(ctx, pl, lag_lo, lag_hi))) return rc;
    if (n_pw && decimation_applies(ctx, pl, lag_lo, lag_hi) && (rc = ensure_stg_groups(ctx))) return rc;
    size_t v_elems = (size_t)pl.Nc * n_pw;
    if (n_pw && decimation_applies(ctx, pl, lag_lo, lag_hi))      // G + V' of the pairs, then the tiled spectra of the stations
        v_elems = std::max(v_elems, dec_spectra_offset(pl, n_pw) + (cols_only_plan(pl) ? 0 : (size_t)pl.Nc * n_sw));
    if (n_pw && (rc = ensure(ctx, ctx->v, sizeof(float2) * v_elems))) return rc;
    return TDOA_OK;
}

// second sweep of the two-sweep column pass: the G = N2 / 256 rows kb + 256 a of every column, in place
void launch_col_finish(hipStream_t st, float2 *tz, const FftPlan &pl, int n_sw)
{
    const dim3 grid(pl.N1 / 512, 256, n_sw), blk(256);
    if (pl.N2 == 4096) hipLaunchKernelGGL(k_fwd_col_finish<16>, grid, blk, 0, st, tz, pl);
    else if (pl.N2 == 2560) hipLaunchKernelGGL(k_fwd_col_finish<10>, grid, blk, 0, st, tz, pl);
    else if (pl.N2 == 3072) hipLaunchKernelGGL(k_fwd_col_finish<12>, grid, blk, 0, st, tz, pl);
    else hipLaunchKernelGGL(k_fwd_col_finish<8>, grid, blk, 0, st, tz, pl);
}

// ---- mode B core: run stats + forward + inverse + peak over prepared descriptors
// sw/pw descriptors are already in device memory; maxlen = longest window.
int run_fm_batch(tdoa_ctx *ctx, const SWDesc *d_sw, int n_sw, int maxlen, const PWDesc *d_pw, int n_pw,
                 unsigned long long *d_keys, const FftPlan &pl, int lag_lo, int lag_hi, float *lag_dump,
                 float dump_scale, double sum_len, float *fine_raw = nullptr, int pairs_per_window = 0,
                 const QuadDesc *d_quads = nullptr, int n_quads = 0, bool allow_fused_k1 = true,
                 const SWDesc *d_sw_stats = nullptr,      // d_sw_stats: the windows K1 and its statistics run over when the
                                                          // transforms see truncated ones (TDOA_LAGS_GO); default: d_sw
                 bool equal_len = false)                  // every station-window of the batch has `maxlen` samples
{
    int rc;
    const int pieces = std::max(1, (maxlen + kDemodPiece - 1) / kDemodPiece);
    const long long code_stride = ((long long)maxlen + 15) / 8 * 8;      // rows stay 16-byte aligned
    if ((rc = ensure(ctx, ctx->partials, sizeof(StatsPartial) * (size_t)n_sw))) return rc;
    if ((rc = ensure(ctx, ctx->stats, sizeof(FmStats) * (size_t)n_sw))) return rc;
    if ((rc = ensure(ctx, ctx->tz, sizeof(float2) * (size_t)pl.Zs * n_sw))) return rc;
    if ((rc = reserve_fm_batch(ctx, n_sw, maxlen, n_pw, pl, lag_lo, lag_hi, allow_fused_k1))) return rc;
    auto *stats = static_cast<FmStats *>(ctx->stats.p);
    const int *codes = nullptr;
    auto *tz = static_cast<float2 *>(ctx->tz.p);
    auto *v = static_cast<float2 *>(ctx->v.p);
    hipStream_t st = ctx->stream;
    const double nc8 = 8.0 * (double)pl.Nc;

    // hot-size kernels (fft_radix16.hpp) when the plan allows, else the any-size kernels of
    // fft_stockham.hpp
    const bool row16 = pl.N1 == 4096 && !ctx->force_generic;
    const bool col16 = row16 && pl.N2 == 256;
    const bool col2pass = row16 && (pl.N2 == 4096 || pl.N2 == 2048 || pl.N2 == 2560 || pl.N2 == 3072);   // 256-point sub-transforms + G-point finish (two sweeps)
    const int col16x = row16 && pl.N2 >= 16 && pl.N2 <= 128 ? pl.N2 / 16 : 0;   // short columns: k_fwd_col16x_c16<F>
    const int colx = row16 && (pl.N2 == 512 || pl.N2 == 1024) ? pl.N2 / 256 : 0;   // last radix of k_fwd_colx_c16
    int np = 0, nn = 0;
    {
        const long long n_real = 2 * pl.Nc;
        np = lag_hi >= 0 ? (int)((lag_hi / 2) / pl.N1) + 1 : 0;
        nn = lag_lo < 0 ? pl.N2 - (int)(((n_real + lag_lo) / 2) / pl.N1) : 0;
    }
    // short-lag form: the inverse row kernel emits its shares of the few column sums that can hold a lag and V is
    // never written (needs lag_lo - 1 .. lag_hi + 1 inside [-512 fk, 512 fk - 1] for the refinement neighbours)
    int fk = 0;
    if (row16 && ctx->short_lag) {
        const int reach = std::max(lag_hi + 1, -(lag_lo - 1));
        fk = reach <= 511 ? 1 : reach <= 1023 ? 2 : reach <= 2047 ? 4 : reach <= 4095 ? 8 : 0;
    }
    const bool pruned = !ctx->force_generic && pl.N1 >= 128 && pl.N2 <= 4096 && np + nn <= kPruneMax && np + nn <= pl.N2 &&
                        lag_hi < pl.Nc && lag_lo > -pl.Nc;
    // decimated inverse (general form on 4096 x 256 and 4096 x 512 plans)
    const bool decim = pruned && fk == 0 && decimation_applies(ctx, pl, lag_lo, lag_hi) &&
                       ctx->dec_nc == pl.Nc && ctx->dec_reach == std::max(lag_hi + 1, -(lag_lo - 1));
    // segment form (search ranges up to 1024 lags): overlap-save over 4096-point frames entirely in LDS; neither the
    // column pass nor TZ nor V rows are touched.  Its chunk sums and lag array live where the short-lag form keeps its
    // shares (inside this pair-window's V row), which bounds the chunk count by N2 / 2.
    int seg_chunks = 0;
    const int seg_pq = segment_pq(ctx, pl, lag_lo, lag_hi, n_pw);
    // quads (two station transforms per segment serve up to four pair-windows) when that is fewer transforms than one
    // per pair-window
    const bool seg_quads = ctx->segment_quads && d_quads && n_quads > 0 && 2 * n_quads < n_pw;
    if (seg_pq) {
        const int hop = 4096 - 512 * seg_pq;
        const int frames = (maxlen + hop - 1) / hop;
        const int trips = seg_quads ? frames : (frames + 1) / 2;        // the pair kernel takes two frames per trip
        const int units = seg_quads ? n_quads : n_pw;
        // chunks per unit: the grid runs in rounds of 2 workgroups per CU (64 KB LDS, 128 VGPRs x 512 threads); cost
        // model = rounds x (trips of the longest chunk + 1 for the prologue and the final inverse transform).  The model
        // is flat over a wide range (measured: 10 ... 36 chunks within 2 % on cfg2); among the near-ties take the most
        // chunks -- more, shorter workgroups balance better than one round of long ones (5 chunks: 4 % slower).
        const int c_max = std::max(1, std::min({trips / 8, pl.N2 / 2 - 1, (8192 + units - 1) / units}));
        const long long slots = 2ll * ctx->n_cu;
        auto cost = [&](int c) { return (double)(((long long)c * units + slots - 1) / slots) * ((trips + c - 1) / c + 1); };
        double best = cost(1);
        for (int c = 2; c <= c_max; c++) best = std::min(best, cost(c));
        for (int c = 1; c <= c_max; c++)
            if (cost(c) <= 1.03 * best) seg_chunks = c;
        if (ctx->seg_chunks_override > 0) seg_chunks = std::max(1, std::min({ctx->seg_chunks_override, trips, pl.N2 / 2 - 1}));
    }
    // its code rows at 3 bytes per code (round 4; the gate and the smoother work on int32 rows)
    const bool seg_pack3 = seg_chunks > 0 && ctx->seg_pack3 && !ctx->prm.k1_gate && ctx->prm.k1_smooth <= 1;
    const bool fused_k1 = fused_k1_applies(ctx, pl, lag_lo, lag_hi, n_pw, allow_fused_k1);
    // single-look K1 (k1_single_look.hpp): no statistics pre-pass.  Needs windows of one length, the peak picked by
    // k_small_col_peak or a pruned column kernel, and head / tail runs of K samples that do not meet.
    const int once_kmax = once_k_max(lag_lo, lag_hi), once_n1 = once_k1(lag_lo, lag_hi);
    const bool once = ctx->k1_once && fused_k1 && equal_len && !d_sw_stats && n_pw > 0 && !seg_chunks && fk == 0 && pruned &&
                      once_kmax < maxlen / 2 && once_kmax + 1 <= kOncePiece * kOnceMaxPieces;
    ctx->once_active = once;
    OnceCorr oc{};
    auto *once_tiles = static_cast<OnceTile *>(ctx->once_tiles.p);
    if (once) {
        oc.edges = static_cast<const float *>(ctx->once_edges.p);
        oc.fin = static_cast<const OnceFin *>(ctx->once_fin.p);
        oc.slot_gain = static_cast<double *>(ctx->slot_gain.p);
        oc.k1 = once_n1;
        oc.raw_per_unit = (float)(8.0 * (double)pl.Nc);          // raw = 4 N sum w w, N = 2 Nc
        oc.k_max = once_kmax;
        // 4096 samples per window -> the estimates (m0, s0) the column kernels normalise with; then the running sums of the
        // first and the last K samples of every window (streamed like k_fm_demod, 2 x (K + 1) samples per window)
        const int pieces = (once_kmax + 1 + kOncePiece - 1) / kOncePiece;
        ProfScope ps(ctx, TDOA_K_STATS, (2.0 * (2.0 * (once_kmax + 1) + (double)kOnceRuns * (kOnceRun + 1)) + 8.0 * (once_kmax + 1)) * n_sw);
        hipLaunchKernelGGL(k_once_estimate, dim3(n_sw), dim3(kOnceRuns), 0, st, d_sw, static_cast<const int *>(ctx->k1_direct.p), stats);
        const int blocks = std::max(1, std::min(2 * n_sw, ctx->n_cu));
        hipLaunchKernelGGL(k_once_edges, dim3(blocks), dim3(kDemodThreads), kK1DirectBytes, st, d_sw, n_sw,
                           static_cast<const int *>(ctx->k1_direct.p), stats, static_cast<float *>(ctx->once_edges.p), once_kmax,
                           once_n1, pieces);
    } else {
        // K1: capture bytes -> exact window statistics (fused: nothing else; the column pass evaluates the discriminator
        // itself) and, materialised, the 24-bit phase codes as int32
        ProfScope ps(ctx, TDOA_K_STATS, (fused_k1 ? 2.0 : seg_pack3 ? 5.0 : 6.0) * sum_len);
        codes = launch_k1(ctx, st, d_sw_stats ? d_sw_stats : d_sw, n_sw, maxlen, pieces, code_stride, !fused_k1, seg_pack3);
    }
    const size_t lds_col = sizeof(float2) * 2 * (size_t)pl.N2 * pl.C;
    const size_t lds_row = sizeof(float2) * 2 * (size_t)pl.N1;
    const size_t lds_row2 = sizeof(float2) * 4 * (size_t)pl.N1;
    const size_t lds_col16 = sizeof(float2) * 256 * 32;
    const size_t lds_pair16 = sizeof(float2) * 2 * kRowLds;
    if (!seg_chunks) {
        // two-sweep column pass (N2 = 2048, 4096): 8 Nc written, read and written again -- SURVEY's third pass
        ProfScope ps(ctx, TDOA_K_FWD_COL, (fused_k1 ? 2.0 : 4.0) * sum_len + (col2pass ? 3.0 : 1.0) * nc8 * n_sw);
        const auto *qtable = static_cast<const int *>(ctx->k1_quad.p);
#define TDOA_COL256(SUB, ONCE_)                                                                                       \
    hipLaunchKernelGGL((k_fwd_col256_k1<SUB, ONCE_>), dim3(ctx->n_cu), dim3(1024), kColK1Lds, st, d_sw, qtable, stats, tz, pl,    \
                       n_sw, ONCE_ ? once_tiles : static_cast<OnceTile *>(nullptr))
        if (fused_k1 && col16) {
            if (once) TDOA_COL256(false, true);
            else TDOA_COL256(false, false);
        }
        else if (fused_k1 && col2pass) {
            if (once) TDOA_COL256(true, true);
            else TDOA_COL256(true, false);
#undef TDOA_COL256
            launch_col_finish(st, tz, pl, n_sw);
        }
        else if (fused_k1 && colx == 2) {
            if (once)
                hipLaunchKernelGGL(k_fwd_col512_k1<true>, dim3(ctx->n_cu), dim3(1024), kCol512Lds, st, d_sw, qtable, stats, tz, pl,
                                   n_sw, once_tiles);
            else
                hipLaunchKernelGGL(k_fwd_col512_k1<false>, dim3(ctx->n_cu), dim3(1024), kCol512Lds, st, d_sw, qtable, stats, tz, pl,
                                   n_sw, static_cast<OnceTile *>(nullptr));
        }
        else if (col16)
            hipLaunchKernelGGL(k_fwd_col256_c16<false>, dim3(pl.N1 / 32, n_sw), dim3(512), lds_col16, st, d_sw, codes,
                               code_stride, stats, tz, pl);
        else if (col2pass) {
            hipLaunchKernelGGL(k_fwd_col256_c16<true>, dim3(pl.N1 / 32, n_sw, pl.N2 / 256), dim3(512), lds_col16, st,
                               d_sw, codes, code_stride, stats, tz, pl);
            launch_col_finish(st, tz, pl, n_sw);
        }
        else if (col16x == 1)
            hipLaunchKernelGGL(k_fwd_col16x_c16<1>, dim3(pl.N1 / 256, n_sw), dim3(256), 0, st, d_sw, codes, code_stride,
                               stats, tz, pl);
        else if (col16x == 2)
            hipLaunchKernelGGL(k_fwd_col16x_c16<2>, dim3(pl.N1 / 128, n_sw), dim3(256), 0, st, d_sw, codes, code_stride,
                               stats, tz, pl);
        else if (col16x == 4)
            hipLaunchKernelGGL(k_fwd_col16x_c16<4>, dim3(pl.N1 / 64, n_sw), dim3(256), 0, st, d_sw, codes, code_stride,
                               stats, tz, pl);
        else if (col16x == 8)
            hipLaunchKernelGGL(k_fwd_col16x_c16<8>, dim3(pl.N1 / 32, n_sw), dim3(256), 0, st, d_sw, codes, code_stride,
                               stats, tz, pl);
        else if (colx == 2)
            hipLaunchKernelGGL(k_fwd_colx_c16<2>, dim3(pl.N1 / 16, n_sw), dim3(512), lds_col16, st, d_sw, codes,
                               code_stride, stats, tz, pl);
        else if (colx == 4)
            hipLaunchKernelGGL(k_fwd_colx_c16<4>, dim3(pl.N1 / 8, n_sw), dim3(512), lds_col16, st, d_sw, codes,
                               code_stride, stats, tz, pl);
        else
            hipLaunchKernelGGL(k_fwd_col_c16, dim3(pl.N1 / pl.C, n_sw), dim3(256), lds_col, st, d_sw, codes,
                               code_stride, stats, tz, pl);
    }
    if (once) {
        // the tiles' exact sums -> the window statistics (bit-identical to the pre-pass's), eps and g of every station-window
        ProfScope ps(ctx, TDOA_K_STATS, sizeof(OnceTile) * (double)once_tiles_per_sw(pl) * n_sw);
        hipLaunchKernelGGL(k_once_final, dim3(n_sw), dim3(256), 0, st, d_sw, once_tiles, once_tiles_per_sw(pl), stats,
                           static_cast<OnceFin *>(ctx->once_fin.p), n_sw);
    }
    // XCD-aware 1-D grid of the pair kernel when every window of the group carries the same `pairs_per_window` > S pairs
    // (window-major sharding with more pairs than stations): see k_inv_row_pair4096
    int xcd_pairs = 0;
    unsigned int xcd_grid = 0;
    // ... or when a window's spectra are too large to wait in the Infinity Cache for their second reader (cfg3: 3 x 134 MB per
    // window, and the plain grid runs ALL rows of one pair-window before the next: 62 ms against 73-75 for its pair-row pass;
    // one workgroup running a group's pair-windows one after the other measured 67)
    if (row16 && ctx->xcd_rows && pairs_per_window > 0 && n_pw % pairs_per_window == 0 && n_sw > 0 && pl.N2 > 2) {
        const int stations = n_sw / (n_pw / pairs_per_window);
        if (pairs_per_window > stations ||
            (pairs_per_window > 1 && (size_t)stations * (size_t)pl.Nc * sizeof(float2) > ((size_t)64 << 20))) {
            const long long groups = (long long)(n_pw / pairs_per_window) * (pl.N2 / 2 - 1);
            const long long blocks = (groups + 7) / 8 * 8 * pairs_per_window;
            if (blocks < (1ll << 31)) { xcd_pairs = pairs_per_window; xcd_grid = (unsigned int)blocks; }
        }
    }
    if (!seg_chunks) {
        ProfScope ps(ctx, TDOA_K_FWD_ROW, 2.0 * nc8 * n_sw);
        if (row16 && decim && staged_walk_blocks(ctx, pl, n_sw, n_pw, pairs_per_window))     // unpacked spectra in blocks of 64 columns (k_pair_decimate_staged)
            hipLaunchKernelGGL(k_fwd_row4096_unpack<false>, dim3(pl.N2 / 2, n_sw), dim3(512), sizeof(float2) * 2 * kRowLds, st, tz, pl,
                               v + dec_spectra_offset(pl, n_pw), fused_k1 && (col16 || colx == 2), kStgBlockCols);
        else if (row16 && decim && dec_walks_columns(ctx, pl, n_sw, n_pw, pairs_per_window))     // unpacked spectra back into their rows (k_pair_decimate_cols walks the columns)
            hipLaunchKernelGGL(k_fwd_row4096_unpack<true>, dim3(pl.N2 / 2, n_sw), dim3(512), sizeof(float2) * 2 * kRowLds, st, tz, pl,
                               tz, fused_k1 && (col16 || colx == 2), 0);
        else if (row16 && decim)     // unpacked spectra in COLS-column tiles behind G and V' in the V workspace (k_pair_decimate16 streams them)
            hipLaunchKernelGGL(k_fwd_row4096_unpack<false>, dim3(pl.N2 / 2, n_sw), dim3(512), sizeof(float2) * 2 * kRowLds, st, tz, pl,
                               v + dec_spectra_offset(pl, n_pw), fused_k1 && (col16 || colx == 2), 0);
        else if (row16)
            hipLaunchKernelGGL(k_fwd_row4096, dim3(pl.N2, n_sw), dim3(256), 0, st, tz, pl, fused_k1 && (col16 || colx == 2));
        else
            hipLaunchKernelGGL(k_fwd_row, dim3(pl.N2, n_sw), dim3(256), lds_row, st, tz, pl);
    }
    if (n_pw && seg_chunks) {
        const int hop = 4096 - 512 * seg_pq;
        const double frames = (double)((maxlen + hop - 1) / hop);
        const float mul = (float)(4.0 * 2.0 * (double)pl.Nc / 4096.0);          // 4 N / M
        const size_t lds_seg = sizeof(float2) * 2 * kRow8Lds;
        const double code_bytes = seg_pack3 ? 3.0 : 4.0;
#define TDOA_SEGMENTS_AS(PQ, PACK)                                                                                   \
    do {                                                                                                             \
        if (seg_quads) {                                                                                             \
            ProfScope ps(ctx, TDOA_K_INV_ROW, 4.0 * code_bytes * 4096.0 * frames * n_quads);  /* four frames of codes */ \
            hipLaunchKernelGGL((k_xcorr_segments_quad<PQ, PACK>), dim3(seg_chunks, n_quads), dim3(512), lds_seg, st, d_sw, \
                               d_quads, codes, code_stride, stats, v, pl, seg_chunks);                               \
        } else {                                                                                                     \
            ProfScope ps(ctx, TDOA_K_INV_ROW, 2.0 * code_bytes * 4096.0 * frames * n_pw);   /* two frames of codes */  \
            hipLaunchKernelGGL((k_xcorr_segments<PQ, PACK>), dim3(seg_chunks, n_pw), dim3(512), lds_seg, st, d_sw, d_pw, codes, \
                               code_stride, stats, v, pl, seg_chunks);                                               \
        }                                                                                                            \
    } while (0)
#define TDOA_SEGMENTS(PQ)                                                                                            \
    do {                                                                                                             \
        if (seg_pack3) TDOA_SEGMENTS_AS(PQ, true);                                                                   \
        else TDOA_SEGMENTS_AS(PQ, false);                                                                            \
        {                                                                                                            \
            ProfScope ps(ctx, TDOA_K_INV_COL, 4.0 * 512.0 * PQ * (seg_chunks + 1) * n_pw);                            \
            hipLaunchKernelGGL(k_segments_reduce<PQ>, dim3(2 * PQ + 1, n_pw), dim3(256), 0, st, v, d_keys, d_pw, pl,  \
                               seg_chunks, mul, lag_lo, lag_hi, lag_dump, dump_scale);                               \
        }                                                                                                            \
        if (fine_raw) ctx->prof_last = -1;          /* unscoped launch: the next scope records its own start */     \
        if (fine_raw)                                                                                                \
            hipLaunchKernelGGL(k_refine_segments<PQ>, dim3((n_pw + 63) / 64), dim3(64), 0, st, v, d_keys, d_pw, pl,   \
                               n_pw, fine_raw);                                                                      \
    } while (0)
        if (seg_pq == 1) TDOA_SEGMENTS(1);
        else if (seg_pq == 2) TDOA_SEGMENTS(2);
        else TDOA_SEGMENTS(4);
#undef TDOA_SEGMENTS
#undef TDOA_SEGMENTS_AS
    } else if (n_pw && decim) {
        // decimated inverse: K3 + FIR decimation of the pair's spectrum (one read of the two station spectra), then the
        // R = Nc/16-point inverse on the small plan (rows, pruned column pass with the window divided out, K5)
        FftPlan ps2;
        if ((rc = make_plan(2 * (pl.Nc / kDecD), true, &ps2))) return fail(ctx, rc, "decimated plan");
        const size_t rc_pts = (size_t)(pl.Nc / kDecD);
        float2 *g = v, *vs = v + rc_pts * (size_t)n_pw;                 // compact: [n_pw][R] each, inside the V workspace
        int np2 = 0, nn2 = 0;
        {
            const long long n_real2 = 2 * ps2.Nc;
            np2 = lag_hi >= 0 ? (int)((lag_hi / 2) / ps2.N1) + 1 : 0;
            nn2 = lag_lo < 0 ? ps2.N2 - (int)(((n_real2 + lag_lo) / 2) / ps2.N1) : 0;
        }
        {
            ProfScope ps(ctx, TDOA_K_INV_ROW, 2.0 * nc8 * n_pw + 8.0 * (double)rc_pts * n_pw);      // two spectra read, G written
            float2 *edges = v + dec_edge_offset(pl, n_pw), *spectra = v + dec_spectra_offset(pl, n_pw);
            // pair-windows of a window that share station tiles on one XCD (k_pair_decimate16): when the batch is uniform and
            // a window's spectra are too many to come from on-die memory for their other readers (ctx->xcd_pair_mb: cfg5, 16
            // stations x 16.8 MB: its step 258 -> 237 ms in round 3; cfg4, 8 x 8.4 MB: -1.3 % since round 4; cfg2: plain grid)
            int gp = 0;
            dim3 grid(pl.N2 / 2, n_pw);
            if (ctx->xcd_rows && pairs_per_window > 1 && n_pw % pairs_per_window == 0 && n_sw > 0 &&
                (size_t)(n_sw / (n_pw / pairs_per_window)) * (size_t)pl.Nc * sizeof(float2) > ((size_t)ctx->xcd_pair_mb << 20)) {
                const long long groups = (long long)(n_pw / pairs_per_window) * (pl.N2 / 2);
                const long long blocks = (groups + 7) / 8 * 8 * pairs_per_window;
                if (blocks < (1ll << 31)) { gp = pairs_per_window; grid = dim3((unsigned int)blocks); }
            }
            // W_N^DK, DK = N2 / 8 bins between a thread's consecutive elements of a tile (N = 2 Nc)
            const double ang = -2.0 * M_PI * (double)(pl.N2 / 8) / (2.0 * (double)pl.Nc);
            const float2 rot = make_float2((float)std::cos(ang), (float)std::sin(ang));
            const int stg_s = staged_walk_stations(ctx, pl, n_sw, n_pw, pairs_per_window);
            const bool stg_blk = staged_walk_blocks(ctx, pl, n_sw, n_pw, pairs_per_window);
            if (dec_walks_columns(ctx, pl, n_sw, n_pw, pairs_per_window) && stg_s) {
#if TDOA_HAVE_DEC_COLS
                // One loader wave, the other waves of at most sixteen walk one pair each; the share-out of the window's pairs comes
                // from build_stg_groups.  What the geometry is chosen for is the BARRIER: one per phase stops all sixteen waves, and
                // the pair step of BASELINE config 4 took 4.33 / 3.57 / 3.38 ms with 2 / 4 / 8 rows per phase (the ring's depth
                // made no difference: 3, 6 or 8 phases of two rows all 4.3 ms) -- so the most rows per phase of which TWO phases
                // fit the workgroup's share of the LDS: 8 rows up to eight station slots; small workgroups (three pairs: four
                // waves) leave room for their neighbours on the CU.
                const int P = pairs_per_window, n_win = n_pw / P;
                // the FOLDED form (dec_staged.hpp: no loader wave, up to sixteen walks, the last waves bring one station each): blocked
                // spectra, a two-phase ring -- where sixteen walks per workgroup make FEWER workgroups (16 stations: eight groups
                // instead of nine, cfg5 pair step 73.5 -> 70.2 ms; 8 stations: 16 + 12 walks measured 3.21 ms against 3.12 for
                // 14 + 14 next to a loader wave, and keep the loader)
                const bool folded = stg_blk && ctx->stg_folded && !ctx->stg_loaders && ctx->stg_bufs <= 2 && ctx->stg_tab16[stg_s].slots <= 8 &&
                                    (ctx->stg_tab16[stg_s].count < ctx->stg_tab[stg_s].count || ctx->stg_folded_always);
                const auto &tab = folded ? ctx->stg_tab16[stg_s] : ctx->stg_tab[stg_s];
                const int groups = tab.count, n_cw = tab.max_n, slots = tab.slots;
                const int n_lw = folded ? 0 : std::max(1, std::min(ctx->stg_loaders ? ctx->stg_loaders : 1, std::min(4, slots)));
                // (few-station batches wait for memory rather than for the barrier: eight rows per phase there as well, and on the
                //  blocked plans a third phase in the ring where two workgroups still share a CU's LDS -- cfg2: 0.594 -> 0.571 ms;
                //  a fourth, or a third on the in-place plans, lost: cfg2 0.63, cfg3 17.8 against 16.3)
                const int wgs_by_waves = std::max(1, kStgMaxWaves / (n_cw + n_lw));
                const int budget = wgs_by_waves >= 2 ? 80 * 1024 : kStgLdsBytes;
                int rows = ctx->stg_rows;
                if (!rows) rows = 2 * 8 * slots * 1024 <= kStgLdsBytes ? 8 : 2 * 4 * slots * 1024 <= kStgLdsBytes ? 4 : 2;
                const int per_phase = n_lw ? rows * ((slots + n_lw - 1) / n_lw) : rows;
                int nb = folded ? 2 : ctx->stg_bufs ? ctx->stg_bufs : stg_blk ? std::max(2, std::min(3, budget / (rows * slots * 1024))) : 2;
                if (rows * slots * 1024 * 2 > kStgLdsBytes) return fail(ctx, TDOA_ERR_INVALID, "TDOA_DEC_STAGED_ROWS: two phases do not fit the LDS ring");
                nb = std::min(nb, kStgLdsBytes / (rows * slots * 1024));
                nb = std::max(2, std::min(nb, 2 + kStgMaxInFlight / per_phase));
                const int n_items = n_win * 32;
                const unsigned int blocks = (unsigned int)((n_items + 7) / 8 * 8) * (unsigned int)groups;
                const size_t lds = (size_t)nb * rows * slots * 1024;
                const float *tp = static_cast<const float *>(ctx->dec_taps.p);
                const StgGroup *gt = static_cast<const StgGroup *>(ctx->stg_groups.p) + tab.off;
                const dim3 sblock(64 * (n_cw + n_lw));
#define TDOA_STAGED_R(N2V, RV)                                                                                       \
    hipLaunchKernelGGL((k_pair_decimate_staged<N2V, RV>), dim3(blocks), sblock, lds, st, d_pw, stg_blk ? spectra : tz, g, edges, pl, tp, gt, \
                       n_items, P, slots, n_cw, groups, nb, stg_blk ? (long long)pl.Nc : (long long)pl.Zs, (int)stg_blk)
#define TDOA_STAGED(N2V)                                                                                              \
    do {                                                                                                              \
        if (rows == 8) TDOA_STAGED_R(N2V, 8);                                                                         \
        else if (rows == 4) TDOA_STAGED_R(N2V, 4);                                                                    \
        else TDOA_STAGED_R(N2V, 2);                                                                                   \
    } while (0)
                if (pl.N2 == 256) TDOA_STAGED(256);
                else if (pl.N2 == 512) TDOA_STAGED(512);
                else if (pl.N2 == 2048) TDOA_STAGED(2048);
                else if (pl.N2 == 2560) TDOA_STAGED(2560);
                else if (pl.N2 == 3072) TDOA_STAGED(3072);
                else TDOA_STAGED(4096);
#undef TDOA_STAGED_R
#undef TDOA_STAGED
#endif
            } else if (dec_walks_columns(ctx, pl, n_sw, n_pw, pairs_per_window)) {
#if TDOA_HAVE_DEC_COLS
                const dim3 sgrid(32, (unsigned int)((n_pw + kDecWavesPerWg - 1) / kDecWavesPerWg)), sblock(64 * kDecWavesPerWg);
                const float *tp = static_cast<const float *>(ctx->dec_taps.p);
                if (pl.N2 == 256) hipLaunchKernelGGL(k_pair_decimate_cols<256>, sgrid, sblock, 0, st, d_pw, tz, g, edges, pl, tp, n_pw);
                else if (pl.N2 == 512) hipLaunchKernelGGL(k_pair_decimate_cols<512>, sgrid, sblock, 0, st, d_pw, tz, g, edges, pl, tp, n_pw);
                else if (pl.N2 == 2048) hipLaunchKernelGGL(k_pair_decimate_cols<2048>, sgrid, sblock, 0, st, d_pw, tz, g, edges, pl, tp, n_pw);
                else if (pl.N2 == 2560) hipLaunchKernelGGL(k_pair_decimate_cols<2560>, sgrid, sblock, 0, st, d_pw, tz, g, edges, pl, tp, n_pw);
                else if (pl.N2 == 3072) hipLaunchKernelGGL(k_pair_decimate_cols<3072>, sgrid, sblock, 0, st, d_pw, tz, g, edges, pl, tp, n_pw);
                else hipLaunchKernelGGL(k_pair_decimate_cols<4096>, sgrid, sblock, 0, st, d_pw, tz, g, edges, pl, tp, n_pw);
#endif
            } else if (pl.N2 == 256)
                hipLaunchKernelGGL(k_pair_decimate16<8>, grid, dim3(512), sizeof(float2) * 2 * 16 * kDecPitch, st,
                                   d_pw, spectra, g, edges, pl, static_cast<const float *>(ctx->dec_taps.p), ps2.N2, gp, n_pw, rot);
            else
                hipLaunchKernelGGL(k_pair_decimate16<9>, grid, dim3(512), sizeof(float2) * 2 * 16 * kDecPitch, st,
                                   d_pw, spectra, g, edges, pl, static_cast<const float *>(ctx->dec_taps.p), ps2.N2, gp, n_pw, rot);
        }
        {
            ProfScope ps(ctx, TDOA_K_INV_COL, 3.0 * 8.0 * (double)rc_pts * n_pw);
            const int by_col = dec_walks_columns(ctx, pl, n_sw, n_pw, pairs_per_window) ? 1 : 0;
            if (ctx->small_fused && np2 == 3 && nn2 == 3 && ps2.odd == 1 && ps2.N1 == 4096 && ps2.N2 >= 8 && (n_pw >= 1024 || ctx->small_fused_always)) {
                // rows, column sums and K5 in one pass, V' never written (the reference's 20 000 lags on the 4096 x 16 / x 32 small
                // plans).  One workgroup per pair-window and CU at a time: for batches of a thousand pair-windows and more -- cfg5
                // (4500 per launch, 32 rows each) 25.2 -> 21.7 ms per step, cfg4 (2772, 16 rows) 1.26 -> 1.24; cfg2's 297 pair-windows
                // are one round and a tail of such workgroups (0.165 -> 0.253 ms) and keep the two kernels.
                hipLaunchKernelGGL(k_small_rows_col_peak, dim3(n_pw), dim3(512), sizeof(float2) * 2 * kRow8Lds, st, g,
                                   v + dec_edge_offset(pl, n_pw), d_keys, d_pw, ps2, pl.N2, by_col, lag_lo, lag_hi, lag_dump, dump_scale,
                                   static_cast<const float *>(ctx->dec_gain.p), oc);
            } else {
            hipLaunchKernelGGL(k_inv_rows_plain_r8, dim3(ps2.N2 / 2, n_pw), dim3(512), sizeof(float2) * 2 * kRow8Lds, st, g,
                               v + dec_edge_offset(pl, n_pw), vs, ps2, pl.N2, by_col);
            if (np2 == 3 && nn2 == 3)          // the reference's 20 000 lags on either small plan
                hipLaunchKernelGGL((k_small_col_peak<3, 3>), dim3(ps2.N1 / 256, n_pw), dim3(256), 0, st, vs, d_keys, d_pw, ps2, lag_lo,
                                   lag_hi, np2, nn2, lag_dump, dump_scale, static_cast<const float *>(ctx->dec_gain.p), oc);
            else
                hipLaunchKernelGGL((k_small_col_peak<0, 0>), dim3(ps2.N1 / 256, n_pw), dim3(256), 0, st, vs, d_keys, d_pw, ps2, lag_lo,
                                   lag_hi, np2, nn2, lag_dump, dump_scale, static_cast<const float *>(ctx->dec_gain.p), oc);
            }
        }
    } else if (n_pw) {
        {
            ProfScope ps(ctx, TDOA_K_INV_ROW, 3.0 * nc8 * n_pw);     // SURVEY's model: two spectra read, V written, per pair
            if (fk) {
#define TDOA_PAIR_ROWS(FK)                                                                                           \
    do {                                                                                                             \
        if (pl.N2 > 2)                                                                                               \
            hipLaunchKernelGGL((k_inv_row_pair4096<false, FK>), xcd_pairs ? dim3(xcd_grid) : dim3(pl.N2 / 2 - 1, n_pw), \
                               dim3(256), lds_pair16, st, d_pw, tz, v, pl, xcd_pairs, n_pw);                         \
        hipLaunchKernelGGL((k_inv_row_pair4096<true, FK>), dim3(1, n_pw), dim3(256), lds_pair16, st, d_pw, tz, v, pl, \
                           0, n_pw);                                                                                 \
    } while (0)
                if (fk == 1) TDOA_PAIR_ROWS(1);
                else if (fk == 2) TDOA_PAIR_ROWS(2);
                else if (fk == 4) TDOA_PAIR_ROWS(4);
                else TDOA_PAIR_ROWS(8);
            } else if (row16) {
                TDOA_PAIR_ROWS(0);
#undef TDOA_PAIR_ROWS
            } else {
                hipLaunchKernelGGL(k_inv_row_pair, dim3(pl.N2 / 2, n_pw), dim3(256), lds_row2, st, d_pw, tz, v, pl);
            }
        }
        {
            ProfScope ps(ctx, TDOA_K_INV_COL, fk ? 8.0 * 256 * fk * pl.N2 * n_pw : nc8 * n_pw);
            if (fk == 1)
                hipLaunchKernelGGL(k_fused_reduce<1>, dim3(2, n_pw), dim3(256), 0, st, v, d_keys, d_pw, pl, lag_lo, lag_hi,
                                   lag_dump, dump_scale);
            else if (fk == 2)
                hipLaunchKernelGGL(k_fused_reduce<2>, dim3(4, n_pw), dim3(256), 0, st, v, d_keys, d_pw, pl, lag_lo, lag_hi,
                                   lag_dump, dump_scale);
            else if (fk == 4)
                hipLaunchKernelGGL(k_fused_reduce<4>, dim3(8, n_pw), dim3(256), 0, st, v, d_keys, d_pw, pl, lag_lo, lag_hi,
                                   lag_dump, dump_scale);
            else if (fk == 8)
                hipLaunchKernelGGL(k_fused_reduce<8>, dim3(16, n_pw), dim3(256), 0, st, v, d_keys, d_pw, pl, lag_lo, lag_hi,
                                   lag_dump, dump_scale);
            else if (pruned) {
                const dim3 grid(pl.N1 / 128, n_pw), blk(256);
                const size_t lds_wtab = sizeof(float2) * (size_t)pl.N2;
#define TDOA_PRUNED(NP, NN)                                                                                      \
    hipLaunchKernelGGL((k_inv_col_pruned<NP, NN>), grid, blk, lds_wtab, st, v, d_keys, d_pw, pl, lag_lo, lag_hi, \
                       lag_dump, dump_scale, oc)
                const bool fixed = (pl.N2 & 31) == 0;     // the compile-time forms read 32 rows per trip unguarded
                if (fixed && np == 3 && nn == 3) TDOA_PRUNED(3, 3);
                else if (fixed && np == 1 && nn == 1) TDOA_PRUNED(1, 1);
                else if (fixed && np == 2 && nn == 2) TDOA_PRUNED(2, 2);
                else if (fixed && np == 4 && nn == 4) TDOA_PRUNED(4, 4);
                else
                    hipLaunchKernelGGL(k_inv_col_pruned_any, grid, blk, lds_wtab, st, v, d_keys, d_pw, pl, lag_lo, lag_hi, np,
                                       nn, lag_dump, dump_scale, oc);
#undef TDOA_PRUNED
            }
            else
                hipLaunchKernelGGL(k_inv_col_peak, dim3(pl.N1 / pl.C, n_pw), dim3(256), lds_col, st, v, d_keys,
                                   d_pw, pl, lag_lo, lag_hi, lag_dump, dump_scale);
        }
    }
    if (n_pw && fine_raw) {   // V (or the short-lag array) of this batch is still in place: peak neighbours for the parabola
        const dim3 g1((n_pw + 63) / 64), b1(64);
        ctx->prof_last = -1;      // unscoped launches: the next scope records its own start
        if (seg_chunks) { /* done above: k_refine_segments */ }
        else if (fk == 1) hipLaunchKernelGGL(k_refine_fused<1>, g1, b1, 0, st, v, d_keys, d_pw, pl, n_pw, fine_raw);
        else if (fk == 2) hipLaunchKernelGGL(k_refine_fused<2>, g1, b1, 0, st, v, d_keys, d_pw, pl, n_pw, fine_raw);
        else if (fk == 4) hipLaunchKernelGGL(k_refine_fused<4>, g1, b1, 0, st, v, d_keys, d_pw, pl, n_pw, fine_raw);
        else if (fk == 8) hipLaunchKernelGGL(k_refine_fused<8>, g1, b1, 0, st, v, d_keys, d_pw, pl, n_pw, fine_raw);
        else if (decim) {      // the row-pass output of the small plan is still in place behind G; window divided out per lag
            FftPlan ps2;
            if ((rc = make_plan(2 * (pl.Nc / kDecD), true, &ps2))) return fail(ctx, rc, "decimated plan");
            hipLaunchKernelGGL(k_refine_peaks, dim3(n_pw), dim3(64), 0, st, v + (size_t)(pl.Nc / kDecD) * (size_t)n_pw, d_keys, d_pw,
                               ps2, fine_raw, static_cast<const float *>(ctx->dec_gain.p), oc);
        }
        else hipLaunchKernelGGL(k_refine_peaks, dim3(n_pw), dim3(64), 0, st, v, d_keys, d_pw, pl, fine_raw, static_cast<const float *>(nullptr), oc);
    }
    HIPCHK(ctx, hipGetLastError());
    return TDOA_OK;
}

// raise the dynamic-LDS limit of every kernel that needs more than the default once per context
int allow_big_lds(tdoa_ctx *ctx)
{
    int rc;
    const size_t all = 136 * 1024;   // largest dynamic request: 128 KiB (kLdsCap tiles, generic row pair); static LDS comes on top
    if ((rc = set_lds(ctx, k_once_edges, all))) return rc;
    if ((rc = set_lds(ctx, k_fm_demod<true>, all))) return rc;
    if ((rc = set_lds(ctx, k_fm_demod<false>, all))) return rc;
    if ((rc = set_lds(ctx, k_fwd_col512_k1<false>, all))) return rc;
    if ((rc = set_lds(ctx, k_fwd_col512_k1<true>, all))) return rc;
    if ((rc = set_lds(ctx, k_fwd_row4096_unpack<false>, all))) return rc;
    if ((rc = set_lds(ctx, k_fwd_row4096_unpack<true>, all))) return rc;
    if ((rc = set_lds(ctx, (k_fwd_col256_k1<false, false>), all))) return rc;
    if ((rc = set_lds(ctx, (k_fwd_col256_k1<true, false>), all))) return rc;
    if ((rc = set_lds(ctx, (k_fwd_col256_k1<false, true>), all))) return rc;
    if ((rc = set_lds(ctx, (k_fwd_col256_k1<true, true>), all))) return rc;
    if ((rc = set_lds(ctx, k_fwd_col_c16, all))) return rc;
    if ((rc = set_lds(ctx, k_fwd_row, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_col_peak, all))) return rc;
    if ((rc = set_lds(ctx, k_fwd_col256_c16<false>, all))) return rc;
    if ((rc = set_lds(ctx, k_fwd_col256_c16<true>, all))) return rc;
    if ((rc = set_lds(ctx, k_fwd_colx_c16<2>, all))) return rc;
    if ((rc = set_lds(ctx, k_fwd_colx_c16<4>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair4096<false, 0>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair4096<true, 0>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair4096<false, 1>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair4096<true, 1>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair4096<false, 2>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair4096<true, 2>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair4096<false, 4>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair4096<true, 4>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair4096<false, 8>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_row_pair4096<true, 8>, all))) return rc;
    if ((rc = set_lds(ctx, (k_fm_demod<true, true>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments<1, false>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments<2, false>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments<4, false>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments_quad<1, false>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments_quad<2, false>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments_quad<4, false>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments<1, true>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments<2, true>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments<4, true>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments_quad<1, true>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments_quad<2, true>), all))) return rc;
    if ((rc = set_lds(ctx, (k_xcorr_segments_quad<4, true>), all))) return rc;
    if ((rc = set_lds(ctx, k_pair_decimate16<8>, all))) return rc;
    if ((rc = set_lds(ctx, k_pair_decimate16<9>, all))) return rc;
    if ((rc = set_lds(ctx, k_inv_rows_plain_r8, all))) return rc;
    if ((rc = set_lds(ctx, k_small_rows_col_peak, all))) return rc;
#if TDOA_HAVE_DEC_COLS
#define TDOA_STG_LDS(N2V, RV)                                                                    \
    if ((rc = set_lds(ctx, (k_pair_decimate_staged<N2V, RV>), all))) return rc;
#define TDOA_STG_LDS_N(N2V) TDOA_STG_LDS(N2V, 2) TDOA_STG_LDS(N2V, 4) TDOA_STG_LDS(N2V, 8)
    TDOA_STG_LDS_N(256) TDOA_STG_LDS_N(512) TDOA_STG_LDS_N(2048) TDOA_STG_LDS_N(2560) TDOA_STG_LDS_N(3072) TDOA_STG_LDS_N(4096)
#undef TDOA_STG_LDS_N
#undef TDOA_STG_LDS
#endif
    return TDOA_OK;
}

int check_ctx(tdoa_ctx *ctx)
{
    if (!ctx) return TDOA_ERR_INVALID;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, TDOA_ERR_HIP, "hipSetDevice", e);
    return TDOA_OK;
}

// Transform length for `need` = window + search range samples.  The reference's rule is the next power of two
// (processor.go:563 -- dead code there, processor.go:638, and a design hint here: any N >= need gives the same linear
// correlation).  Ten-second windows at 2 Msps need 20 020 000 points: 2^25 = 33 554 432 is 40 % zero padding that every pass
// moves; 5 x 2^22 = 20 971 520 (packed 4096 x 2560) holds them with 4.5 %.  That plan exists where the step runs the fused /
// two-sweep column pass and the decimated inverse as a column walk (decimation_applies): everything else -- short search
// ranges, TDOA_LAGS_GO, TDOA_NO_DECIMATE, the any-size kernels -- keeps the power of two, and so does TDOA_POW2_ONLY=1 /
// TDOA_DEBUG_POW2_ONLY (the A/B switch).
long long choose_fft_size(const tdoa_ctx *ctx, long long need, int lag_lo, int lag_hi, int zpad, FftPlan *pl, int *rc)
{
    const long long p = std::max<long long>(next_pow2(need), 64);
    if (!ctx->pow2_only && p == (1ll << 25)) {
        // 5 x 2^22 = 20 971 520 (4096 x 2560), then 3 x 2^23 = 25 165 824 (4096 x 3072: windows of 10.5 to 12.6 s at 2 Msps)
        for (const long long cand : {5ll << 22, 3ll << 23}) {
            FftPlan q;
            if (need <= cand && make_plan(cand, true, &q, zpad) == TDOA_OK && decimation_applies(ctx, q, lag_lo, lag_hi)) {
                *pl = q;
                *rc = TDOA_OK;
                return cand;
            }
        }
    }
    *rc = make_plan(p, true, pl, zpad);
    return p;
}

// timeDomainCorrelation's block count for a template of lt samples (processor.go:691: starts 0, cb, 2 cb, ... < lt - cb)
long long go_blocks(long long lt, long long cb) { return lt > cb ? (lt - cb + cb - 1) / cb : 0; }

// copy two host IQ windows into scratch and build 2 sw + 1 pw descriptors
// (corr_len1: samples of the first window the transforms see, <= n1; the descriptors with the full lengths follow at
// d_sw + 2 for K1 and its statistics)
int stage_pair_u8(tdoa_ctx *ctx, const uint8_t *iq1, size_t n1, const uint8_t *iq2, size_t n2,
                  SWDesc **d_sw, PWDesc **d_pw, size_t corr_len1)
{
    int rc;
    size_t b1 = (2 * n1 + 15) & ~(size_t)15;
    if ((rc = ensure(ctx, ctx->scratch_a, b1 + 2 * n2 + 16))) return rc;
    auto *base = static_cast<uint8_t *>(ctx->scratch_a.p);
    if (n1) HIPCHK(ctx, hipMemcpyAsync(base, iq1, 2 * n1, hipMemcpyHostToDevice, ctx->stream));
    if (n2) HIPCHK(ctx, hipMemcpyAsync(base + b1, iq2, 2 * n2, hipMemcpyHostToDevice, ctx->stream));
    SWDesc sw[4] = {{base, (int32_t)corr_len1, 0}, {base + b1, (int32_t)n2, 0}, {base, (int32_t)n1, 0}, {base + b1, (int32_t)n2, 0}};
    PWDesc pw = {0, 1, 0, (int32_t)corr_len1};
    if ((rc = ensure(ctx, ctx->sw_desc, sizeof(sw)))) return rc;
    if ((rc = ensure(ctx, ctx->pw_desc, sizeof(pw)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->sw_desc.p, sw, sizeof(sw), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->pw_desc.p, &pw, sizeof(pw), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // sw/pw are stack objects
    *d_sw = static_cast<SWDesc *>(ctx->sw_desc.p);
    *d_pw = static_cast<PWDesc *>(ctx->pw_desc.p);
    return TDOA_OK;
}

int fm_pair(tdoa_ctx *ctx, const uint8_t *iq1, size_t n1, const uint8_t *iq2, size_t n2, int max_lag,
            tdoa_peak *peak, double *lags_out, tdoa_fine_peak *fine = nullptr, double gate = 0.0)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (max_lag < 1 || (!peak && !lags_out && !fine)) return fail(ctx, TDOA_ERR_INVALID, "bad argument");
    if (n1 == 0 || n2 == 0) {   // processor.go:622-625 behaviour: (0, 0.0)
        if (peak) *peak = tdoa_peak{0, 0.0f, 0.0};
        if (fine) *fine = tdoa_fine_peak{0.0, 0.0f, {0.0f, 0.0f, 0.0f}, gate >= 0.0 ? 1 : 0, 0};
        if (lags_out) std::fill(lags_out, lags_out + (2 * max_lag - 1), 0.0);
        return TDOA_OK;
    }
    if (n1 > 0x7fffffff / 2 || n2 > 0x7fffffff / 2) return fail(ctx, TDOA_ERR_UNSUPPORTED, "window too long");
    // TDOA_LAGS_GO: template = the shorter input (ties: the first), its first B corr_block samples, lags [0, eff)
    const bool go = ctx->prm.lag_mode == TDOA_LAGS_GO;
    if (go && fine) return fail(ctx, TDOA_ERR_UNSUPPORTED, "sub-sample refinement with TDOA_LAGS_GO");
    if (go && n2 < n1) {                                      // processor.go:650-655
        std::swap(iq1, iq2);
        std::swap(n1, n2);
    }
    const int nl = 2 * max_lag - 1;
    size_t corr_len = n1;
    int lag_lo = -(max_lag - 1), lag_hi = max_lag - 1;
    if (go) {
        const long long blocks = go_blocks((long long)n1, ctx->prm.corr_block);
        if (blocks == 0) {                                    // processor.go:708-717: no block, (0, 0.0)
            if (peak) *peak = tdoa_peak{0, 0.0f, 0.0};
            if (lags_out) std::fill(lags_out, lags_out + nl, 0.0);
            return TDOA_OK;
        }
        corr_len = (size_t)(blocks * ctx->prm.corr_block);
        const long long eff = std::max<long long>(1, std::min<long long>(max_lag, (long long)n2 - (long long)n1));   // :668-678
        lag_lo = 0;
        lag_hi = (int)eff - 1;
    }
    FftPlan pl;
    const long long n = choose_fft_size(ctx, (long long)std::max(n1, n2) + max_lag, lag_lo, lag_hi, ctx->zpad, &pl, &rc);
    if (rc) return fail(ctx, rc, "FFT size unsupported");
    ctx->plan = pl;            // tdoa_plan_info reports the plan of the last call, pair calls included
    ctx->plan_n = n;
    SWDesc *d_sw;
    PWDesc *d_pw;
    if ((rc = stage_pair_u8(ctx, iq1, n1, iq2, n2, &d_sw, &d_pw, corr_len))) return rc;
    if ((rc = ensure(ctx, ctx->keys, sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(ctx, ctx->scales, sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->peaks, sizeof(PeakOut)))) return rc;
    if ((rc = ensure(ctx, ctx->slot_gain, sizeof(double)))) return rc;
    if (fine) {
        if ((rc = ensure(ctx, ctx->fine_raw, 3 * sizeof(float)))) return rc;
        if ((rc = ensure(ctx, ctx->fine, sizeof(FineOut)))) return rc;
    }
    const int n_dump = lag_hi - lag_lo + 1;                   // the kernels write lag d at dump[d - lag_lo]
    float *dump = nullptr;
    if (lags_out) {
        if ((rc = ensure(ctx, ctx->lagdump, sizeof(float) * (size_t)n_dump))) return rc;
        dump = static_cast<float *>(ctx->lagdump.p);
        HIPCHK(ctx, hipMemsetAsync(dump, 0, sizeof(float) * (size_t)n_dump, ctx->stream));
    }
    double scale = 1.0 / (4.0 * (double)n * std::sqrt((double)corr_len));
    HIPCHK(ctx, hipMemsetAsync(ctx->keys.p, 0, sizeof(unsigned long long), ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->scales.p, &scale, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ctx->prof_last = -1;
    rc = run_fm_batch(ctx, d_sw, 2, (int)std::max(n1, n2), d_pw, 1, static_cast<unsigned long long *>(ctx->keys.p),
                      pl, lag_lo, lag_hi, dump, 1.0f, (double)(n1 + n2),
                      fine ? static_cast<float *>(ctx->fine_raw.p) : nullptr, 0, nullptr, 0, n1 >= 2 && n2 >= 2 && corr_len >= 2,
                      corr_len != n1 ? d_sw + 2 : nullptr, n1 == n2 && corr_len == n1);
    if (rc) return rc;
    const double *slot_gain = ctx->once_active ? static_cast<const double *>(ctx->slot_gain.p) : nullptr;
    hipLaunchKernelGGL(k_decode_peaks, dim3(1), dim3(64), 0, ctx->stream,
                       static_cast<unsigned long long *>(ctx->keys.p), static_cast<double *>(ctx->scales.p),
                       static_cast<PeakOut *>(ctx->peaks.p), 1, slot_gain);
    tdoa_peak pk;
    HIPCHK(ctx, hipMemcpyAsync(&pk, ctx->peaks.p, sizeof(pk), hipMemcpyDeviceToHost, ctx->stream));
    tdoa_fine_peak fk;
    if (fine) {
        hipLaunchKernelGGL(k_decode_fine, dim3(1), dim3(64), 0, ctx->stream,
                           static_cast<unsigned long long *>(ctx->keys.p), static_cast<double *>(ctx->scales.p),
                           static_cast<float *>(ctx->fine_raw.p), static_cast<FineOut *>(ctx->fine.p), gate, 1, slot_gain);
        HIPCHK(ctx, hipMemcpyAsync(&fk, ctx->fine.p, sizeof(fk), hipMemcpyDeviceToHost, ctx->stream));
    }
    double pair_gain = 1.0;                                   // single-look K1: the lag array lacks g_t g_s like the key does
    if (lags_out && slot_gain)
        HIPCHK(ctx, hipMemcpyAsync(&pair_gain, slot_gain, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<float> hl;
    if (lags_out) {
        hl.resize(n_dump);
        HIPCHK(ctx, hipMemcpyAsync(hl.data(), dump, sizeof(float) * (size_t)n_dump, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    if (peak) *peak = pk;
    if (fine) *fine = fk;
    if (lags_out) {                                           // layout [2 max_lag - 1]: lag d at d + max_lag - 1
        std::fill(lags_out, lags_out + nl, 0.0);
        for (int i = 0; i < n_dump; i++) lags_out[i + lag_lo + (max_lag - 1)] = (double)hl[i] * scale * pair_gain;
    }
    return TDOA_OK;
}

}  // namespace

// ===========================================================================
// lifecycle
// ===========================================================================
extern "C" {

void tdoa_default_params(tdoa_params *p)
{
    if (!p) return;
    p->sample_rate = 2000000.0;   // processor.go:440
    p->max_lag = 20000;           // processor.go:633
    p->corr_block = 1000;         // processor.go:682
    p->weak_threshold = 0.001;    // processor.go:476
    p->window_len = 2000000;      // processor.go:772
    p->device = 0;
    p->windows_per_batch = 0;
    p->k1_smooth = 0;
    p->k1_gate = 0;
    p->lag_mode = TDOA_LAGS_SIGNED;
    p->reserved = 0;
}

int tdoa_abi_version(void) { return TDOA_ABI_VERSION; }

int tdoa_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *tdoa_strerror(int status)
{
    switch (status) {
        case TDOA_OK: return "ok";
        case TDOA_ERR_INVALID: return "invalid argument";
        case TDOA_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
        case TDOA_ERR_HIP: return "HIP runtime error";
        case TDOA_ERR_NOMEM: return "out of device memory";
        case TDOA_ERR_UNSUPPORTED: return "unsupported size";
        case TDOA_ERR_STATE: return "invalid call order";
        case TDOA_ERR_SINGULAR: return "singular Jacobian matrix";
        default: return "unknown status";
    }
}

const char *tdoa_last_error(const tdoa_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

const char *tdoa_kernel_name(int k)
{
    static const char *names[TDOA_K_COUNT] = {"k_fm_demod", "k_fwd_col", "k_fwd_row", "k_inv_row_pair",
                                              "k_inv_col_peak", "k_decode_peaks"};
    return (k >= 0 && k < TDOA_K_COUNT) ? names[k] : "";
}

int tdoa_create(const tdoa_params *p, tdoa_ctx **out)
{
    if (!out) return TDOA_ERR_INVALID;
    *out = nullptr;
    tdoa_params prm;
    if (p)
        prm = *p;
    else
        tdoa_default_params(&prm);
    if (prm.max_lag < 1 || prm.corr_block < 1 || prm.window_len < 2 || !(prm.sample_rate > 0) || prm.k1_smooth < 0 ||
        prm.k1_smooth > 2001 || (prm.lag_mode != TDOA_LAGS_SIGNED && prm.lag_mode != TDOA_LAGS_GO))
        return TDOA_ERR_INVALID;
    int ndev = tdoa_device_count();
    if (ndev <= 0 || prm.device < 0 || prm.device >= ndev) return TDOA_ERR_NO_DEVICE;
    if (hipSetDevice(prm.device) != hipSuccess) return TDOA_ERR_NO_DEVICE;
    tdoa_ctx *ctx = new (std::nothrow) tdoa_ctx();
    if (!ctx) return TDOA_ERR_NOMEM;
    ctx->prm = prm;
    ctx->device = prm.device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, prm.device) == hipSuccess && cus > 0)
            ctx->n_cu = cus;
        size_t total_mem = 0;
        if (hipDeviceTotalMem(&total_mem, prm.device) == hipSuccess && total_mem > 0)
            ctx->workspace_limit = std::max(8.0 * 1073741824.0, (double)total_mem / 3.0);
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return TDOA_ERR_HIP;
    }
    if (allow_big_lds(ctx) != TDOA_OK) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return TDOA_ERR_HIP;
    }
    {   // K1 angle tables, once per context (the first-octant table is the host's source for the two the kernels use)
        std::vector<int32_t> tab, direct, quad;
        k1_build_table_host(tab, direct, quad);
        void *dt = nullptr, *dq = nullptr;
        if (hipMalloc(&dt, kK1DirectBytes) != hipSuccess) {
            (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return TDOA_ERR_NOMEM;
        }
        ctx->k1_direct.p = dt;
        ctx->k1_direct.cap = kK1DirectBytes;
        if (hipMalloc(&dq, kK1QuadrantBytes) != hipSuccess) {
            tdoa_destroy(ctx);
            return TDOA_ERR_NOMEM;
        }
        ctx->k1_quad.p = dq;
        ctx->k1_quad.cap = kK1QuadrantBytes;
        if (hipMemcpy(dq, quad.data(), kK1QuadrantBytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dt, direct.data(), kK1DirectBytes, hipMemcpyHostToDevice) != hipSuccess) {
            tdoa_destroy(ctx);
            return TDOA_ERR_HIP;
        }
    }
    // run-time switches are read ONCE here (a captured graph must not depend on an environment that changes later)
    if (const char *e = std::getenv("TDOA_NO_GRAPH")) ctx->use_graph = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_NO_SHORT_LAG")) ctx->short_lag = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_NO_SEGMENT_FORM")) ctx->segment_form = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_NO_SEGMENT_QUADS")) ctx->segment_quads = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_NO_DECIMATE")) ctx->decimate = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_NO_K1_ONCE")) ctx->k1_once = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_POW2_ONLY")) ctx->pow2_only = e[0] == '1';
    if (const char *e = std::getenv("TDOA_ZPAD")) {
        const int v = std::atoi(e);
        ctx->zpad = v < 0 ? 0 : v > 4096 ? 4096 : v & ~15;      // rows stay 128-byte aligned (the finish sweep reads 16-byte pairs)
    }
    if (const char *e = std::getenv("TDOA_NO_FUSED_K1")) ctx->fused_k1 = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_NO_DEC_COLS")) ctx->dec_cols = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_DEC_COLS_ALWAYS")) ctx->dec_cols_always = e[0] == '1';
    if (const char *e = std::getenv("TDOA_NO_DEC_STAGED")) ctx->dec_staged = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_NO_SMALL_FUSED")) ctx->small_fused = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_SMALL_FUSED_ALWAYS")) ctx->small_fused_always = e[0] == '1';
    if (const char *e = std::getenv("TDOA_NO_STG_FOLDED")) ctx->stg_folded = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_STG_FOLDED_ALWAYS")) ctx->stg_folded_always = e[0] == '1';
    if (const char *e = std::getenv("TDOA_NO_STG_BLOCKS")) ctx->stg_blocks = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_DEC_STAGED_LOADERS")) ctx->stg_loaders = std::max(0, std::min(4, std::atoi(e)));
    if (const char *e = std::getenv("TDOA_DEC_STAGED_ROWS")) ctx->stg_rows = std::atoi(e) == 8 ? 8 : std::atoi(e) == 4 ? 4 : std::atoi(e) == 2 ? 2 : 0;
    if (const char *e = std::getenv("TDOA_DEC_STAGED_CW")) ctx->stg_cw = std::max(0, std::min(15, std::atoi(e)));
    if (const char *e = std::getenv("TDOA_DEC_STAGED_BUFS")) ctx->stg_bufs = std::max(0, std::min(16, std::atoi(e)));
    if (const char *e = std::getenv("TDOA_NO_SEG_PACK3")) ctx->seg_pack3 = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_SEG_CHUNKS")) ctx->seg_chunks_override = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("TDOA_DEBUG_MEMSET_NODES")) ctx->memset_nodes = e[0] == '1';
    if (const char *e = std::getenv("TDOA_NO_XCD_ROWS")) ctx->xcd_rows = !(e[0] == '1');
    if (const char *e = std::getenv("TDOA_XCD_PAIR_MB")) ctx->xcd_pair_mb = std::max(0, std::atoi(e));
    *out = ctx;
    return TDOA_OK;
}

void tdoa_destroy(tdoa_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    prof_collect(ctx);
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
    clear_graph_marks(ctx);
    if (ctx->graph) (void)hipGraphDestroy(ctx->graph);
    tdoa_capture_clear(ctx);
    DevBuf *bufs[] = {&ctx->k1_direct, &ctx->k1_quad, &ctx->sw_desc, &ctx->pw_desc, &ctx->partials, &ctx->stats, &ctx->codes, &ctx->codes_lp, &ctx->k1_power, &ctx->dec_taps, &ctx->dec_gain, &ctx->stg_groups, &ctx->tz, &ctx->v, &ctx->keys,
                      &ctx->scales, &ctx->peaks, &ctx->scratch_a, &ctx->scratch_b, &ctx->lagdump,
                      &ctx->ex_a, &ctx->ex_b, &ctx->ex_c, &ctx->ex_d, &ctx->ex_part,
                      &ctx->g_sw_desc, &ctx->g_pw_desc, &ctx->g_quad_desc, &ctx->g_scales, &ctx->g_keys, &ctx->fine_raw, &ctx->fine, &ctx->qual,
                      &ctx->once_edges, &ctx->once_tiles, &ctx->once_fin, &ctx->slot_gain};
    for (DevBuf *b : bufs) release(*b);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ===========================================================================
// mode B
// ===========================================================================

// device buffer of `bytes` for a station's capture: the station's previous owned buffer when it is large enough
// (re-uploading captures of the same size keeps the pointers, hence the captured graph), else a new allocation
static int capture_buffer(tdoa_ctx *ctx, int station, size_t bytes, uint8_t **out)
{
    if ((size_t)station >= ctx->caps.size()) ctx->caps.resize(station + 1);
    auto &c = ctx->caps[station];
    if (c.owned && c.dev && c.cap_bytes >= bytes) {
        *out = const_cast<uint8_t *>(c.dev);
        return TDOA_OK;
    }
    if (c.owned && c.dev) (void)hipFree(const_cast<uint8_t *>(c.dev));
    c = tdoa_ctx::Capture{};
    void *d = nullptr;
    const hipError_t e = hipMalloc(&d, bytes + 64);
    if (e != hipSuccess) return fail(ctx, TDOA_ERR_NOMEM, "hipMalloc capture", e);
    c.dev = static_cast<const uint8_t *>(d);
    c.owned = true;
    c.cap_bytes = bytes;
    c.n = 0;
    *out = static_cast<uint8_t *>(d);
    return TDOA_OK;
}

// host memory (src) or file (fd, from offset 0) -> device, through the context's staged uploader
static int staged_upload(tdoa_ctx *ctx, uint8_t *dst, const uint8_t *src, int fd, size_t bytes)
{
    if (bytes == 0) return TDOA_OK;
    if (bytes < (1u << 20) && src) {   // small: one plain copy beats waking threads
        HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        return TDOA_OK;
    }
    int n_threads = 4;   // measured: 2-4 copy threads reach ~40 GB/s on the bench node, 12 fall back to 25
    if (const char *e = std::getenv("TDOA_UPLOAD_THREADS")) n_threads = std::atoi(e);
    if (!ctx->uploader.init(ctx->device, n_threads)) {
        (void)hipGetLastError();
        return fail(ctx, TDOA_ERR_HIP, "staging buffers for the uploader");
    }
    const int st = ctx->uploader.run(dst, src, fd, bytes);
    if (st == 2) return fail(ctx, TDOA_ERR_INVALID, "failed to read data");
    if (st) {
        (void)hipGetLastError();
        return fail(ctx, TDOA_ERR_HIP, "host to device copy failed");
    }
    return TDOA_OK;
}

int tdoa_capture_upload(tdoa_ctx *ctx, int station, const uint8_t *iq, size_t n_samples)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (station < 0 || station > 1023 || (!iq && n_samples)) return fail(ctx, TDOA_ERR_INVALID, "bad station/iq");
    uint8_t *d = nullptr;
    if ((rc = capture_buffer(ctx, station, 2 * n_samples, &d))) return rc;
    ctx->caps[station].n = 0;                                  // not valid until the copy has finished
    if ((rc = staged_upload(ctx, d, iq, -1, 2 * n_samples))) return rc;
    ctx->caps[station].n = n_samples;
    return TDOA_OK;
}

// Sharded ingest: a rank of a multi-GPU job only reads the windows it owns (tdoa_process(rank, world)), so it only
// needs those bytes in its HBM.  The station's buffer has the full capture's size (window offsets stay what they are);
// the samples outside the uploaded ranges are never read by that rank.
int tdoa_capture_upload_range(tdoa_ctx *ctx, int station, size_t total_samples, size_t first_sample, const uint8_t *iq,
                              size_t n_samples)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (station < 0 || station > 1023 || (!iq && n_samples) || first_sample > total_samples ||
        n_samples > total_samples - first_sample)
        return fail(ctx, TDOA_ERR_INVALID, "bad station/iq/range");
    uint8_t *d = nullptr;
    if ((rc = capture_buffer(ctx, station, 2 * total_samples, &d))) return rc;
    const size_t had = ctx->caps[station].n;
    ctx->caps[station].n = 0;                                  // not valid until the copy has finished
    if ((rc = staged_upload(ctx, d + 2 * first_sample, iq, -1, 2 * n_samples))) return rc;
    (void)had;
    ctx->caps[station].n = total_samples;
    return TDOA_OK;
}

int tdoa_capture_upload_file(tdoa_ctx *ctx, int station, const char *path, size_t *n_samples)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (station < 0 || station > 1023 || !path) return fail(ctx, TDOA_ERR_INVALID, "bad station/path");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(ctx, TDOA_ERR_INVALID, "failed to open file");
    if (std::fseek(f, 0, SEEK_END) != 0) { std::fclose(f); return fail(ctx, TDOA_ERR_INVALID, "failed to get file size"); }
    const long long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (size < 0) { std::fclose(f); return fail(ctx, TDOA_ERR_INVALID, "failed to get file size"); }
    const size_t n = (size_t)size / 2;                       // processor.go:182
    uint8_t *d = nullptr;
    if ((rc = capture_buffer(ctx, station, 2 * n, &d))) { std::fclose(f); return rc; }
    ctx->caps[station].n = 0;
    rc = staged_upload(ctx, d, nullptr, fileno(f), 2 * n);
    std::fclose(f);
    if (rc) return rc;
    ctx->caps[station].n = n;
    if (n_samples) *n_samples = n;
    return TDOA_OK;
}

int tdoa_capture_attach_device(tdoa_ctx *ctx, int station, const void *dev_iq, size_t n_samples)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (station < 0 || station > 1023 || !dev_iq || ((uintptr_t)dev_iq & 1)) return fail(ctx, TDOA_ERR_INVALID, "bad station/pointer");
    if ((size_t)station >= ctx->caps.size()) ctx->caps.resize(station + 1);
    auto &c = ctx->caps[station];
    if (c.owned && c.dev) (void)hipFree(const_cast<uint8_t *>(c.dev));
    c.dev = static_cast<const uint8_t *>(dev_iq);
    c.n = n_samples;
    c.owned = false;
    c.cap_bytes = 0;
    return TDOA_OK;
}

int tdoa_synth_capture(tdoa_ctx *ctx, int station, size_t block_samples, double ref_freq, double tgt_freq,
                       double noise_level, const double station_lle[3], const double tx_lle[3], double tx_power,
                       uint64_t seed)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (station < 0 || station > 1023 || block_samples < 2 || !station_lle || !tx_lle)
        return fail(ctx, TDOA_ERR_INVALID, "bad argument");
    uint8_t *d = nullptr;
    if ((rc = capture_buffer(ctx, station, 6 * block_samples, &d))) return rc;
    ctx->caps[station].n = 0;
    // simulator.go:104-120: distance -> travel time -> carrier phase; amplitude power/d*0.1
    double a[3], b[3];
    geo::latlon_to_ecef(station_lle[0], station_lle[1], station_lle[2], a);
    geo::latlon_to_ecef(tx_lle[0], tx_lle[1], tx_lle[2], b);
    const double dist = geo::range(a, b);
    const double travel = dist / geo::kC;
    const double fs = ctx->prm.sample_rate;
    const double phase = 2 * geo::kPi * tgt_freq * travel;
    const double amp = tx_power / dist * 0.1;
    SynthBlock blk[3] = {
        {2 * geo::kPi * ref_freq / fs, 0.0, 0.01, noise_level, seed, 1},      // simulator.go:126-128
        {2 * geo::kPi * tgt_freq / fs, phase, amp, noise_level, seed, 2},     // simulator.go:131-133
        {2 * geo::kPi * ref_freq / fs, 0.0, 0.01, noise_level, seed, 3},      // simulator.go:136-138
    };
    const long long n = (long long)block_samples;
    for (int k = 0; k < 3; k++)
        hipLaunchKernelGGL(k_synth_tone_block, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           d + 2 * n * k, n, blk[k]);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->caps[station].n = 3 * block_samples;
    return TDOA_OK;
}

int tdoa_synth_weak_capture(tdoa_ctx *ctx, int station, size_t block_samples, double ref_freq, double tgt_freq,
                            const double station_lle[3], const double tx_lle[3], double ref_power, double tgt_power,
                            uint64_t seed)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (station < 0 || station > 1023 || block_samples < 2 || !station_lle || !tx_lle)
        return fail(ctx, TDOA_ERR_INVALID, "bad argument");
    uint8_t *d = nullptr;
    if ((rc = capture_buffer(ctx, station, 6 * block_samples, &d))) return rc;
    ctx->caps[station].n = 0;
    // weak_signal_simulator.go:155-173: distance -> travel time -> carrier phases; amplitudes power/d*0.1
    double a[3], b[3];
    geo::latlon_to_ecef(station_lle[0], station_lle[1], station_lle[2], a);
    geo::latlon_to_ecef(tx_lle[0], tx_lle[1], tx_lle[2], b);
    const double dist = geo::range(a, b);
    const double travel = dist / geo::kC;
    const double fs = ctx->prm.sample_rate;
    const double ref_amp = ref_power / dist * 0.1, tgt_amp = tgt_power / dist * 0.1;
    // weak profile, weak_signal_simulator.go:180-186; the strong block adds 0.001 sigma of noise only (:141-143)
    SynthWeakBlock weak = {2 * geo::kPi * ref_freq / fs, 2 * geo::kPi * ref_freq * travel, ref_amp, ref_amp * 0.8, 0.001,
                           ref_amp * 5.0, 0.05 / fs, ref_amp * 0.1, seed, 1, 1};
    SynthWeakBlock strong = {2 * geo::kPi * tgt_freq / fs, 2 * geo::kPi * tgt_freq * travel, tgt_amp, 0.001, 0.0, 0.0, 0.0,
                             0.0, seed, 2, 0};
    const long long n = (long long)block_samples;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    hipLaunchKernelGGL(k_synth_weak_block, grid, blk, 0, ctx->stream, d, n, weak);                    // block 1: weak reference
    hipLaunchKernelGGL(k_synth_weak_block, grid, blk, 0, ctx->stream, d + 2 * n, n, strong);          // block 2: strong target
    weak.block_id = 3;
    hipLaunchKernelGGL(k_synth_weak_block, grid, blk, 0, ctx->stream, d + 4 * n, n, weak);            // block 3: weak reference
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->caps[station].n = 3 * block_samples;
    return TDOA_OK;
}

int tdoa_capture_download(tdoa_ctx *ctx, int station, size_t first_sample, size_t n_samples, uint8_t *out)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (station < 0 || (size_t)station >= ctx->caps.size() || !ctx->caps[station].dev || !out)
        return fail(ctx, TDOA_ERR_INVALID, "no such capture");
    const auto &c = ctx->caps[station];
    if (first_sample > c.n || n_samples > c.n - first_sample) return fail(ctx, TDOA_ERR_INVALID, "range outside capture");
    HIPCHK(ctx, hipMemcpy(out, c.dev + 2 * first_sample, 2 * n_samples, hipMemcpyDeviceToHost));
    return TDOA_OK;
}

int tdoa_capture_clear(tdoa_ctx *ctx)
{
    if (!ctx) return TDOA_ERR_INVALID;
    for (auto &c : ctx->caps)
        if (c.owned && c.dev) (void)hipFree(const_cast<uint8_t *>(c.dev));
    ctx->caps.clear();
    return TDOA_OK;
}

static int window_geometry(const tdoa_ctx *ctx, long long *block, long long *wlen, int *wpb)
{
    if (ctx->caps.size() < 2) return TDOA_ERR_STATE;
    size_t nmin = (size_t)-1;
    for (auto &c : ctx->caps) {
        if (!c.dev) return TDOA_ERR_STATE;
        nmin = std::min(nmin, c.n);
    }
    // every capture is cut into its OWN thirds (processor.go:214 takes len(signal)/3 per file); the window grid
    // comes from the shortest one, so captures of unequal length still pair block k window w with block k window w
    long long b = (long long)(nmin / 3);
    if (b < 2) return TDOA_ERR_UNSUPPORTED;
    long long l = std::min<long long>(ctx->prm.window_len, b);
    *block = b;
    *wlen = l;
    *wpb = (int)std::max<long long>(1, b / l);
    return TDOA_OK;
}

int tdoa_num_windows(const tdoa_ctx *ctx, int *windows_per_block, int *n_windows_total)
{
    if (!ctx) return TDOA_ERR_INVALID;
    long long b, l;
    int wpb;
    int rc = window_geometry(ctx, &b, &l, &wpb);
    if (rc) return rc;
    if (windows_per_block) *windows_per_block = wpb;
    if (n_windows_total) *n_windows_total = 3 * wpb;
    return TDOA_OK;
}

int tdoa_num_pairs(const tdoa_ctx *ctx)
{
    if (!ctx) return 0;
    int s = (int)ctx->caps.size();
    return s * (s - 1) / 2;
}

int tdoa_plan_info(const tdoa_ctx *ctx, int64_t *fft_n, int32_t *n1, int32_t *n2)
{
    if (!ctx || !ctx->plan_n) return TDOA_ERR_STATE;
    if (fft_n) *fft_n = ctx->plan_n;
    if (n1) *n1 = ctx->plan.N1;
    if (n2) *n2 = ctx->plan.N2;
    return TDOA_OK;
}

static int process_impl(tdoa_ctx *ctx, int rank, int world, tdoa_peak *out_host, void *out_dev,
                        tdoa_fine_peak *fine_host, double gate)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, TDOA_ERR_INVALID, "bad rank/world");
    long long block, wlen;
    int wpb;
    if ((rc = window_geometry(ctx, &block, &wlen, &wpb))) return fail(ctx, rc, "captures missing or too small");
    const int S = (int)ctx->caps.size();
    const int P = S * (S - 1) / 2;
    const int W = 3 * wpb;
    // TDOA_LAGS_GO: every window has the same length, so timeDomainCorrelation evaluates lag 0 only (processor.go:668-678)
    const bool go = ctx->prm.lag_mode == TDOA_LAGS_GO;
    const int lag_lo = go ? 0 : -(ctx->prm.max_lag - 1), lag_hi = go ? 0 : ctx->prm.max_lag - 1;
    FftPlan pl;
    const long long n = choose_fft_size(ctx, wlen + ctx->prm.max_lag, lag_lo, lag_hi, ctx->zpad, &pl, &rc);
    if (rc) return fail(ctx, rc, "FFT size unsupported");
    ctx->plan = pl;
    ctx->plan_n = n;
    // (TDOA_LAGS_GO) ... over the first B corr_block samples; with the template cut there the signal's samples beyond do not enter
    // lag 0 either, so every station-window is cut for the transforms (K1 and its statistics see the whole window)
    if (go && fine_host) return fail(ctx, TDOA_ERR_UNSUPPORTED, "sub-sample refinement with TDOA_LAGS_GO");
    const long long corr_len = go ? go_blocks(wlen, ctx->prm.corr_block) * ctx->prm.corr_block : wlen;

    // Sharding: window-major -- rank r owns the windows wid = r (mod world), so a station-window
    // is transformed once and reused by all its pairs.  With fewer windows than ranks that would leave ranks idle:
    // then the (window, pair) units u = wid*P + p are dealt u = r (mod world) and a rank transforms only the
    // stations its pairs need (station spectra are duplicated across ranks).
    const bool pair_major = W < world;
    auto owns = [&](int wid, int p) { return pair_major ? ((wid * P + p) % world) == rank : (wid % world) == rank; };
    std::vector<int> mine;
    for (int w = 0; w < W; w++) {
        bool any = false;
        for (int p = 0; p < P && !any; p++) any = owns(w, p);
        if (any) mine.push_back(w);
    }
    // default: every window of this rank in one launch group (launch tails cost more than cache residency gains), bounded
    // by a third of the device's memory for the workspace (96 GB of an MI355X's 288: cfg4's 99 windows x 36 spectra are one
    // group of 30 GB; round 3 stopped at 24 GiB and ran them as 85 + 14)
    int per_batch = ctx->prm.windows_per_batch > 0 ? ctx->prm.windows_per_batch : (int)std::max<size_t>(mine.size(), 1);
    // what reserve_fm_batch asks for per window (+ 1/8: ensure() rounds every buffer up): TZ, the V workspace in the form this
    // plan's pair step uses (dec_spectra_offset + the tiled spectra behind the tile form), code rows where K1 is materialised
    double bytes_per_window = 8.0 * (double)pl.Zs * S + 8.0 * (double)pl.Nc * P;
    if (decimation_applies(ctx, pl, lag_lo, lag_hi))
        bytes_per_window = 8.0 * (double)pl.Zs * S +
                           8.0 * std::max((double)pl.Nc * P, (double)dec_spectra_offset(pl, P) + (cols_only_plan(pl) ? 0.0 : (double)pl.Nc * S));
    if (!fused_k1_applies(ctx, pl, lag_lo, lag_hi, P, true)) bytes_per_window += 4.0 * (double)(wlen + 16) * S * (ctx->prm.k1_smooth > 1 ? 2 : 1);
    bytes_per_window *= 1.125;
    // the bound: a third of the device (tdoa_create), and not more than is FREE now plus what this context already holds of it
    // (captures attached by the caller, other contexts, other ranks on the same card all count against the device)
    double limit = ctx->workspace_limit;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double held = (double)ctx->tz.cap + (double)ctx->v.cap + (double)ctx->codes.cap + (double)ctx->codes_lp.cap;
            limit = std::min(limit, std::max(0.0, (double)free_b + held - 1073741824.0));      // 1 GiB stays free: descriptors, edges, the runtime
        } else {
            (void)hipGetLastError();
        }
    }
    per_batch = (int)std::max(1.0, std::min<double>(per_batch, limit / bytes_per_window));
    // per_batch * S and per_batch * P become gridDim.y of the FFT kernels (HIP limit 65535)
    if (std::max(S, P) > 65535) return fail(ctx, TDOA_ERR_UNSUPPORTED, "too many station pairs for one launch group");
    per_batch = std::max(1, std::min(per_batch, 65535 / std::max(S, P)));
    if (!mine.empty()) {                     // groups of equal size (99 windows at most 85 at a time: 50 + 49, not 85 + 14)
        const int groups = ((int)mine.size() + per_batch - 1) / per_batch;
        per_batch = ((int)mine.size() + groups - 1) / groups;
    }

    // all descriptors, uploaded once; window wi of this rank owns sw[sw_off[wi] .. sw_off[wi+1]) and likewise pw
    std::vector<SWDesc> sw;
    std::vector<PWDesc> pw;
    std::vector<QuadDesc> quads;
    std::vector<size_t> sw_off(mine.size() + 1, 0), pw_off(mine.size() + 1, 0), q_off(mine.size() + 1, 0);
    for (size_t wi = 0; wi < mine.size(); wi++) {
        const int wid = mine[wi];
        const size_t batch_base = sw_off[wi - (wi % (size_t)per_batch)];   // PWDesc indices are relative to the batch
        std::vector<int> slot(S, -1);
        int p = 0;
        for (int i = 0; i < S; i++)
            for (int j = i + 1; j < S; j++, p++) {
                if (!owns(wid, p)) continue;
                for (int s : {i, j})
                    if (slot[s] < 0) {
                        const long long off = (long long)(wid / wpb) * (long long)(ctx->caps[s].n / 3) + (long long)(wid % wpb) * wlen;
                        slot[s] = (int)(sw.size() - batch_base);
                        sw.push_back(SWDesc{ctx->caps[s].dev + 2 * off, (int32_t)corr_len, 0});
                    }
                pw.push_back(PWDesc{slot[i], slot[j], wid * P + p, (int32_t)corr_len});
            }
        sw_off[wi + 1] = sw.size();
        pw_off[wi + 1] = pw.size();
        // segment form: the quad cover of this window's pairs (the same for every window under window-major sharding).
        // Every pair-window of the rank is in exactly one quad; run_fm_batch takes the quads of a batch or none of them.
        // (the greedy cover costs O(P S^2) per quad: beyond kMaxQuadStations stations the segment form stays pair by pair)
        if (pw_off[wi + 1] > pw_off[wi] && S <= kMaxQuadStations) {
            std::vector<int> owned;
            std::vector<std::pair<int, int>> st_pairs;
            p = 0;
            for (int i = 0; i < S; i++)
                for (int j = i + 1; j < S; j++, p++)
                    if (owns(wid, p)) { owned.push_back(p); st_pairs.emplace_back(i, j); }
            auto it = ctx->quad_cache.find(owned);
            if (it == ctx->quad_cache.end()) it = ctx->quad_cache.emplace(owned, build_segment_quads(S, st_pairs)).first;
            const int pw_base = (int)(pw_off[wi] - pw_off[wi - (wi % (size_t)per_batch)]);    // batch-relative pair-window index
            for (const StationQuad &q : it->second) {
                QuadDesc d{slot[q.a], q.b >= 0 ? slot[q.b] : -1, slot[q.c], q.d >= 0 ? slot[q.d] : -1, {-1, -1, -1, -1}};
                for (int o = 0; o < 4; o++)
                    if (q.pair[o] >= 0) d.pw[o] = pw_base + q.pair[o];
                quads.push_back(d);
            }
        }
        q_off[wi + 1] = quads.size();
    }
    const size_t slots = (size_t)W * P;
    if (go && corr_len == 0) {                                 // windows of at most one block: (0, 0.0) everywhere (:708-717)
        if (out_host) std::memset(out_host, 0, sizeof(tdoa_peak) * slots);
        if (out_dev) HIPCHK(ctx, hipMemset(out_dev, 0, sizeof(PeakOut) * slots));
        return TDOA_OK;
    }
    hipStream_t st = ctx->stream;
    const int n_first = (int)std::min<size_t>(per_batch, mine.size());
    if ((rc = ensure(ctx, ctx->peaks, sizeof(PeakOut) * slots))) return rc;
    if ((rc = ensure(ctx, ctx->g_keys, sizeof(unsigned long long) * slots))) return rc;
    if ((rc = ensure(ctx, ctx->g_scales, sizeof(double) * slots))) return rc;
    if ((rc = ensure(ctx, ctx->slot_gain, sizeof(double) * slots))) return rc;
    // TDOA_LAGS_GO: a second copy of the station-window descriptors with the full window length follows the first
    if ((rc = ensure(ctx, ctx->g_sw_desc, sizeof(SWDesc) * std::max<size_t>(2 * sw.size(), 1)))) return rc;
    if ((rc = ensure(ctx, ctx->g_pw_desc, sizeof(PWDesc) * std::max<size_t>(pw.size(), 1)))) return rc;
    if ((rc = ensure(ctx, ctx->g_quad_desc, sizeof(QuadDesc) * std::max<size_t>(quads.size(), 1)))) return rc;
    if (fine_host) {
        if ((rc = ensure(ctx, ctx->fine_raw, 3 * sizeof(float) * slots))) return rc;
        if ((rc = ensure(ctx, ctx->fine, sizeof(FineOut) * slots))) return rc;
    }
    if (n_first && (rc = reserve_fm_batch(ctx, n_first * S, (int)wlen, n_first * P, pl, lag_lo, lag_hi, true)))
        return rc;
    auto *d_sw = static_cast<SWDesc *>(ctx->g_sw_desc.p);
    auto *d_pw = static_cast<PWDesc *>(ctx->g_pw_desc.p);
    auto *d_quads = static_cast<QuadDesc *>(ctx->g_quad_desc.p);
    auto *d_keys = static_cast<unsigned long long *>(ctx->g_keys.p);
    auto *d_scales = static_cast<double *>(ctx->g_scales.p);

    // everything the launches depend on: same key => the captured graph can be replayed as is
    std::vector<uint64_t> key = {(uint64_t)S, (uint64_t)rank, (uint64_t)world, (uint64_t)per_batch, (uint64_t)wlen,
                                 (uint64_t)ctx->prm.max_lag | ((uint64_t)ctx->prm.k1_smooth << 32) | ((uint64_t)(ctx->prm.k1_gate != 0) << 62) |
                                     ((uint64_t)go << 61), (uint64_t)block,
                                 (uint64_t)ctx->force_generic | ((uint64_t)ctx->short_lag << 1) |
                                     ((uint64_t)ctx->segment_form << 3) | ((uint64_t)ctx->xcd_rows << 4) |
                                     ((uint64_t)ctx->segment_quads << 6) |
                                     ((uint64_t)ctx->decimate << 8) | ((uint64_t)ctx->fused_k1 << 9) | ((uint64_t)ctx->k1_once << 11) | ((uint64_t)ctx->seg_pack3 << 12) | ((uint64_t)ctx->dec_cols << 13) | ((uint64_t)ctx->dec_cols_always << 14) | ((uint64_t)ctx->pow2_only << 15) | ((uint64_t)ctx->dec_staged << 7) | ((uint64_t)ctx->stg_cw << 58) | ((uint64_t)ctx->stg_loaders << 54) | ((uint64_t)ctx->stg_blocks << 53) | ((uint64_t)ctx->stg_folded << 50) | ((uint64_t)ctx->stg_folded_always << 49) | ((uint64_t)ctx->small_fused << 52) | ((uint64_t)ctx->small_fused_always << 51) | ((uint64_t)ctx->stg_rows << 28) | ((uint64_t)ctx->stg_bufs << 32) |
                                     ((uint64_t)ctx->memset_nodes << 10) | ((uint64_t)ctx->seg_chunks_override << 16) | ((uint64_t)ctx->xcd_pair_mb << 40),
                                 ctx->alloc_gen, (uint64_t)(fine_host != nullptr), 0};
    std::memcpy(&key.back(), &gate, sizeof(double));
    for (auto &c : ctx->caps) {
        key.push_back((uint64_t)(uintptr_t)c.dev);
        key.push_back((uint64_t)c.n);
    }
    key.push_back(ctx->graph_prof ? 0x100000000ull | ctx->prof_mask : 0ull);      // an instrumented step is a different graph
    const bool graph_ok = ctx->use_graph && !ctx->profiling;
    const bool replay = graph_ok && ctx->graph_exec && key == ctx->graph_key;

    std::vector<SWDesc> sw_full;
    if (!replay) {
        std::vector<double> scales(slots, 1.0 / (4.0 * (double)n * std::sqrt((double)corr_len)));
        HIPCHK(ctx, hipMemcpyAsync(d_scales, scales.data(), sizeof(double) * slots, hipMemcpyHostToDevice, st));
        if (!sw.empty()) {
            HIPCHK(ctx, hipMemcpyAsync(d_sw, sw.data(), sizeof(SWDesc) * sw.size(), hipMemcpyHostToDevice, st));
            if (go) {
                sw_full = sw;
                for (auto &d : sw_full) d.len = (int32_t)wlen;
                HIPCHK(ctx, hipMemcpyAsync(d_sw + sw.size(), sw_full.data(), sizeof(SWDesc) * sw.size(), hipMemcpyHostToDevice, st));
            }
            HIPCHK(ctx, hipMemcpyAsync(d_pw, pw.data(), sizeof(PWDesc) * pw.size(), hipMemcpyHostToDevice, st));
        }
        if (!quads.empty())
            HIPCHK(ctx, hipMemcpyAsync(d_quads, quads.data(), sizeof(QuadDesc) * quads.size(), hipMemcpyHostToDevice, st));

        HIPCHK(ctx, hipStreamSynchronize(st));   // host vectors go out of scope below
    }

    auto enqueue = [&]() -> int {
        ctx->prof_last = -1;
        if (ctx->memset_nodes)               // probe only (TDOA_DEBUG_MEMSET_NODES=1)
            (void)hipMemsetAsync(d_keys, 0, sizeof(unsigned long long) * slots, st);
        else
            hipLaunchKernelGGL(k_zero_u64, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, d_keys, slots);
        float *fine_raw = fine_host ? static_cast<float *>(ctx->fine_raw.p) : nullptr;
        for (size_t w0 = 0; w0 < mine.size(); w0 += per_batch) {
            const int nw = (int)std::min<size_t>(per_batch, mine.size() - w0);
            const int n_sw = (int)(sw_off[w0 + nw] - sw_off[w0]), n_pw = (int)(pw_off[w0 + nw] - pw_off[w0]);
            const int r = run_fm_batch(ctx, d_sw + sw_off[w0], n_sw, (int)wlen, d_pw + pw_off[w0], n_pw, d_keys, pl,
                                       lag_lo, lag_hi, nullptr, 1.0f,
                                       (double)wlen * n_sw, fine_raw, pair_major ? 0 : P, d_quads + q_off[w0],
                                       (int)(q_off[w0 + nw] - q_off[w0]), corr_len >= 2,
                                       go ? d_sw + sw.size() + sw_off[w0] : nullptr, !go);      // every window has wlen samples
            if (r) return r;
        }
        // (every batch of a step takes the same path: same plan, same lag range, same lengths)
        const double *slot_gain = ctx->once_active ? static_cast<const double *>(ctx->slot_gain.p) : nullptr;
        if (fine_raw) ctx->prof_last = -1;
        if (fine_raw)
            hipLaunchKernelGGL(k_decode_fine, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, d_keys, d_scales,
                               fine_raw, static_cast<FineOut *>(ctx->fine.p), gate, (int)slots, slot_gain);
        ProfScope ps(ctx, TDOA_K_PEAK, 32.0 * (double)slots);
        hipLaunchKernelGGL(k_decode_peaks, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, d_keys, d_scales,
                           static_cast<PeakOut *>(ctx->peaks.p), (int)slots, slot_gain);
        return TDOA_OK;
    };

    if (replay) {
        ctx->once_active = ctx->graph_once;
        HIPCHK(ctx, hipGraphLaunch(ctx->graph_exec, st));
    } else if (graph_ok) {
        if (ctx->graph_exec) { (void)hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
        if (ctx->graph) { (void)hipGraphDestroy(ctx->graph); ctx->graph = nullptr; }
        ctx->graph_key.clear();
        HIPCHK(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        clear_graph_marks(ctx);
        ctx->capturing = true;
        rc = enqueue();
        ctx->capturing = false;
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(st, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return fail(ctx, TDOA_ERR_HIP, "hipStreamEndCapture", e);
        ctx->graph = g;
        // the step was captured from ONE stream: it must come out as one dependency chain -- every node but the first has
        // a predecessor (a node without one would replay unordered against the kernels that feed or consume it)
        {
            size_t n_nodes = 0, n_edges = 0, n_roots = 0;
            HIPCHK(ctx, hipGraphGetNodes(g, nullptr, &n_nodes));
            HIPCHK(ctx, hipGraphGetEdges(g, nullptr, nullptr, &n_edges));
            HIPCHK(ctx, hipGraphGetRootNodes(g, nullptr, &n_roots));
            std::vector<hipGraphNode_t> nodes(n_nodes);
            if (n_nodes) HIPCHK(ctx, hipGraphGetNodes(g, nodes.data(), &n_nodes));
            int memsets = 0;
            for (hipGraphNode_t nd : nodes) {
                hipGraphNodeType ty;
                if (hipGraphNodeGetType(nd, &ty) == hipSuccess && ty == hipGraphNodeTypeMemset) memsets++;
            }
            ctx->graph_nodes = (int)n_nodes;
            ctx->graph_edges = (int)n_edges;
            ctx->graph_roots = (int)n_roots;
            ctx->graph_memsets = memsets;
            if (n_nodes && (n_roots != 1 || n_edges + 1 < n_nodes))
                return fail(ctx, TDOA_ERR_STATE, "captured step is not one dependency chain");
            if (memsets && !ctx->memset_nodes) return fail(ctx, TDOA_ERR_STATE, "captured step holds a memset node");
        }
        // graph-mode profiling: an event-record node before the first and after the last kernel of every marked scope
        for (auto &m : ctx->graph_marks) {
            if (!m.before || !m.last || m.last == m.before) { m.e0 = m.e1 = nullptr; continue; }
            HIPCHK(ctx, hipEventCreate(&m.e0));
            HIPCHK(ctx, hipEventCreate(&m.e1));
            auto splice = [&](hipGraphNode_t after, hipEvent_t ev) -> hipError_t {      // after -> [record ev] -> after's successors
                size_t nd = 0;
                hipError_t r = hipGraphNodeGetDependentNodes(after, nullptr, &nd);
                if (r != hipSuccess) return r;
                std::vector<hipGraphNode_t> succ(nd);
                if (nd && (r = hipGraphNodeGetDependentNodes(after, succ.data(), &nd)) != hipSuccess) return r;
                hipGraphNode_t rec = nullptr;
                for (hipGraphNode_t sn : succ)
                    if ((r = hipGraphRemoveDependencies(g, &after, &sn, 1)) != hipSuccess) return r;
                if ((r = hipGraphAddEventRecordNode(&rec, g, &after, 1, ev)) != hipSuccess) return r;
                for (hipGraphNode_t sn : succ)
                    if ((r = hipGraphAddDependencies(g, &rec, &sn, 1)) != hipSuccess) return r;
                return hipSuccess;
            };
            HIPCHK(ctx, splice(m.last, m.e1));       // (the later place first: `before` keeps its successor until then)
            HIPCHK(ctx, splice(m.before, m.e0));
        }
        HIPCHK(ctx, hipGraphInstantiate(&ctx->graph_exec, g, nullptr, nullptr, 0));
        ctx->graph_key = key;
        ctx->graph_once = ctx->once_active;
        HIPCHK(ctx, hipGraphLaunch(ctx->graph_exec, st));
    } else {
        if ((rc = enqueue())) return rc;
    }
    if (out_dev)
        HIPCHK(ctx, hipMemcpyAsync(out_dev, ctx->peaks.p, sizeof(PeakOut) * slots, hipMemcpyDeviceToDevice, st));
    if (out_host)
        HIPCHK(ctx, hipMemcpyAsync(out_host, ctx->peaks.p, sizeof(PeakOut) * slots, hipMemcpyDeviceToHost, st));
    if (fine_host)
        HIPCHK(ctx, hipMemcpyAsync(fine_host, ctx->fine.p, sizeof(FineOut) * slots, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    prof_collect(ctx);
    if (graph_ok && ctx->graph_prof)            // the replay just finished: read the event-record nodes of this step
        for (auto &m : ctx->graph_marks) {
            float ms = 0;
            if (m.e0 && m.e1 && hipEventElapsedTime(&ms, m.e0, m.e1) == hipSuccess) {
                ctx->prof_ms[m.kernel] += ms;
                ctx->prof_launches[m.kernel] += 1;
                ctx->prof_bytes[m.kernel] += m.bytes;
            }
        }
    return TDOA_OK;
}

int tdoa_process(tdoa_ctx *ctx, int rank, int world, tdoa_peak *out_host, void *out_dev)
{
    return process_impl(ctx, rank, world, out_host, out_dev, nullptr, 0.0);
}

int tdoa_process_fine(tdoa_ctx *ctx, int rank, int world, double gate_samples, tdoa_peak *out_host,
                      tdoa_fine_peak *fine_host)
{
    if (!fine_host || !(gate_samples >= 0.0)) return fail(ctx, TDOA_ERR_INVALID, "fine_host is NULL or gate < 0");
    return process_impl(ctx, rank, world, out_host, nullptr, fine_host, gate_samples);
}

int tdoa_fm_xcorr_fine_u8(tdoa_ctx *ctx, const uint8_t *iq1, size_t n1, const uint8_t *iq2, size_t n2, int max_lag,
                          double gate_samples, tdoa_peak *peak, tdoa_fine_peak *fine)
{
    if (!fine || !(gate_samples >= 0.0)) return fail(ctx, TDOA_ERR_INVALID, "fine is NULL or gate < 0");
    return fm_pair(ctx, iq1, n1, iq2, n2, max_lag, peak, nullptr, fine, gate_samples);
}

// fast_analyzer.go:139-155 and collector.go:224 from the exact integer sums, in the reference's expression order
static void finalize_quality(const QualAcc &q, long long n_samples, tdoa_window_quality *out)
{
    std::memset(out, 0, sizeof(*out));
    out->n_samples = n_samples;
    if (n_samples <= 0) return;
    const double n = (double)n_samples;
    const double isum = (double)q.si, qsum = (double)q.sq, isq = (double)q.sii, qsq = (double)q.sqq;   // exact (< 2^53)
    out->i_avg = isum / n;
    out->q_avg = qsum / n;
    out->i_std = std::sqrt((isq / n) - (out->i_avg * out->i_avg));
    out->q_std = std::sqrt((qsq / n) - (out->q_avg * out->q_avg));
    const double pm = std::sqrt(out->i_std * out->i_std + out->q_std * out->q_std);
    out->power_level = pm <= 1e-10 ? -100.0 : 20 * std::log10(pm);
    // sum of (b-127.5)^2 over both components: every term is a multiple of 1/4, so the f64 running sum is exact
    const double sumsq = (isq + qsq) - 255.0 * (isum + qsum) + 2.0 * n * 16256.25;
    out->mean_power = sumsq / n;
    out->i_min = (int32_t)q.imin; out->i_max = (int32_t)q.imax; out->q_min = (int32_t)q.qmin; out->q_max = (int32_t)q.qmax;
    out->has_clipping = (q.imin == 0 || q.imax == 255 || q.qmin == 0 || q.qmax == 255) ? 1 : 0;
    out->has_overload = (out->i_std < 2 || out->q_std < 2) ? 1 : 0;
}

static int run_quality(tdoa_ctx *ctx, const SWDesc *d_sw, int n_sw, long long max_len, std::vector<QualAcc> *host)
{
    int rc;
    if ((rc = ensure(ctx, ctx->qual, sizeof(QualAcc) * (size_t)std::max(n_sw, 1)))) return rc;
    auto *acc = static_cast<QualAcc *>(ctx->qual.p);
    host->resize(n_sw);
    if (n_sw == 0) return TDOA_OK;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_quality_init, dim3((n_sw + 255) / 256), dim3(256), 0, st, acc, n_sw);
    const unsigned chunks = (unsigned)std::max<long long>(1, (max_len + kQualChunk - 1) / kQualChunk);
    hipLaunchKernelGGL(k_window_quality, dim3(chunks, n_sw), dim3(256), 0, st, d_sw, acc);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(host->data(), acc, sizeof(QualAcc) * (size_t)n_sw, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return TDOA_OK;
}

int tdoa_window_quality_all(tdoa_ctx *ctx, int rank, int world, tdoa_window_quality *out_host)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (world < 1 || rank < 0 || rank >= world || !out_host) return fail(ctx, TDOA_ERR_INVALID, "bad rank/world/out");
    long long block, wlen;
    int wpb;
    if ((rc = window_geometry(ctx, &block, &wlen, &wpb))) return fail(ctx, rc, "captures missing or too small");
    const int S = (int)ctx->caps.size();
    const int W = 3 * wpb;
    std::vector<int> mine;
    for (int w = rank; w < W; w += world) mine.push_back(w);
    std::vector<SWDesc> sw(mine.size() * (size_t)S);
    for (size_t wi = 0; wi < mine.size(); wi++) {
        const int wid = mine[wi];
        for (int s = 0; s < S; s++) {
            const long long off = (long long)(wid / wpb) * (long long)(ctx->caps[s].n / 3) + (long long)(wid % wpb) * wlen;
            sw[wi * S + s] = SWDesc{ctx->caps[s].dev + 2 * off, (int32_t)wlen, 0};
        }
    }
    if ((rc = ensure(ctx, ctx->g_sw_desc, sizeof(SWDesc) * std::max<size_t>(sw.size(), 1)))) return rc;
    ctx->graph_key.clear();     // the descriptor buffer of a captured tdoa_process graph is being rewritten
    if (!sw.empty())
        HIPCHK(ctx, hipMemcpyAsync(ctx->g_sw_desc.p, sw.data(), sizeof(SWDesc) * sw.size(), hipMemcpyHostToDevice, ctx->stream));
    std::vector<QualAcc> acc;
    if ((rc = run_quality(ctx, static_cast<const SWDesc *>(ctx->g_sw_desc.p), (int)sw.size(), wlen, &acc))) return rc;
    std::memset(out_host, 0, sizeof(tdoa_window_quality) * (size_t)W * S);
    for (size_t wi = 0; wi < mine.size(); wi++)
        for (int s = 0; s < S; s++) finalize_quality(acc[wi * S + s], wlen, &out_host[(size_t)mine[wi] * S + s]);
    return TDOA_OK;
}

int tdoa_window_quality_u8(tdoa_ctx *ctx, const uint8_t *iq, size_t n_samples, tdoa_window_quality *out)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (!out || (!iq && n_samples) || n_samples > 0x7fffffff) return fail(ctx, TDOA_ERR_INVALID, "bad argument");
    if (n_samples == 0) { finalize_quality(QualAcc{}, 0, out); return TDOA_OK; }
    if ((rc = ensure(ctx, ctx->scratch_a, 2 * n_samples + 16))) return rc;
    if ((rc = ensure(ctx, ctx->sw_desc, sizeof(SWDesc)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->scratch_a.p, iq, 2 * n_samples, hipMemcpyHostToDevice, ctx->stream));
    const SWDesc d{static_cast<const uint8_t *>(ctx->scratch_a.p), (int32_t)n_samples, 0};
    HIPCHK(ctx, hipMemcpyAsync(ctx->sw_desc.p, &d, sizeof(d), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // d is a stack object
    std::vector<QualAcc> acc;
    if ((rc = run_quality(ctx, static_cast<const SWDesc *>(ctx->sw_desc.p), 1, (long long)n_samples, &acc))) return rc;
    finalize_quality(acc[0], (long long)n_samples, out);
    return TDOA_OK;
}

int tdoa_process_u8(tdoa_ctx *ctx, const uint8_t *const *station_iq, const size_t *n_samples, int n_stations,
                    tdoa_peak *out)
{
    if (!ctx || !station_iq || !n_samples || n_stations < 2 || !out) return fail(ctx, TDOA_ERR_INVALID, "bad argument");
    int rc = TDOA_OK;
    if ((int)ctx->caps.size() != n_stations) rc = tdoa_capture_clear(ctx);   // else the uploads reuse the buffers
    for (int s = 0; s < n_stations && !rc; s++) rc = tdoa_capture_upload(ctx, s, station_iq[s], n_samples[s]);
    if (rc) return rc;
    return tdoa_process(ctx, 0, 1, out, nullptr);
}

int tdoa_fm_xcorr_u8(tdoa_ctx *ctx, const uint8_t *iq1, size_t n1, const uint8_t *iq2, size_t n2, int max_lag,
                     tdoa_peak *peak)
{
    if (!peak) return fail(ctx, TDOA_ERR_INVALID, "peak is NULL");
    return fm_pair(ctx, iq1, n1, iq2, n2, max_lag, peak, nullptr);
}

int tdoa_fm_xcorr_lags_u8(tdoa_ctx *ctx, const uint8_t *iq1, size_t n1, const uint8_t *iq2, size_t n2, int max_lag,
                          double *lags_out)
{
    if (!lags_out) return fail(ctx, TDOA_ERR_INVALID, "lags_out is NULL");
    return fm_pair(ctx, iq1, n1, iq2, n2, max_lag, nullptr, lags_out);
}

int tdoa_fm_preprocess_u8(tdoa_ctx *ctx, const uint8_t *iq, size_t n, float *out_f32, tdoa_fm_stats *stats)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (!iq || n == 0 || n > 0x3fffffff) return fail(ctx, TDOA_ERR_INVALID, "bad argument");
    if ((rc = ensure(ctx, ctx->scratch_a, 2 * n + 16))) return rc;
    if ((rc = ensure(ctx, ctx->scratch_b, sizeof(float) * n))) return rc;
    if ((rc = ensure(ctx, ctx->sw_desc, sizeof(SWDesc)))) return rc;
    const int pieces = (int)((n + kDemodPiece - 1) / kDemodPiece);
    const long long code_stride = ((long long)n + 15) / 8 * 8;
    if ((rc = ensure(ctx, ctx->partials, sizeof(StatsPartial)))) return rc;
    if ((rc = ensure(ctx, ctx->stats, sizeof(FmStats)))) return rc;
    if ((rc = ensure(ctx, ctx->codes, sizeof(int) * (size_t)code_stride))) return rc;
    hipStream_t st = ctx->stream;
    SWDesc sw = {static_cast<uint8_t *>(ctx->scratch_a.p), (int32_t)n, 0};
    HIPCHK(ctx, hipMemcpyAsync(ctx->scratch_a.p, iq, 2 * n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->sw_desc.p, &sw, sizeof(sw), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    auto *d_sw = static_cast<SWDesc *>(ctx->sw_desc.p);
    if (ctx->prm.k1_smooth > 1 && (rc = ensure(ctx, ctx->codes_lp, sizeof(int) * (size_t)code_stride))) return rc;
    if (ctx->prm.k1_gate && (rc = ensure(ctx, ctx->k1_power, sizeof(unsigned long long)))) return rc;
    // no output array wanted and no option that needs the codes: the reduce-only pass of the fused path
    const bool stats_only = !out_f32 && ctx->prm.k1_smooth <= 1 && !ctx->prm.k1_gate;
    int *codes_used = launch_k1(ctx, st, d_sw, 1, (int)n, pieces, code_stride, !stats_only);
    if (codes_used)
        hipLaunchKernelGGL(k_fm_dump, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, st, d_sw,
                           codes_used, static_cast<FmStats *>(ctx->stats.p),
                           static_cast<float *>(ctx->scratch_b.p));
    HIPCHK(ctx, hipGetLastError());
    if (out_f32) HIPCHK(ctx, hipMemcpyAsync(out_f32, ctx->scratch_b.p, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    if (stats) HIPCHK(ctx, hipMemcpyAsync(stats, ctx->stats.p, sizeof(FmStats), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return TDOA_OK;
}

// ===========================================================================
// measurement
// ===========================================================================
int tdoa_debug_force_generic(tdoa_ctx *ctx, int on)
{
    if (!ctx) return TDOA_ERR_INVALID;
    ctx->force_generic = on != 0;
    return TDOA_OK;
}

int tdoa_debug_segment_quads(int n_stations, const int32_t *pairs, int n_pairs, int32_t *quads_out, int max_quads)
{
    if (n_stations < 2 || n_stations > kMaxQuadStations || n_pairs < 0 || (n_pairs && !pairs) || max_quads < 0 || (max_quads && !quads_out))
        return -TDOA_ERR_INVALID;
    std::vector<std::pair<int, int>> pr;
    for (int i = 0; i < n_pairs; i++) {
        const int a = pairs[2 * i], c = pairs[2 * i + 1];
        if (a < 0 || c < 0 || a >= n_stations || c >= n_stations || a == c) return -TDOA_ERR_INVALID;
        pr.emplace_back(a, c);
    }
    const std::vector<StationQuad> q = build_segment_quads(n_stations, pr);
    if ((int)q.size() > max_quads) return -TDOA_ERR_INVALID;
    for (size_t i = 0; i < q.size(); i++) {
        int32_t *o = quads_out + 8 * i;
        o[0] = q[i].a; o[1] = q[i].b; o[2] = q[i].c; o[3] = q[i].d;
        for (int k = 0; k < 4; k++) o[4 + k] = q[i].pair[k];
    }
    return (int)q.size();
}

int tdoa_debug_staged_groups(int n_stations, int max_pairs, uint32_t *masks_out, int32_t *counts_out, uint8_t *pairs_out, int max_groups)
{
    if (n_stations < 2 || n_stations > kStgMaxStations || max_pairs < 1 || max_pairs > kStgMaxWaves || max_groups < 0 ||
        (max_groups && (!masks_out || !counts_out || !pairs_out)))
        return -TDOA_ERR_INVALID;
    const std::vector<StgGroup> g = build_stg_groups(n_stations, max_pairs, max_pairs == kStgMaxWaves);      // (16: the folded form's table)
    if ((int)g.size() > max_groups) return -TDOA_ERR_INVALID;
    for (size_t i = 0; i < g.size(); i++) {
        masks_out[i] = g[i].mask;
        counts_out[i] = g[i].n;
        std::memcpy(pairs_out + 16 * i, g[i].pair, 16);
    }
    return (int)g.size();
}

int tdoa_debug_graph_info(tdoa_ctx *ctx, int32_t info[4], const char *dot_path)
{
    if (!ctx || !info) return TDOA_ERR_INVALID;
    if (!ctx->graph) return fail(ctx, TDOA_ERR_STATE, "no captured step");
    info[0] = ctx->graph_nodes;
    info[1] = ctx->graph_edges;
    info[2] = ctx->graph_roots;
    info[3] = ctx->graph_memsets;
    if (dot_path && dot_path[0]) {
        HIPCHK(ctx, hipGraphDebugDotPrint(ctx->graph, dot_path, hipGraphDebugDotFlagsVerbose));
        // the parameters of the memset nodes (probe builds only), read back from the graph itself: <dot_path>.memsets
        size_t n_nodes = 0;
        HIPCHK(ctx, hipGraphGetNodes(ctx->graph, nullptr, &n_nodes));
        std::vector<hipGraphNode_t> nodes(n_nodes);
        if (n_nodes) HIPCHK(ctx, hipGraphGetNodes(ctx->graph, nodes.data(), &n_nodes));
        const std::string mp = std::string(dot_path) + ".memsets";
        if (FILE *f = std::fopen(mp.c_str(), "w")) {
            for (hipGraphNode_t nd : nodes) {
                hipGraphNodeType ty;
                hipMemsetParams mpz;
                if (hipGraphNodeGetType(nd, &ty) == hipSuccess && ty == hipGraphNodeTypeMemset &&
                    hipGraphMemsetNodeGetParams(nd, &mpz) == hipSuccess)
                    std::fprintf(f, "memset node: dst %p elementSize %u width %zu height %zu pitch %zu value %u\n", mpz.dst,
                                 mpz.elementSize, mpz.width, mpz.height, mpz.pitch, mpz.value);
            }
            std::fclose(f);
        }
    }
    return TDOA_OK;
}

int tdoa_debug_flags(tdoa_ctx *ctx, unsigned flags)
{
    if (!ctx) return TDOA_ERR_INVALID;
    ctx->force_generic = (flags & TDOA_DEBUG_GENERIC_KERNELS) != 0;
    ctx->short_lag = !(flags & TDOA_DEBUG_NO_SHORT_LAG);
    ctx->segment_form = !(flags & TDOA_DEBUG_NO_SEGMENT_FORM);
    ctx->xcd_rows = !(flags & TDOA_DEBUG_NO_XCD_ROWS);
    ctx->segment_quads = !(flags & TDOA_DEBUG_NO_SEGMENT_QUADS);
    ctx->decimate = !(flags & TDOA_DEBUG_NO_DECIMATE);
    ctx->fused_k1 = !(flags & TDOA_DEBUG_NO_FUSED_K1);
    ctx->k1_once = !(flags & TDOA_DEBUG_NO_K1_ONCE);
    ctx->seg_pack3 = !(flags & TDOA_DEBUG_NO_SEG_PACK3);
    ctx->dec_cols = !(flags & TDOA_DEBUG_NO_DEC_COLS);
    ctx->dec_cols_always = (flags & TDOA_DEBUG_DEC_COLS_ALWAYS) != 0;
    ctx->pow2_only = (flags & TDOA_DEBUG_POW2_ONLY) != 0;
    ctx->dec_staged = !(flags & TDOA_DEBUG_NO_DEC_STAGED);
    ctx->small_fused = !(flags & TDOA_DEBUG_NO_SMALL_FUSED);
    ctx->small_fused_always = (flags & TDOA_DEBUG_SMALL_FUSED_ALWAYS) != 0;
    return TDOA_OK;
}

int tdoa_debug_last_k1(tdoa_ctx *ctx, int sw_index, tdoa_fm_stats *stats, int32_t *single_look)
{
    int rc;
    if ((rc = check_ctx(ctx))) return rc;
    if (sw_index < 0 || (size_t)(sw_index + 1) * sizeof(FmStats) > ctx->stats.cap) return fail(ctx, TDOA_ERR_INVALID, "no such station-window");
    if (stats) HIPCHK(ctx, hipMemcpy(stats, static_cast<FmStats *>(ctx->stats.p) + sw_index, sizeof(FmStats), hipMemcpyDeviceToHost));
    if (single_look) *single_look = ctx->once_active ? 1 : 0;
    return TDOA_OK;
}

int tdoa_profile_enable(tdoa_ctx *ctx, int on)
{
    if (!ctx) return TDOA_ERR_INVALID;
    ctx->profiling = on == 1;
    ctx->graph_prof = on == 2;
    return TDOA_OK;
}

int tdoa_profile_select(tdoa_ctx *ctx, unsigned int scope_mask)
{
    if (!ctx) return TDOA_ERR_INVALID;
    ctx->prof_mask = scope_mask;
    return TDOA_OK;
}

int tdoa_profile_reset(tdoa_ctx *ctx)
{
    if (!ctx) return TDOA_ERR_INVALID;
    prof_collect(ctx);
    for (int k = 0; k < TDOA_K_COUNT; k++) {
        ctx->prof_ms[k] = 0;
        ctx->prof_launches[k] = 0;
        ctx->prof_bytes[k] = 0;
    }
    return TDOA_OK;
}

int tdoa_profile_get(tdoa_ctx *ctx, int kernel, double *total_ms, int64_t *launches, double *algorithmic_bytes)
{
    if (!ctx || kernel < 0 || kernel >= TDOA_K_COUNT) return TDOA_ERR_INVALID;
    if (total_ms) *total_ms = ctx->prof_ms[kernel];
    if (launches) *launches = ctx->prof_launches[kernel];
    if (algorithmic_bytes) *algorithmic_bytes = ctx->prof_bytes[kernel];
    return TDOA_OK;
}

// ===========================================================================
// downstream geodesy / solver (host)
// ===========================================================================
void tdoa_latlon_to_ecef(double lat, double lon, double elev, double xyz[3]) { geo::latlon_to_ecef(lat, lon, elev, xyz); }
void tdoa_ecef_to_latlon(double x, double y, double z, double lle[3]) { geo::ecef_to_latlon(x, y, z, lle); }
int tdoa_solve_3station(const double stations_lle[9], const double *range_diff, double out_lle[3], int *iterations)
{
    if (!stations_lle || !range_diff || !out_lle) return TDOA_ERR_INVALID;
    return geo::solve_3station(stations_lle, range_diff, out_lle, iterations) ? TDOA_ERR_SINGULAR : TDOA_OK;
}

int tdoa_solve_nstation(const double *stations_lle, int n_stations, const double *range_diff, const double *weights,
                        int solve_z, double out_lle[3], int *iterations)
{
    if (!stations_lle || !range_diff || !out_lle) return TDOA_ERR_INVALID;
    const int rc = geo::solve_nstation(stations_lle, n_stations, range_diff, weights, solve_z, 10, 0.5, 1.0, out_lle,
                                       iterations);
    return rc == 0 ? TDOA_OK : (rc == -2 ? TDOA_ERR_UNSUPPORTED : rc == -3 ? TDOA_ERR_INVALID : TDOA_ERR_SINGULAR);
}

int tdoa_solve_surface(const double *stations_lle, int n_stations, const double *range_diff, const double *weights,
                       double height_m, double out_lle[3], int *iterations)
{
    if (!stations_lle || !range_diff || !out_lle || !std::isfinite(height_m)) return TDOA_ERR_INVALID;
    const int rc = geo::solve_surface(stations_lle, n_stations, range_diff, weights, height_m, 20, 1.0, out_lle, iterations);
    return rc == 0 ? TDOA_OK : (rc == -2 ? TDOA_ERR_UNSUPPORTED : rc == -3 ? TDOA_ERR_INVALID : TDOA_ERR_SINGULAR);
}

#ifdef TDOA_STG_TIMING
// measurement build only: read and clear the staged walk's wave-cycle counters (dec_staged.hpp)
int tdoa_debug_stg_prof(unsigned long long *out8)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(tdoa::g_stg_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(tdoa::g_stg_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
}  // extern "C"

#include "exact_reference_api.inc"
