/*
 * tdoa_mi355x.h -- C ABI of the MI355X-native TDOA correlation processor.
 *
 * Drop-in boundary for the correlation hot path of KX0U-Jim/tdoa-geolocation
 * (processor.go / simple_corr.go / fast_analyzer.go).  The reference has no FFI
 * today: every entry point below replaces a Go function that a cgo shim would
 * forward to (INTEGRATION.md shows the shim).  Citations are file:line in the
 * reference.  Plain pointers and sizes only; no C++/torch types.
 *
 * Data layouts (identical to the reference's):
 *   complex64 signal  = n x {float re, float im} interleaved   (Go []complex64)
 *   IQ capture bytes  = n x {uint8 I, uint8 Q} centred at 127.5 (.dat files,
 *                       librtlsdr-2freq/src/rtl_sdr.c:103-146), three equal
 *                       blocks [f1 | f2 | f1]
 *   delay             = samples, relative to the shorter input as template
 *
 * Ownership: the caller owns every pointer it passes; the library copies what
 * it needs during the call and keeps no host pointer afterwards (cgo rule).
 * A context is single-caller; use one context per GPU.  Every function
 * returns a status code (0 = TDOA_OK) and never aborts the host process.
 */
#ifndef TDOA_MI355X_H
#define TDOA_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDOA_ABI_VERSION 4

typedef struct tdoa_ctx tdoa_ctx;

enum {
    TDOA_OK = 0,
    TDOA_ERR_INVALID = 1,     /* bad argument                                  */
    TDOA_ERR_NO_DEVICE = 2,   /* no usable HIP device (no CPU fallback exists) */
    TDOA_ERR_HIP = 3,         /* a HIP runtime call failed (tdoa_last_error)   */
    TDOA_ERR_NOMEM = 4,
    TDOA_ERR_UNSUPPORTED = 5, /* size outside the supported range              */
    TDOA_ERR_STATE = 6,       /* call order (e.g. process before upload)       */
    TDOA_ERR_SINGULAR = 7     /* solver: singular Jacobian (processor.go:997-999) */
};

/* The constants the reference hard-codes, as parameters. */
typedef struct {
    double  sample_rate;     /* 2e6     processor.go:440,488,821,841            */
    int32_t max_lag;         /* 20000   processor.go:633                        */
    int32_t corr_block;      /* 1000    processor.go:682                        */
    double  weak_threshold;  /* 0.001   processor.go:476                        */
    int64_t window_len;      /* 2000000 processor.go:772 (testChunkSize)        */
    int32_t device;          /* HIP device ordinal                              */
    int32_t windows_per_batch; /* station-windows processed per launch group; 0 = auto */
    int32_t k1_smooth;       /* 0 (default): none.  W > 1: centred moving average of W samples (half-window W/2, edges
                                truncated: applyLowPassFilter, processor.go:270-296) on the discriminator output before
                                it is normalised -- the prebuilt reference binary's strong-signal chain is discriminator
                                -> removeDCBias -> applyLowPassFilter(10) -> normalizeSignal */
    int32_t k1_gate;         /* 0 (default): every window through the discriminator (the north-star pipeline).
                                1: the prebuilt reference binary's power gate (preprocessSignal):
                                a station-window whose mean power, mean |(b - 127.5)/127.5|^2, is <= 0.01 takes
                                envelope |x| -> removeDCBias -> normalizeSignal instead of the discriminator chain (and is
                                not smoothed).  The binary's third branch (<= 0.001: band-passed complex samples) has no
                                mode-B counterpart; such windows take the envelope too -- tdoa_window_quality_all
                                reports every window's mean power */
    int32_t lag_mode;        /* which lags mode B searches (K5):
                                TDOA_LAGS_SIGNED (0, default): -max_lag < lag < max_lag, template = the first input, peak =
                                largest |corr|, ties -> smaller |lag|, then the positive lag;
                                TDOA_LAGS_GO (1): the lag set and peak rule of timeDomainCorrelation (processor.go:646-736):
                                template = the SHORTER input (ties: the first, :650-655), only its first B corr_block samples
                                count (B = number of block starts 0, corr_block, ... below Lt - corr_block, :691; B = 0 gives
                                (0, 0.0)), lags 0 <= lag < max(1, min(max_lag, Ls - Lt)) (:668-678), first strictly larger
                                |corr| wins = the lowest lag among equals (:722-725), corr scaled by 1/sqrt(B corr_block)
                                (:719-720).  Windows of tdoa_process have equal lengths: lag 0 only, as in the reference's
                                own call pattern.  Not available with the sub-sample refinement (TDOA_ERR_UNSUPPORTED). */
    int32_t reserved;        /* 0 */
} tdoa_params;

enum { TDOA_LAGS_SIGNED = 0, TDOA_LAGS_GO = 1 };

/* One correlation peak.  lag > 0: the second station of the pair lags the first. */
typedef struct {
    int32_t lag;        /* samples                                           */
    float   abs_corr;   /* |corr| as float (sort/weight key)                 */
    double  corr;       /* signed, reference scale sum/sqrt(n_template)      */
} tdoa_peak;

/* Exact statistics of one FM-discriminated window (mode B preprocessing).  The discriminator output is held as an
 * integer phase code in units of pi/2^23 (-2^23 < code <= 2^23: exact in a float). */
typedef struct {
    int64_t  s1;              /* sum of the codes                                              */
    uint64_t s2_lo, s2_hi;    /* sum of code^2 (128 bits: code^2 < 2^46 per sample)             */
    float    mean, scale;     /* f32(s1 / n); f32(1 / sqrt(variance)), 1 if the variance is 0   */
} tdoa_fm_stats;

/* ---- lifecycle ------------------------------------------------------------ */
void        tdoa_default_params(tdoa_params *p);
int         tdoa_create(const tdoa_params *p, tdoa_ctx **out);
void        tdoa_destroy(tdoa_ctx *ctx);
const char *tdoa_strerror(int status);
const char *tdoa_last_error(const tdoa_ctx *ctx);   /* detail of the last failure */
int         tdoa_abi_version(void);
int         tdoa_device_count(void);

/* ---- mode A: the reference's executed call surface -------------------------
 * Bit-faithful GPU evaluation of the Go functions (same filters, same
 * accumulation order where the Go order is observable).                      */

/* processor.go:166-205 loadIQData conversion: (float32(b)-127.5)/127.5 */
int tdoa_load_iq_u8(tdoa_ctx *ctx, const uint8_t *raw, size_t n_samples, float *out_c64);

/* processor.go:469-499 preprocessSignal (power gate, filter chains, normalise);
 * *weak_chain receives 1 if enhanceWeakSignal (:437) was taken. */
int tdoa_preprocess_c64(tdoa_ctx *ctx, const float *sig, size_t n, float *out_c64, int *weak_chain);

/* processor.go:646-736 timeDomainCorrelation(signal1, signal2, maxLag) -> (delay, corr) */
int tdoa_time_domain_correlation_c64(tdoa_ctx *ctx, const float *s1, size_t n1,
                                     const float *s2, size_t n2, int max_lag,
                                     int32_t *delay, double *corr);

/* processor.go:619-643 crossCorrelate(signal1, signal2) -> (delay, corr):
 * the station-pair call surface (callers :818, :838, correlation_sanity.go:50,55).
 * Empty input returns TDOA_OK with (0, 0.0) like the reference (:622-625). */
int tdoa_cross_correlate_c64(tdoa_ctx *ctx, const float *s1, size_t n1,
                             const float *s2, size_t n2,
                             int32_t *delay, double *corr);
/* the reference's pair loop (processor.go:816-830, :836-850) in one call: every signal goes through
 * preprocessSignal ONCE (crossCorrelate re-does it for every pair a station is in, :629-630), then each pair
 * i < j is correlated; delay[p], corr[p] in the reference's pair order, bit-identical to per-pair calls */
int tdoa_cross_correlate_batch_c64(tdoa_ctx *ctx, const float *const *signals, const size_t *n_samples, int n_signals,
                                   int32_t *delay, double *corr);

/* simple_corr.go:83-160 simpleCorrelate -> (delay, float32 corr) */
int tdoa_simple_correlate_c64(tdoa_ctx *ctx, const float *s1, size_t n1,
                              const float *s2, size_t n2,
                              int32_t *delay, float *corr);

/* fast_analyzer.go:163-227 fastSNRCalculation(samples, totalSamples) -> dB */
int tdoa_fast_snr_u8(tdoa_ctx *ctx, const uint8_t *samples, int total_samples, double *snr_db);

/* fast_analyzer.go:15-24 FastAnalysis / :113-161 fastAnalyzeSamples */
typedef struct {
    int32_t total_samples;
    int32_t has_clipping;     /* any I or Q byte equal to 0 or 255 (:154) */
    int32_t has_overload;     /* I or Q standard deviation below 2 (:155)  */
    int32_t reserved;
    double  i_avg, q_avg, i_std, q_std;
    double  snr_estimate;     /* dB, fastSNRCalculation                    */
    double  power_level;      /* 20*log10(sqrt(i_std^2+q_std^2)), floor -100 (:146-151) */
} tdoa_fast_analysis;
int tdoa_fast_analyze_u8(tdoa_ctx *ctx, const uint8_t *samples, int total_samples, tdoa_fast_analysis *out);
/* fast_analyzer.go:53-111 fastAnalyzeDualFrequencyFile on capture bytes in memory: the first 32768
 * samples of blocks 1 and 3 -> ref, of block 2 -> tgt; TDOA_ERR_INVALID if the capture has < 3 samples */
int tdoa_fast_analyze_capture_u8(tdoa_ctx *ctx, const uint8_t *raw, size_t n_bytes,
                                 tdoa_fast_analysis *ref, tdoa_fast_analysis *tgt);

/* ---- mode B: the north-star pipeline ---------------------------------------
 * u8 IQ -> FM discriminator -> Stockham FFT -> conj-multiply -> inverse FFT
 * -> argmax, batched over (station, window) and (pair, window).
 *
 * A capture is 3 blocks of n = floor(total/3) samples (each capture its own
 * n, as processor.go:214 does per file).  Windows of window_len samples tile
 * each block from its start; the window grid comes from the shortest capture:
 * windows_per_block = max(1, n_min / window_len) (a block shorter than
 * window_len is one window of n_min samples).
 * Window id wid = block * windows_per_block + w; block 1 is the target
 * frequency, blocks 0 and 2 the reference frequency (processor.go:211-233).
 * Pairs are ordered i<j as in processor.go:816-850.                          */

/* copy one station's capture into HBM (ctx-owned).  A few host threads stage the bytes through pinned
 * buffers onto their own copy streams (about 40 GB/s from pageable memory against 15 GB/s for one
 * hipMemcpy); a station's previous buffer is reused when the new capture fits, so repeated
 * uploads of equal-sized captures keep their device addresses (TDOA_UPLOAD_THREADS overrides 4). */
int tdoa_capture_upload(tdoa_ctx *ctx, int station, const uint8_t *iq, size_t n_samples);
/* sharded ingest: upload only samples [first_sample, first_sample + n_samples) of a capture of total_samples -- a rank
 * of a multi-GPU job calls it once per run of windows it owns (tdoa_process(rank, world) reads nothing else), so the
 * host-to-device traffic of a G-rank job is 1/G of the capture per rank instead of all of it */
int tdoa_capture_upload_range(tdoa_ctx *ctx, int station, size_t total_samples, size_t first_sample, const uint8_t *iq,
                              size_t n_samples);
/* the same from a .dat file (collector.go:61 naming, raw u8 I,Q; any size, > 1 GiB safe): the threads
 * pread their chunks straight into the pinned buffers;
 * *n_samples (may be NULL) receives size/2 like processor.go:182 */
int tdoa_capture_upload_file(tdoa_ctx *ctx, int station, const char *path, size_t *n_samples);
/* or attach a buffer that is already in this device's memory (not copied, not freed) */
int tdoa_capture_attach_device(tdoa_ctx *ctx, int station, const void *dev_iq, size_t n_samples);
int tdoa_capture_clear(tdoa_ctx *ctx);
/* synthesise a simulator.go-style capture (tones + uniform noise, carrier-phase delay,
 * [ref | target | ref] blocks of block_samples; simulator.go:100-161) directly in HBM */
int tdoa_synth_capture(tdoa_ctx *ctx, int station, size_t block_samples, double ref_freq, double tgt_freq,
                       double noise_level, const double station_lle[3], const double tx_lle[3],
                       double tx_power, uint64_t seed);
/* the same for weak_signal_simulator.go (BASELINE config 3; weak_signal_simulator.go:89-257): weak reference blocks
 * (Gaussian noise 0.8 A, impulses p = 1e-3 at 5 A, phase drift 0.05 rad/s, DC 0.1 A), strong target block */
int tdoa_synth_weak_capture(tdoa_ctx *ctx, int station, size_t block_samples, double ref_freq, double tgt_freq,
                            const double station_lle[3], const double tx_lle[3], double ref_power, double tgt_power,
                            uint64_t seed);
/* read back part of a capture that lives in HBM */
int tdoa_capture_download(tdoa_ctx *ctx, int station, size_t first_sample, size_t n_samples, uint8_t *out);

int tdoa_num_windows(const tdoa_ctx *ctx, int *windows_per_block, int *n_windows_total);
int tdoa_num_pairs(const tdoa_ctx *ctx);

/* Correlate every pair on every window wid with wid % world == rank (with fewer windows than
 * ranks: every unit u = wid * n_pairs + pair with u % world == rank, so no rank idles).
 * out_host (may be NULL): [n_windows_total][n_pairs] tdoa_peak, entries of
 *   windows owned by other ranks are zero-filled;
 * out_dev (may be NULL): same array in device memory (for an RCCL all-gather
 *   of the per-pair peaks without a host round trip).
 * Search ranges below 4095 lags (params.max_lag) take a shorter inverse that never
 * materialises the full correlation array: about 15 % faster at max_lag <= 2047. */
int tdoa_process(tdoa_ctx *ctx, int rank, int world, tdoa_peak *out_host, void *out_dev);

/* upload + process in one call (host pointers in, host peaks out) */
int tdoa_process_u8(tdoa_ctx *ctx, const uint8_t *const *station_iq, const size_t *n_samples,
                    int n_stations, tdoa_peak *out);

/* single pair of raw IQ windows -> peak (lags -(max_lag-1) .. max_lag-1) */
int tdoa_fm_xcorr_u8(tdoa_ctx *ctx, const uint8_t *iq1, size_t n1, const uint8_t *iq2, size_t n2,
                     int max_lag, tdoa_peak *peak);

/* Sub-sample refinement and physical-plausibility gate (SURVEY section 8 row (f)-4;
 * PROJECT_NOTES.md:29-32: one sample is 500 ns = 150 m of range, max |TDOA| of the
 * deployed stations about 57 us = 114 samples).  Around the integer peak lag d:
 * y_q = s*c[d-1+q], s = sign(c[d]); frac = vertex of the parabola through the three
 * points, (y_m - y_p) / (2 (y_m - 2 y_0 + y_p)) when that curvature is negative, clamped to
 * [-1/2, 1/2], else 0.  The integer peak of tdoa_process is untouched (index parity). */
typedef struct {
    double  delay;      /* lag + frac, samples                               */
    float   frac;       /* vertex offset in [-1/2, 1/2]                      */
    float   y[3];       /* s*c[lag-1], s*c[lag], s*c[lag+1], reference scale */
    int32_t plausible;  /* |delay| <= gate_samples                           */
    int32_t reserved;
} tdoa_fine_peak;
/* tdoa_process + refinement: fine_host [n_windows_total][n_pairs] (zero delay, plausible
 * for windows of other ranks), out_host may be NULL */
int tdoa_process_fine(tdoa_ctx *ctx, int rank, int world, double gate_samples, tdoa_peak *out_host,
                      tdoa_fine_peak *fine_host);
int tdoa_fm_xcorr_fine_u8(tdoa_ctx *ctx, const uint8_t *iq1, size_t n1, const uint8_t *iq2, size_t n2,
                          int max_lag, double gate_samples, tdoa_peak *peak /* may be NULL */,
                          tdoa_fine_peak *fine);

/* Capture-quality statistics of every (window, station) in one streaming pass over the bytes
 * in HBM: the byte statistics of fastAnalyzeSamples
 * (fast_analyzer.go:117-155) and the block power of validateDataFile (collector.go:219-224).
 * Exact integer sums on the GPU, so every field equals the reference's float64 result. */
typedef struct {
    int64_t n_samples;
    double  i_avg, q_avg, i_std, q_std;   /* fast_analyzer.go:139-142                     */
    double  power_level;                  /* dB, floor -100 (:146-151)                    */
    double  mean_power;                   /* mean of (I-127.5)^2+(Q-127.5)^2 (collector.go:219-224) */
    int32_t i_min, i_max, q_min, q_max;
    int32_t has_clipping, has_overload;   /* :154-155                                     */
} tdoa_window_quality;
/* out_host [n_windows_total][n_stations]; windows of other ranks are zero-filled */
int tdoa_window_quality_all(tdoa_ctx *ctx, int rank, int world, tdoa_window_quality *out_host);
/* the same statistics of one host buffer of raw IQ */
int tdoa_window_quality_u8(tdoa_ctx *ctx, const uint8_t *iq, size_t n_samples, tdoa_window_quality *out);

/* inspection hooks used by the parity tests.  tdoa_fm_preprocess_u8: the normalised discriminator output of one window
 * and its statistics; out_f32 == NULL runs the statistics-only pass of the default (fused) path instead of the one that
 * also writes the codes -- both must give the same statistics */
int tdoa_fm_preprocess_u8(tdoa_ctx *ctx, const uint8_t *iq, size_t n, float *out_f32, tdoa_fm_stats *stats);
int tdoa_fm_xcorr_lags_u8(tdoa_ctx *ctx, const uint8_t *iq1, size_t n1, const uint8_t *iq2, size_t n2,
                          int max_lag, double *lags_out /* [2*max_lag-1] */);

/* tests only: run the any-size fallback kernels even where a hot-size kernel exists */
int tdoa_debug_force_generic(tdoa_ctx *ctx, int on);
/* tests / measurements only: pick kernel variants by hand.  flags == 0 is the library's default path; every bit
 * set switches one specialised form OFF.  The same switches can be given once, at tdoa_create time, through the
 * environment (TDOA_NO_SHORT_LAG=1, TDOA_NO_SEGMENT_FORM=1, TDOA_NO_SEGMENT_QUADS=1, TDOA_NO_XCD_ROWS=1,
 * TDOA_NO_DECIMATE=1, TDOA_NO_FUSED_K1=1, TDOA_NO_K1_ONCE=1, TDOA_NO_SEG_PACK3=1, TDOA_NO_DEC_COLS=1, TDOA_POW2_ONLY=1); results are the same to rounding whichever form runs. */
enum {
    TDOA_DEBUG_GENERIC_KERNELS = 1,  /* any-size LDS radix-4 kernels instead of the radix-16 register kernels        */
    TDOA_DEBUG_NO_SHORT_LAG    = 2,  /* general pruned inverse even when the search range is below 4095 lags          */
    TDOA_DEBUG_NO_FUSED_K1     = 4,  /* K1 always writes its 24-bit codes to memory (int32); the forward column kernels
                                        read them instead of evaluating the discriminator on the capture bytes      */
    TDOA_DEBUG_NO_SEGMENT_FORM = 8,  /* no LDS-resident overlap-save correlation for search ranges up to 1024 lags    */
    TDOA_DEBUG_NO_XCD_ROWS     = 16, /* plain 2-D grid of the pair kernel even with more pairs than stations          */
    TDOA_DEBUG_NO_SEGMENT_QUADS = 64, /* segment form one pair-window at a time: no station transforms shared by pairs */
    TDOA_DEBUG_NO_DECIMATE     = 256, /* full inverse transform even where the decimated one applies (4096 x 256 / x 512 plans, search ranges above 4095 lags) */
    TDOA_DEBUG_NO_K1_ONCE      = 512, /* the K1 statistics pre-pass everywhere: no single-look K1 (every capture byte read once,
                                        the mean's residual removed where the lags come out; csrc/k1_single_look.hpp)      */
    TDOA_DEBUG_NO_SEG_PACK3    = 1024, /* the segment form reads int32 code rows instead of the packed 3-byte ones        */
    TDOA_DEBUG_NO_DEC_COLS     = 2048, /* the decimated pair step as 4096-bin tiles in LDS (k_pair_decimate16) instead of the column
                                        walk (csrc/dec_stream.hpp); 4096 x 4096 plans then take the full inverse             */
    TDOA_DEBUG_DEC_COLS_ALWAYS = 4096, /* (switches a form ON) the column walk wherever the decimated inverse
                                        applies, also where the library would pick the tile form (as many pairs as stations)  */
    TDOA_DEBUG_NO_DEC_STAGED   = 16384, /* the column walk one pair-window per wave straight from memory (k_pair_decimate_cols)
                                        instead of one workgroup per window and column block with the stations' rows staged in
                                        LDS (csrc/dec_staged.hpp; environment: TDOA_NO_DEC_STAGED=1)                            */
    TDOA_DEBUG_POW2_ONLY       = 8192, /* transform lengths are powers of two everywhere (the reference's padding rule,
                                        processor.go:563): ten-second windows then run in N = 2^25 instead of 5 x 2^22
                                        (environment: TDOA_POW2_ONLY=1)                                                      */
    TDOA_DEBUG_NO_SMALL_FUSED  = 32768, /* the decimated inverse's small plan as two kernels (row pass -> V' -> column sums + K5)
                                        instead of one workgroup per pair-window that keeps the column sums in registers
                                        (k_small_rows_col_peak; environment: TDOA_NO_SMALL_FUSED=1)                            */
    TDOA_DEBUG_SMALL_FUSED_ALWAYS = 65536 /* (switches a form ON) ... for any number of pair-windows; the library takes it from
                                        1024 pair-windows per launch on (environment: TDOA_SMALL_FUSED_ALWAYS=1)              */
};
int tdoa_debug_flags(tdoa_ctx *ctx, unsigned flags);
/* inspection: the K1 statistics of station-window `sw_index` of the last batch (the order of the batch's descriptors:
 * pair calls 0 = first input, 1 = second; tdoa_process: window-major, stations in the order of first use), and whether
 * that batch took the single-look path (every capture byte read once; statistics from the column kernels' tile sums). */
int tdoa_debug_last_k1(tdoa_ctx *ctx, int sw_index, tdoa_fm_stats *stats, int32_t *single_look);
/* tests only: structure of the hipGraph the last tdoa_process captured -- info = {nodes, edges, root nodes, memset nodes};
 * dot_path (may be NULL): also writes the graph in Graphviz form (hipGraphDebugDotPrint).  The library itself refuses a
 * captured step that is not ONE dependency chain of kernel nodes (TDOA_ERR_STATE), see DESIGN.md section 7. */
int tdoa_debug_graph_info(tdoa_ctx *ctx, int32_t info[4], const char *dot_path);
/* tests only (host, no GPU): the cover of a window's station pairs by "quads" -- two template stations x two signal
 * stations whose two packed transforms per segment serve up to four pairs in the segment form.
 * pairs[2 i], pairs[2 i + 1] = template, signal station of pair i; quads_out gets 8 ints per quad: stations a, b, c, d
 * (-1 = empty slot) and the pair index of (a,c), (a,d), (b,c), (b,d) (-1 = not wanted).  Returns the number of quads
 * (at most n_pairs), or a negative TDOA_ERR_* value (more than 32 stations: the library then runs the segment form one
 * pair-window at a time and builds no quads). */
int tdoa_debug_segment_quads(int n_stations, const int32_t *pairs, int n_pairs, int32_t *quads_out, int max_quads);

/* tests only (host, no GPU): how the LDS-staged column walk of the decimated pair step (csrc/dec_staged.hpp) deals the
 * n_stations (n_stations - 1) / 2 pairs of a window -- numbered (0,1), (0,2), ..., as tdoa_process lays them out -- to
 * workgroups of at most max_pairs walks (1..15 next to a loader wave; 16: the form without one, full workgroups first): group g
 * takes counts_out[g] pairs, pairs_out[16 g ..] their numbers, and stages the stations of masks_out[g] (bit s = station s).
 * More than eight stations: every group stays within eight.  Returns the number of groups, or a negative TDOA_ERR_* value
 * (stations outside 2..16, max_pairs outside 1..16, more than max_groups). */
int tdoa_debug_staged_groups(int n_stations, int max_pairs, uint32_t *masks_out, int32_t *counts_out, uint8_t *pairs_out, int max_groups);

/* ---- downstream (processor.go:125-163, 932-1045), host side ---------------- */
void tdoa_latlon_to_ecef(double lat, double lon, double elev, double xyz[3]);
void tdoa_ecef_to_latlon(double x, double y, double z, double lle[3]);
/* reference 3-station solver, bit-compatible call: range_diff[0]=(0,1), [1]=(0,2) */
int  tdoa_solve_3station(const double stations_lle[9], const double *range_diff,
                         double out_lle[3], int *iterations);
/* N-station generalisation: all n(n-1)/2 range differences (pairs i<j), optional weights,
 * X,Y (Z frozen, like the reference) or X,Y,Z unknowns; centroid start, 0.5 damping,
 * 10 iterations, 1 m stop rule as processor.go:950-1010. */
int  tdoa_solve_nstation(const double *stations_lle, int n_stations, const double *range_diff,
                         const double *weights, int solve_z, double out_lle[3], int *iterations);
/* Ground transmitter: the same residuals and weights with the position held on the ellipsoid at height_m (unknowns
 * latitude, longitude; undamped Gauss-Newton from the stations' centroid, at most 20 iterations, 1 m stop rule).  The
 * reference's frozen ECEF Z (processor.go:1004) is a plane that misses a transmitter north or south of the centroid by
 * hundreds of metres to kilometres; this form is what bench.py checks its multi-GPU fix with. */
int  tdoa_solve_surface(const double *stations_lle, int n_stations, const double *range_diff, const double *weights,
                        double height_m, double out_lle[3], int *iterations);

/* ---- measurement ----------------------------------------------------------- */
enum {
    TDOA_K_STATS = 0, TDOA_K_FWD_COL, TDOA_K_FWD_ROW, TDOA_K_INV_ROW, TDOA_K_INV_COL,
    TDOA_K_PEAK, TDOA_K_COUNT
};
/* on = 1: tdoa_process launches its kernels one by one with HIP events at the boundaries of the selected scopes.
 * on = 2: tdoa_process keeps replaying the whole step as one hipGraph (the default path); the selected scopes are timed by
 *         event-record nodes spliced into the captured graph.  on = 0: off. */
int         tdoa_profile_enable(tdoa_ctx *ctx, int on);
/* which scopes of the profiling path record events: bit k = TDOA_K_* scope k (default: all).  A measurement that needs
 * one kernel's launch durations (bench.py: the dominant kernel's, for the roofline) selects that scope alone, so that the
 * timed steps carry two event records instead of one per kernel boundary. */
int         tdoa_profile_select(tdoa_ctx *ctx, unsigned int scope_mask);
int         tdoa_profile_reset(tdoa_ctx *ctx);
int         tdoa_profile_get(tdoa_ctx *ctx, int kernel, double *total_ms, int64_t *launches,
                             double *algorithmic_bytes);
const char *tdoa_kernel_name(int kernel);
/* plan facts: FFT length N (real), factors N1 x N2 of N/2, passes */
int         tdoa_plan_info(const tdoa_ctx *ctx, int64_t *fft_n, int32_t *n1, int32_t *n2);

#ifdef __cplusplus
}
#endif
#endif
