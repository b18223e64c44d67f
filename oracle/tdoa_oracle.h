/*
 * tdoa_oracle.h -- CPU parity oracle for the TDOA correlation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tdoa-geolocation_amd/ (the product)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * What it is: a plain-C restatement of the algorithm the reference executes in
 * processor.go / simple_corr.go / fast_analyzer.go (KX0U-Jim/tdoa-geolocation),
 * following Go/amd64 floating-point lowering (no FMA, complex64 multiply done
 * in f64 and rounded once, complex64/real division == componentwise f32
 * division, f64 accumulators where the Go source has them).  Compile with
 *     gcc -O2 -ffp-contract=off -fno-fast-math
 * Every function cites the reference file:line it follows.
 *
 * PARITY PINNING STATUS: the reference is Go and there is no Go toolchain in
 * the build image, so it can be neither compiled nor run; it ships no unit
 * tests or numeric golden vectors for any correlation output.  The oracle is
 * pinned against every known answer the reference does hold (baselines
 * 12.29/17.02/10.02 km, simple_corr.go's three acceptance thresholds, the
 * correlation_sanity.go flow, file-size arithmetic, fast_analyzer fallbacks);
 * numeric correlation outputs beyond those are "parity unpinned" -- they rest
 * on this restatement alone.  See DESIGN.md.
 *
 * The "mode B" functions (ob_*) are the oracle for the north-star pipeline
 * (u8 IQ -> FM discriminator -> FFT cross-correlation -> peak pick).  That
 * pipeline does not exist in any .go file; its
 * definition is fixed in DESIGN.md and restated here in f64 time-domain form.
 */
#ifndef TDOA_ORACLE_H
#define TDOA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* complex64 == Go complex64 == interleaved {float re, float im}. */

/* ---- processor.go: load / slice ---------------------------------------- */
void   o_iq_u8_to_c64(const uint8_t *raw, size_t n_samples, float *out_c64);
size_t o_extract_reference(const float *data, size_t n, float *out_c64);
size_t o_extract_target(const float *data, size_t n, float *out_c64);

/* ---- processor.go: preprocessing --------------------------------------- */
double o_signal_power(const float *sig, size_t n);
void   o_remove_dc(const float *sig, size_t n, float *out, float *dc_re, float *dc_im);
void   o_lowpass(const float *sig, size_t n, int window, float *out);
int    o_cutoff_window(double cutoff, double fs);
void   o_lowpass_cutoff(const float *sig, size_t n, double cutoff, double fs, float *out);
void   o_highpass(const float *sig, size_t n, double cutoff, double fs, float *out);
void   o_bandpass(const float *sig, size_t n, double lo, double hi, double fs, float *out);
void   o_notch(const float *sig, size_t n, double f0, double bw, double fs, float *out);
void   o_normalize(const float *sig, size_t n, float *out);
void   o_enhance_weak(const float *sig, size_t n, double fs, float *out);
/* returns 1 if the weak chain was taken, 0 for the standard chain */
int    o_preprocess(const float *sig, size_t n, double fs, float *out);

/* ---- processor.go: correlation ----------------------------------------- */
void   o_time_domain_correlation(const float *s1, size_t n1, const float *s2, size_t n2,
                                 int max_lag, int *delay, double *corr);
void   o_cross_correlate(const float *s1, size_t n1, const float *s2, size_t n2,
                         double fs, int *delay, double *corr);
/* all lags of timeDomainCorrelation (for tests): out[d], d < returned count */
int    o_time_domain_all_lags(const float *s1, size_t n1, const float *s2, size_t n2,
                              int max_lag, double *out, int out_cap);
int    o_next_pow2(int n);
void   o_simple_dft(const float *sig, int n, float *out);
/* frequencyDomainCorrelation is dead code that would panic; returns -1 to say so
 * when the source would index out of range, else 0. */
int    o_frequency_domain_correlation(const float *s1, size_t n1, const float *s2, size_t n2,
                                      int max_lag, int *delay, double *corr);

/* ---- simple_corr.go ------------------------------------------------------ */
void   o_simple_correlate(const float *s1, size_t n1, const float *s2, size_t n2,
                          int *delay, float *corr);

/* ---- fast_analyzer.go ---------------------------------------------------- */
typedef struct {
    int    total_samples;
    double i_avg, q_avg, i_std, q_std;
    double snr_estimate, power_level;
    int    has_clipping, has_overload;
} o_fast_analysis;
void   o_fast_dft(const double *in_c128, int n, double *out_c128);
double o_fast_snr(const uint8_t *samples, int total_samples);
void   o_fast_analyze(const uint8_t *samples, int total_samples, o_fast_analysis *out);
/* fastAnalyzeDualFrequencyFile on an in-memory capture; returns -1 if too small */
double o_block_power(const uint8_t *iq, size_t n_samples);   /* collector.go:219-224 */
int    o_fast_analyze_capture(const uint8_t *raw, size_t n_bytes,
                              o_fast_analysis *ref, o_fast_analysis *tgt);

/* ---- processor.go: geodesy + solver ------------------------------------- */
void   o_latlon_to_ecef(double lat, double lon, double elev, double xyz[3]);
double o_distance3d(const double a[3], const double b[3]);
void   o_ecef_to_latlon(double x, double y, double z, double lle[3]);
/* stations: 3 x {lat,lon,elev}; range_diff: >=2 entries ((0,1),(0,2)).
 * returns 0 ok, -1 singular Jacobian (iteration in *iters). */
int    o_solve_tdoa(const double stations_lle[9], const double *range_diff,
                    double out_lle[3], int *iters);

/* ---- simulators (deterministic restatement) ----------------------------- */
typedef struct {
    double lat, lon, elev;
} o_station;
/* simulator.go simulateStation: writes 3*block_samples IQ pairs (6*block bytes) */
void   o_simulate_station(uint8_t *out, size_t block_samples, double fs,
                          double ref_freq, double tgt_freq, double noise_level,
                          o_station st, o_station tx, double tx_power, uint64_t seed);
/* weak_signal_simulator.go simulateWeakSignalStation */
void   o_simulate_weak_station(uint8_t *out, size_t block_samples, double fs,
                               double ref_freq, double tgt_freq,
                               o_station st, o_station tx,
                               double ref_power, double tgt_power, uint64_t seed);
/* test helper: wide-band noise "FM-like" capture with a true integer sample
 * delay between stations (the simulators only model carrier phase). */
void   o_simulate_delayed_fm(uint8_t *out, size_t n_samples, int delay_samples,
                             double mod_index, double noise_level,
                             uint64_t content_seed, uint64_t noise_seed);
double o_rand_float64(uint64_t seed, uint64_t counter);

/* ---- mode B oracle (north-star pipeline) ----------- */
int32_t ob_octant_code(int mn, int mx);          /* atan(mn/mx) in units of pi/2^23, 0 < mn <= mx <= 255 odd */
int    ob_angle_code(int I, int Q);               /* arg(I + iQ) in units of pi/2^23 */
/* phase codes in units of pi/2^23, -2^23 < code <= 2^23 (+pi for an exactly reversed sample) */
void   ob_discriminate_u8(const uint8_t *iq, size_t n, int32_t *code);
typedef struct {
    int64_t  s1;        /* sum of the phase codes                       */
    uint64_t s2_lo;     /* sum of code^2, low 64 bits                   */
    uint64_t s2_hi;     /* high 64 bits (code^2 < 2^46 per sample)      */
    float    mean;      /* f32((double)s1 / n), in code units           */
    float    scale;     /* f32(1/sqrt(var)), 1.0f if var <= 0           */
    double   var;
} ob_stats;
void   ob_phase_stats(const int32_t *code, size_t n, ob_stats *st);
void   ob_preprocess_u8(const uint8_t *iq, size_t n, float *out, ob_stats *st);
/* optional smoothing of the codes (tdoa_params.k1_smooth; applyLowPassFilter, processor.go:270-296, in integers) */
void   ob_smooth_codes(const int32_t *code, size_t n, int window, int32_t *out);
void   ob_preprocess_smooth_u8(const uint8_t *iq, size_t n, int window, float *out, ob_stats *st);
/* optional power gate of the prebuilt binary (tdoa_params.k1_gate): envelope branch for mean power <= 0.01 */
uint64_t ob_power_sum_u8(const uint8_t *iq, size_t n);
int    ob_envelope_class(uint64_t power_sum, size_t n);
int32_t ob_envelope_code(unsigned I, unsigned Q);
void   ob_preprocess_gate_u8(const uint8_t *iq, size_t n, int window, int gate, float *out, ob_stats *st, int *cls);
/* c[d] = sum_i t[i]*s[i+d] (f64, zero outside), lags -(max_lag-1)..max_lag-1,
 * out[d + max_lag - 1]; scaled by 1/sqrt(nt). */
void   ob_xcorr_all_lags(const float *t, size_t nt, const float *s, size_t ns,
                         int max_lag, double *out);
/* peak rule: max |c|, ties -> smaller |d|, then positive d */
void   ob_pick_peak(const double *c, int max_lag, int *lag, double *corr);
void   ob_xcorr_peak(const float *t, size_t nt, const float *s, size_t ns,
                     int max_lag, int *lag, double *corr);
/* (f)-4: parabolic sub-sample refinement around a peak lag and the plausibility gate */
typedef struct {
    double delay;       /* lag + frac, samples                          */
    double frac;        /* vertex offset in [-1/2, 1/2]                 */
    double y[3];        /* s*c[lag-1], s*c[lag], s*c[lag+1], s=sign(c[lag]) */
    int    plausible;   /* |delay| <= gate                              */
} ob_fine;
double ob_parabola_vertex(double ym, double y0, double yp);
void   ob_refine_peak(const float *t, size_t nt, const float *s, size_t ns, int lag,
                      double gate, ob_fine *out);

#ifdef __cplusplus
}
#endif
#endif
