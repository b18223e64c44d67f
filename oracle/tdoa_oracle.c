/*
 * tdoa_oracle.c -- CPU parity oracle (TEST INFRASTRUCTURE, see tdoa_oracle.h).
 *
 * Plain-C restatement of the reference's executed algorithm.  Citations are
 * file:line in KX0U-Jim/tdoa-geolocation.  "parity unpinned" for numeric
 * correlation outputs: the reference has no golden vectors (see header).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 *
 * Go/amd64 lowering rules mirrored here:
 *   - no FMA contraction (mulss + addss are separate)           -> -ffp-contract=off
 *   - complex64 * complex64: operands widened to f64, one rounding back to f32
 *   - complex64 / complex(real,0): runtime.complex128div in f64, rounded to f32
 *   - complex64 +,-: componentwise f32
 *   - float->byte/int conversions truncate toward zero
 */
#include "tdoa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Go calls math.Cos and math.Sin separately.  gcc would merge cos(x) and sin(x) of one
 * argument into a single sincos() call, whose results differ from sin()/cos() in the
 * last bit for a few arguments; route sin through a volatile copy to keep two calls. */
static inline double sin_sep(double x)
{
    volatile double v = x;
    return sin(v);
}

/* ------------------------------------------------------------------------ */
/* Go complex helpers                                                        */
/* ------------------------------------------------------------------------ */

/* complex64 multiply as the Go compiler lowers it on amd64: compute in f64,
 * round each component once to f32. */
static inline void c64_mul(float ar, float ai, float br, float bi, float *re, float *im)
{
    double r = (double)ar * (double)br - (double)ai * (double)bi;
    double i = (double)ar * (double)bi + (double)ai * (double)br;
    *re = (float)r;
    *im = (float)i;
}

/* complex64 / complex(float32(c), 0): runtime.complex128div (Smith's
 * algorithm) with imag(m) == 0, evaluated in f64 and rounded to f32. */
static inline void c64_div_real(float nr, float ni, float c, float *re, float *im)
{
    double mr = (double)c, mi = 0.0;
    double ratio = mi / mr;
    double denom = mr + ratio * mi;
    double e = ((double)nr + (double)ni * ratio) / denom;
    double f = ((double)ni - (double)nr * ratio) / denom;
    *re = (float)e;
    *im = (float)f;
}

/* ------------------------------------------------------------------------ */
/* processor.go: load / slice                                                */
/* ------------------------------------------------------------------------ */

/* processor.go:195-201 loadIQData conversion: (float32(b) - 127.5) / 127.5,
 * a true IEEE f32 division per component. */
void o_iq_u8_to_c64(const uint8_t *raw, size_t n_samples, float *out)
{
    for (size_t i = 0; i < n_samples; i++) {
        float iv = ((float)raw[2 * i] - 127.5f) / 127.5f;
        float qv = ((float)raw[2 * i + 1] - 127.5f) / 127.5f;
        out[2 * i] = iv;
        out[2 * i + 1] = qv;
    }
}

/* processor.go:208-238 extractReferenceSignal: blocks 1 and 3 concatenated;
 * block size = total/3; if 0 the input is returned unchanged. */
size_t o_extract_reference(const float *data, size_t n, float *out)
{
    size_t bs = n / 3;
    if (bs == 0) {
        memcpy(out, data, n * 2 * sizeof(float));
        return n;
    }
    memcpy(out, data, bs * 2 * sizeof(float));
    if (2 * bs < n)
        memcpy(out + 2 * bs, data + 2 * (2 * bs), bs * 2 * sizeof(float));
    else
        memset(out + 2 * bs, 0, bs * 2 * sizeof(float));
    return 2 * bs;
}

/* processor.go:241-267 extractTargetSignal: block 2. */
size_t o_extract_target(const float *data, size_t n, float *out)
{
    size_t bs = n / 3;
    if (bs == 0) {
        memcpy(out, data, n * 2 * sizeof(float));
        return n;
    }
    if (2 * bs <= n)
        memcpy(out, data + 2 * bs, bs * 2 * sizeof(float));
    else
        memset(out, 0, bs * 2 * sizeof(float));
    return bs;
}

/* ------------------------------------------------------------------------ */
/* processor.go: preprocessing                                               */
/* ------------------------------------------------------------------------ */

/* processor.go:322-333 calculateSignalPower: f32 re*re+im*im (two f32
 * multiplies, one f32 add), widened, accumulated sequentially in f64. */
double o_signal_power(const float *sig, size_t n)
{
    if (n == 0)
        return 0.0;
    double power = 0.0;
    for (size_t i = 0; i < n; i++) {
        float re = sig[2 * i], im = sig[2 * i + 1];
        float p = re * re + im * im;
        power += (double)p;
    }
    return power / (double)n;
}

/* processor.go:299-319 removeDCBias: sequential f32 complex sum, mean by
 * complex division with a real divisor, componentwise f32 subtract. */
void o_remove_dc(const float *sig, size_t n, float *out, float *dc_re, float *dc_im)
{
    if (n == 0) {
        if (dc_re) *dc_re = 0.0f;
        if (dc_im) *dc_im = 0.0f;
        return;
    }
    float sr = 0.0f, si = 0.0f;
    for (size_t i = 0; i < n; i++) {
        sr = sr + sig[2 * i];
        si = si + sig[2 * i + 1];
    }
    float mr, mi;
    c64_div_real(sr, si, (float)n, &mr, &mi);
    for (size_t i = 0; i < n; i++) {
        out[2 * i] = sig[2 * i] - mr;
        out[2 * i + 1] = sig[2 * i + 1] - mi;
    }
    if (dc_re) *dc_re = mr;
    if (dc_im) *dc_im = mi;
}

/* processor.go:270-296 applyLowPassFilter: centred moving average with
 * half-window window/2, edge-truncated; every output is a fresh sequential
 * f32 sum in ascending j, divided by complex(float32(count),0). */
void o_lowpass(const float *sig, size_t n, int window, float *out)
{
    if (window <= 1) {
        if (out != sig)
            memcpy(out, sig, n * 2 * sizeof(float));
        return;
    }
    long half = window / 2;
    long len = (long)n;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < len; i++) {
        float sr = 0.0f, si = 0.0f;
        int count = 0;
        long j0 = i - half, j1 = i + half;
        if (j0 < 0) j0 = 0;
        if (j1 > len - 1) j1 = len - 1;
        for (long j = j0; j <= j1; j++) {
            sr = sr + sig[2 * j];
            si = si + sig[2 * j + 1];
            count++;
        }
        float re = 0.0f, im = 0.0f;
        if (count > 0)
            c64_div_real(sr, si, (float)count, &re, &im);
        out[2 * i] = re;
        out[2 * i + 1] = im;
    }
}

/* processor.go:400-406 window size from a cut-off: int(fs/(2*fc)) in [3,1000]. */
int o_cutoff_window(double cutoff, double fs)
{
    double w = fs / (2 * cutoff);
    int window;
    if (!(w < 2147483647.0))
        window = 2147483647; /* Go's float->int of +Inf/huge is implementation-defined; clamp follows anyway */
    else
        window = (int)w;
    if (window < 3) window = 3;
    if (window > 1000) window = 1000;
    return window;
}

/* processor.go:397-409 */
void o_lowpass_cutoff(const float *sig, size_t n, double cutoff, double fs, float *out)
{
    o_lowpass(sig, n, o_cutoff_window(cutoff, fs), out);
}

/* processor.go:384-394 applyHighPassFilter: signal - lowpass(signal). */
void o_highpass(const float *sig, size_t n, double cutoff, double fs, float *out)
{
    float *lp = (float *)malloc((n ? n : 1) * 2 * sizeof(float));
    o_lowpass_cutoff(sig, n, cutoff, fs, lp);
    for (size_t i = 0; i < 2 * n; i++)
        out[i] = sig[i] - lp[i];
    free(lp);
}

/* processor.go:354-381 applyBandpassFilter: high-pass stage if lo > 0,
 * low-pass stage if hi < fs/2 (strict). */
void o_bandpass(const float *sig, size_t n, double lo, double hi, double fs, float *out)
{
    if (n == 0)
        return;
    float *tmp = (float *)malloc(n * 2 * sizeof(float));
    if (lo > 0)
        o_highpass(sig, n, lo, fs, tmp);
    else
        memcpy(tmp, sig, n * 2 * sizeof(float));
    if (hi < fs / 2)
        o_lowpass_cutoff(tmp, n, hi, fs, out);
    else
        memcpy(out, tmp, n * 2 * sizeof(float));
    free(tmp);
}

/* processor.go:412-434 applyNotchFilter: x - bandpass(x)*0.8 with the band
 * clamped to [0, fs/2]; the *0.8 is a complex64 multiply by complex64(0.8+0i). */
void o_notch(const float *sig, size_t n, double f0, double bw, double fs, float *out)
{
    double lo = f0 - bw / 2, hi = f0 + bw / 2;
    if (lo < 0) lo = 0;
    if (hi > fs / 2) hi = fs / 2;
    float *band = (float *)malloc((n ? n : 1) * 2 * sizeof(float));
    o_bandpass(sig, n, lo, hi, fs, band);
    for (size_t i = 0; i < n; i++) {
        float pr, pi;
        c64_mul(band[2 * i], band[2 * i + 1], 0.8f, 0.0f, &pr, &pi);
        out[2 * i] = sig[2 * i] - pr;
        out[2 * i + 1] = sig[2 * i + 1] - pi;
    }
    free(band);
}

/* processor.go:336-351 normalizeSignal: scale = float32(1/sqrt(power)) with
 * the f64 power of calculateSignalPower; unchanged if power <= 0. */
void o_normalize(const float *sig, size_t n, float *out)
{
    double power = o_signal_power(sig, n);
    if (power <= 0) {
        if (out != sig)
            memcpy(out, sig, n * 2 * sizeof(float));
        return;
    }
    float scale = (float)(1.0 / sqrt(power));
    for (size_t i = 0; i < 2 * n; i++)
        out[i] = sig[i] * scale;
}

/* processor.go:437-466 enhanceWeakSignal. */
void o_enhance_weak(const float *sig, size_t n, double fs, float *out)
{
    float *a = (float *)malloc((n ? n : 1) * 2 * sizeof(float));
    float *b = (float *)malloc((n ? n : 1) * 2 * sizeof(float));
    o_remove_dc(sig, n, a, NULL, NULL);          /* :443 */
    o_notch(a, n, 60, 5, fs, b);                 /* :446 */
    o_notch(b, n, 120, 5, fs, a);                /* :447 */
    o_notch(a, n, 1000000, 50000, fs, b);        /* :448 */
    o_bandpass(b, n, 100.0, 40000.0, fs, a);     /* :453-457 */
    o_lowpass(a, n, 50, b);                      /* :460 */
    o_normalize(b, n, out);                      /* :463 */
    free(a);
    free(b);
}

/* processor.go:469-499 preprocessSignal: power gate at 0.001. */
int o_preprocess(const float *sig, size_t n, double fs, float *out)
{
    double p0 = o_signal_power(sig, n);
    if (p0 < 0.001) {
        o_enhance_weak(sig, n, fs, out);
        return 1;
    }
    float *a = (float *)malloc((n ? n : 1) * 2 * sizeof(float));
    float *b = (float *)malloc((n ? n : 1) * 2 * sizeof(float));
    o_remove_dc(sig, n, a, NULL, NULL);          /* :485 */
    o_bandpass(a, n, 500, 50000, fs, b);         /* :489 */
    o_lowpass(b, n, 100, a);                     /* :492 */
    o_normalize(a, n, out);                      /* :495 */
    free(a);
    free(b);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* processor.go: correlation                                                 */
/* ------------------------------------------------------------------------ */

/* one lag of processor.go:686-726; returns number of blocks used */
static int tdc_one_lag(const float *tpl, long tl, const float *sg, long sl, long delay, double *out)
{
    const long block = 1000; /* :682 */
    double correlation = 0.0;
    int nblocks = 0;
    for (long bs = 0; bs < tl - block; bs += block) {
        long be = bs + block;
        if (delay + be > sl)
            break;
        double bc = 0.0;
        for (long i = bs; i < be; i++) {
            float tr = tpl[2 * i], ti = tpl[2 * i + 1];
            float sr = sg[2 * (delay + i)], si = sg[2 * (delay + i) + 1];
            float p = tr * sr + ti * si; /* :705, f32 then widened */
            bc += (double)p;
        }
        bc /= (double)block;
        correlation += bc;
        nblocks++;
    }
    if (nblocks > 0) {
        correlation /= (double)nblocks;
        double gain = sqrt((double)(nblocks * block));
        correlation *= gain;
    }
    *out = correlation;
    return nblocks;
}

static long tdc_setup(const float *s1, size_t n1, const float *s2, size_t n2, int max_lag,
                      const float **tpl, long *tl, const float **sg, long *sl)
{
    *tpl = s1; *tl = (long)n1; *sg = s2; *sl = (long)n2;
    if (n1 > n2) { *tpl = s2; *tl = (long)n2; *sg = s1; *sl = (long)n1; } /* :650-655 */
    long ml = max_lag;
    if (ml > *sl - *tl) ml = *sl - *tl;   /* :668-670 */
    if (ml < 1) ml = 1;                    /* :673-675 */
    return ml;
}

/* processor.go:646-736 timeDomainCorrelation. */
void o_time_domain_correlation(const float *s1, size_t n1, const float *s2, size_t n2,
                               int max_lag, int *delay, double *corr)
{
    const float *tpl, *sg;
    long tl, sl;
    long ml = tdc_setup(s1, n1, s2, n2, max_lag, &tpl, &tl, &sg, &sl);
    int best_delay = 0;
    double best = 0.0;
    double *vals = (double *)malloc((size_t)ml * sizeof(double));
    int *nb = (int *)malloc((size_t)ml * sizeof(int));
#pragma omp parallel for schedule(dynamic, 16)
    for (long d = 0; d < ml; d++)
        nb[d] = tdc_one_lag(tpl, tl, sg, sl, d, &vals[d]);
    for (long d = 0; d < ml; d++) {
        if (nb[d] > 0 && fabs(vals[d]) > fabs(best)) { /* :722-725 strictly greater */
            best = vals[d];
            best_delay = (int)d;
        }
    }
    free(vals);
    free(nb);
    *delay = best_delay;
    *corr = best;
}

int o_time_domain_all_lags(const float *s1, size_t n1, const float *s2, size_t n2,
                           int max_lag, double *out, int out_cap)
{
    const float *tpl, *sg;
    long tl, sl;
    long ml = tdc_setup(s1, n1, s2, n2, max_lag, &tpl, &tl, &sg, &sl);
    if (ml > out_cap) ml = out_cap;
#pragma omp parallel for schedule(dynamic, 16)
    for (long d = 0; d < ml; d++)
        tdc_one_lag(tpl, tl, sg, sl, d, &out[d]);
    return (int)ml;
}

/* processor.go:619-643 crossCorrelate. */
void o_cross_correlate(const float *s1, size_t n1, const float *s2, size_t n2,
                       double fs, int *delay, double *corr)
{
    if (n1 == 0 || n2 == 0) { /* :622-625 */
        *delay = 0;
        *corr = 0.0;
        return;
    }
    float *p1 = (float *)malloc(n1 * 2 * sizeof(float));
    float *p2 = (float *)malloc(n2 * 2 * sizeof(float));
    o_preprocess(s1, n1, fs, p1);
    o_preprocess(s2, n2, fs, p2);
    o_time_domain_correlation(p1, n1, p2, n2, 20000, delay, corr); /* :633-636 */
    free(p1);
    free(p2);
}

/* processor.go:502-512 */
int o_next_pow2(int n)
{
    if (n <= 1)
        return 1;
    int p = 1;
    while (p < n)
        p <<= 1;
    return p;
}

/* processor.go:515-536 simpleFFT: O(n^2) forward DFT, twiddle in f64 rounded
 * to f32, complex64 multiply-accumulate. */
void o_simple_dft(const float *sig, int n, float *out)
{
    if (n <= 1) {
        if (n == 1) { out[0] = sig[0]; out[1] = sig[1]; }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < n; k++) {
        float sr = 0.0f, si = 0.0f;
        for (int j = 0; j < n; j++) {
            double angle = -2.0 * M_PI * (double)((long)k * (long)j) / (double)n;
            float tr = (float)cos(angle), ti = (float)sin_sep(angle);
            float pr, pi;
            c64_mul(sig[2 * j], sig[2 * j + 1], tr, ti, &pr, &pi);
            sr = sr + pr;
            si = si + pi;
        }
        out[2 * k] = sr;
        out[2 * k + 1] = si;
    }
}

/* processor.go:539-616 frequencyDomainCorrelation -- DEAD CODE in the
 * reference (crossCorrelate never calls it, :638).  Restated for the record:
 * with max_lag >= 1 the first loop iteration reads correlation[fftSize-0],
 * which is out of range in Go (panic).  Returns -1 in that case. */
int o_frequency_domain_correlation(const float *s1, size_t n1, const float *s2, size_t n2,
                                   int max_lag, int *delay, double *corr)
{
    (void)s1;
    (void)s2;
    *delay = 0;
    *corr = 0.0;
    if (n1 == 0 || n2 == 0)
        return 0;
    int chunk = 1024;
    if ((size_t)chunk > n1) chunk = (int)n1;
    if ((size_t)chunk > n2) chunk = (int)n2;
    int fft_size = o_next_pow2(chunk + max_lag);
    int search = max_lag;
    if (search > fft_size / 2) search = fft_size / 2;
    if (search >= 1)
        return -1; /* :605-606 correlation[fftSize-0] -> index out of range */
    return 0;
}

/* ------------------------------------------------------------------------ */
/* simple_corr.go                                                            */
/* ------------------------------------------------------------------------ */

/* simple_corr.go:83-160 simpleCorrelate: f32 accumulators, per-lag signal
 * power, corr/(sqrt(Pt)*sqrt(Ps)) with each sqrt done in f64 and rounded. */
void o_simple_correlate(const float *s1, size_t n1, const float *s2, size_t n2,
                        int *delay, float *corr)
{
    *delay = 0;
    *corr = 0.0f;
    if (n1 == 0 || n2 == 0)
        return;
    const float *tpl = s1, *sg = s2;
    long tl = (long)n1, sl = (long)n2;
    if (n1 > n2) { tpl = s2; tl = (long)n2; sg = s1; sl = (long)n1; }
    long ml = 1000;
    if (ml > sl - tl) ml = sl - tl;
    if (ml < 1) ml = 1;
    float tpow = 0.0f;
    for (long i = 0; i < tl; i++) {
        float re = tpl[2 * i], im = tpl[2 * i + 1];
        tpow = tpow + (re * re + im * im);
    }
    int best_delay = 0;
    float best = 0.0f;
    for (long d = 0; d < ml; d++) {
        float c = 0.0f, spow = 0.0f;
        for (long i = 0; i < tl; i++) {
            if (d + i >= sl)
                break;
            float tr = tpl[2 * i], ti = tpl[2 * i + 1];
            float sr = sg[2 * (d + i)], si = sg[2 * (d + i) + 1];
            c = c + (tr * sr + ti * si);
            spow = spow + (sr * sr + si * si);
        }
        if (tpow > 0 && spow > 0) {
            float nc = c / ((float)sqrt((double)tpow) * (float)sqrt((double)spow));
            if (fabs((double)nc) > fabs((double)best)) {
                best = nc;
                best_delay = (int)d;
            }
        }
    }
    *delay = best_delay;
    *corr = best;
}

/* ------------------------------------------------------------------------ */
/* fast_analyzer.go                                                          */
/* ------------------------------------------------------------------------ */

/* fast_analyzer.go:229-253 fastDFT: complex128, table twiddles, idx=(k*i)%n. */
void o_fast_dft(const double *in, int n_in, double *out)
{
    int n = n_in;
    if (n > 8192) n = 8192;
    double *tw = (double *)malloc((size_t)(n ? n : 1) * 2 * sizeof(double));
    for (int k = 0; k < n; k++) {
        double angle = -2 * M_PI * (double)k / (double)n;
        tw[2 * k] = cos(angle);
        tw[2 * k + 1] = sin_sep(angle);
    }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < n; k++) {
        double sr = 0.0, si = 0.0;
        for (int i = 0; i < n; i++) {
            int idx = (int)(((long)k * (long)i) % n);
            double ar = in[2 * i], ai = in[2 * i + 1];
            double br = tw[2 * idx], bi = tw[2 * idx + 1];
            double pr = ar * br - ai * bi;
            double pi = ar * bi + ai * br;
            sr = sr + pr;
            si = si + pi;
        }
        out[2 * k] = sr;
        out[2 * k + 1] = si;
    }
    free(tw);
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* Go math.Hypot shape: p*sqrt(1+(q/p)^2) with p=max, q=min. */
static double go_hypot(double p, double q)
{
    p = fabs(p);
    q = fabs(q);
    if (p < q) { double t = p; p = q; q = t; }
    if (p == 0)
        return 0;
    q = q / p;
    return p * sqrt(1 + q * q);
}

/* fast_analyzer.go:163-227 fastSNRCalculation. */
double o_fast_snr(const uint8_t *samples, int total)
{
    int asz = 8192;
    if (total < asz) asz = total;
    if (asz <= 0)
        return -20.0;
    int start = (total - asz) / 2;
    const uint8_t *ab = samples + (size_t)start * 2;
    double *cs = (double *)malloc((size_t)asz * 2 * sizeof(double));
    for (int i = 0; i < asz; i++) {
        cs[2 * i] = ((double)ab[2 * i] - 127.5) / 127.5;
        cs[2 * i + 1] = ((double)ab[2 * i + 1] - 127.5) / 127.5;
    }
    for (int i = 0; i < asz; i++) { /* :183-186 Hann, complex128 multiply by (w,0) */
        double w = 0.5 - 0.5 * cos(2 * M_PI * (double)i / (double)(asz - 1));
        double sr = cs[2 * i], si = cs[2 * i + 1];
        cs[2 * i] = w * sr - 0.0 * si;
        cs[2 * i + 1] = w * si + 0.0 * sr;
    }
    double *ft = (double *)malloc((size_t)asz * 2 * sizeof(double));
    o_fast_dft(cs, asz, ft);
    double *psd = (double *)malloc((size_t)asz * sizeof(double));
    double *sorted = (double *)malloc((size_t)asz * sizeof(double));
    for (int i = 0; i < asz; i++) {
        double a = go_hypot(ft[2 * i], ft[2 * i + 1]);
        psd[i] = a * a;
        sorted[i] = psd[i];
    }
    qsort(sorted, (size_t)asz, sizeof(double), cmp_double);
    double sig_thr = sorted[(int)(0.9 * (double)asz)];
    double noise_thr = sorted[(int)(0.4 * (double)asz)];
    double sp = 0, np = 0;
    int sc = 0, nc = 0;
    for (int i = 0; i < asz; i++) {
        if (psd[i] >= sig_thr) { sp += psd[i]; sc++; }
        else if (psd[i] <= noise_thr) { np += psd[i]; nc++; }
    }
    if (sc > 0) sp /= (double)sc;
    if (nc > 0) np /= (double)nc;
    free(cs); free(ft); free(psd); free(sorted);
    if (np > 0 && sp > np)
        return 10 * log10(sp / np);
    return -20.0;
}

/* fast_analyzer.go:113-161 fastAnalyzeSamples. */
void o_fast_analyze(const uint8_t *s, int total, o_fast_analysis *a)
{
    memset(a, 0, sizeof(*a));
    a->total_samples = total;
    double isum = 0, qsum = 0, isq = 0, qsq = 0;
    uint8_t imin = 255, imax = 0, qmin = 255, qmax = 0;
    for (int i = 0; i < total; i++) {
        uint8_t iv = s[2 * i], qv = s[2 * i + 1];
        double fi = (double)iv, fq = (double)qv;
        isum += fi; qsum += fq;
        isq += fi * fi; qsq += fq * fq;
        if (iv < imin) imin = iv;
        if (iv > imax) imax = iv;
        if (qv < qmin) qmin = qv;
        if (qv > qmax) qmax = qv;
    }
    double n = (double)total;
    a->i_avg = isum / n;
    a->q_avg = qsum / n;
    a->i_std = sqrt((isq / n) - (a->i_avg * a->i_avg));
    a->q_std = sqrt((qsq / n) - (a->q_avg * a->q_avg));
    double pm = sqrt(a->i_std * a->i_std + a->q_std * a->q_std);
    if (pm <= 1e-10)
        a->power_level = -100.0;
    else
        a->power_level = 20 * log10(pm);
    a->has_clipping = (imin == 0 || imax == 255 || qmin == 0 || qmax == 255);
    a->has_overload = (a->i_std < 2 || a->q_std < 2);
    a->snr_estimate = o_fast_snr(s, total);
}

/* collector.go:219-224 validateDataFile block power: mean of (I-127.5)^2 + (Q-127.5)^2 over the
 * first n samples, float64 running sum in sample order. */
double o_block_power(const uint8_t *iq, size_t n)
{
    double sum_sq = 0.0;
    for (size_t j = 0; j < n; j++) {
        double iv = (double)iq[2 * j] - 127.5;
        double qv = (double)iq[2 * j + 1] - 127.5;
        sum_sq += iv * iv + qv * qv;
    }
    return sum_sq / (double)n;
}

/* fast_analyzer.go:53-111 fastAnalyzeDualFrequencyFile on bytes in memory. */
int o_fast_analyze_capture(const uint8_t *raw, size_t n_bytes,
                           o_fast_analysis *ref, o_fast_analysis *tgt)
{
    long total = (long)(n_bytes / 2);
    long bs = total / 3;
    if (bs == 0)
        return -1;
    long asz = 32768;
    if (bs < asz) asz = bs;
    uint8_t *rb = (uint8_t *)malloc((size_t)asz * 4);
    uint8_t *tb = (uint8_t *)malloc((size_t)asz * 2);
    memcpy(rb, raw, (size_t)asz * 2);
    memcpy(rb + asz * 2, raw + bs * 2 * 2, (size_t)asz * 2);
    memcpy(tb, raw + bs * 2, (size_t)asz * 2);
    o_fast_analyze(rb, (int)(asz * 2), ref);
    o_fast_analyze(tb, (int)asz, tgt);
    free(rb);
    free(tb);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* processor.go: geodesy + solver                                            */
/* ------------------------------------------------------------------------ */

/* processor.go:125-148 */
void o_latlon_to_ecef(double lat, double lon, double elev, double xyz[3])
{
    const double a = 6378137.0;
    const double f = 1.0 / 298.257223563;
    double e2 = 2 * f - f * f;
    double lat_r = lat * M_PI / 180;
    double lon_r = lon * M_PI / 180;
    double sl = sin_sep(lat_r), cl = cos(lat_r), so = sin_sep(lon_r), co = cos(lon_r);
    double N = a / sqrt(1 - e2 * sl * sl);
    xyz[0] = (N + elev) * cl * co;
    xyz[1] = (N + elev) * cl * so;
    xyz[2] = (N * (1 - e2) + elev) * sl;
}

/* processor.go:151-156 */
double o_distance3d(const double p1[3], const double p2[3])
{
    double dx = p2[0] - p1[0], dy = p2[1] - p1[1], dz = p2[2] - p1[2];
    return sqrt(dx * dx + dy * dy + dz * dz);
}

/* processor.go:1023-1045 */
void o_ecef_to_latlon(double x, double y, double z, double lle[3])
{
    const double a = 6378137.0;
    const double f = 1.0 / 298.257223563;
    const double e2 = 2 * f - f * f;
    double p = sqrt(x * x + y * y);
    double lon = atan2(y, x);
    double lat = atan2(z, p * (1 - e2));
    for (int i = 0; i < 5; i++) {
        double N = a / sqrt(1 - e2 * sin_sep(lat) * sin_sep(lat));
        double elev = p / cos(lat) - N;
        lat = atan2(z, p * (1 - e2 * N / (N + elev)));
    }
    double N = a / sqrt(1 - e2 * sin_sep(lat) * sin_sep(lat));
    double elev = p / cos(lat) - N;
    lle[0] = lat * 180.0 / M_PI;
    lle[1] = lon * 180.0 / M_PI;
    lle[2] = elev;
}

/* processor.go:932-1020 solveTDOA: 3 stations hard-wired, uses range
 * differences (0,1) and (0,2), damped (0.5) 2x2 Newton in ECEF X,Y, Z frozen,
 * 10 iterations, 1 m tolerance, centroid start. */
int o_solve_tdoa(const double st[9], const double *rd, double out[3], int *iters)
{
    double s[3][3];
    for (int i = 0; i < 3; i++)
        o_latlon_to_ecef(st[3 * i], st[3 * i + 1], st[3 * i + 2], s[i]);
    double clat = (st[0] + st[3] + st[6]) / 3.0;
    double clon = (st[1] + st[4] + st[7]) / 3.0;
    double celev = (st[2] + st[5] + st[8]) / 3.0;
    double x[3];
    o_latlon_to_ecef(clat, clon, celev, x);
    int it;
    for (it = 0; it < 10; it++) {
        double r1 = sqrt((x[0] - s[0][0]) * (x[0] - s[0][0]) + (x[1] - s[0][1]) * (x[1] - s[0][1]) + (x[2] - s[0][2]) * (x[2] - s[0][2]));
        double r2 = sqrt((x[0] - s[1][0]) * (x[0] - s[1][0]) + (x[1] - s[1][1]) * (x[1] - s[1][1]) + (x[2] - s[1][2]) * (x[2] - s[1][2]));
        double r3 = sqrt((x[0] - s[2][0]) * (x[0] - s[2][0]) + (x[1] - s[2][1]) * (x[1] - s[2][1]) + (x[2] - s[2][2]) * (x[2] - s[2][2]));
        double res1 = (r2 - r1) - rd[0];
        double res2 = (r3 - r1) - rd[1];
        if (fabs(res1) < 1.0 && fabs(res2) < 1.0)
            break;
        double dx1 = (x[0] - s[0][0]) / r1, dy1 = (x[1] - s[0][1]) / r1;
        double dx2 = (x[0] - s[1][0]) / r2, dy2 = (x[1] - s[1][1]) / r2;
        double dx3 = (x[0] - s[2][0]) / r3, dy3 = (x[1] - s[2][1]) / r3;
        double J11 = dx2 - dx1, J12 = dy2 - dy1;
        double J21 = dx3 - dx1, J22 = dy3 - dy1;
        double det = J11 * J22 - J12 * J21;
        if (fabs(det) < 1e-10) {
            if (iters) *iters = it;
            return -1;
        }
        double dx = (-res1 * J22 + res2 * J12) / det;
        double dy = (res1 * J21 - res2 * J11) / det;
        double dz = 0.0;
        double step = 0.5;
        x[0] += step * dx;
        x[1] += step * dy;
        x[2] += step * dz;
    }
    if (iters) *iters = it;
    o_ecef_to_latlon(x[0], x[1], x[2], out);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* deterministic simulators                                                  */
/* ------------------------------------------------------------------------ */

/* Counter-based uniform in [0,1) with 53 bits (splitmix64 finaliser).  The
 * reference uses time-seeded math/rand (simulator.go:225), so its outputs are
 * not reproducible; the restatement keeps the distributions and replaces the
 * generator. */
static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

double o_rand_float64(uint64_t seed, uint64_t counter)
{
    uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ULL * (counter + 1));
    z = mix64(z ^ 0xD6E8FEB86659FD93ULL);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

/* standard normal via Box-Muller on two counter-based uniforms (replaces
 * rand.NormFloat64's ziggurat). */
static double rand_norm(uint64_t seed, uint64_t counter)
{
    double u1 = o_rand_float64(seed, 2 * counter);
    double u2 = o_rand_float64(seed, 2 * counter + 1);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

/* simulator.go:148-161 quantisation: f32 x*127.5+127.5 (separate mul, add),
 * clamp to [0,255], byte() truncation. */
static inline uint8_t quantize_u8(float v)
{
    float q = v * 127.5f + 127.5f;
    if (q < 0) q = 0;
    if (q > 255) q = 255;
    return (uint8_t)q;
}

/* simulator.go:34-64 calculateDistance3D */
static double sim_distance(o_station a, o_station b)
{
    double p1[3], p2[3];
    o_latlon_to_ecef(a.lat, a.lon, a.elev, p1);
    o_latlon_to_ecef(b.lat, b.lon, b.elev, p2);
    return o_distance3d(p1, p2);
}

/* simulator.go:67-98: tone cos/sin(omega*t+phase) rounded to f32, plus
 * uniform noise level*(2U-1) rounded to f32, f32 add. */
static void sim_block_perfect(uint8_t *out, size_t n, double freq, double fs, double amp,
                              double phase, double noise, uint64_t seed, uint64_t block_id)
{
    double omega = 2 * M_PI * freq;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        double t = (double)i / fs;
        float re = (float)(amp * cos(omega * t + phase));
        float im = (float)(amp * sin_sep(omega * t + phase));
        uint64_t ctr = (block_id << 40) + (uint64_t)i;
        float ni = (float)(noise * (2 * o_rand_float64(seed, 2 * ctr) - 1));
        float nq = (float)(noise * (2 * o_rand_float64(seed, 2 * ctr + 1) - 1));
        re = re + ni;
        im = im + nq;
        out[2 * i] = quantize_u8(re);
        out[2 * i + 1] = quantize_u8(im);
    }
}

/* simulator.go:100-180 simulateStation: delay is a carrier PHASE offset only
 * (2*pi*f_tgt*travel), amplitude power/d*0.1; blocks [ref | target | ref],
 * ref tone amplitude 0.01 phase 0. */
void o_simulate_station(uint8_t *out, size_t bsamp, double fs, double ref_freq, double tgt_freq,
                        double noise, o_station st, o_station tx, double tx_power, uint64_t seed)
{
    double dist = sim_distance(st, tx);
    const double c = 299792458.0;
    double travel = dist / c;
    double phase = 2 * M_PI * tgt_freq * travel;
    double amp = tx_power / dist;
    amp *= 0.1;
    sim_block_perfect(out, bsamp, ref_freq, fs, 0.01, 0.0, noise, seed, 1);
    sim_block_perfect(out + 2 * bsamp, bsamp, tgt_freq, fs, amp, phase, noise, seed, 2);
    sim_block_perfect(out + 4 * bsamp, bsamp, ref_freq, fs, 0.01, 0.0, noise, seed, 3);
}

typedef struct {
    double gaussian, impulse_p, impulse_level, drift, dc;
} noise_profile;

/* weak_signal_simulator.go:89-126 generateWeakSignal.  Deviation (documented):
 * phaseDrift is the closed form (i+1)*drift/fs instead of the running f64 sum
 * so that blocks can be generated in parallel. */
static void sim_block_weak(uint8_t *out, size_t n, double freq, double fs, double amp, double phase,
                           noise_profile np, uint64_t seed, uint64_t block_id)
{
    double omega = 2 * M_PI * freq;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        double t = (double)i / fs;
        double drift = (double)(i + 1) * (np.drift / fs);
        double cur = omega * t + phase + drift;
        double re = amp * cos(cur), im = amp * sin_sep(cur);
        re += np.dc;
        im += np.dc;
        uint64_t ctr = ((block_id << 40) + (uint64_t)i) * 8;
        if (np.gaussian > 0) {
            re += np.gaussian * rand_norm(seed, ctr);
            im += np.gaussian * rand_norm(seed, ctr + 1);
        }
        if (o_rand_float64(seed, 2 * (ctr + 2)) < np.impulse_p) {
            re += np.impulse_level * (2 * o_rand_float64(seed, 2 * (ctr + 3)) - 1);
            im += np.impulse_level * (2 * o_rand_float64(seed, 2 * (ctr + 4)) - 1);
        }
        out[2 * i] = quantize_u8((float)re);
        out[2 * i + 1] = quantize_u8((float)im);
    }
}

/* weak_signal_simulator.go:129-148 generateStrongSignal */
static void sim_block_strong(uint8_t *out, size_t n, double freq, double fs, double amp, double phase,
                             uint64_t seed, uint64_t block_id)
{
    double omega = 2 * M_PI * freq;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        double t = (double)i / fs;
        double re = amp * cos(omega * t + phase), im = amp * sin_sep(omega * t + phase);
        uint64_t ctr = ((block_id << 40) + (uint64_t)i) * 8;
        re += 0.001 * rand_norm(seed, ctr);
        im += 0.001 * rand_norm(seed, ctr + 1);
        out[2 * i] = quantize_u8((float)re);
        out[2 * i + 1] = quantize_u8((float)im);
    }
}

/* weak_signal_simulator.go:151-257 simulateWeakSignalStation. */
void o_simulate_weak_station(uint8_t *out, size_t bsamp, double fs, double ref_freq, double tgt_freq,
                             o_station st, o_station tx, double ref_power, double tgt_power, uint64_t seed)
{
    double dist = sim_distance(st, tx);
    const double c = 299792458.0;
    double travel = dist / c;
    double ref_phase = 2 * M_PI * ref_freq * travel;
    double tgt_phase = 2 * M_PI * tgt_freq * travel;
    double ref_amp = ref_power / dist * 0.1;
    double tgt_amp = tgt_power / dist * 0.1;
    noise_profile weak = { ref_amp * 0.8, 0.001, ref_amp * 5.0, 0.05, ref_amp * 0.1 };
    sim_block_weak(out, bsamp, ref_freq, fs, ref_amp, ref_phase, weak, seed, 1);
    sim_block_strong(out + 2 * bsamp, bsamp, tgt_freq, fs, tgt_amp, tgt_phase, seed, 2);
    sim_block_weak(out + 4 * bsamp, bsamp, ref_freq, fs, ref_amp, ref_phase, weak, seed, 3);
}

/* Test helper (not in the reference): constant-envelope FM of a smooth random
 * message, with a true integer sample delay, plus per-station uniform noise.
 * The message at absolute index k depends only on (content_seed, k), so two
 * stations generated with different delays see time-shifted copies. */
void o_simulate_delayed_fm(uint8_t *out, size_t n, int delay, double mod_index, double noise,
                           uint64_t content_seed, uint64_t noise_seed)
{
    /* Phase modulation by a short-memory (one-pole) message, so the waveform at
     * absolute index k does not depend on where generation started (512-sample
     * warm-up: 0.95^512 ~ 4e-12). */
    const long warm = 512;
    const long base = 1L << 30;
    const double msg_rms = 0.09245003270420485; /* sqrt(0.05^2/(1-0.95^2)/3) */
    double s = 0.0;
    long k0 = base - (long)delay - warm;
    for (long k = k0; k < base - (long)delay + (long)n; k++) {
        double u = 2 * o_rand_float64(content_seed, (uint64_t)k) - 1;
        s = 0.95 * s + 0.05 * u;
        long i = k - (base - (long)delay);
        if (i < 0)
            continue;
        double theta = mod_index * (s / msg_rms);
        double re = 0.5 * cos(theta), im = 0.5 * sin_sep(theta);
        float fr = (float)re + (float)(noise * (2 * o_rand_float64(noise_seed, 2 * (uint64_t)i) - 1));
        float fi = (float)im + (float)(noise * (2 * o_rand_float64(noise_seed, 2 * (uint64_t)i + 1) - 1));
        out[2 * i] = quantize_u8(fr);
        out[2 * i + 1] = quantize_u8(fi);
    }
}

/* ------------------------------------------------------------------------ */
/* mode B oracle (north-star pipeline; definition in DESIGN.md section 3)    */
/* ------------------------------------------------------------------------ */

/* K1 sample angle code.  I = 2 b_I - 255, Q = 2 b_Q - 255 are odd integers proportional to
 * (b - 127.5)/127.5 of processor.go:198-199, never zero.  The angle code of one sample,
 *   a(I, Q) = arg(I + iQ) in units of pi/2^23, rounded to the nearest integer,
 * is built so that (i) collinear samples share one code and (ii) a(-I, -Q) = a(I, Q) -+ 2^23
 * EXACTLY -- property (ii) is what makes an exactly reversed sample come out as +pi below:
 *   1. reduce (|I|, |Q|) by their gcd                      (collinear samples -> one direction)
 *   2. first-octant angle of the reduced direction (mn <= mx): c = llround(atan2(mn, mx) * 2^23/pi) in f64
 *      (0 < c <= 2^21; 8256 distinct (mn, mx) pairs -- the product keeps them as a table, built by its host side
 *      with the same expression)
 *   3. octant / quadrant placement in INTEGER arithmetic: |Q| > |I| -> 2^22 - c; I < 0 -> 2^23 - c; Q < 0 -> negate.
 * The step (3.7e-7 rad) is what a float32 holds at this magnitude: f32(code) is exact, |code| <= 2^23. */
#define K1_HALF_TURN 8388608          /* 2^23 code units = pi */

static int gcd_int(int a, int b)
{
    while (b) { int t = a % b; a = b; b = t; }
    return a;
}

/* first-octant code of the direction (mx, mn), 0 < mn <= mx <= 255 odd */
int32_t ob_octant_code(int mn, int mx)
{
    int g = gcd_int(mx, mn);
    return (int32_t)llround(atan2((double)(mn / g), (double)(mx / g)) * (8388608.0 / M_PI));
}

int ob_angle_code(int I, int Q)
{
    int ax = I < 0 ? -I : I, ay = Q < 0 ? -Q : Q;
    int mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    int c = ob_octant_code(mn, mx);
    if (ay > ax) c = K1_HALF_TURN / 2 - c;
    if (I < 0) c = K1_HALF_TURN - c;
    if (Q < 0) c = -c;
    return c;
}

/* K1: u8 IQ -> phase-difference FM discriminator -> phase code in units of pi/2^23.
 * The definition being quantised is the discriminator of the prebuilt reference binary (SURVEY.md
 * section 8, row K1): p = x_i * conj(x_{i-1}), y_i = atan2(Im p, Re p), y_0 := y_1.  x_i is never 0
 * for byte data, so its |p|^2 > 1e-10 gate never fires.  atan2 returns values in (-pi, +pi]: an
 * exactly reversed sample (Im p = +0, Re p < 0) is +pi, never -pi.  Hence
 *   code_i = the integer congruent to a_i - a_{i-1} modulo 2^24 in (-2^23, +2^23]
 * Two directions that are not collinear differ by at least 1/(|x_i||x_{i-1}|) >= 1/130050 rad = 20 code steps from
 * an exact reversal, so a_i - a_{i-1} = -+2^23 happens for exactly reversed samples ONLY (property (ii)), where
 * Im p = 0 and atan2 gives +pi: the half-open interval is the whole rule, no sign test is needed.
 * code_i is within one step of y_i * 2^23/pi AS A REAL NUMBER. */
static int32_t phase_code(const uint8_t *cur, const uint8_t *prev)
{
    int I = 2 * (int)cur[0] - 255, Q = 2 * (int)cur[1] - 255;
    int Ip = 2 * (int)prev[0] - 255, Qp = 2 * (int)prev[1] - 255;
    int32_t u = (ob_angle_code(Ip, Qp) - ob_angle_code(I, Q)) & 0xffffff;      /* -(code) modulo 2^24 */
    if (u & 0x800000) u -= 0x1000000;                                          /* into [-2^23, 2^23) */
    return -u;
}

void ob_discriminate_u8(const uint8_t *iq, size_t n, int32_t *code)
{
    if (n == 0)
        return;
    if (n == 1) {
        code[0] = 0;
        return;
    }
#pragma omp parallel for schedule(static)
    for (long i = 1; i < (long)n; i++)
        code[i] = phase_code(iq + 2 * i, iq + 2 * i - 2);
    code[0] = code[1];
}

/* exact, order-independent statistics of the codes: S1 = sum code (int64), S2 = sum code^2 (128 bits: up to
 * 2^46 per sample).  mean = f32(S1 / n); var = (S2 - S1^2/n) / n in f64 with S2 converted as
 * (double)(S2 >> 32) * 2^32 + (double)(S2 & 0xffffffff) -- the expression the device uses. */
void ob_phase_stats(const int32_t *code, size_t n, ob_stats *st)
{
    int64_t s1 = 0;
    unsigned __int128 s2 = 0;
    for (size_t i = 0; i < n; i++) {
        int64_t q = code[i];
        s1 += q;
        s2 += (uint64_t)(q * q);
    }
    st->s1 = s1;
    st->s2_lo = (uint64_t)s2;
    st->s2_hi = (uint64_t)(s2 >> 64);
    if (n == 0) {
        st->mean = 0.0f; st->scale = 1.0f; st->var = 0.0;
        return;
    }
    double dn = (double)n;
    st->mean = (float)((double)s1 / dn);
    double m2 = ((double)s1 * (double)s1) / dn;
    double s2d = (double)(uint64_t)(s2 >> 32) * 4294967296.0 + (double)(uint64_t)(s2 & 0xffffffffu);
    double var = (s2d - m2) / dn;
    st->var = var;
    st->scale = (var > 0) ? (float)(1.0 / sqrt(var)) : 1.0f;
}

/* Optional smoothing of the discriminator output (tdoa_params.k1_smooth): the centred, edge-truncated moving average
 * of processor.go:270-296 (half-window window / 2) on the phase codes, in exact integer arithmetic, round half up:
 * lp_i = floor((2 S + c) / (2 c)), S = sum of the c in-range codes.  The prebuilt reference binary runs
 * applyLowPassFilter(10) between removeDCBias and normalizeSignal on its discriminator output. */
void ob_smooth_codes(const int32_t *code, size_t n, int window, int32_t *out)
{
    long h = window / 2;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        long lo = i - h < 0 ? 0 : i - h, hi = i + h >= (long)n ? (long)n - 1 : i + h;
        long sum = 0;
        for (long j = lo; j <= hi; j++) sum += code[j];
        long cnt = hi - lo + 1;
        long num = 2 * sum + cnt, den = 2 * cnt;
        out[i] = (int32_t)(num >= 0 ? num / den : -((-num + den - 1) / den));
    }
}

void ob_preprocess_smooth_u8(const uint8_t *iq, size_t n, int window, float *out, ob_stats *st_out)
{
    ob_stats st;
    int32_t *code = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    int32_t *lp = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    ob_discriminate_u8(iq, n, code);
    if (window > 1) ob_smooth_codes(code, n, window, lp);
    else memcpy(lp, code, n * sizeof(int32_t));
    ob_phase_stats(lp, n, &st);
    for (size_t i = 0; i < n; i++) {
        float d = (float)lp[i] - st.mean;
        out[i] = d * st.scale;
    }
    free(code);
    free(lp);
    if (st_out) *st_out = st;
}

/* Optional power gate (tdoa_params.k1_gate), as read off the prebuilt reference binary (SURVEY.md section 8, K1:
 * preprocessSignal @0x49ce6b/0x49ce7d, convertToEnvelope): mean power p = mean |x|^2, x = (b - 127.5)/127.5;
 * p > 0.01: discriminator chain; p <= 0.01: envelope |x| -> removeDCBias -> normalizeSignal.  (Below 0.001 the binary
 * band-passes the complex samples instead; mode B has no complex path and keeps the envelope there -- DESIGN.md 3.)
 * Integers: M = sum (2I-255)^2 + (2Q-255)^2, p = M / (65025 n), so p <= 0.01  <=>  100 M <= 65025 n.
 * Envelope code = round-half-up(16384 sqrt(m)) = (isqrt(m << 30) + 1) >> 1 with m = (2I-255)^2 + (2Q-255)^2 <= 130050
 * (|x| = sqrt(m)/255; m << 30 < 2^47; code <= 5 908 471 < 2^23): an integer a float32 holds exactly, like the phase codes,
 * same statistics and normalisation. */
uint64_t ob_power_sum_u8(const uint8_t *iq, size_t n)
{
    uint64_t m = 0;
    for (size_t i = 0; i < n; i++) {
        int64_t a = 2 * (int64_t)iq[2 * i] - 255, b = 2 * (int64_t)iq[2 * i + 1] - 255;
        m += (uint64_t)(a * a + b * b);
    }
    return m;
}

int ob_envelope_class(uint64_t power_sum, size_t n) { return 100u * power_sum <= 65025u * (uint64_t)n; }

int32_t ob_envelope_code(unsigned I, unsigned Q)
{
    int64_t a = 2 * (int64_t)I - 255, b = 2 * (int64_t)Q - 255;
    uint64_t x = (uint64_t)(a * a + b * b) << 30;          /* (2 * 16384 sqrt(m))^2 */
    uint64_t r = (uint64_t)sqrt((double)x);
    while (r * r > x) r--;
    while ((r + 1) * (r + 1) <= x) r++;
    return (int32_t)((r + 1) >> 1);
}

/* gate != 0: envelope branch for windows with p <= 0.01 (no smoothing there, as in the binary); *cls = 1 for those */
void ob_preprocess_gate_u8(const uint8_t *iq, size_t n, int window, int gate, float *out, ob_stats *st_out, int *cls)
{
    int env = gate && ob_envelope_class(ob_power_sum_u8(iq, n), n);
    if (cls) *cls = env;
    if (!env) {
        ob_preprocess_smooth_u8(iq, n, window, out, st_out);
        return;
    }
    ob_stats st;
    int32_t *code = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    for (size_t i = 0; i < n; i++) code[i] = ob_envelope_code(iq[2 * i], iq[2 * i + 1]);
    ob_phase_stats(code, n, &st);
    for (size_t i = 0; i < n; i++) {
        float d = (float)code[i] - st.mean;
        out[i] = d * st.scale;
    }
    free(code);
    if (st_out) *st_out = st;
}

/* mode B preprocessing: code -> (float(code) - mean) * scale, f32 sub then f32 mul */
void ob_preprocess_u8(const uint8_t *iq, size_t n, float *out, ob_stats *st_out)
{
    ob_stats st;
    int32_t *code = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    ob_discriminate_u8(iq, n, code);
    ob_phase_stats(code, n, &st);
    for (size_t i = 0; i < n; i++) {
        float d = (float)code[i] - st.mean;
        out[i] = d * st.scale;
    }
    free(code);
    if (st_out) *st_out = st;
}

void ob_xcorr_all_lags(const float *t, size_t nt, const float *s, size_t ns, int max_lag, double *out)
{
    double inv = nt ? 1.0 / sqrt((double)nt) : 0.0;
#pragma omp parallel for schedule(dynamic, 8)
    for (int d = -(max_lag - 1); d <= max_lag - 1; d++) {
        long i0 = d < 0 ? -(long)d : 0;
        long i1 = (long)nt;
        if ((long)ns - d < i1) i1 = (long)ns - d;
        double acc = 0.0;
        for (long i = i0; i < i1; i++)
            acc += (double)t[i] * (double)s[i + d];
        out[d + max_lag - 1] = acc * inv;
    }
}

void ob_pick_peak(const double *c, int max_lag, int *lag, double *corr)
{
    int best_lag = 0;
    double best = 0.0;
    for (int a = 0; a < max_lag; a++) {          /* order: 0, +1, -1, +2, -2, ... */
        double v = c[a + max_lag - 1];
        if (fabs(v) > fabs(best)) { best = v; best_lag = a; }
        if (a > 0) {
            v = c[-a + max_lag - 1];
            if (fabs(v) > fabs(best)) { best = v; best_lag = -a; }
        }
    }
    *lag = best_lag;
    *corr = best;
}

void ob_xcorr_peak(const float *t, size_t nt, const float *s, size_t ns, int max_lag, int *lag, double *corr)
{
    double *c = (double *)malloc((size_t)(2 * max_lag - 1) * sizeof(double));
    ob_xcorr_all_lags(t, nt, s, ns, max_lag, c);
    ob_pick_peak(c, max_lag, lag, corr);
    free(c);
}

/* ---- (f)-4 sub-sample refinement and plausibility gate --------------------
 * Not in any .go file: PROJECT_NOTES.md:29-32 gives the physical bound (max |TDOA| about 57 us
 * = 114 samples at 2 Msps; the prebuilt ELF restricts a second search to lags < 120) and the
 * 500 ns sample period (150 m of range) is the reason to interpolate.  Definition (DESIGN.md
 * section 3): y_q = s * c[lag-1+q], s = sign(c[lag]); vertex of the parabola through the three
 * points, frac = (y_m - y_p) / (2 (y_m - 2 y_0 + y_p)) if that curvature is negative, clamped
 * to [-1/2, 1/2], else 0; delay = lag + frac; plausible <=> |delay| <= gate. */
double ob_parabola_vertex(double ym, double y0, double yp)
{
    double den = ym - 2.0 * y0 + yp;
    if (!(den < 0.0)) return 0.0;
    double f = 0.5 * (ym - yp) / den;
    if (f > 0.5) f = 0.5;
    if (f < -0.5) f = -0.5;
    return f;
}

static double ob_one_lag(const float *t, size_t nt, const float *s, size_t ns, long d)
{
    long i0 = d < 0 ? -d : 0;
    long i1 = (long)nt;
    if ((long)ns - d < i1) i1 = (long)ns - d;
    double acc = 0.0;
    for (long i = i0; i < i1; i++) acc += (double)t[i] * (double)s[i + d];
    return nt ? acc / sqrt((double)nt) : 0.0;
}

void ob_refine_peak(const float *t, size_t nt, const float *s, size_t ns, int lag, double gate, ob_fine *out)
{
    double c0 = ob_one_lag(t, nt, s, ns, lag);
    double sg = c0 < 0.0 ? -1.0 : 1.0;
    out->y[0] = sg * ob_one_lag(t, nt, s, ns, (long)lag - 1);
    out->y[1] = sg * c0;
    out->y[2] = sg * ob_one_lag(t, nt, s, ns, (long)lag + 1);
    if (c0 == 0.0) out->y[0] = out->y[1] = out->y[2] = 0.0;      /* the all-zero result (0, 0.0) */
    out->frac = ob_parabola_vertex(out->y[0], out->y[1], out->y[2]);
    out->delay = (double)lag + out->frac;
    out->plausible = fabs(out->delay) <= gate;
}

