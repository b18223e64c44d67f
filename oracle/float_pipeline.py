"""Independent float64 statement of the north-star pipeline -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

This is the SECOND anchor of mode B ("next round" item 1).  It shares no code with
oracle/tdoa_oracle.c's ob_* functions and none with the device kernels: no angle table, no 16-bit
phase code, no ob_theta polynomial.  It follows SURVEY.md section 8, row K1 -- the discriminator as it
was read off the prebuilt reference binary (`convertToInstantaneousFrequency` @0x49d120; the only place
in the reference where an FM discriminator exists) -- and the padding / cross-power hints of the dead
`frequencyDomainCorrelation` (processor.go:539-616):

    x_i = ((I_i - 127.5)/127.5, (Q_i - 127.5)/127.5)            processor.go:198-199
    p   = x_i * conj(x_{i-1})
    y_i = atan2(Im p, Re p)  if |p|^2 > 1e-10 else 0 ;  y_0 := y_1
          with p evaluated EXACTLY: x_i = (2 b - 255)/255 is a small odd integer over 255, so p is an exact
          integer over 255^2 and atan2 sees its true sign.  That matters for exactly reversed samples
          (Im p = 0, Re p < 0 -- 2 % of the samples of a simulator.go capture, whose I/Q only take the values
          +-1, +-3 LSB): atan2(+0, negative) = +pi.  The binary evaluates p in f64 from f32-rounded x and gets
          the same +0 whenever |x_i| = |x_{i-1}|; for collinear samples of different magnitude its Im p is
          rounding noise of either sign, i.e. +-pi at random -- a floating-point accident at the one point
          where atan2 is discontinuous, not a definition.  (numpy's own complex multiply uses FMA and flips
          even the equal-magnitude case, which is why the product is spelled out on integers here.)
    v   = (y - mean(y)) / sqrt(var(y))                           removeDCBias + normalizeSignal, f64
    c[d] = (1/sqrt(L_t)) sum_i v_t[i] v_s[i+d]                   full-overlap linear correlation by f64 FFT,
                                                                 N = nextPow2(L_t + L_s)  (processor.go:563 rule)
    peak = largest |c[d]| over |d| < max_lag; ties -> smaller |d|, then positive d
                                                                 (processor.go:596-611 scan order, :722-725 strict >)

Everything is numpy float64.
"""
import numpy as np


def discriminate(iq_u8):
    """u8 IQ -> instantaneous phase difference in radians (float64), the ELF's definition."""
    b = np.asarray(iq_u8, dtype=np.uint8)
    n = b.size // 2
    if n == 0:
        return np.zeros(0)
    if n == 1:
        return np.zeros(1)
    i = 2 * b[0:2 * n:2].astype(np.int64) - 255            # 255 * x.re, exact
    q = 2 * b[1:2 * n:2].astype(np.int64) - 255            # 255 * x.im
    re = i[1:] * i[:-1] + q[1:] * q[:-1]                   # 255^2 * Re(x_i conj(x_{i-1}))
    im = q[1:] * i[:-1] - i[1:] * q[:-1]                   # 255^2 * Im(...)
    mag2 = (re.astype(np.float64) ** 2 + im.astype(np.float64) ** 2) / 255.0 ** 4
    y = np.where(mag2 > 1e-10, np.arctan2(im.astype(np.float64), re.astype(np.float64)), 0.0)
    return np.concatenate([[y[0]], y])


def normalise(y):
    """zero mean, unit variance (population), float64; variance <= 0 leaves the scale at 1"""
    y = np.asarray(y, dtype=np.float64)
    if y.size == 0:
        return y
    m = y.mean()
    var = ((y - m) ** 2).mean()
    return (y - m) * (1.0 / np.sqrt(var) if var > 0 else 1.0)


def lowpass(y, window):
    """applyLowPassFilter (processor.go:270-296): centred moving average, half-window window // 2, taps outside the
    signal dropped (the divisor is the number of taps that are inside)"""
    y = np.asarray(y, dtype=np.float64)
    h = int(window) // 2
    if window <= 1 or y.size == 0:
        return y
    c = np.concatenate([[0.0], np.cumsum(y)])
    i = np.arange(y.size)
    lo, hi = np.maximum(i - h, 0), np.minimum(i + h, y.size - 1)
    return (c[hi + 1] - c[lo]) / (hi - lo + 1)


def mean_power(iq_u8):
    """mean |x|^2 of x = (b - 127.5) / 127.5 -- the quantity the prebuilt binary's preprocessSignal gates on"""
    b = np.asarray(iq_u8, dtype=np.float64).reshape(-1, 2)
    x = (b - 127.5) / 127.5
    return float((x * x).sum(axis=1).mean()) if len(x) else 0.0


def envelope(iq_u8):
    """the binary's convertToEnvelope: sqrt(re^2 + im^2) of x = (b - 127.5) / 127.5"""
    b = np.asarray(iq_u8, dtype=np.float64).reshape(-1, 2)
    x = (b - 127.5) / 127.5
    return np.sqrt((x * x).sum(axis=1))


def preprocess(iq_u8, smooth=0, gate=False):
    """smooth = 0: discriminator -> zero mean -> unit variance.  smooth = W: the prebuilt binary's strong-signal chain in
    ITS order: discriminator -> removeDCBias -> applyLowPassFilter(W) -> normalizeSignal (scale to
    unit mean power, no second mean removal).  gate: the binary's power gate -- mean power <= 0.01 takes
    envelope -> removeDCBias -> normalizeSignal instead (its third branch, <= 0.001, is not restated: DESIGN.md 3)"""
    if gate and mean_power(iq_u8) <= 0.01:
        return normalise(envelope(iq_u8))
    y = discriminate(iq_u8)
    if smooth <= 1:
        return normalise(y)
    if y.size == 0:
        return y
    v = lowpass(y - y.mean(), smooth)
    p = (v * v).mean()
    return v * (1.0 / np.sqrt(p) if p > 0 else 1.0)


def xcorr_lags(t, s, max_lag):
    """c[d] for d = -(max_lag-1) .. max_lag-1 (index d + max_lag - 1), float64 FFT evaluation"""
    t = np.asarray(t, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64)
    if t.size == 0 or s.size == 0:
        return np.zeros(2 * max_lag - 1)
    n = 1
    while n < t.size + s.size:
        n <<= 1
    r = np.fft.irfft(np.conj(np.fft.rfft(t, n)) * np.fft.rfft(s, n), n)
    d = np.arange(-(max_lag - 1), max_lag)
    c = r[d % n] / np.sqrt(float(t.size))
    # lags that leave no overlap are exactly zero by definition (the FFT leaves rounding noise there)
    c[(d >= s.size) | (d <= -t.size)] = 0.0
    return c


def pick_peak(c, max_lag):
    """(lag, corr): largest |c|; ties -> smaller |lag|, then the positive lag; NaN never wins; all-zero -> (0, 0.0)"""
    c = np.asarray(c, dtype=np.float64)
    d = np.arange(-(max_lag - 1), max_lag)
    mag = np.where(np.isnan(c), -1.0, np.abs(c))
    top = mag.max()
    if not top > 0.0:
        return 0, 0.0
    cand = d[mag == top]
    order = np.lexsort((cand < 0, np.abs(cand)))       # primary |d|, secondary: positive first
    lag = int(cand[order[0]])
    return lag, float(c[lag + max_lag - 1])


# ---- a test signal for the envelope branch of the optional power gate ----
def am_capture(n, delay, amp, seed, station):
    """amplitude-modulated carrier of mean power ~amp^2 (a moderate-signal capture for the binary's envelope branch):
    a common low-pass random message delayed by `delay` samples, independent phase walk and noise per station"""
    rng = np.random.default_rng(seed)
    msg = np.convolve(rng.standard_normal(n + 4096 + 64), np.ones(32) / np.sqrt(32.0), mode="same")
    own = np.random.default_rng(1000 * seed + station)
    m = msg[2048 - delay:2048 - delay + n]
    phase = np.cumsum(own.standard_normal(n) * 0.2)
    x = amp * (1.0 + 0.5 * np.tanh(m)) * np.exp(1j * phase) + (own.standard_normal(n) + 1j * own.standard_normal(n)) * amp * 0.05
    iq = np.empty(2 * n, dtype=np.uint8)
    iq[0::2] = np.clip(np.trunc(x.real * 127.5 + 127.5), 0, 255).astype(np.uint8)
    iq[1::2] = np.clip(np.trunc(x.imag * 127.5 + 127.5), 0, 255).astype(np.uint8)
    return iq


def xcorr_peak_u8(iq_t, iq_s, max_lag, smooth=0, gate=False):
    """u8 IQ of the template and the signal -> (lag, corr, all lags)"""
    c = xcorr_lags(preprocess(iq_t, smooth, gate), preprocess(iq_s, smooth, gate), max_lag)
    lag, corr = pick_peak(c, max_lag)
    return lag, corr, c
