"""ctypes view of oracle/libtdoa_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (tdoa-geolocation_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtdoa_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "tdoa_oracle.c")
    hdr = os.path.join(_HERE, "tdoa_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


class FastAnalysis(C.Structure):
    _fields_ = [("total_samples", C.c_int),
                ("i_avg", C.c_double), ("q_avg", C.c_double),
                ("i_std", C.c_double), ("q_std", C.c_double),
                ("snr_estimate", C.c_double), ("power_level", C.c_double),
                ("has_clipping", C.c_int), ("has_overload", C.c_int)]


class Station(C.Structure):
    _fields_ = [("lat", C.c_double), ("lon", C.c_double), ("elev", C.c_double)]


class BStats(C.Structure):
    _fields_ = [("s1", C.c_int64), ("s2_lo", C.c_uint64), ("s2_hi", C.c_uint64),
                ("mean", C.c_float), ("scale", C.c_float), ("var", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        L = _lib
        fp, u8p, dp = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        sz = C.c_size_t
        L.o_signal_power.restype = C.c_double
        L.o_signal_power.argtypes = [fp, sz]
        L.o_extract_reference.restype = sz
        L.o_extract_target.restype = sz
        L.o_distance3d.restype = C.c_double
        L.o_fast_snr.restype = C.c_double
        L.o_rand_float64.restype = C.c_double
        L.o_rand_float64.argtypes = [C.c_uint64, C.c_uint64]
        L.ob_octant_code.restype = C.c_int32
        L.ob_octant_code.argtypes = [C.c_int, C.c_int]
        L.ob_angle_code.restype = C.c_int
        L.ob_angle_code.argtypes = [C.c_int, C.c_int]
        L.o_lowpass.argtypes = [fp, sz, C.c_int, fp]
        L.o_cutoff_window.argtypes = [C.c_double, C.c_double]
        L.o_lowpass_cutoff.argtypes = [fp, sz, C.c_double, C.c_double, fp]
        L.o_highpass.argtypes = [fp, sz, C.c_double, C.c_double, fp]
        L.o_bandpass.argtypes = [fp, sz, C.c_double, C.c_double, C.c_double, fp]
        L.o_notch.argtypes = [fp, sz, C.c_double, C.c_double, C.c_double, fp]
        L.o_enhance_weak.argtypes = [fp, sz, C.c_double, fp]
        L.o_preprocess.argtypes = [fp, sz, C.c_double, fp]
        L.o_time_domain_correlation.argtypes = [fp, sz, fp, sz, C.c_int, ip, dp]
        L.o_time_domain_all_lags.argtypes = [fp, sz, fp, sz, C.c_int, dp, C.c_int]
        L.o_cross_correlate.argtypes = [fp, sz, fp, sz, C.c_double, ip, dp]
        L.o_simple_correlate.argtypes = [fp, sz, fp, sz, ip, fp]
        L.o_simulate_station.argtypes = [u8p, sz, C.c_double, C.c_double, C.c_double, C.c_double,
                                         Station, Station, C.c_double, C.c_uint64]
        L.o_simulate_weak_station.argtypes = [u8p, sz, C.c_double, C.c_double, C.c_double,
                                              Station, Station, C.c_double, C.c_double, C.c_uint64]
        L.o_simulate_delayed_fm.argtypes = [u8p, sz, C.c_int, C.c_double, C.c_double,
                                            C.c_uint64, C.c_uint64]
        L.o_latlon_to_ecef.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.o_ecef_to_latlon.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.ob_xcorr_all_lags.argtypes = [fp, sz, fp, sz, C.c_int, dp]
        L.ob_xcorr_peak.argtypes = [fp, sz, fp, sz, C.c_int, ip, dp]
        L.ob_parabola_vertex.argtypes = [C.c_double, C.c_double, C.c_double]
        L.ob_parabola_vertex.restype = C.c_double
        L.ob_refine_peak.argtypes = [fp, sz, fp, sz, C.c_int, C.c_double, C.POINTER(Fine)]
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _c64(a):
    """complex64 array -> contiguous float32 view [2n]."""
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a, a.view(np.float32)


# --- processor.go ------------------------------------------------------------

def iq_u8_to_c64(raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    n = raw.size // 2
    out = np.empty(n, dtype=np.complex64)
    lib().o_iq_u8_to_c64(_u8(raw), C.c_size_t(n), _f(out.view(np.float32)))
    return out


def extract_reference(data):
    a, v = _c64(data)
    out = np.empty(max(a.size, 1), dtype=np.complex64)
    m = lib().o_extract_reference(_f(v), C.c_size_t(a.size), _f(out.view(np.float32)))
    return out[:m]


def extract_target(data):
    a, v = _c64(data)
    out = np.empty(max(a.size, 1), dtype=np.complex64)
    m = lib().o_extract_target(_f(v), C.c_size_t(a.size), _f(out.view(np.float32)))
    return out[:m]


def signal_power(sig):
    a, v = _c64(sig)
    return lib().o_signal_power(_f(v), a.size)


def remove_dc(sig):
    a, v = _c64(sig)
    out = np.empty_like(a)
    re, im = C.c_float(), C.c_float()
    lib().o_remove_dc(_f(v), C.c_size_t(a.size), _f(out.view(np.float32)), C.byref(re), C.byref(im))
    return out, complex(re.value, im.value)


def lowpass(sig, window):
    a, v = _c64(sig)
    out = np.empty_like(a)
    lib().o_lowpass(_f(v), a.size, int(window), _f(out.view(np.float32)))
    return out


def cutoff_window(cutoff, fs=2e6):
    return lib().o_cutoff_window(float(cutoff), float(fs))


def highpass(sig, cutoff, fs=2e6):
    a, v = _c64(sig)
    out = np.empty_like(a)
    lib().o_highpass(_f(v), a.size, cutoff, fs, _f(out.view(np.float32)))
    return out


def bandpass(sig, lo, hi, fs=2e6):
    a, v = _c64(sig)
    out = np.empty_like(a)
    lib().o_bandpass(_f(v), a.size, lo, hi, fs, _f(out.view(np.float32)))
    return out


def notch(sig, f0, bw, fs=2e6):
    a, v = _c64(sig)
    out = np.empty_like(a)
    lib().o_notch(_f(v), a.size, f0, bw, fs, _f(out.view(np.float32)))
    return out


def normalize(sig):
    a, v = _c64(sig)
    out = np.empty_like(a)
    lib().o_normalize(_f(v), C.c_size_t(a.size), _f(out.view(np.float32)))
    return out


def enhance_weak(sig, fs=2e6):
    a, v = _c64(sig)
    out = np.empty_like(a)
    lib().o_enhance_weak(_f(v), a.size, fs, _f(out.view(np.float32)))
    return out


def preprocess(sig, fs=2e6):
    a, v = _c64(sig)
    out = np.empty_like(a)
    weak = lib().o_preprocess(_f(v), a.size, fs, _f(out.view(np.float32)))
    return out, bool(weak)


def time_domain_correlation(s1, s2, max_lag):
    a, va = _c64(s1)
    b, vb = _c64(s2)
    d, c = C.c_int(), C.c_double()
    lib().o_time_domain_correlation(_f(va), a.size, _f(vb), b.size, int(max_lag), C.byref(d), C.byref(c))
    return d.value, c.value


def time_domain_all_lags(s1, s2, max_lag):
    a, va = _c64(s1)
    b, vb = _c64(s2)
    out = np.zeros(max(int(max_lag), 1), dtype=np.float64)
    m = lib().o_time_domain_all_lags(_f(va), a.size, _f(vb), b.size, int(max_lag), _d(out), out.size)
    return out[:m]


def cross_correlate(s1, s2, fs=2e6):
    a, va = _c64(s1)
    b, vb = _c64(s2)
    d, c = C.c_int(), C.c_double()
    lib().o_cross_correlate(_f(va), a.size, _f(vb), b.size, fs, C.byref(d), C.byref(c))
    return d.value, c.value


def next_pow2(n):
    return lib().o_next_pow2(int(n))


def simple_dft(sig):
    a, v = _c64(sig)
    out = np.empty_like(a)
    lib().o_simple_dft(_f(v), C.c_int(a.size), _f(out.view(np.float32)))
    return out


def frequency_domain_correlation(s1, s2, max_lag):
    a, va = _c64(s1)
    b, vb = _c64(s2)
    d, c = C.c_int(), C.c_double()
    rc = lib().o_frequency_domain_correlation(_f(va), C.c_size_t(a.size), _f(vb), C.c_size_t(b.size),
                                              int(max_lag), C.byref(d), C.byref(c))
    return rc, d.value, c.value


def simple_correlate(s1, s2):
    a, va = _c64(s1)
    b, vb = _c64(s2)
    d, c = C.c_int(), C.c_float()
    lib().o_simple_correlate(_f(va), a.size, _f(vb), b.size, C.byref(d), C.byref(c))
    return d.value, c.value


# --- fast_analyzer.go ----------------------------------------------------------

def fast_snr(samples_u8, total_samples):
    s = np.ascontiguousarray(samples_u8, dtype=np.uint8)
    return lib().o_fast_snr(_u8(s), C.c_int(int(total_samples)))


def fast_analyze(samples_u8, total_samples):
    s = np.ascontiguousarray(samples_u8, dtype=np.uint8)
    fa = FastAnalysis()
    lib().o_fast_analyze(_u8(s), C.c_int(int(total_samples)), C.byref(fa))
    return fa


def block_power(iq_u8):
    s = np.ascontiguousarray(iq_u8, dtype=np.uint8)
    f = lib().o_block_power
    f.restype = C.c_double
    return f(_u8(s), C.c_size_t(s.size // 2))


def fast_analyze_capture(raw_u8):
    s = np.ascontiguousarray(raw_u8, dtype=np.uint8)
    ref, tgt = FastAnalysis(), FastAnalysis()
    rc = lib().o_fast_analyze_capture(_u8(s), C.c_size_t(s.size), C.byref(ref), C.byref(tgt))
    return rc, ref, tgt


# --- geodesy / solver -----------------------------------------------------------

def latlon_to_ecef(lat, lon, elev):
    out = np.zeros(3)
    lib().o_latlon_to_ecef(lat, lon, elev, _d(out))
    return out


def distance3d(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return lib().o_distance3d(_d(a), _d(b))


def ecef_to_latlon(x, y, z):
    out = np.zeros(3)
    lib().o_ecef_to_latlon(x, y, z, _d(out))
    return out


def solve_tdoa(stations_lle, range_diff):
    st = np.ascontiguousarray(stations_lle, dtype=np.float64).reshape(9)
    rd = np.ascontiguousarray(range_diff, dtype=np.float64)
    out = np.zeros(3)
    it = C.c_int()
    rc = lib().o_solve_tdoa(_d(st), _d(rd), _d(out), C.byref(it))
    return rc, out, it.value


# --- simulators ---------------------------------------------------------------------

STATIONS = {  # lat-lon-table.csv / simulator.go:195-218
    "162400000": (41.25703803095629, -95.95512763589404, 349.07),
    "kx0u": (41.18660274289527, -95.96064116595667, 355.69),
    "n3pay": (41.24669616513154, -96.08366304481238, 329.0),
    "kf0mtl": (41.32916620016985, -96.03513381562004, 373.18),
}
COLLECTORS = ["kx0u", "n3pay", "kf0mtl"]     # simulator.go:272
DEFAULT_TX = (41.20, -96.00, 400.0)          # simulator.go:229 example
SEED_BASE = 0x5D0A0000


def simulate_station(name_or_lle, block_samples, seed, fs=2e6, ref_freq=162.4e6, tgt_freq=101.7e6,
                     noise=0.01, tx=DEFAULT_TX, tx_power=1000.0):
    lle = STATIONS[name_or_lle] if isinstance(name_or_lle, str) else name_or_lle
    out = np.empty(6 * block_samples, dtype=np.uint8)
    lib().o_simulate_station(_u8(out), block_samples, fs, ref_freq, tgt_freq, noise,
                             Station(*lle), Station(*tx), tx_power, seed)
    return out


def simulate_weak_station(name_or_lle, block_samples, seed, fs=2e6, ref_freq=162.4e6, tgt_freq=92.3e6,
                          tx=DEFAULT_TX, ref_power=10.0, tgt_power=1000.0):
    lle = STATIONS[name_or_lle] if isinstance(name_or_lle, str) else name_or_lle
    out = np.empty(6 * block_samples, dtype=np.uint8)
    lib().o_simulate_weak_station(_u8(out), block_samples, fs, ref_freq, tgt_freq,
                                  Station(*lle), Station(*tx), ref_power, tgt_power, seed)
    return out


def simulate_delayed_fm(n_samples, delay, content_seed, noise_seed, mod_index=1.0, noise=0.02):
    out = np.empty(2 * n_samples, dtype=np.uint8)
    lib().o_simulate_delayed_fm(_u8(out), n_samples, int(delay), mod_index, noise, content_seed, noise_seed)
    return out


def rand_float64(seed, counter):
    return lib().o_rand_float64(seed, counter)


# --- mode B ------------------------------------------------------------------------

def b_octant_code(mn, mx):
    return lib().ob_octant_code(int(mn), int(mx))


def b_angle_code(i, q):
    return lib().ob_angle_code(int(i), int(q))


def b_discriminate(iq_u8):
    """u8 IQ -> phase codes in units of pi/2^23, -2^23 < code <= 2^23 (int32)."""
    s = np.ascontiguousarray(iq_u8, dtype=np.uint8)
    n = s.size // 2
    out = np.empty(n, dtype=np.int32)
    lib().ob_discriminate_u8(_u8(s), C.c_size_t(n), out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def b_phase_stats(code):
    p = np.ascontiguousarray(code, dtype=np.int32)
    st = BStats()
    lib().ob_phase_stats(p.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(p.size), C.byref(st))
    return st


def b_preprocess(iq_u8):
    s = np.ascontiguousarray(iq_u8, dtype=np.uint8)
    n = s.size // 2
    out = np.empty(n, dtype=np.float32)
    st = BStats()
    lib().ob_preprocess_u8(_u8(s), C.c_size_t(n), _f(out), C.byref(st))
    return out, st


def b_preprocess_smooth(iq_u8, window):
    """mode B preprocessing with the optional moving average on the phase codes (tdoa_params.k1_smooth)"""
    s = np.ascontiguousarray(iq_u8, dtype=np.uint8)
    n = s.size // 2
    out = np.empty(n, dtype=np.float32)
    st = BStats()
    lib().ob_preprocess_smooth_u8(_u8(s), C.c_size_t(n), C.c_int(int(window)), _f(out), C.byref(st))
    return out, st


def b_preprocess_gate(iq_u8, window=0, gate=1):
    """mode B preprocessing with the prebuilt binary's power gate (tdoa_params.k1_gate): returns (signal, stats, cls),
    cls = 1 when the window took the envelope branch (mean power <= 0.01)"""
    s = np.ascontiguousarray(iq_u8, dtype=np.uint8)
    n = s.size // 2
    out = np.empty(n, dtype=np.float32)
    st = BStats()
    cls = C.c_int()
    lib().ob_preprocess_gate_u8(_u8(s), C.c_size_t(n), C.c_int(int(window)), C.c_int(int(gate)), _f(out), C.byref(st),
                                C.byref(cls))
    return out, st, cls.value


def b_envelope_code(i, q):
    f = lib().ob_envelope_code
    f.restype = C.c_int32
    return f(C.c_uint(int(i)), C.c_uint(int(q)))


def b_xcorr_all_lags(t, s, max_lag):
    t = np.ascontiguousarray(t, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float32)
    out = np.empty(2 * max_lag - 1, dtype=np.float64)
    lib().ob_xcorr_all_lags(_f(t), t.size, _f(s), s.size, int(max_lag), _d(out))
    return out


def b_pick_peak(c, max_lag):
    c = np.ascontiguousarray(c, dtype=np.float64)
    lag, corr = C.c_int(), C.c_double()
    lib().ob_pick_peak(_d(c), C.c_int(int(max_lag)), C.byref(lag), C.byref(corr))
    return lag.value, corr.value


def b_xcorr_peak(t, s, max_lag):
    t = np.ascontiguousarray(t, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float32)
    lag, corr = C.c_int(), C.c_double()
    lib().ob_xcorr_peak(_f(t), t.size, _f(s), s.size, int(max_lag), C.byref(lag), C.byref(corr))
    return lag.value, corr.value


class Fine(C.Structure):
    _fields_ = [("delay", C.c_double), ("frac", C.c_double), ("y", C.c_double * 3), ("plausible", C.c_int)]


def b_parabola_vertex(ym, y0, yp):
    return lib().ob_parabola_vertex(float(ym), float(y0), float(yp))


def b_refine_peak(t, s, lag, gate):
    """ob_refine_peak -> dict(delay, frac, y[3], plausible)"""
    t = np.ascontiguousarray(t, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float32)
    f = Fine()
    lib().ob_refine_peak(_f(t), t.size, _f(s), s.size, int(lag), float(gate), C.byref(f))
    return dict(delay=f.delay, frac=f.frac, y=np.array(list(f.y)), plausible=bool(f.plausible))


def b_xcorr_peak_fft(t, s, max_lag):
    """float64 FFT evaluation of ob_xcorr_all_lags + ob_pick_peak, for sizes the
    direct form cannot finish in seconds.  Same definition, different algorithm."""
    t = np.asarray(t, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64)
    n = 1
    while n < t.size + s.size:
        n <<= 1
    ft = np.fft.rfft(t, n)
    fs_ = np.fft.rfft(s, n)
    r = np.fft.irfft(np.conj(ft) * fs_, n)
    lags = np.arange(-(max_lag - 1), max_lag)
    c = r[lags % n] / np.sqrt(float(t.size))
    return b_pick_peak(c, max_lag) + (c,)
